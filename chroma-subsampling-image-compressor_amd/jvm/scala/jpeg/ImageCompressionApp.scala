package jpeg

import java.io.File

/** Drop-in for `sbt "Test / runMain jpeg.ImageCompressionApp ..."` (src/test/scala/jpeg/ImageCompressorTopApp.scala:18-216):
  * same keys (`--input --a --b --yq --cbq --crq --sf --op1 --op2 --op3`, space-separated pairs), same defaults (note
  * sf = 8 and the order spatial, color, chroma), same banner and output naming -- but the frame goes through one fused HIP
  * launch (ImageCompressorTop.process over JNI) instead of a pixel-per-clock treadle simulation. */
object ImageCompressionApp {

  /** Same 11 parameters, same order as ImageCompressorTopApp.scala:23-37. */
  def processImage(
      inputImagePath: String, outputImagePath: String,
      chromaParamA: Int, chromaParamB: Int,
      yTargetBits: Int, cbTargetBits: Int, crTargetBits: Int,
      spatialFactorToUse: Int,
      op1: ProcessingStep.Type, op2: ProcessingStep.Type, op3: ProcessingStep.Type): Unit = {
    val in = ImageProcessorModel.readImage(inputImagePath)
    val f = spatialFactorToUse
    val hasSpatial = Seq(op1, op2, op3).contains(ProcessingStep.SpatialSampling)
    val finalW = if (hasSpatial) in.width / f else in.width         // :44-45
    val finalH = if (hasSpatial) in.height / f else in.height
    if (hasSpatial && (in.width % f != 0 || in.height % f != 0))
      println(s"[WARN] Image dimensions (${in.width}x${in.height}) are not perfectly divisible by spatialFactor ($f). SpatialDownsampler might truncate.")
    val top = new ImageCompressorTop(in.width, in.height, chromaParamA, chromaParamB, yTargetBits, cbTargetBits, crTargetBits, f, op1, op2, op3)
    val stream = try top.process(in.argb) finally top.close()       // ceil(W/f) * ceil(H/f) pixels, row-major
    // the harness keeps the first finalW * finalH pixels of the output stream and lays them out finalW per row; what it
    // never collected stays magenta (:108-124, :133-142)
    val out = Array.fill(finalW * finalH)(0xFFFF00FF)
    System.arraycopy(stream, 0, out, 0, math.min(stream.length, out.length))
    ImageProcessorModel.writeImage(Image(finalW, finalH, out), outputImagePath)
  }

  def parseProcessingStep(name: String): ProcessingStep.Type = name.toLowerCase match {   // :154-161
    case "spatial" | "spatialsampling" => ProcessingStep.SpatialSampling
    case "color" | "colorquantization" => ProcessingStep.ColorQuantization
    case "chroma" | "chromasubsampling" => ProcessingStep.ChromaSubsampling
    case _ => throw new IllegalArgumentException(s"Unknown processing step: $name. Use 'spatial', 'color', or 'chroma'.")
  }

  /** The reference formats its order tag from a ChiselEnum value, whose toString is "ProcessingStep(1=SpatialSampling)":
    * `.split('.').last.take(2)` yields "Pr" for every step (:188).  Kept, because the file name is the interface. */
  private def orderTag(step: ProcessingStep.Type): String = s"ProcessingStep(${step.id}=$step)".split('.').last.take(2)

  def main(args: Array[String]): Unit = {
    val argsMap = args.sliding(2, 2).collect { case Array(k, v) if k.startsWith("--") => k -> v }.toMap   // :149-151
    val inputPath = argsMap.getOrElse("--input", "test_images/in128x128.png")
    val a = argsMap.getOrElse("--a", "4").toInt
    val b = argsMap.getOrElse("--b", "4").toInt
    val yq = argsMap.getOrElse("--yq", "8").toInt
    val cbq = argsMap.getOrElse("--cbq", "8").toInt
    val crq = argsMap.getOrElse("--crq", "8").toInt
    val sf = argsMap.getOrElse("--sf", "8").toInt
    val op1 = parseProcessingStep(argsMap.getOrElse("--op1", "spatial"))
    val op2 = parseProcessingStep(argsMap.getOrElse("--op2", "color"))
    val op3 = parseProcessingStep(argsMap.getOrElse("--op3", "chroma"))
    val imageName = new File(inputPath).getName.takeWhile(_ != '.')

    val rule = "----------------------------------------------------"
    println(rule); println("Image Compressor Application Parameters:"); println(rule)
    println(s"Input Image: $inputPath")
    println(s"Selected Chroma Subsampling (J:a:b): 4:$a:$b")
    println(s"Selected Quantization Bits (Y/Cb/Cr): $yq/$cbq/$crq")
    println(s"Selected Spatial Downsampling Factor: $sf")
    println(s"Selected Pipeline Order: $op1 -> $op2 -> $op3")
    println(rule)

    val outDir = "APP_OUTPUT"
    val order = s"order-${orderTag(op1)}-${orderTag(op2)}-${orderTag(op3)}"
    val outputPath = s"$outDir/${imageName}_processed_chroma4-$a-${b}_Y${yq}Cb${cbq}Cr${crq}_sf${sf}_$order.png"
    new File(outDir).mkdirs()
    if (!new File(inputPath).exists()) println(s"[ERROR] Input image not found: $inputPath")      // :197-199, not an exception
    else {
      processImage(inputPath, outputPath, a, b, yq, cbq, crq, sf, op1, op2, op3)
      println(s"Image processing complete. Output saved to: $outputPath")
    }
  }
}
