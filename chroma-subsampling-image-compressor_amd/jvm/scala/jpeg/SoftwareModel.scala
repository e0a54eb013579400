package jpeg

/** Single-threaded, allocation-free software model of the whole pixel stream on packed ARGB frames: the "Scala/JVM CPU
  * path" that BASELINE.json wants timed beside the GPU (SURVEY.md 8d, CPU baseline (3)).  Written from SURVEY.md App. A --
  * the stages are the reference's state machines run in op order on a stream of (y, cb, cr) ints:
  *   forward     App. A.1   RGB2YCbCr.scala:33-65 (floor) / :95-121 (trunc)
  *   chroma      App. A.3   ChromaSubsampler.scala:37-65   sample-and-hold, counters wrap at the FULL width / height
  *   spatial     App. A.4   SpatialDownsampler.scala:17-55 keep (row % f == 0 && col % f == 0)
  *   quantiser   App. A.2   ColorQuantizer.scala:29-44      v & (0xFF << (8 - bits))
  *   inverse     App. A.5   YCbCr2RGB.scala:17-26
  * It exists to be timed and to cross-check the GPU from the JVM side; it is not used by ImageCompressorTop. */
final class SoftwareModel(width: Int, height: Int, a: Int, b: Int, yBits: Int, cbBits: Int, crBits: Int, factor: Int,
                          op1: ProcessingStep.Type, op2: ProcessingStep.Type, op3: ProcessingStep.Type, floorRounding: Boolean = true) {
  NativeBackend.validate(NativeBackend.pack(width, height, a, b, yBits, cbBits, crBits, factor, Seq(op1.id, op2.id, op3.id),
    if (floorRounding) NativeBackend.FloorHw else NativeBackend.TruncSw, NativeBackend.FmtArgb, strictDivisible = false))

  val outWidth: Int = (width + factor - 1) / factor
  val outHeight: Int = (height + factor - 1) / factor
  private val h = 4 / a
  private val v = if (b == 0) 2 else 1
  private val maskY = (0xFF << (8 - yBits)) & 0xFF
  private val maskCb = (0xFF << (8 - cbBits)) & 0xFF
  private val maskCr = (0xFF << (8 - crBits)) & 0xFF
  private val order = Array(op1.id, op2.id, op3.id)

  /** ARGB frame (width * height ints) -> reconstructed ARGB frame (outWidth * outHeight ints), alpha = 255. */
  def process(argb: Array[Int]): Array[Int] = {
    require(argb.length == width * height, s"expected ${width * height} pixels, got ${argb.length}")
    val out = new Array[Int](outWidth * outHeight)
    // stage state, as in the RTL: chroma latch + its two counters, the decimator's two counters
    var lastCb = 0; var lastCr = 0; var cPix = 0; var cLine = 0
    var sCol = 0; var sRow = 0
    var n = 0
    var i = 0
    while (i < argb.length) {
      val px = argb(i)
      var (y, cb, cr) = Fixed8.forward((px >> 16) & 255, (px >> 8) & 255, px & 255, floorRounding)
      var alive = true
      var k = 0
      while (k < 3 && alive) {
        order(k) match {
          case 1 =>                                             // SpatialSampling
            alive = (sCol % factor == 0) && (sRow % factor == 0)
            sCol += 1
            if (sCol == width) { sCol = 0; sRow += 1; if (sRow == height) sRow = 0 }
          case 2 =>                                             // ColorQuantization
            y &= maskY; cb &= maskCb; cr &= maskCr
          case _ =>                                             // ChromaSubsampling
            if (cPix % h == 0 && cLine % v == 0) { lastCb = cb; lastCr = cr } else { cb = lastCb; cr = lastCr }
            cPix += 1
            if (cPix == width) { cPix = 0; cLine += 1; if (cLine == height) cLine = 0 }
        }
        k += 1
      }
      if (alive) {
        val (r, g, bl) = Fixed8.inverse(y, cb, cr)
        out(n) = 0xFF000000 | (r << 16) | (g << 8) | bl
        n += 1
      }
      i += 1
    }
    out
  }
}

/** `scala jpeg.SoftwareModelBench [width height seconds]`: Mpixel/s of the model on synthetic frames (the counter-based
  * generator of SURVEY.md 8d), one thread, after JIT warm-up -- the number bench.py's cpu_baseline.jvm field asks for. */
object SoftwareModelBench {
  private def fmix32(x0: Int): Int = {
    var x = x0
    x ^= x >>> 16; x *= 0x85ebca6b; x ^= x >>> 13; x *= 0xc2b2ae35; x ^= x >>> 16
    x
  }

  def main(args: Array[String]): Unit = {
    val w = if (args.length > 0) args(0).toInt else 8192
    val hgt = if (args.length > 1) args(1).toInt else 8192
    val budget = if (args.length > 2) args(2).toDouble else 10.0
    val frame = Array.tabulate(w * hgt)(i => 0xFF000000 | (fmix32(i + 20250629 * 0x9E3779B9) & 0xFFFFFF))
    val model = new SoftwareModel(w, hgt, 2, 0, 8, 8, 8, 2, ProcessingStep.ChromaSubsampling, ProcessingStep.SpatialSampling,
      ProcessingStep.ColorQuantization)
    var sink = 0
    for (_ <- 0 until 3) sink ^= model.process(frame)(0)        // warm-up: let C2 compile the loop
    var frames = 0
    val t0 = System.nanoTime()
    while ((System.nanoTime() - t0) * 1e-9 < budget) { sink ^= model.process(frame)(0); frames += 1 }
    val secs = (System.nanoTime() - t0) * 1e-9
    println(f"""{"kind": "scala-jvm", "frames": $frames, "width": $w, "height": $hgt, "seconds": $secs%.2f, """ +
      f""""Mpixels_per_s": ${frames.toDouble * w * hgt / secs / 1e6}%.2f, "cores": 1, "java": "${System.getProperty("java.version")}", "sink": $sink}""")
  }
}
