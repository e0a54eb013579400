package jpeg

/** GPU-backed stand-ins for the reference generators, same names and parameter lists, written from
  * scratch for the C ABI (no Chisel).  Construction validates exactly like the reference's require()s
  * (IllegalArgumentException); instead of a Decoupled `io` bundle the classes expose process(). */
object ProcessingStep extends Enumeration {
  type Type = Value
  val NoOp, SpatialSampling, ColorQuantization, ChromaSubsampling = Value   // ordinals 0..3
}

case class ImageProcessorParams(width: Int, height: Int, factor: Int, chromaParamA: Int, chromaParamB: Int) {
  private[jpeg] def packed(outFormat: Int) = NativeBackend.pack(width, height, chromaParamA, chromaParamB, 8, 8, 8,
    factor, Seq(3, 1, 2), NativeBackend.FloorHw, outFormat, strictDivisible = true)
  NativeBackend.validate(packed(NativeBackend.FmtArgb))   // the five require()s incl. divisibility
}

class ImageCompressorTop(
    width: Int, height: Int,
    chroma_param_a_config: Int, chroma_param_b_config: Int,
    yTargetQuantBitsConfig: Int, cbTargetQuantBitsConfig: Int, crTargetQuantBitsConfig: Int,
    downFactorConfig: Int,
    op1Type: ProcessingStep.Type, op2Type: ProcessingStep.Type, op3Type: ProcessingStep.Type,
    rounding: Int = NativeBackend.FloorHw, device: Int = 0
) extends AutoCloseable {
  private def packed(fmt: Int) = NativeBackend.pack(width, height, chroma_param_a_config, chroma_param_b_config,
    yTargetQuantBitsConfig, cbTargetQuantBitsConfig, crTargetQuantBitsConfig, downFactorConfig,
    Seq(op1Type.id, op2Type.id, op3Type.id), rounding, fmt, strictDivisible = false)
  NativeBackend.validate(packed(NativeBackend.FmtArgb))
  private val dims = NativeBackend.outDims(packed(NativeBackend.FmtArgb))
  val outWidth: Int = dims(0); val outHeight: Int = dims(1)
  private lazy val rgbPlan = NativeBackend.planCreate(packed(NativeBackend.FmtArgb), device)
  private lazy val yccPlan = NativeBackend.planCreate(packed(NativeBackend.FmtYcc), device)
  private lazy val planarPlan = NativeBackend.planCreate(packed(NativeBackend.FmtPlanar), device)
  /** csic_planar_layout of these parameters: plane sizes and offsets of processPlanar's buffer (no device needed). */
  lazy val planarLayout: NativeBackend.PlanarLayout = NativeBackend.planarLayoutOf(packed(NativeBackend.FmtPlanar))
  private var opened = Set.empty[Long]

  /** ARGB frame in -> reconstructed ARGB frame out (DUT output through YCbCrUtils.ycbcr2rgb). */
  def process(argb: Array[Int]): Array[Int] = run(rgbPlan, argb)
  /** ARGB frame in -> io.out's PixelYCbCrBundle stream, packed Y | Cb << 8 | Cr << 16. */
  def processYCbCr(argb: Array[Int]): Array[Int] = run(yccPlan, argb)
  /** ARGB frame in -> the subsampled planar frame buffer: one Y byte per output pixel at planarLayout.yOffset, one Cb / Cr byte per
    * chroma SAMPLE POINT at cbOffset / crOffset (4:2:0: 1.5 bytes per pixel) -- the format the reference's README describes
    * (README.md:35-46) and ChromaSubsampler.scala:57-65 never builds. */
  def processPlanar(argb: Array[Int]): Array[Byte] = {
    opened += planarPlan
    val out = new Array[Byte](planarLayout.frameBytes.toInt)
    NativeBackend.processPlanar(planarPlan, argb, out)
    out
  }

  private def run(plan: Long, argb: Array[Int]): Array[Int] = {
    opened += plan
    val out = new Array[Int](outWidth * outHeight)
    NativeBackend.process(plan, argb, out)
    out
  }
  override def close(): Unit = { opened.foreach(NativeBackend.planDestroy); opened = Set.empty }
}

class ImageProcessor(p: ImageProcessorParams, device: Int = 0)
  extends ImageCompressorTop(p.width, p.height, p.chromaParamA, p.chromaParamB, 8, 8, 8, p.factor,
    ProcessingStep.ChromaSubsampling, ProcessingStep.SpatialSampling, ProcessingStep.ColorQuantization,
    NativeBackend.FloorHw, device)
