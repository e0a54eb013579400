package jpeg

/** JNI entry points of libcsic_jni.so (jvm/jni/csic_jni.c), 1:1 over include/csic.h.
  * `params` is an Array[Int](16) in csic_params field order. */
object NativeBackend {
  System.loadLibrary("csic_jni")
  @native def validate(params: Array[Int]): Unit
  @native def planCreate(params: Array[Int], device: Int): Long
  @native def planDestroy(handle: Long): Unit
  @native def outDims(params: Array[Int]): Array[Int]
  @native def process(handle: Long, argbIn: Array[Int], out: Array[Int]): Unit
  /** the 14 fields of csic_planar_layout (include/csic.h) in declaration order; needs no device */
  @native def planarLayout(params: Array[Int]): Array[Long]
  /** a plan created with outFormat = FmtPlanar: one frame -> its planar frame buffer (PlanarLayout.frameBytes bytes) */
  @native def processPlanar(handle: Long, argbIn: Array[Int], out: Array[Byte]): Unit

  val FloorHw = 0; val TruncSw = 1
  val FmtArgb = 0; val FmtYcc = 1; val FmtPlanar = 2

  def pack(width: Int, height: Int, a: Int, b: Int, yq: Int, cbq: Int, crq: Int, sf: Int,
           ops: Seq[Int], rounding: Int, outFormat: Int, strictDivisible: Boolean): Array[Int] =
    Array(width, height, a, b, yq, cbq, crq, sf, ops(0), ops(1), ops(2), rounding,
          /*sampling*/ 0, /*in_format*/ FmtArgb, outFormat, if (strictDivisible) 1 else 0)

  /** csic_planar_layout: one Y byte per output pixel, one Cb / Cr byte per chroma sample point (4:2:0: 1.5 bytes per pixel) --
    * the wire format the reference's README describes (README.md:35-46) and ChromaSubsampler never builds. */
  final case class PlanarLayout(yWidth: Int, yHeight: Int, chromaWidth: Int, chromaHeight: Int, moduleWidth: Int, holdH: Int, holdV: Int,
                                replayLast: Boolean, chromaSamples: Long, yOffset: Long, cbOffset: Long, crOffset: Long,
                                frameBytes: Long, payloadBytes: Long)
  def planarLayoutOf(params: Array[Int]): PlanarLayout = {
    val v = planarLayout(params)
    PlanarLayout(v(0).toInt, v(1).toInt, v(2).toInt, v(3).toInt, v(4).toInt, v(5).toInt, v(6).toInt, v(7) != 0, v(8), v(9), v(10), v(11), v(12), v(13))
  }
}
