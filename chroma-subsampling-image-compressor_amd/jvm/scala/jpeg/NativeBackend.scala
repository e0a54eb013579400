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

  val FloorHw = 0; val TruncSw = 1
  val FmtArgb = 0; val FmtYcc = 1

  def pack(width: Int, height: Int, a: Int, b: Int, yq: Int, cbq: Int, crq: Int, sf: Int,
           ops: Seq[Int], rounding: Int, outFormat: Int, strictDivisible: Boolean): Array[Int] =
    Array(width, height, a, b, yq, cbq, crq, sf, ops(0), ops(1), ops(2), rounding,
          /*sampling*/ 0, /*in_format*/ FmtArgb, outFormat, if (strictDivisible) 1 else 0)
}
