package jpeg

/** The reference's small pure-Scala colour models, written from the normative arithmetic (SURVEY.md App. A.1 / A.5),
  * with the names its host code calls:
  *   YCbCrUtils.rgbToYCbCr   <- RGB2YCbCr.scala:95-121   the TRUNC_SW form: (x + 128) / 256, Scala's `/` truncates toward zero
  *   YCbCrUtils.ycbcr2rgb    <- RGB2YCbCr.scala:123-132 == YCbCr2RGB.scala:17-26 (c = y, NOT y - 16: the pair is lossy)
  *   ReferenceModel.rgb2ycbcr<- ReferenceModel.scala:8-19 the FLOOR_HW form: (x + 128) >> 8, what the RTL computes
  * Scalar helpers for tests and for the per-pixel call sites a maintainer keeps (ImageCompressorTopApp.scala:118);
  * whole frames go through ImageCompressorTop.process on the GPU, or through SoftwareModel when a CPU baseline is wanted. */
private[jpeg] object Fixed8 {
  /** rows Y, Cb, Cr of the forward matrix, times 256 */
  val Forward: Array[Int] = Array(77, 150, 29, -43, -85, 128, 128, -107, -21)

  @inline def sat8(v: Int): Int = if (v < 0) 0 else if (v > 255) 255 else v

  /** numerator of channel `ch` (0 = Y, 1 = Cb, 2 = Cr) with the +128 rounding bias already added */
  @inline def biased(ch: Int, r: Int, g: Int, b: Int): Int =
    Forward(3 * ch) * r + Forward(3 * ch + 1) * g + Forward(3 * ch + 2) * b + 128

  /** FLOOR_HW: arithmetic shift; TRUNC_SW: integer division that truncates toward zero.  They differ by one LSB in Cb / Cr
    * whenever the biased numerator is negative and not a multiple of 256 -- 74 % of all colours; Y never differs. */
  @inline def forward(r: Int, g: Int, b: Int, floor: Boolean): (Int, Int, Int) = {
    def q(t: Int): Int = if (floor) t >> 8 else t / 256
    (sat8(q(biased(0, r, g, b))), sat8(q(biased(1, r, g, b)) + 128), sat8(q(biased(2, r, g, b)) + 128))
  }

  @inline def inverse(y: Int, cb: Int, cr: Int): (Int, Int, Int) = {
    val d = cb - 128
    val e = cr - 128
    val luma = 298 * y + 128
    (sat8((luma + 409 * e) >> 8), sat8((luma - 100 * d - 208 * e) >> 8), sat8((luma + 516 * d) >> 8))
  }
}

object YCbCrUtils {
  def rgbToYCbCr(r_in: Int, g_in: Int, b_in: Int): (Int, Int, Int) = Fixed8.forward(r_in, g_in, b_in, floor = false)
  def ycbcr2rgb(y: Int, cb: Int, cr: Int): (Int, Int, Int) = Fixed8.inverse(y, cb, cr)
}

object ReferenceModel {
  case class PixelRGB(r: Int, g: Int, b: Int)
  case class PixelYCbCr(y: Int, cb: Int, cr: Int)

  def rgb2ycbcr(p: PixelRGB): PixelYCbCr = {
    val (y, cb, cr) = Fixed8.forward(p.r, p.g, p.b, floor = true)
    PixelYCbCr(y, cb, cr)
  }
}
