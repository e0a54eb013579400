package jpeg

import java.awt.image.BufferedImage
import java.io.File
import javax.imageio.ImageIO

/** Packed-ARGB stand-in for scrimage's ImmutableImage / MutableImage: what the reference's harness reads through
  * `image.pixel(x, y).red()` etc. (ImageCompressorTopApp.scala:86-89) is an Array[Int] of 0xAARRGGBB here. */
final case class Image(width: Int, height: Int, argb: Array[Int]) {
  require(argb.length == width * height, s"expected ${width * height} pixels, got ${argb.length}")
  def pixel(x: Int, y: Int): (Int, Int, Int) = { val v = argb(y * width + x); ((v >> 16) & 255, (v >> 8) & 255, v & 255) }
}

/** The reference's host I/O helper (src/test/scala/jpeg/ImageProcessorModel.scala:9-53), same method names, over
  * javax.imageio instead of scrimage 4.1.1 (no third-party jar needed on the GPU box).  Like the original it holds no
  * arithmetic.  Decoding takes straight 8-bit samples (getRGB on a TYPE_INT_ARGB copy): input alpha is ignored downstream
  * and ancillary colour chunks are not applied -- the behaviour the reference's golden images pin (SURVEY.md 8c). */
object ImageProcessorModel {
  type PixelType = Seq[Int]                    // ImageProcessorModel.scala:11
  type ImageType = Seq[Seq[PixelType]]         // :12

  def readImage(file: String): Image = {       // :14-16
    val bi = ImageIO.read(new File(file))
    if (bi == null) throw new java.io.IOException(s"cannot decode $file")
    val w = bi.getWidth; val h = bi.getHeight
    Image(w, h, bi.getRGB(0, 0, w, h, null, 0, w))
  }

  def writeImage(image: Image, file: String): Unit = {   // :18-22
    val out = new File(file)
    Option(out.getAbsoluteFile.getParentFile).foreach(_.mkdirs())
    val bi = new BufferedImage(image.width, image.height, BufferedImage.TYPE_INT_RGB)
    bi.setRGB(0, 0, image.width, image.height, image.argb, 0, image.width)
    if (!ImageIO.write(bi, "png", out)) throw new java.io.IOException(s"no PNG writer for $file")
  }

  def writeImage(pixels: Array[Int], p: ImageProcessorParams, file: String): Unit =     // :24-28
    writeImage(Image(p.width, p.height, pixels), file)

  /** Defaults chroma to 4:4:4 (a = b = 4), as the reference does when only the decimator is of interest (:33-41). */
  def getImageParams(image: Image, numPixelsPerCycle: Int): ImageProcessorParams =
    ImageProcessorParams(width = image.width, height = image.height, factor = numPixelsPerCycle, chromaParamA = 4, chromaParamB = 4)

  def getImagePixels(image: Image): ImageType =           // :43-52, [row][col] = Seq(r, g, b)
    Seq.tabulate(image.height, image.width) { (r, c) => val (red, green, blue) = image.pixel(c, r); Seq(red, green, blue) }
}
