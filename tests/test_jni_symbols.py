"""The JNI shim cannot be compiled in this image (no JDK, no jni.h), so the link between the Scala natives and
the C exports is checked at source level: the names the JVM will look up are DERIVED from NativeBackend.scala by
the JNI name-mangling rules and compared with what csic_jni.c defines (VERDICT r01 weak item 3 / ADVICE).

JNI spec, "Resolving Native Method Names": Java_ + mangled fully-qualified class name + _ + mangled method name,
with '/' -> '_', '_' -> '_1', ';' -> '_2', '[' -> '_3' and any non-ASCII-alphanumeric char -> _0xxxx (so '$' ->
'_00024').  A Scala `object X` compiles its `@native def`s to instance methods of class `X$`, hence the `_00024`
and a `jobject` (not `jclass`) second parameter."""
import os
import re

from conftest import ROOT

JVM = os.path.join(ROOT, "chroma-subsampling-image-compressor_amd", "jvm")


def _mangle(s: str) -> str:
    out = []
    for ch in s:
        if ch.isascii() and ch.isalnum():
            out.append(ch)
        elif ch in "/.":
            out.append("_")
        elif ch == "_":
            out.append("_1")
        elif ch == ";":
            out.append("_2")
        elif ch == "[":
            out.append("_3")
        else:
            out.append("_0%04x" % ord(ch))
    return "".join(out)


def _scala_natives():
    src = open(os.path.join(JVM, "scala", "jpeg", "NativeBackend.scala")).read()
    pkg = re.search(r"^package\s+([\w.]+)", src, re.M).group(1)
    m = re.search(r"^(object|class)\s+(\w+)", src, re.M)
    kind, name = m.group(1), m.group(2)
    cls = f"{pkg}.{name}" + ("$" if kind == "object" else "")
    natives = re.findall(r"@native\s+def\s+(\w+)\s*\(([^)]*)\)\s*:\s*([\w\[\]]+)", src)
    return kind, cls, natives


def test_jni_exports_match_the_names_the_jvm_resolves():
    kind, cls, natives = _scala_natives()
    assert natives, "no @native methods found"
    want = {f"Java_{_mangle(cls)}_{_mangle(name)}": (name, args) for name, args, _ in natives}
    csrc = open(os.path.join(JVM, "jni", "csic_jni.c")).read()
    defs = re.findall(r"JNIEXPORT\s+(\w+)\s+JNICALL\s+(\w+)\s*\(\s*JNIEnv\s*\*\s*\w+\s*,\s*(\w+)\s+\w+\s*([^)]*)\)", csrc)
    got = {sym: (ret, recv, rest) for ret, sym, recv, rest in defs}
    assert set(got) == set(want), (sorted(got), sorted(want))
    # receiver: natives of a Scala object are instance methods of the module class -> jobject
    for sym, (_, recv, _) in got.items():
        assert recv == ("jobject" if kind == "object" else "jclass"), (sym, recv)
    # arity and JNI types of the remaining parameters
    jtype = {"Int": "jint", "Long": "jlong", "Array[Int]": "jintArray", "Array[Byte]": "jbyteArray", "Array[Long]": "jlongArray", "Unit": "void"}
    for name, args, ret in natives:
        sym = f"Java_{_mangle(cls)}_{_mangle(name)}"
        scala_types = [a.split(":")[1].strip() for a in args.split(",") if a.strip()]
        c_types = [a.split()[0] for a in got[sym][2].lstrip(",").split(",") if a.strip()]
        assert c_types == [jtype[t] for t in scala_types], (sym, c_types, scala_types)
        assert got[sym][0] == jtype[ret], (sym, got[sym][0], ret)


def test_mangling_rules():
    assert _mangle("jpeg.NativeBackend$") == "jpeg_NativeBackend_00024"
    assert _mangle("a_b") == "a_1b"


def test_integration_doc_shows_the_same_symbols():
    """INTEGRATION.md presents the binding to the reference's maintainers: it must not advertise the unmangled form."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert "Java_jpeg_NativeBackend_00024_" in doc
    assert not re.search(r"Java_jpeg_NativeBackend_(?!00024_)\w", doc)


# ---- the rest of the Scala host surface, checked at source level: no JDK / scalac in the image ---------------
def _scala(name):
    return open(os.path.join(JVM, "scala", "jpeg", name)).read()


def _strip(src):
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r'"""(?:.|\n)*?"""', '""', src)
    src = re.sub(r'"(?:\\.|[^"\\\n])*"', '""', src)
    return re.sub(r"//[^\n]*", "", src)


def test_scala_sources_are_balanced_and_in_package_jpeg():
    files = sorted(f for f in os.listdir(os.path.join(JVM, "scala", "jpeg")) if f.endswith(".scala"))
    assert files == ["GpuImageCompressor.scala", "HostModels.scala", "ImageCompressionApp.scala", "ImageProcessorModel.scala",
                     "NativeBackend.scala", "SoftwareModel.scala"]
    for f in files:
        src = _strip(_scala(f))
        assert re.search(r"^package jpeg\s*$", src, re.M), f
        for a, b in ("()", "[]", "{}"):
            assert src.count(a) == src.count(b), (f, a, src.count(a), src.count(b))


def test_scala_surface_keeps_the_reference_names():
    """object / method names and parameter lists a user of the reference calls; paths relative to
    /root/reference/src."""
    m = _strip(_scala("ImageProcessorModel.scala"))                   # test/scala/jpeg/ImageProcessorModel.scala:9-53
    assert "object ImageProcessorModel" in m
    assert re.search(r"type PixelType\s*=\s*Seq\[Int\]", m) and re.search(r"type ImageType\s*=\s*Seq\[Seq\[PixelType\]\]", m)
    assert re.search(r"def readImage\(file: String\)", m)
    assert len(re.findall(r"def writeImage\(", m)) == 2 and re.search(r"def writeImage\(\w+: Array\[Int\], p: ImageProcessorParams, file: String\)", m)
    assert re.search(r"def getImageParams\(image: Image, numPixelsPerCycle: Int\): ImageProcessorParams", m)
    assert re.search(r"chromaParamA = 4, chromaParamB = 4", m) and re.search(r"def getImagePixels\(image: Image\): ImageType", m)
    assert "mkdirs()" in m and "scrimage" not in m.replace("scrimage's", "")

    h = _strip(_scala("HostModels.scala"))                            # main/scala/jpeg/RGB2YCbCr.scala:94-133, ReferenceModel.scala:5-19
    assert re.search(r"object YCbCrUtils\s*\{", h) and re.search(r"def rgbToYCbCr\(r_in: Int, g_in: Int, b_in: Int\): \(Int, Int, Int\)", h)
    assert re.search(r"def ycbcr2rgb\(y: Int, cb: Int, cr: Int\): \(Int, Int, Int\)", h)
    assert re.search(r"object ReferenceModel\s*\{", h) and "case class PixelRGB(r: Int, g: Int, b: Int)" in h
    assert "case class PixelYCbCr(y: Int, cb: Int, cr: Int)" in h and re.search(r"def rgb2ycbcr\(p: PixelRGB\): PixelYCbCr", h)
    assert re.search(r"Array\(77, 150, 29, -43, -85, 128, 128, -107, -21\)", h)
    for c in ("298", "409", "100", "208", "516"):                                     # App. A.5
        assert c in h
    assert "t >> 8" in h and "t / 256" in h                                           # both roundings, by name
    assert re.search(r"rgbToYCbCr.*floor = false", h) and re.search(r"Fixed8\.forward\(p\.r, p\.g, p\.b, floor = true\)", h)

    a = _strip(_scala("ImageCompressionApp.scala"))                   # test/scala/jpeg/ImageCompressorTopApp.scala:23-37, 149-215
    assert "object ImageCompressionApp" in a
    sig = re.search(r"def processImage\(([^)]*)\)", a, re.S).group(1)
    names = [p.split(":")[0].strip() for p in sig.split(",")]
    assert names == ["inputImagePath", "outputImagePath", "chromaParamA", "chromaParamB", "yTargetBits", "cbTargetBits", "crTargetBits",
                     "spatialFactorToUse", "op1", "op2", "op3"]
    raw = _scala("ImageCompressionApp.scala")
    for key, default in (("--input", "test_images/in128x128.png"), ("--a", "4"), ("--b", "4"), ("--yq", "8"), ("--cbq", "8"), ("--crq", "8"),
                         ("--sf", "8"), ("--op1", "spatial"), ("--op2", "color"), ("--op3", "chroma")):
        assert f'getOrElse("{key}", "{default}")' in raw, key
    assert '"spatial" | "spatialsampling"' in raw and "Unknown processing step" in raw and "[ERROR] Input image not found" in raw
    assert "0xFFFF00FF" in raw and "APP_OUTPUT" in raw and "_processed_chroma4-" in raw and "args.sliding(2, 2)" in raw

    g = _strip(_scala("GpuImageCompressor.scala"))                    # main/scala/jpeg/ImageCompressorTop.scala:11-25
    ctor = re.search(r"class ImageCompressorTop\(([^)]*)\)", g, re.S).group(1)
    assert [p.split(":")[0].strip() for p in ctor.split(",")][:11] == [
        "width", "height", "chroma_param_a_config", "chroma_param_b_config", "yTargetQuantBitsConfig", "cbTargetQuantBitsConfig",
        "crTargetQuantBitsConfig", "downFactorConfig", "op1Type", "op2Type", "op3Type"]
    assert re.search(r"case class ImageProcessorParams\(width: Int, height: Int, factor: Int, chromaParamA: Int, chromaParamB: Int\)", g)
    assert re.search(r"val NoOp, SpatialSampling, ColorQuantization, ChromaSubsampling = Value", g)

    sw = _strip(_scala("SoftwareModel.scala"))                        # the "Scala/JVM CPU path" baseline, SURVEY.md 8d
    assert "final class SoftwareModel(" in sw and "def process(argb: Array[Int]): Array[Int]" in sw and "object SoftwareModelBench" in sw
    assert "0xFF << (8 - yBits)" in sw and "cPix % h == 0 && cLine % v == 0" in sw and "sCol % factor == 0" in sw


def test_order_tag_quirk_is_reproduced_in_the_scala_app():
    """ImageCompressorTopApp.scala:188 prints `Pr` for every step (ChiselEnum.toString); a scala.Enumeration would print `Sp`/`Co`/`Ch`,
    so the Scala app formats the ChiselEnum spelling explicitly -- same expression shape as the Python app's _order_tag."""
    raw = _scala("ImageCompressionApp.scala")
    assert 's"ProcessingStep(${step.id}=$step)".split(\'.\').last.take(2)' in raw
    import csic_amd.app as app
    assert app._order_tag(app.ProcessingStep.SpatialSampling) == "Pr"
