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
    jtype = {"Int": "jint", "Long": "jlong", "Array[Int]": "jintArray", "Unit": "void"}
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
