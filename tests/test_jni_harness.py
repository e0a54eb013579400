"""jvm/jni/csic_jni.c through a compiler and a fake JNIEnv.

The image has no JDK, so the shim is compiled against tests/cpp/jni_stub/jni.h (the JNI types plus the function-table slots
the shim uses, at the indices the JNI specification gives them) into tests/cpp/libcsic_jni.so, and tests/cpp/jni_harness.c
calls every Java_jpeg_NativeBackend_00024_* export with a JNIEnv whose arrays are malloc'd int[] objects that MOVE after every
region copy, whose ThrowNew records class and message, and whose critical-region entry points are counted (must stay 0: the
shim may not hold one across a GPU round trip).  A compile-and-logic check of the shim -- not a JVM ABI proof.

Reference behaviour replayed: require() -> IllegalArgumentException at construction (SpatialDownsamplerSpec.scala:147-151), the
ImageProcessor integration flow in16x16.png -> 4:2:0, sf 2 (SpatialDownsamplerSpec.scala:172-227, golden ip_420_sf2_16.png)
and the app flow in128x128.png -> 4:2:2, sf 2 (ImageCompressorTopApp.scala:189-190, golden app_422_888_sf2_128.png)."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

PKG = os.path.join(ROOT, "chroma-subsampling-image-compressor_amd")
CPP = os.path.join(ROOT, "tests", "cpp")
SHIM_SRC = os.path.join(PKG, "jvm", "jni", "csic_jni.c")
SHIM = os.path.join(CPP, "libcsic_jni.so")
EXE = os.path.join(CPP, "jni_harness")
STUB = os.path.join(CPP, "jni_stub")
FLAGS = ["-std=c11", "-O1", "-g", "-Wall", "-Wextra", "-Werror", "-I" + STUB, "-I" + os.path.join(ROOT, "include")]
LINK = ["-L" + PKG, "-lcsic_hip", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"]


def _build():
    lib = os.path.join(PKG, "libcsic_hip.so")
    assert os.path.exists(lib), "build libcsic_hip.so first (python -c 'import __graft_entry__ as g; g.build()')"
    hdrs = [os.path.join(STUB, "jni.h"), os.path.join(ROOT, "include", "csic.h"), lib]
    if not os.path.exists(SHIM) or os.path.getmtime(SHIM) < max(os.path.getmtime(d) for d in [SHIM_SRC] + hdrs):
        subprocess.check_call(["gcc"] + FLAGS + ["-fPIC", "-shared", "-fvisibility=hidden", SHIM_SRC] + LINK + ["-o", SHIM])
    src = os.path.join(CPP, "jni_harness.c")
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(d) for d in [src, SHIM] + hdrs):
        subprocess.check_call(["gcc"] + FLAGS + [src, "-L" + CPP, "-lcsic_jni", "-Wl,-rpath," + CPP] + LINK + ["-o", EXE])
    return EXE


def _exports():
    out = subprocess.check_output(["nm", "-D", "--defined-only", SHIM], text=True)
    return sorted(l.split()[-1] for l in out.splitlines() if " T " in l)


def test_the_shim_compiles_and_exports_exactly_the_natives_of_the_scala_object():
    _build()
    scala = open(os.path.join(PKG, "jvm", "scala", "jpeg", "NativeBackend.scala")).read()
    natives = re.findall(r"@native\s+def\s+(\w+)", scala)
    assert natives and _exports() == sorted("Java_jpeg_NativeBackend_00024_" + n for n in natives)


def test_no_critical_region_in_the_shim():
    """JNI spec, Get/ReleasePrimitiveArrayCritical: no blocking call inside a critical region -- and a frame is a GPU round trip."""
    code = re.sub(r"/\*.*?\*/", "", open(SHIM_SRC).read(), flags=re.S)
    assert "PrimitiveArrayCritical" not in code
    assert "csic_process_host" not in code and "csic_pipeline_submit" in code and "GetIntArrayRegion" in code


def test_every_export_through_a_fake_jnienv_without_a_gpu():
    r = subprocess.run([_build(), "cpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout + r.stderr
    assert "critical regions 0" in r.stdout


@pytest.mark.gpu
def test_golden_flows_through_the_jni_exports_on_the_gpu():
    g = os.path.join(ROOT, "tests", "golden")
    args = [_build(), "gpu", os.path.join(g, "inputs", "in16.png"), os.path.join(g, "outputs", "ip_420_sf2_16.png"),
            os.path.join(g, "inputs", "in128.png"), os.path.join(g, "outputs", "app_422_888_sf2_128.png")]
    r = subprocess.run(args, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout + r.stderr
    assert "ImageProcessor integration flow: 16x16 -> 8x8 bit-exact" in r.stdout
    assert "ImageCompressionApp flow: 128x128 -> 64x64 bit-exact" in r.stdout
    assert "critical regions 0" in r.stdout
