"""Builds tests/cpp/host_test.cpp (g++, against include/csic.hpp + libcsic_hip.so) and runs it:
the C++ host layer must throw on every reference require() without a GPU, and reproduce the
reference's known-answer vectors on one."""
import os
import subprocess

import pytest

from conftest import ROOT

PKG = os.path.join(ROOT, "chroma-subsampling-image-compressor_amd")
EXE = os.path.join(ROOT, "tests", "cpp", "host_test")


def _build():
    src = os.path.join(ROOT, "tests", "cpp", "host_test.cpp")
    lib = os.path.join(PKG, "libcsic_hip.so")
    assert os.path.exists(lib), "build libcsic_hip.so first (python -c 'import __graft_entry__ as g; g.build()')"
    deps = [src, os.path.join(ROOT, "include", "csic.hpp"), os.path.join(ROOT, "include", "csic.h"), lib]
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(d) for d in deps):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"), src,
                               "-L" + PKG, "-lcsic_hip", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib", "-o", EXE])
    return EXE


def _run(mode, tmp_path):
    golden = os.path.join(ROOT, "tests", "golden")
    args = [_build(), mode, os.path.join(golden, "inputs", "in16.png"),
            os.path.join(golden, "outputs", "ip_420_sf2_16.png"), str(tmp_path / "cpp_out.png")]
    r = subprocess.run(args, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout + r.stderr


def test_cpp_host_layer_requires_and_png(tmp_path):
    _run("cpu", tmp_path)


@pytest.mark.gpu
def test_cpp_host_layer_kats_and_integration_flow_on_gpu(tmp_path):
    _run("gpu", tmp_path)


@pytest.mark.gpu
def test_frame_graph_c_abi_from_native_hip_host_code():
    """tests/cpp/graph_test.hip: csic_frame_graph_* from C++ with the HIP runtime (hipcc), both backends, producer and
    consumer on the launch stream, no Python between the host code and the C ABI."""
    src = os.path.join(ROOT, "tests", "cpp", "graph_test.hip")
    exe = os.path.join(ROOT, "tests", "cpp", "graph_test")
    lib = os.path.join(PKG, "libcsic_hip.so")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(lib)):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"), src,
                               "-L" + PKG, "-lcsic_hip", "-Wl,-rpath," + PKG, "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout + r.stderr
    assert "backend DIRECT" in r.stdout and "backend HIP" in r.stdout and "backend FUSED" in r.stdout
