"""The range-checking build (libcsic_hip_debug.so: `make -C chroma-subsampling-image-compressor_amd/csrc debug`, -DCSIC_DEBUG).
GPU AddressSanitizer is not available on the pool, so this is the device-side sanitizer stand-in: every global
access of the pixel kernels is checked against the frame extent and traps.  Two things are shown here, in CHILD processes
(CSIC_LIB selects the library at import time, and a trap ends its process):
  * the checks are live: one checked read outside the frame kills the child (and the same read inside the frame does not);
  * a broad parity sample through every kernel family runs clean under them.
The whole `-m gpu` suite under the debug library is run by tools/run_debug_suite.sh (log: profiles/r04_gpu_tests_debug.log)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

PKG = os.path.join(ROOT, "chroma-subsampling-image-compressor_amd")
DEBUG_LIB = os.path.join(PKG, "libcsic_hip_debug.so")


def _child(code, timeout=300):
    env = dict(os.environ, CSIC_LIB=DEBUG_LIB, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


@pytest.fixture(scope="module", autouse=True)
def _built():
    if not os.path.exists(DEBUG_LIB):
        subprocess.check_call(["make", "-C", os.path.join(PKG, "csrc"), "-s", "debug"])


PROBE = """
import ctypes as C, sys, torch, csic_amd as csic
N = csic._native; lib = N.lib()
assert lib.csic_debug_build() == 1, "not the debug build"
buf = torch.arange(64 * 8, dtype=torch.int32, device="cuda:0")
sink = torch.zeros(1, dtype=torch.int32, device="cuda:0")
off = int(sys.argv[1]) if len(sys.argv) > 1 else %d
N.check(lib.csic_debug_probe_device(C.c_void_p(buf.data_ptr()), 64, 4, off, C.c_void_p(sink.data_ptr()), None))
torch.cuda.synchronize()
print("read", int(sink.item()))
"""


def test_the_product_library_is_not_the_debug_build():
    import csic_amd
    assert csic_amd._native.lib().csic_debug_build() == 0


def test_a_checked_read_inside_the_frame_passes_and_one_outside_traps():
    ok = _child(PROBE % (64 * 4 - 1))                       # the frame's last pixel: extent = (H - 1) * pitch + W = 256
    assert ok.returncode == 0 and "read 255" in ok.stdout, ok.stdout + ok.stderr
    bad = _child(PROBE % (64 * 4))                          # one past it -- still inside the ALLOCATION (512 words): only the check can object
    assert bad.returncode != 0 and "read" not in bad.stdout, bad.stdout + bad.stderr
    neg = _child(PROBE % -1)
    assert neg.returncode != 0 and "read" not in neg.stdout, neg.stdout + neg.stderr


SAMPLE = """
import itertools, numpy as np, torch, csic_amd as csic
from oracle import oracle as orc
N = csic._native
assert N.lib().csic_debug_build() == 1
rng = np.random.default_rng(404)
modes = [(4, 4), (2, 2), (2, 0), (1, 1), (4, 0), (1, 0)]
orders = list(itertools.permutations((1, 2, 3)))
kernels = set()
for it in range(260):
    W, H = int(rng.integers(1, 120)), int(rng.integers(1, 50))
    if rng.random() < 0.4: W = (W + 7) // 8 * 8
    a, b = modes[int(rng.integers(0, 6))]
    f = int(rng.choice([1, 2, 4, 8]))
    op = orders[int(rng.integers(0, 6))]
    avg = rng.random() < 0.25
    if avg: op = (3, 1, 2)
    planar = rng.random() < 0.3
    argb = rng.integers(0, 1 << 32, W * H, dtype=np.uint32)
    p = orc.OracleParams(width=W, height=H, chroma_a=a, chroma_b=b, y_bits=7, cb_bits=6, cr_bits=5, factor=f, op=op)
    cp = csic.make_c_params(W, H, a, b, 7, 6, 5, f, op, out_format=2 if planar else 0, sampling=csic.Sampling.AVG if avg else csic.Sampling.HOLD_DECIMATE)
    with csic.Plan(cp, 0) as pl:
        for variant in (0, 5, 7, 9, 10, 11) if not avg else (0, 8):
            pl.tune(N.TUNE_VARIANT, variant)
            kernels.add(pl.kernel_name.split("<")[0])
            want = orc.process(p, argb, form="avg" if avg else "stream")
            if planar:
                d = torch.from_numpy(argb.view(np.int32)).cuda()
                got = pl.reconstruct_device(pl.process_device(d)).cpu().numpy().view(np.uint32)
            else:
                got = pl.process_host(argb)
            assert np.array_equal(got, want), (pl.kernel_name, W, H, a, b, f, op, avg, planar, variant)
torch.cuda.synchronize()
print("clean", len(kernels), sorted(kernels))
"""


def test_every_kernel_family_runs_clean_under_the_range_checks():
    r = _child(SAMPLE, timeout=900)
    assert r.returncode == 0 and "clean" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    for fam in ("k_avg", "k_dec", "k_decflat", "k_f1flat", "k_f1x4", "k_flatgen", "k_generic", "k_planar_flat", "k_planar_strided", "k_planar_avg_f1", "k_planar_avg_gen"):
        assert f"'{fam}'" in r.stdout, (fam, r.stdout)
