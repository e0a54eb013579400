"""pytest configuration: markers, import path, shared fixtures.

`-m "not gpu"` : oracle vs the reference's golden vectors/KATs, host logic, C-ABI symbol checks.
`-m gpu`       : parity tests proper -- the HIP path, called through the C-ABI, against the oracle.
Nothing here reads /root/reference; fixtures are the committed copies under tests/golden/.
"""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _ensure_native_built():
    """The .so files are build artefacts (git-ignored).  On a fresh checkout build them before any test
    module imports the package: hipcc cross-compiles gfx950 without a GPU, gcc builds the oracle."""
    import subprocess
    pkg = os.path.join(ROOT, "chroma-subsampling-image-compressor_amd")
    lib = os.path.join(pkg, "libcsic_hip.so")
    srcs = [os.path.join(pkg, "csrc", f) for f in os.listdir(os.path.join(pkg, "csrc"))
            if f.endswith((".hip", ".cpp", ".h"))] + [os.path.join(ROOT, "include", "csic.h")]
    if not os.path.exists(lib) or os.path.getmtime(lib) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", os.path.join(pkg, "csrc"), "-s"])
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])


_ensure_native_built()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as fh:
        return json.load(fh)


def load_png_rgb(path):
    """Decode a PNG to (H, W, 3) uint8: straight 8-bit samples, alpha dropped, gAMA/cHRM not applied
    (equivalent to scrimage's pixel.red()/green()/blue())."""
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGB"), dtype=np.uint8)


@pytest.fixture(scope="session")
def input_images(manifest):
    return {k: load_png_rgb(os.path.join(GOLDEN, v["file"])) for k, v in manifest["inputs"].items()}


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as orc
    orc.build()
    return orc
