#!/usr/bin/env python3
"""tools/srchash.py -- one sha256 over every source file that determines what the HIP kernels do and which one
a plan selects (csrc/*.hip, *.cpp, *.h and include/csic.h), in a fixed order.

tools/pmc_traffic.py stores it next to each counter-derived HBM byte count in profiles/pmc_traffic.json;
bench.py recomputes it and reports `roofline.traffic` only when the two agree -- so a kernel change can never
keep reporting last round's counter bytes.  `python tools/srchash.py` prints it."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_sha256(root: str = ROOT) -> str:
    csrc = os.path.join(root, "chroma-subsampling-image-compressor_amd", "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".cpp", ".h")))
    files.append(os.path.join(root, "include", "csic.h"))
    h = hashlib.sha256()
    for path in files:
        h.update(os.path.basename(path).encode() + b"\0")
        with open(path, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    return h.hexdigest()


if __name__ == "__main__":
    print(kernel_source_sha256())
