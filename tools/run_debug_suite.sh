#!/bin/bash
# tools/run_debug_suite.sh [OUT] -- the whole `-m gpu` suite with the range-checking build loaded instead of the product library
# (CSIC_LIB; see tests/test_gpu_debug_build.py).  The C / C++ hosts under tests/cpp link libcsic_hip.so by name and are left to the
# ordinary run; everything that goes through the Python host -- every parity test -- runs under the checks.  On the GPU box.
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-$ROOT/gpurun_out/r04/gpu_tests_debug.log}
mkdir -p "$(dirname "$OUT")"
cd "$ROOT"
[ -f chroma-subsampling-image-compressor_amd/libcsic_hip_debug.so ] || make -C chroma-subsampling-image-compressor_amd/csrc -s debug || exit 1
{
  echo "# CSIC_LIB=chroma-subsampling-image-compressor_amd/libcsic_hip_debug.so python -m pytest tests -q -m gpu  (range-checked kernels, -DCSIC_DEBUG)"
  CSIC_LIB=$ROOT/chroma-subsampling-image-compressor_amd/libcsic_hip_debug.so timeout -k 10 1100 python -m pytest tests -q -m gpu \
      --deselect tests/test_gpu_debug_build.py::test_the_product_library_is_not_the_debug_build -p no:cacheprovider
} > "$OUT" 2>&1
rc=$?
tail -5 "$OUT"
exit $rc
