#!/bin/bash
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/overlap2
mkdir -p "$OUT"
cd /tmp
timeout -k 10 150 $ROOT/tools/ubench_overlap cfg5 30 256 > "$OUT/ub_cfg5_t256.log" 2>&1 || exit 1
timeout -k 10 150 $ROOT/tools/ubench_overlap cfg5 30 128 > "$OUT/ub_cfg5_t128.log" 2>&1 || exit 1
timeout -k 10 150 $ROOT/tools/ubench_overlap stripe8 30 256 > "$OUT/ub_stripe8_t256.log" 2>&1 || exit 1
HIP_FORCE_DEV_KERNARG=1 timeout -k 10 150 $ROOT/tools/ubench_overlap cfg5 30 256 > "$OUT/ub_cfg5_t256_devkernarg1.log" 2>&1 || exit 1
HIP_FORCE_DEV_KERNARG=0 timeout -k 10 150 $ROOT/tools/ubench_overlap cfg5 30 256 > "$OUT/ub_cfg5_t256_devkernarg0.log" 2>&1 || exit 1
GPU_MAX_HW_QUEUES=8 timeout -k 10 150 $ROOT/tools/ubench_overlap cfg5 30 256 > "$OUT/ub_cfg5_t256_hwq8.log" 2>&1 || exit 1
cd "$ROOT"
grep -h "mchain\|eager1 \|cap-chain\|batched\|^cfg5\|^stripe" $OUT/*.log
