#!/bin/bash
# tools/overlap_experiments.sh -- run on the GPU box: (1) ubench_overlap on cfg5 frames and 1/8 stripes,
# (2) kernel-trace timelines of the cfg 5 frame graph, (3) HIP graph-queue knobs.  Output: gpurun_out/overlap/
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/overlap
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
for thr in 256 128 64; do
  timeout -k 10 120 $ROOT/tools/ubench_overlap cfg5 30 $thr > "$OUT/ub_cfg5_t$thr.log" 2>&1 || exit 1
done
timeout -k 10 120 $ROOT/tools/ubench_overlap stripe8 30 256 > "$OUT/ub_stripe8_t256.log" 2>&1 || exit 1
timeout -k 10 120 $ROOT/tools/ubench_overlap stripe8 30 128 > "$OUT/ub_stripe8_t128.log" 2>&1 || exit 1
for cfg in "256 1" "128 64" "128 8" "256 64"; do
  set -- $cfg
  d="$OUT/trace_t$1_b$2"
  rocprofv3 --kernel-trace --output-format csv -d "$d" -o trace -- python3 $ROOT/tools/graph_trace.py run $1 $2 > "$d.log" 2>&1 || exit 1
  python3 $ROOT/tools/graph_trace.py analyze "$d" >> "$OUT/trace_summary.jsonl" || exit 1
done
for q in 1 2 8 16; do
  DEBUG_HIP_FORCE_GRAPH_QUEUES=$q timeout -k 10 120 $ROOT/tools/ubench_overlap cfg5 20 128 > "$OUT/ub_cfg5_t128_gq$q.log" 2>&1 || exit 1
done
GPU_MAX_HW_QUEUES=8 timeout -k 10 120 $ROOT/tools/ubench_overlap cfg5 20 128 > "$OUT/ub_cfg5_t128_hwq8.log" 2>&1 || exit 1
GPU_MAX_HW_QUEUES=8 DEBUG_HIP_FORCE_GRAPH_QUEUES=8 timeout -k 10 120 $ROOT/tools/ubench_overlap cfg5 20 128 > "$OUT/ub_cfg5_t128_hwq8_gq8.log" 2>&1 || exit 1
cd "$ROOT"
tail -n 30 "$OUT/ub_cfg5_t128.log"
cat "$OUT/trace_summary.jsonl"
