#!/bin/bash
# tools/profile_counters.sh TAG -- SQ and TCC hardware counters of the headline kernel (separate --pmc passes, no
# tracing domains), on the GPU box.  Outputs: gpurun_out/prof_TAG_counters/; condensed by tools/pmc_counters.py.
set -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${TAG}_counters
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="python3 $ROOT/bench.py --batch-frames 1 --steps 300 --warmup 50 --prewarm-ms 50 --no-cpu-baseline"
rocprofv3 -L > "$OUT/counters_available.txt" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY --output-format csv -d "$OUT/sq1" -o sq1 -- $BENCH > "$OUT/sq1.json" 2> "$OUT/sq1.err"
rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM --output-format csv -d "$OUT/sq2" -o sq2 -- $BENCH > "$OUT/sq2.json" 2> "$OUT/sq2.err"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d "$OUT/tcc" -o tcc -- $BENCH > "$OUT/tcc.json" 2> "$OUT/tcc.err"
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d "$OUT/grbm" -o grbm -- $BENCH > "$OUT/grbm.json" 2> "$OUT/grbm.err"
cd "$ROOT"
ls -la "$OUT" | tail -12
for f in sq1 sq2 tcc grbm; do tail -2 "$OUT/$f.err"; done
