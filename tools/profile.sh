#!/bin/bash
# tools/profile.sh TAG [KEY [bench flags...]] -- run on the GPU box (via gpurun): rocprofv3 kernel trace + stats of bench.py
# and the two PMC passes for HBM traffic (separate passes: FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2).
# Outputs land in gpurun_out/prof_TAG/ (KEY cfg4) or prof_TAG_KEY/; tools/pmc_traffic.py condenses them into profiles/.
# KEY is a bench.py --config name, or any name followed by the bench flags it stands for, e.g.
#   tools/profile.sh r03 sq1000_csq --config sq1000 --frames-per-step 1024
set -o pipefail
TAG=${1:-r01}
CONFIG=${2:-cfg4}            # key of the outputs (and of the profiles/pmc_traffic.json entry)
shift; shift
FLAGS="$*"
[ -n "$FLAGS" ] || FLAGS="--config $CONFIG"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
if [ "$CONFIG" != "cfg4" ]; then OUT=${OUT}_$CONFIG; fi
mkdir -p "$OUT"
export TMPDIR=/tmp
# the sources this binary was built from, hashed HERE (on the box that profiles), not where the results are collected
python3 "$ROOT/tools/srchash.py" > "$OUT/source_sha256.txt"
cd /tmp
# the kernel trace runs bench.py exactly as the driver does (defaults); the PMC passes skip the CPU baseline
BENCH="python3 $ROOT/bench.py $FLAGS --no-cpu-baseline --batch-frames 1 --steps 600 --warmup 100"
if [ "$CONFIG" = "cfg4" ]; then TRACE="python3 $ROOT/bench.py"; else TRACE="$BENCH"; fi
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $TRACE > "$OUT/trace_bench.json" 2> "$OUT/trace.err" &&
{ find "$OUT/trace" -name '*kernel_trace.csv' -size +8M -delete; true; } &&   # the --stats summary is what is kept (the default run's `sustained` leg is ~200k launches)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o fetch -- $BENCH > "$OUT/fetch_bench.json" 2> "$OUT/fetch.err" &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o write -- $BENCH > "$OUT/write_bench.json" 2> "$OUT/write.err" &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/calib_fetch" -o fetch -- python3 $ROOT/tools/pmc_calib.py > "$OUT/calib_fetch.log" 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/calib_write" -o write -- python3 $ROOT/tools/pmc_calib.py > "$OUT/calib_write.log" 2>&1
rc=$?
cd "$ROOT"
find "$OUT" -name '*.csv' | head -50
ls -la "$OUT"
exit $rc
