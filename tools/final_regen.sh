#!/bin/bash
# tools/final_regen.sh TAG measure|bench -- on the GPU box: everything whose committed copy is keyed on the source hash or quotes
# the final build, in two calls:
#   measure : GPU tests, smoke, the host-I/O stage, rocprofv3 + PMC passes (10 configs); then, in the build container,
#             tools/collect_artifacts.py TAG (writes profiles/pmc_traffic.json for the new hash)
#   bench   : the driver's bench command (finds that traffic entry), five repeats of it and the --no-sustained kernel trace;
#             then tools/collect_artifacts.py TAG and tools/collect_final.sh TAG.
set -o pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
if [ "${2:-measure}" = measure ]; then
python -m pytest tests -m gpu -q > "$OUT/gpu_tests.log" 2>&1 || { tail -20 "$OUT/gpu_tests.log"; exit 1; }
tail -1 "$OUT/gpu_tests.log"
python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1 || { tail -5 "$OUT/smoke.log"; exit 1; }
bash tools/artifacts.sh host $TAG > "$OUT/host_stage.log" 2>&1 || { tail -5 "$OUT/host_stage.log"; exit 1; }
echo "host stage done"
bash tools/artifacts.sh profile $TAG > "$OUT/profile_stage.log" 2>&1 || { tail -5 "$OUT/profile_stage.log"; exit 1; }
bash tools/artifacts.sh profile_rows $TAG >> "$OUT/profile_stage.log" 2>&1 || { tail -5 "$OUT/profile_stage.log"; exit 1; }
echo "profiles done"
exit 0
fi
python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err" || { tail -5 "$OUT/bench.err"; exit 1; }
: > "$OUT/bench_repeat.jsonl"
for i in 1 2 3 4 5; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline >> "$OUT/bench_repeat.jsonl" 2>> "$OUT/bench.err" || exit 1; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_ns" -o trace -- python3 "$ROOT/bench.py" --no-sustained > "$OUT/trace_no_sustained_bench.json" 2> "$OUT/trace_ns.err" || { tail -5 "$OUT/trace_ns.err"; exit 1; }
cd "$ROOT"
python - "$OUT" <<'PY'
import json, sys
r = json.load(open(sys.argv[1] + "/bench.json"))
print("bench:", r["value"], r["ms_per_step"], r["roofline"]["frac"], r["roofline"]["traffic"], r.get("sustained", {}).get("roofline_frac"))
PY
