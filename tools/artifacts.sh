#!/bin/bash
# tools/artifacts.sh STAGE [TAG] -- regenerates the committed measurement artefacts of a round with the CURRENT
# binary, on the GPU box (via gpurun); outputs under gpurun_out/<TAG>/, copied to profiles/ by tools/collect_artifacts.py.
#   bench    : bench.py for every BASELINE config + extras -> bench_all_configs.jsonl
#   profile  : rocprofv3 kernel-trace/stats + PMC passes for cfg4, cfg5, 8k_444_f1, 8k_420_f1 (tools/profile.sh)
#   small    : tools/small_launch.py, tools/ubench_overlap, tools/ubench_aql, frame-graph kernel traces
#   host     : tools/host_io.py
#   sweep    : tools/sweep.py, tools/ubench
#   stripes  : bench.py --stripe-of 2/4/8 -> bench_stripe_of.jsonl
set -o pipefail
STAGE=${1:-bench}
TAG=${2:-r04}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
export TMPDIR=/tmp
B="python bench.py --no-cpu-baseline --direct"
case "$STAGE" in
bench)
  J=$OUT/bench_all_configs.jsonl; : > "$J"
  python bench.py --cpu-budget 5 --direct --one-launch >> "$J" 2> "$OUT/bench.err" || exit 1
  for c in cfg5 8k_444_f1 8k_420_f1 avg_8k_420_sf2 avg_4k_420_sf4 cfg2 cfg3; do $B --config $c >> "$J" 2>> "$OUT/bench.err" || exit 1; done
  BN="python bench.py --no-cpu-baseline"
  $BN --config cfg5 --per-frame-graph --issue hip --graph-branches 1 --steps 1000 --warmup 200 >> "$J" 2>> "$OUT/bench.err" || exit 1
  $BN --config cfg5 --per-frame-graph --issue hip --steps 1000 --warmup 200 >> "$J" 2>> "$OUT/bench.err" || exit 1
  $BN --config cfg5 --per-frame-graph --issue direct --steps 1000 --warmup 200 >> "$J" 2>> "$OUT/bench.err" || exit 1
  $BN --config cfg5 --per-frame-graph --issue direct --direct-queues 4 --steps 1000 --warmup 200 >> "$J" 2>> "$OUT/bench.err" || exit 1
  $BN --config cfg5 --per-frame-graph --issue fused --steps 1000 --warmup 200 >> "$J" 2>> "$OUT/bench.err" || exit 1
  $B --config cfg2 --frames-per-step 4096 >> "$J" 2>> "$OUT/bench.err" || exit 1
  $B --config cfg3 --frames-per-step 1024 >> "$J" 2>> "$OUT/bench.err" || exit 1
  $B --config sq1000 --order scq --frames-per-step 1024 >> "$J" 2>> "$OUT/bench.err" || exit 1
  $B --config sq1024 --order scq --frames-per-step 1024 >> "$J" 2>> "$OUT/bench.err" || exit 1
  $B --config sq1000 --frames-per-step 1024 >> "$J" 2>> "$OUT/bench.err" || exit 1      # the SAME 1000x1000 frames, chroma before spatial: k_decflat
  # round 4: the planar output format (forward kernels; each line carries its `reconstruct` object) ...
  for c in planar_8k_420_f1 planar_8k_420_f1_avg planar_cfg4; do $BN --config $c >> "$J" 2>> "$OUT/bench.err" || exit 1; done
  # ... and the AVG tile kernel on frames it used to refuse, each next to its nearest whole-tile neighbour (256 frames per launch)
  for c in avg_1366x768_sf4 avg_1368x768_sf4 avg_1001_sf8 avg_1000_sf8 avg_1922x1082_sf2 avg_1920x1080_sf2; do
    $BN --config $c --frames-per-step 256 >> "$J" 2>> "$OUT/bench.err" || exit 1; done
  $BN --config avg_1366x768_sf4 --frames-per-step 256 --variant 8 >> "$J" 2>> "$OUT/bench.err" || exit 1   # A/B: rounds 1-3's rule (k_avg_generic for anything but whole tiles)
  $BN --config avg_1001_sf8 --frames-per-step 256 --variant 8 >> "$J" 2>> "$OUT/bench.err" || exit 1
  # planar + AVG with a decimating factor: k_avg's tile body with the planar sink, and (variant 9) the one-position-per-lane kernel it replaced
  $BN --config planar_cfg4_avg >> "$J" 2>> "$OUT/bench.err" || exit 1
  $BN --config planar_cfg4_avg --variant 9 >> "$J" 2>> "$OUT/bench.err" || exit 1
  wc -l "$J"
  ;;
profile)
  for c in cfg4 cfg5 8k_444_f1 8k_420_f1 planar_8k_420_f1 planar_8k_420_f1_avg planar_cfg4_avg avg_8k_420_sf2; do
    bash tools/profile.sh $TAG $c > "$OUT/profile_$c.log" 2>&1 || { tail -5 "$OUT/profile_$c.log"; exit 1; }
    echo "profiled $c"
  done
  ;;
profile_rows)
  # the weakest shapes: 1000-pixel rows in both order classes next to the aligned 1024 shape
  bash tools/profile.sh $TAG sq1000_csq --config sq1000 --frames-per-step 1024 > "$OUT/profile_sq1000_csq.log" 2>&1 || { tail -5 "$OUT/profile_sq1000_csq.log"; exit 1; }
  bash tools/profile.sh $TAG sq1000_csq_kdec --config sq1000 --frames-per-step 1024 --variant 5 > "$OUT/profile_sq1000_csq_kdec.log" 2>&1 || { tail -5 "$OUT/profile_sq1000_csq_kdec.log"; exit 1; }
  bash tools/profile.sh $TAG sq1000_scq_kgeneric --config sq1000 --order scq --frames-per-step 1024 --variant 7 > "$OUT/profile_sq1000_scq_kgeneric.log" 2>&1 || { tail -5 "$OUT/profile_sq1000_scq_kgeneric.log"; exit 1; }
  bash tools/profile.sh $TAG sq1000_scq --config sq1000 --order scq --frames-per-step 1024 > "$OUT/profile_sq1000_scq.log" 2>&1 || { tail -5 "$OUT/profile_sq1000_scq.log"; exit 1; }
  bash tools/profile.sh $TAG sq1024_scq --config sq1024 --order scq --frames-per-step 1024 > "$OUT/profile_sq1024_scq.log" 2>&1 || { tail -5 "$OUT/profile_sq1024_scq.log"; exit 1; }
  bash tools/profile.sh $TAG sq1024_csq --config sq1024 --frames-per-step 1024 > "$OUT/profile_sq1024_csq.log" 2>&1 || { tail -5 "$OUT/profile_sq1024_csq.log"; exit 1; }
  echo "profiled sq1000/sq1024 rows"
  ;;
small)
  rm -f "$OUT/small_launch.jsonl"
  timeout -k 10 600 python tools/small_launch.py --threads 0,128 --out "$OUT/small_launch.jsonl" > "$OUT/small_launch.log" 2>&1 || exit 1
  ( cd /tmp
    timeout -k 10 150 $ROOT/tools/ubench_overlap cfg5 30 256 > "$OUT/ubench_overlap_cfg5.log" 2>&1 || exit 1
    timeout -k 10 150 $ROOT/tools/ubench_overlap stripe8 30 256 > "$OUT/ubench_overlap_stripe8.log" 2>&1 || exit 1
    timeout -k 10 150 $ROOT/tools/ubench_aql cfg5 20 256 > "$OUT/ubench_aql_cfg5.log" 2>&1 || exit 1
    timeout -k 10 150 $ROOT/tools/ubench_aql stripe8 20 256 > "$OUT/ubench_aql_stripe8.log" 2>&1 || exit 1
    : > "$OUT/cfg5_graph_trace_summary.jsonl"
    for cfg in "hip 1" "hip 4" "direct 4"; do
      set -- $cfg
      d="$OUT/cfg5_graph_trace_$1_b$2"
      rocprofv3 --kernel-trace --stats --output-format csv -d "$d" -o trace -- python3 $ROOT/tools/graph_trace.py run 256 $2 $1 > "$d.log" 2>&1 || exit 1
      python3 $ROOT/tools/graph_trace.py analyze "$d" >> "$OUT/cfg5_graph_trace_summary.jsonl" || exit 1
    done ) || exit 1
  cat "$OUT/cfg5_graph_trace_summary.jsonl"
  ;;
host)
  timeout -k 10 900 python tools/host_io.py "$OUT/host_io.json" > "$OUT/host_io.log" 2>&1 || { tail -5 "$OUT/host_io.log"; exit 1; }
  tail -40 "$OUT/host_io.log"
  ;;
sweep)
  timeout -k 10 900 python tools/sweep.py > "$OUT/sweep.md" 2> "$OUT/sweep.err" || exit 1
  tail -3 "$OUT/sweep.md"
  ( cd /tmp && timeout -k 10 600 $ROOT/tools/ubench > "$OUT/ubench.log" 2>&1 ) || exit 1
  tail -3 "$OUT/ubench.log"
  ;;
stripes)
  # what ONE rank of the N-GPU strong-scaling run does, measured alone on this GPU (bench.py --stripe-of N)
  J=$OUT/bench_stripe_of.jsonl; : > "$J"
  # with the driver's own flags (K = 20 steps, W = 5), N = 1 first, then one rank's share of N = 2, 4, 8
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --one-launch >> "$J" 2>> "$OUT/bench.err" || exit 1
  for n in 2 4 8; do python bench.py --stripe-of $n --steps 20 --warmup 5 --no-cpu-baseline >> "$J" 2>> "$OUT/bench.err" || exit 1; done
  python - "$J" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    r = json.loads(l)
    print(r["config"]["stripe_rows_per_gpu"], r["config"]["issue"], r["ms_per_step"], r["ms_per_launch"], r["roofline"]["frac"],
          {k: r[k]["ms_per_launch"] for k in ("hip_streams", "direct_dispatch", "serial_launches", "direct_host_ordered") if k in r},
          (r.get("direct_dispatch") or {}).get("host_ordered", {}).get("ms_per_launch"))
PY
  ;;
*) echo "unknown stage $STAGE"; exit 2 ;;
esac
