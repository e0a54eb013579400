#!/bin/bash
# tools/busy_streams.sh [TAG] -- cfg 5 as 64 per-frame launches per step, every launch backend, beside K = 0, 1, 2 OTHER
# HIP streams that run real kernels (64 MiB copies back to back) for the whole timed region.
# The busy streams take HBM bandwidth from everybody; the question is which backend loses more than its share.
set -o pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
J=$OUT/busy_streams.jsonl; : > "$J"
for k in 0 1 2; do
  for mode in "--issue hip" "--issue direct" "--issue direct --direct-queues 4" "--issue fused"; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --config cfg5 --per-frame-graph $mode --steps 400 --warmup 100 --busy-streams $k \
        >> "$J" 2>> "$OUT/busy_streams.err" || { tail -3 "$OUT/busy_streams.err"; exit 1; }
  done
  timeout -k 10 200 python bench.py --no-cpu-baseline --config cfg5 --steps 400 --warmup 100 --busy-streams $k \
      >> "$J" 2>> "$OUT/busy_streams.err" || { tail -3 "$OUT/busy_streams.err"; exit 1; }
done
python - "$J" > "$OUT/busy_streams.md" <<'PY'
import json, sys
rows = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")]
print("| issue | launch queues/chains | busy streams | ms per 64-frame step (HIP events) | % of 8 TB/s | busy streams outlasted the timed region | host-ordered % |")
print("|---|---|---|---|---|---|---|")
for r in rows:
    c, ro = r["config"], r["roofline"]
    b = r.get("busy_streams") or {}
    ho = (r.get("direct_host_ordered") or {}).get("roofline_frac_rank0")
    what = c["issue"] if c["launches_per_step"] > 1 else "one batched launch (csic_process_batch_device)"
    q = c["launch"].split(" user-mode")[0].split("on ")[-1] if "user-mode" in c["launch"] else (c["launch"].split(" hipGraph")[0].split(": ")[-1] if "hipGraph" in c["launch"] else "1")
    print(f"| {what} | {q} | {b.get('streams', 0)} | {ro['kernel_ms_avg'] * c['launches_per_step']:.4f} | {100 * ro['frac']:.1f} | "
          f"{b.get('outlasted_timed_region', '-')} | {'' if ho is None else round(100 * ho, 1)} |")
PY
cat "$OUT/busy_streams.md"
