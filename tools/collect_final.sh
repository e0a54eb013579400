#!/bin/bash
# tools/collect_final.sh [TAG] -- build container, after tools/final_regen.sh + tools/collect_artifacts.py: the files the
# collector does not know by name, then BASELINE.md section 3 from the committed lines.
TAG=${1:-r03}
cd "$(dirname "$0")/.."
cp gpurun_out/$TAG/trace_ns/trace_kernel_stats.csv profiles/${TAG}_kernel_stats_no_sustained.csv
cp gpurun_out/$TAG/trace_no_sustained_bench.json profiles/${TAG}_trace_no_sustained_bench.json
git checkout -- profiles/${TAG}_fuzz.log 2>/dev/null      # gpurun_out's fuzz.log is the LAST run; the committed one is the first (profiles/README.md)
python tools/render_baseline_section3.py
python tools/srchash.py | cut -c1-12
