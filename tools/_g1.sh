set -o pipefail
mkdir -p gpurun_out/r04
python -m pytest tests -x -q -m gpu > gpurun_out/r04/gpu_tests_1.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r04/gpu_tests_1.log
( cd /tmp && timeout -k 10 120 $GRAFT_REPO_ROOT/tools/ubench_unaligned > $GRAFT_REPO_ROOT/gpurun_out/r04/ubench_unaligned.log 2>&1 ); echo "unaligned rc=$?"; cat gpurun_out/r04/ubench_unaligned.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04/bench_1.json 2> gpurun_out/r04/bench_1.err; echo "bench rc=$?"; tail -c 3000 gpurun_out/r04/bench_1.json
for c in avg_1366x768_sf4 avg_1368x768_sf4 avg_1001_sf8 avg_1000_sf8 avg_1922x1082_sf2 avg_1920x1080_sf2; do
  python bench.py --no-cpu-baseline --config $c --frames-per-step 256 --no-verify >> gpurun_out/r04/avg_before.jsonl 2>> gpurun_out/r04/avg_before.err || echo "fail $c"
done
python - <<'PY'
import json
for l in open('gpurun_out/r04/avg_before.jsonl'):
    r = json.loads(l); print(r['config']['workload'][:40], r['config']['kernel'], r['roofline']['frac'], r['ms_per_launch'])
PY
