#!/bin/bash
# tools/rccl_ws1.sh [TAG] -- on the 1-GPU box: bring up a REAL RCCL communicator (backend nccl, world size 1) under the same
# launcher the driver uses for N > 1, and run bench.py's barriers and device-tensor all-reduces through it while the launch
# engine works beside it.  The launcher starts before anything touches the GPU.
#   line 1: rank 0's stripe of the 8-way strong split (8192x1024, issue=direct), process group formed
#   line 2: the same without a process group (the 4.07 us reference)
#   line 3: N = 1 headline launches with the communicator alive
set -o pipefail
TAG=${1:-r03}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd "$ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
LOG=$OUT/${TAG}_rccl_ws1.log
: > "$LOG"
run() { echo "## $*" >> "$LOG"; timeout -k 10 300 "$@" >> "$LOG" 2>&1; }
L="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port"
run $L 29531 bench.py --gpus 1 --force-pg --backend nccl --stripe-of 8 --steps 20 --warmup 5 --no-cpu-baseline || { tail -5 "$LOG"; exit 1; }
run python bench.py --gpus 1 --stripe-of 8 --steps 20 --warmup 5 --no-cpu-baseline || { tail -5 "$LOG"; exit 1; }
run $L 29532 bench.py --gpus 1 --force-pg --backend nccl --steps 20 --warmup 5 --no-cpu-baseline || { tail -5 "$LOG"; exit 1; }
python - "$LOG" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        r = json.loads(l)
        c = r["config"]
        print(c["backend"], "| formed", c["world_size_formed"], "| issue", c["issue"], "| stripe rows", c["stripe_rows_per_gpu"],
              "| ms/launch", r["ms_per_launch"], "| frac", r["roofline"]["frac"], "|", c.get("issue_note"))
PY
