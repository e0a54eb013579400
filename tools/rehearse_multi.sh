#!/bin/bash
# tools/rehearse_multi.sh [TAG] -- on the 1-GPU box: the N > 1 launcher path, twice, two ranks sharing GPU 0
# over gloo.  Logs: gpurun_out/<TAG>_multi_rehearsal.{halo,bench}.log (copied to profiles/).
#   (a) StripedImageCompressorTop, real HIP Plan, CUDA rows, aligned stripes + unaligned row_splits (_exchange_halo)
#   (b) bench.py --gpus $NP --backend gloo: strong split of ONE 8192x8192 frame as `value`, weak beside it
set -o pipefail
TAG=${1:-r02}
NP=${2:-2}               # ranks sharing GPU 0 (4 rehearses the N = 4 path: issue = direct as the `value` path)
SUF=""; [ "$NP" != "2" ] && SUF="_np$NP"
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd "$ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
if [ "$NP" = "2" ]; then          # (the halo cases are written for two ranks)
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $NP --master-addr 127.0.0.1 --master-port 29517 \
    tools/rehearse_multi.py halo > "$OUT/${TAG}_multi_rehearsal${SUF}.halo.log" 2>&1
rc=$?
tail -n 4 "$OUT/${TAG}_multi_rehearsal${SUF}.halo.log"
[ $rc -eq 0 ] || exit $rc
fi
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $NP --master-addr 127.0.0.1 --master-port 29518 \
    bench.py --gpus $NP --backend gloo --steps 20 --warmup 5 > "$OUT/${TAG}_multi_rehearsal${SUF}.bench.log" 2>&1
rc=$?
tail -n 2 "$OUT/${TAG}_multi_rehearsal${SUF}.bench.log"
[ $rc -eq 0 ] || exit $rc
# (c) the same run with the launch engine failing on the LAST rank only, inside the timed region (CSIC_BENCH_INJECT_FAIL, the
#     hook tests/test_bench_protocol.py drives on the CPU): every rank must finish, with the next issue mode and a note
ISSUE=direct; [ "$NP" = "2" ] && ISSUE=hip
CSIC_BENCH_INJECT_FAIL="rank=$((NP-1)),issue=$ISSUE,phase=timed" timeout -k 10 400 python -m torch.distributed.run --nnodes=1 \
    --nproc-per-node $NP --master-addr 127.0.0.1 --master-port 29520 \
    bench.py --gpus $NP --backend gloo --steps 20 --warmup 5 --no-side > "$OUT/${TAG}_multi_rehearsal${SUF}.inject.log" 2>&1
rc=$?
tail -n 1 "$OUT/${TAG}_multi_rehearsal${SUF}.inject.log" | python -c "
import json, sys
l = json.loads(sys.stdin.readline())
print('inject:', l['config']['issue'], '|', l['config'].get('issue_note'))
assert l['config']['issue'] != '$ISSUE' and 'failed on 1 rank' in l['config']['issue_note']"
rc2=$?
[ $rc -eq 0 ] && exit $rc2
exit $rc
