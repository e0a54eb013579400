#!/bin/bash
# tools/probe_pitch_headline.sh -- the headline step (cfg4, one 8192x8192 frame per launch, serial launches) with packed rows
# and with rows padded in HBM (bench.py --pitch-pad), 256- and 128-thread blocks.  One JSON line each -> stdout.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
for pad in 0 64 256 512 1024; do
  for thr in 0 128; do
    python bench.py --no-cpu-baseline --pitch-pad $pad --block-threads $thr --batch-frames 1 --steps 2000 --warmup 300 2>/dev/null || exit 1
  done
done
