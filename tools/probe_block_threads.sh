#!/bin/bash
# tools/probe_block_threads.sh -- 256- vs 128- vs 64-thread blocks (CSIC_TUNE_BLOCK_THREADS) on the big bench configs, serial
# launches, three interleaved repeats of the headline.  One bench.py JSON line per run -> stdout.
cd "${GRAFT_REPO_ROOT:-$(pwd)}"
for rep in 1 2 3; do
  for thr in 0 128 64; do
    python bench.py --no-cpu-baseline --block-threads $thr 2>/dev/null || exit 1
  done
done
for cfg in cfg5 8k_444_f1 8k_420_f1 avg_8k_420_sf2; do
  for thr in 0 128 64; do
    python bench.py --no-cpu-baseline --config $cfg --block-threads $thr 2>/dev/null || exit 1
  done
done
