#!/usr/bin/env python3
"""tools/marker_trace.py -- cfg 5 from files with the library's roctx ranges on (CSIC_ROCTX=1), for
    CSIC_ROCTX=1 rocprofv3 --marker-trace --kernel-trace --stats -d <dir> -o trace -- python3 tools/marker_trace.py run [n]
and `python3 tools/marker_trace.py summarize <dir> <out.md>` condenses the profiler's marker / kernel statistics into the table
that is committed under profiles/ (the analogue of the reference's WriteVcdAnnotation, ImageCompressorTopApp.scala:67: a
timeline of what the harness does around the DUT).  `run` writes n (default 16) 4K PNGs to a temporary directory and pushes them
through ImageCompressionApp.processImages (csic_process_png_files: decoder / encoder pools) and through the staged
FramePipeline (csic_pipeline_*: enqueue_h2d / launch_kernel / enqueue_d2h / wait_gpu ranges)."""
import csv
import glob
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(n):
    import numpy as np
    import csic_amd as csic
    PS = csic.ProcessingStep
    M = csic.ImageProcessorModel
    W, H = 3840, 2160
    tmp = tempfile.mkdtemp(prefix="csic_marker_")
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:H, 0:W]
    ins = []
    for k in range(n):
        noise = rng.integers(0, 8, (H, W, 3), dtype=np.uint32)
        r = (((xx + 3 * k) >> 2) & 255) ^ noise[..., 0]
        g = (((yy + 5 * k) >> 1) & 255) ^ noise[..., 1]
        b = (((xx + yy) >> 3) & 255) ^ noise[..., 2]
        path = os.path.join(tmp, f"f{k:02d}.png")
        M.writeImage(csic.Image((0xFF000000 | (r << 16) | (g << 8) | b).astype(np.uint32)), path)
        ins.append(path)
    args = (2, 0, 3, 3, 2, 4, PS.ChromaSubsampling, PS.SpatialSampling, PS.ColorQuantization)
    st = csic.ImageCompressionApp.processImages(ins, [os.path.join(tmp, "out", f"o{k:02d}.png") for k in range(n)], *args)
    print("pooled:", {k: st[k] for k in ("frames", "wall_s", "decode_threads", "encode_threads", "slots")})
    # the staged host-frame pipeline: hipMemcpyAsync H2D -> kernel -> D2H per slot
    top = csic.ImageCompressorTop(W, H, *args)
    with csic.FramePipeline(top.plan(), depth=3, zero_copy=False) as pipe:
        frames = [M.readImage(p).argb for p in ins[:4]]
        done = 0
        for k in range(12):
            if pipe.pending == 3:
                pipe.collect(); done += 1
            pipe.acquire_input()[...] = frames[k % 4]
            pipe.submit()
        while pipe.pending:
            pipe.collect(); done += 1
    top.close()
    print("staged pipeline frames:", done)


def summarize(d, out):
    rows = []
    for pat, title in (("*marker_api_stats.csv", "roctx ranges (host threads)"), ("*kernel_stats.csv", "kernels")):
        for path in sorted(glob.glob(os.path.join(d, "**", pat), recursive=True)):
            with open(path, newline="") as fh:
                rd = list(csv.DictReader(fh))
            rows.append((title, os.path.basename(path), rd))
    with open(out, "w") as fh:
        fh.write("# cfg 5 from files under `CSIC_ROCTX=1 rocprofv3 --marker-trace --kernel-trace --stats` (tools/marker_trace.py)\n\n")
        for title, name, rd in rows:
            fh.write(f"## {title} -- {name}\n\n| name | calls | total ms | average us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
            for r in rd:
                g = lambda *ks: next((r[k] for k in ks if k in r), "")
                tot, avg, mn, mx = (float(g("TotalDurationNs", "TotalDuration(ns)") or 0), float(g("AverageNs", "Average(ns)") or 0),
                                    float(g("MinNs", "Min(ns)") or 0), float(g("MaxNs", "Max(ns)") or 0))
                fh.write(f"| `{g('Name')[:90]}` | {g('Calls')} | {tot / 1e6:.3f} | {avg / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {g('Percentage')} |\n")
            fh.write("\n")
    print(open(out).read())


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "summarize":
        summarize(sys.argv[2], sys.argv[3])
    else:
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 16)
