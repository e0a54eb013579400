#!/usr/bin/env python3
"""tools/host_io.py -- the SURVEY.md 8(d) side lines, as one JSON artefact (profiles/rNN_host_io.json): everything
that sits either side of the device-resident hot path and is therefore excluded from bench.py's `value`.

  * raw PCIe: pinned H2D / D2H copy rates of one 8192x8192 frame (torch pinned tensors, hipMemcpyAsync)
  * csic_process_host (pageable, synchronous H2D + kernel + D2H)
  * csic_pipeline_* staged (pinned staging + hipMemcpyAsync) and zero-copy (the kernel reads/writes pinned host
    memory over PCIe; dead rows never cross the bus), several depths
  * the library's PNG codec (csic_png_*, zlib): decode and encode Mpixel/s on the reference's three input images
    (tests/golden/inputs) and on one synthetic 3840x2160 frame

    python tools/host_io.py [out.json]
"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import csic_amd as csic  # noqa: E402


def best_of(fn, n=5):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts), sum(ts) / len(ts)


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "host_io.json")
    W = H = 8192
    res = {"frame": f"{W}x{H} ARGB (256 MiB in, 64 MiB out at sf=2)", "host_cores": os.cpu_count(), "pcie": {}, "host_entry_points": {},
           "png_codec": {}}
    rng = np.random.default_rng(0)
    frame = rng.integers(0, 1 << 32, W * H, dtype=np.uint32).reshape(H, W)

    # ---- raw PCIe ---------------------------------------------------------------------------------
    pin_in = torch.from_numpy(frame.view(np.int32)).pin_memory()
    d = torch.empty_like(pin_in, device="cuda:0")
    pin_out = torch.empty((H // 2, W // 2), dtype=torch.int32).pin_memory()
    d_out = torch.empty((H // 2, W // 2), dtype=torch.int32, device="cuda:0")

    def h2d():
        d.copy_(pin_in, non_blocking=True)
        torch.cuda.synchronize()

    def d2h():
        pin_out.copy_(d_out, non_blocking=True)
        torch.cuda.synchronize()
    b, _ = best_of(h2d)
    res["pcie"]["h2d_pinned_GBps"] = round(W * H * 4 / b / 1e9, 1)
    b, _ = best_of(d2h)
    res["pcie"]["d2h_pinned_GBps"] = round(pin_out.numel() * 4 / b / 1e9, 1)

    # ---- host entry points --------------------------------------------------------------------------
    pl = csic.Plan(csic.make_c_params(W, H, 2, 0, 8, 8, 8, 2, (3, 1, 2)), 0)
    full_bytes = W * H * 4 + (W // 2) * (H // 2) * 4
    live_bytes = W * (H // 2) * 4 + (W // 2) * (H // 2) * 4
    b, m = best_of(lambda: pl.process_host(frame), 4)
    res["host_entry_points"]["csic_process_host (pageable, synchronous)"] = {
        "ms_per_frame": round(m * 1e3, 2), "Mpixels_per_s": round(W * H / m / 1e6), "PCIe_GBps": round(full_bytes / m / 1e9, 1)}
    for depth, zc in ((1, False), (3, False), (1, True), (2, True), (4, True)):
        with csic.FramePipeline(pl, depth, zero_copy=zc) as pipe:
            for _ in range(depth):
                pipe.acquire_input()[...] = frame
                pipe.submit()
            while pipe.pending:
                pipe.collect()
            n = 40
            t0 = time.perf_counter()
            for _ in range(n):
                if pipe.pending == depth:
                    pipe.collect()
                pipe.acquire_input()
                pipe.submit()
            while pipe.pending:
                pipe.collect()
            dt = (time.perf_counter() - t0) / n
        bus = live_bytes if zc else full_bytes
        res["host_entry_points"][f"csic_pipeline depth {depth}, {'zero-copy kernel' if zc else 'pinned staging + hipMemcpyAsync'}"] = {
            "ms_per_frame": round(dt * 1e3, 2), "Mpixels_per_s": round(W * H / dt / 1e6), "PCIe_GBps_crossing": round(bus / dt / 1e9, 1)}
    pl.close()

    # ---- PNG codec --------------------------------------------------------------------------------
    M = csic.ImageProcessorModel
    inputs = [os.path.join(ROOT, "tests", "golden", "inputs", f) for f in sorted(os.listdir(os.path.join(ROOT, "tests", "golden", "inputs")))]
    tmp = tempfile.mkdtemp()
    synth = os.path.join(tmp, "synth_3840x2160.png")
    yy, xx = np.mgrid[0:2160, 0:3840]
    smooth = (0xFF000000 | ((xx & 255) << 16) | ((yy & 255) << 8) | (((xx + yy) >> 1) & 255)).astype(np.uint32)
    M.writeImage(csic.Image(smooth), synth)
    for path in inputs + [synth]:
        img = M.readImage(path)
        npx = img.width * img.height
        reps = max(3, min(200, int(2e7 // npx)))

        def dec():
            for _ in range(reps):
                M.readImage(path)
        outp = os.path.join(tmp, "o.png")

        def enc():
            for _ in range(reps):
                M.writeImage(img, outp)
        bd, _ = best_of(dec, 3)
        be, _ = best_of(enc, 3)
        res["png_codec"][os.path.basename(path)] = {
            "size": f"{img.width}x{img.height}", "file_bytes": os.path.getsize(path),
            "decode_Mpixels_per_s": round(npx * reps / bd / 1e6, 1), "encode_Mpixels_per_s": round(npx * reps / be / 1e6, 1)}
    res["note"] = ("none of these is bench.py's `value` (device-resident frames); 1 host thread for the codec; the reference "
                   "decodes with scrimage and feeds one pixel per simulated clock (ImageProcessorModel.scala:14-52)")
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    with open(out_path, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
