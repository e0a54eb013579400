#!/usr/bin/env python3
"""tools/probe_stream_launch.py -- cost of one stream-ordered DIRECT launch (gate + awaits through HIP signal memory)
against the host-ordered submit/wait of the same graph, for 1/2/4 queues and graphs of 1..64 cfg 5 frames."""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import csic_amd as csic
N = csic._native
W, H, n = 3840, 2160, 64
plan = csic.Plan(csic.make_c_params(W, H, 2, 0, 3, 3, 2, 4, (3, 1, 2)), 0)
ipx, opx = W * H, plan.out_width * plan.out_height
d_in = torch.empty(n * ipx, dtype=torch.int32, device="cuda:0")
d_out = torch.empty(n * opx, dtype=torch.int32, device="cuda:0")
st = torch.cuda.current_stream()
N.check(N.lib().csic_synth_frame_device(C.c_void_p(d_in.data_ptr()), d_in.numel(), 0, 20250629, C.c_void_p(st.cuda_stream)))
torch.cuda.synchronize()
for nf in (4, 64):
    for q in (1, 2, 3, 4):
        if q > nf:
            continue
        g = csic.FrameGraph(plan, [d_in[k * ipx:(k + 1) * ipx] for k in range(nf)], [d_out[k * opx:(k + 1) * opx] for k in range(nf)],
                            branches=q, backend="direct")
        for _ in range(20):
            g.launch(st)
        torch.cuda.synchronize()
        reps = 200
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(st)
        for _ in range(reps):
            g.launch(st)
        e1.record(st)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        us_stream = e0.elapsed_time(e1) * 1e3 / reps
        t3 = time.perf_counter()
        for _ in range(reps):
            g.submit()
        g.wait()
        us_host = (time.perf_counter() - t3) * 1e6 / reps
        print(json.dumps({"frames": nf, "queues": q, "stream_ordered_launch_us": round(us_stream, 2), "host_enqueue_us": round((t1 - t0) * 1e6 / reps, 2),
                          "wall_us": round((t2 - t0) * 1e6 / reps, 2), "host_ordered_submit_us": round(us_host, 2)}), flush=True)
        g.close()
