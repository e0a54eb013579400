import ctypes as C, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, csic_amd as csic
N = csic._native
for (W, H, a, b, f) in [(1000, 1000, 1, 1, 4), (1000, 1000, 1, 1, 8), (1000, 1000, 2, 2, 8), (1000, 1000, 2, 0, 8), (1366, 768, 2, 0, 2), (1366, 768, 2, 0, 4), (1918, 1080, 2, 0, 4), (3838, 2160, 2, 0, 4), (8190, 4096, 2, 0, 4)]:
    cp = csic.make_c_params(W, H, a, b, 8, 8, 8, f, (1, 3, 2))
    pl = csic.Plan(cp, 0)
    alg = pl.algorithmic_bytes
    nfr = max(1, min(4096, (768 << 20) // alg))
    ring = 4
    ins = [torch.empty(nfr * W * H, dtype=torch.int32, device="cuda:0") for _ in range(ring)]
    outs = [torch.empty(nfr * pl.out_width * pl.out_height, dtype=torch.int32, device="cuda:0") for _ in range(ring)]
    st = torch.cuda.current_stream(); sh = C.c_void_p(st.cuda_stream)
    for t in ins:
        N.check(N.lib().csic_synth_frame_device(C.c_void_p(t.data_ptr()), t.numel(), 0, 1, sh))
    res = []
    for variant, thr in ((7, 0), (0, 0), (0, 128), (0, 64), (7, 0), (0, 0)):
        pl.tune(N.TUNE_VARIANT, variant); pl.tune(N.TUNE_BLOCK_THREADS, thr)
        def run(n):
            for i in range(n):
                N.lib().csic_process_batch_device(pl._h, C.c_void_p(ins[i % ring].data_ptr()), C.c_void_p(outs[i % ring].data_ptr()), nfr, sh)
        run(10); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st); run(30); e1.record(st); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 30)
        res.append(f"{pl.kernel_name.split('<')[0]}{'/T' + str(thr) if thr else ''} {100 * alg * nfr / (best * 1e-3) / 8e12:.1f}")
    print(f"{W}x{H} 4:{a}:{b} f={f} s>c x{nfr}: " + " | ".join(res), flush=True)
    pl.close(); del ins, outs; torch.cuda.empty_cache()
