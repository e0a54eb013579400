// tools/ubench_aql.hip -- feasibility probe (not part of the product): independent per-frame launches submitted as
// AQL kernel-dispatch packets on a user-mode HSA queue of our own, WITHOUT the barrier bit that every HIP stream
// (and every hipGraph chain) sets -- so the command processor may start frame k+1 while frame k is still draining.
//
// The kernels are the ones HIP already loaded: their kernel objects are looked up in the process's loaded
// executables (hsa_ven_amd_loader_iterate_executables) by the name hipKernelNameRefByPtr gives.
// usage: ubench_aql [cfg5|stripe8|stripe16] [rounds] [block_threads]
#include "csic_kernels.hip"

#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <hsa/hsa_ven_amd_loader.h>
#include <hsa/amd_hsa_signal.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <string>
#include <vector>

using namespace csic;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); exit(1);} } while (0)
#define HK(x) do { hsa_status_t e = (x); if (e != HSA_STATUS_SUCCESS && e != HSA_STATUS_INFO_BREAK) { const char *m = nullptr; hsa_status_string(e, &m); printf("%s: %s (line %d)\n", #x, m ? m : "?", __LINE__); exit(1);} } while (0)

struct Found {
    hsa_agent_t agent;
    std::string want;
    uint64_t kernel_object = 0;
    uint32_t kernarg_size = 0, group_size = 0, private_size = 0;
    bool ok = false;
};

static hsa_status_t sym_cb(hsa_executable_t, hsa_agent_t, hsa_executable_symbol_t sym, void *data)
{
    Found *f = static_cast<Found *>(data);
    hsa_symbol_kind_t kind;
    if (hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_TYPE, &kind) != HSA_STATUS_SUCCESS || kind != HSA_SYMBOL_KIND_KERNEL)
        return HSA_STATUS_SUCCESS;
    uint32_t len = 0;
    hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_NAME_LENGTH, &len);
    std::string name(len, '\0');
    hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_NAME, name.data());
    if (name == f->want || name == f->want + ".kd") {
        hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &f->kernel_object);
        hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &f->kernarg_size);
        hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &f->group_size);
        hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &f->private_size);
        f->ok = true;
        return HSA_STATUS_INFO_BREAK;
    }
    return HSA_STATUS_SUCCESS;
}

static hsa_status_t exec_cb(hsa_executable_t ex, void *data)
{
    Found *f = static_cast<Found *>(data);
    hsa_status_t s = hsa_executable_iterate_agent_symbols(ex, f->agent, sym_cb, data);
    return f->ok ? HSA_STATUS_INFO_BREAK : (s == HSA_STATUS_INFO_BREAK ? HSA_STATUS_SUCCESS : s);
}

static hsa_status_t agent_cb(hsa_agent_t a, void *data)
{
    hsa_device_type_t t;
    hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU) { *static_cast<hsa_agent_t *>(data) = a; return HSA_STATUS_INFO_BREAK; }
    return HSA_STATUS_SUCCESS;
}

int main(int argc, char **argv)
{
    const std::string what = argc > 1 ? argv[1] : "cfg5";
    const int rounds = argc > 2 ? atoi(argv[2]) : 20;
    const int tpb = argc > 3 ? atoi(argv[3]) : 0;
    const int N = 64;
    csic_params p;
    if (what == "cfg5") { csic_params_default(&p, 3840, 2160); p.factor = 4; p.y_bits = 3; p.cb_bits = 3; p.cr_bits = 2; }
    else if (what == "stripe16") { csic_params_default(&p, 8192, 512); p.factor = 2; }
    else { csic_params_default(&p, 8192, 1024); p.factor = 2; }
    p.chroma_a = 2; p.chroma_b = 0;
    csic_plan *pl = nullptr;
    if (csic_plan_create(&p, 0, &pl) != CSIC_OK) { printf("plan: %s\n", csic_last_error()); return 1; }
    if (tpb) csic_plan_tune(pl, CSIC_TUNE_BLOCK_THREADS, tpb);
    size_t ipx, opx;
    plan_sizes(pl, &ipx, &opx);
    int64_t alg = 0;
    csic_algorithmic_bytes(&p, &alg);
    uint32_t *din, *dout, *dref;
    CK(hipMalloc(&din, ipx * 4 * N));
    CK(hipMalloc(&dout, opx * 4 * N));
    CK(hipMalloc(&dref, opx * 4 * N));
    csic_synth_frame_device(din, (int64_t)ipx * N, 0, 20250629u, nullptr);
    std::vector<LaunchDesc> d(N);
    for (int k = 0; k < N; ++k) prepare_launch(pl, din + (size_t)k * ipx, dout + (size_t)k * opx, 1, 0, 0, &d[k]);
    // reference output through HIP (also forces the code object to be loaded)
    csic_process_batch_device(pl, din, dref, N, nullptr);
    CK(hipDeviceSynchronize());
    uint64_t ref_sum = 0;
    csic_checksum_device(dref, (int64_t)opx * N, &ref_sum, nullptr);

    const char *kname = hipKernelNameRefByPtr((const void *)d[0].fn, nullptr);
    printf("%s: %s\n  device symbol: %s\n", what.c_str(), csic_plan_kernel_name(pl), kname ? kname : "(null)");
    if (!kname) return 1;

    HK(hsa_init());
    Found f;
    f.want = kname;
    HK(hsa_iterate_agents(agent_cb, &f.agent));
    hsa_ven_amd_loader_1_03_pfn_t loader;
    HK(hsa_system_get_major_extension_table(HSA_EXTENSION_AMD_LOADER, 1, sizeof loader, &loader));
    HK(loader.hsa_ven_amd_loader_iterate_executables(exec_cb, &f));
    if (!f.ok) { printf("kernel symbol not found among the loaded executables\n"); return 1; }
    printf("  kernel_object 0x%llx kernarg %u B group %u B private %u B (sizeof KArgs = %zu)\n", (unsigned long long)f.kernel_object,
           f.kernarg_size, f.group_size, f.private_size, sizeof(KArgs));
    if (f.private_size != 0 || f.group_size != 0) { printf("unexpected scratch/LDS use\n"); return 1; }

    const int MAXQ = 8;
    hsa_queue_t *qs[MAXQ];
    hsa_signal_t dones[MAXQ];
    for (int i = 0; i < MAXQ; ++i) {
        HK(hsa_queue_create(f.agent, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &qs[i]));
        HK(hsa_signal_create(1, 0, nullptr, &dones[i]));
    }
    hsa_queue_t *q = qs[0];
    // kernarg blocks in device memory: explicit KArgs followed by the implicit (hidden) arguments, zero except the
    // block counts / group sizes of code-object-v5's layout
    const size_t kstride = (f.kernarg_size + 255) & ~size_t(255);
    std::vector<uint8_t> hk(kstride * N, 0);
    const size_t hidden = (sizeof(KArgs) + 7) & ~size_t(7);
    for (int k = 0; k < N; ++k) {
        uint8_t *b = hk.data() + k * kstride;
        memcpy(b, &d[k].args, sizeof(KArgs));
        if (hidden + 24 <= f.kernarg_size) {
            uint32_t bc[3] = {d[k].grid.x, d[k].grid.y, d[k].grid.z};
            uint16_t gs[3] = {(uint16_t)d[k].block.x, (uint16_t)d[k].block.y, (uint16_t)d[k].block.z};
            memcpy(b + hidden, bc, 12);
            memcpy(b + hidden + 12, gs, 6);
        }
    }
    uint8_t *dk;
    CK(hipMalloc(&dk, hk.size()));
    CK(hipMemcpy(dk, hk.data(), hk.size(), hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());

    hsa_signal_t done;
    HK(hsa_signal_create(1, 0, nullptr, &done));
    const uint32_t mask = q->size - 1;
    auto *ring = static_cast<hsa_kernel_dispatch_packet_t *>(q->base_address);

    auto submit = [&](int nrounds, bool barrier_bit, int acq, int rel) {
        const int total = nrounds * N;
        hsa_signal_store_relaxed(done, 1);
        uint64_t idx = hsa_queue_add_write_index_relaxed(q, total + 1);
        while (idx + total + 1 - hsa_queue_load_read_index_scacquire(q) > q->size) {}
        for (int i = 0; i < total; ++i) {
            const int k = i % N;
            hsa_kernel_dispatch_packet_t *pk = &ring[(idx + i) & mask];
            pk->workgroup_size_x = d[k].block.x; pk->workgroup_size_y = d[k].block.y; pk->workgroup_size_z = d[k].block.z;
            pk->reserved0 = 0;
            pk->grid_size_x = d[k].grid.x * d[k].block.x; pk->grid_size_y = d[k].grid.y * d[k].block.y; pk->grid_size_z = d[k].grid.z * d[k].block.z;
            pk->private_segment_size = 0; pk->group_segment_size = 0;
            pk->kernel_object = f.kernel_object;
            pk->kernarg_address = dk + k * kstride;
            pk->reserved2 = 0;
            pk->completion_signal.handle = 0;
            const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) |
                                    ((barrier_bit || i == 0 ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                                    (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
            const uint16_t setup = 3 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
            __atomic_store_n(reinterpret_cast<uint32_t *>(pk), header | (uint32_t(setup) << 16), __ATOMIC_RELEASE);
        }
        // closing barrier-AND packet: barrier bit -> waits for every earlier packet, then signals `done`
        auto *bp = reinterpret_cast<hsa_barrier_and_packet_t *>(&ring[(idx + total) & mask]);
        memset(reinterpret_cast<uint8_t *>(bp) + 4, 0, sizeof *bp - 4);
        bp->completion_signal = done;
        const uint16_t bh = (HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                            (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
        __atomic_store_n(reinterpret_cast<uint32_t *>(bp), uint32_t(bh), __ATOMIC_RELEASE);
        hsa_signal_store_screlease(q->doorbell_signal, idx + total);
        const hsa_signal_value_t v = hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, 2000000000ull /* timeout in ticks */, HSA_WAIT_STATE_BLOCKED);
        if (v >= 1) { printf("TIMEOUT waiting for the queue (signal %lld)\n", (long long)v); exit(2); }
    };

    // Q queues: frame i goes to queue i % Q; every queue ends with its own closing barrier + completion signal
    auto submit_mq = [&](int nrounds, int Q) {
        const int total = nrounds * N;
        uint64_t idx[MAXQ];
        int cnt[MAXQ];
        for (int j = 0; j < Q; ++j) {
            cnt[j] = total / Q + (j < total % Q ? 1 : 0);
            hsa_signal_store_relaxed(dones[j], 1);
            idx[j] = hsa_queue_add_write_index_relaxed(qs[j], cnt[j] + 1);
        }
        int pos[MAXQ] = {0};
        for (int i = 0; i < total; ++i) {
            const int k = i % N, j = i % Q;
            auto *rg = static_cast<hsa_kernel_dispatch_packet_t *>(qs[j]->base_address);
            hsa_kernel_dispatch_packet_t *pk = &rg[(idx[j] + pos[j]) & (qs[j]->size - 1)];
            pk->workgroup_size_x = d[k].block.x; pk->workgroup_size_y = d[k].block.y; pk->workgroup_size_z = d[k].block.z;
            pk->reserved0 = 0;
            pk->grid_size_x = d[k].grid.x * d[k].block.x; pk->grid_size_y = d[k].grid.y * d[k].block.y; pk->grid_size_z = d[k].grid.z * d[k].block.z;
            pk->private_segment_size = 0; pk->group_segment_size = 0;
            pk->kernel_object = f.kernel_object;
            pk->kernarg_address = dk + k * kstride;
            pk->reserved2 = 0;
            pk->completion_signal.handle = 0;
            const int first = pos[j] == 0;
            const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (first << HSA_PACKET_HEADER_BARRIER) |
                                    ((first ? HSA_FENCE_SCOPE_AGENT : HSA_FENCE_SCOPE_NONE) << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                                    (HSA_FENCE_SCOPE_NONE << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
            const uint16_t setup = 3 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
            __atomic_store_n(reinterpret_cast<uint32_t *>(pk), header | (uint32_t(setup) << 16), __ATOMIC_RELEASE);
            ++pos[j];
        }
        for (int j = 0; j < Q; ++j) {
            auto *rg = static_cast<hsa_kernel_dispatch_packet_t *>(qs[j]->base_address);
            auto *bp = reinterpret_cast<hsa_barrier_and_packet_t *>(&rg[(idx[j] + cnt[j]) & (qs[j]->size - 1)]);
            memset(reinterpret_cast<uint8_t *>(bp) + 4, 0, sizeof *bp - 4);
            bp->completion_signal = dones[j];
            const uint16_t bh = (HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                (HSA_FENCE_SCOPE_NONE << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
            __atomic_store_n(reinterpret_cast<uint32_t *>(bp), uint32_t(bh), __ATOMIC_RELEASE);
        }
        for (int j = 0; j < Q; ++j) hsa_signal_store_screlease(qs[j]->doorbell_signal, idx[j] + cnt[j]);
        for (int j = 0; j < Q; ++j) {
            const hsa_signal_value_t v = hsa_signal_wait_scacquire(dones[j], HSA_SIGNAL_CONDITION_LT, 1, 2000000000ull, HSA_WAIT_STATE_BLOCKED);
            if (v >= 1) { printf("TIMEOUT waiting for queue %d\n", j); exit(2); }
        }
    };
    auto report_mq = [&](int Q, int per_submit) {
        CK(hipMemset(dout, 0, opx * 4 * N));
        CK(hipDeviceSynchronize());
        submit_mq(1, Q);
        uint64_t sum = 0;
        csic_checksum_device(dout, (int64_t)opx * N, &sum, nullptr);
        for (int w = 0; w < 10; ++w) submit_mq(8, Q);
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < rounds; r += per_submit) submit_mq(per_submit, Q);
        const auto t1 = std::chrono::steady_clock::now();
        const double us = std::chrono::duration<double, std::micro>(t1 - t0).count() / (rounds * N);
        printf("aql %d queue(s), barrier=0, %3d frames per submit+wait  %7.3f us/frame  %5.1f%% of 8 TB/s   output %s\n", Q, per_submit * N, us,
               alg / 8e12 * 1e6 / us * 100, sum == ref_sum ? "bit-exact vs HIP launch" : "MISMATCH");
        fflush(stdout);
    };

    auto report = [&](const char *name, bool barrier_bit, int acq, int rel) {
        CK(hipMemset(dout, 0, opx * 4 * N));
        CK(hipDeviceSynchronize());
        submit(1, barrier_bit, acq, rel);
        uint64_t sum = 0;
        csic_checksum_device(dout, (int64_t)opx * N, &sum, nullptr);
        for (int w = 0; w < 10; ++w) submit(8, barrier_bit, acq, rel);              // clock conditioning
        const auto t0 = std::chrono::steady_clock::now();
        submit(rounds, barrier_bit, acq, rel);
        const auto t1 = std::chrono::steady_clock::now();
        const double us = std::chrono::duration<double, std::micro>(t1 - t0).count() / (rounds * N);
        printf("%-34s %7.3f us/frame (host wall, %d frames)  %5.1f%% of 8 TB/s   output %s\n", name, us, rounds * N,
               alg / 8e12 * 1e6 / us * 100, sum == ref_sum ? "bit-exact vs HIP launch" : "MISMATCH");
        fflush(stdout);
    };
    report("aql barrier=1 acq/rel agent", true, HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_AGENT);
    report("aql barrier=0 acq/rel agent", false, HSA_FENCE_SCOPE_AGENT, HSA_FENCE_SCOPE_AGENT);
    report("aql barrier=0 acq none rel agent", false, HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_AGENT);
    report("aql barrier=0 acq/rel none", false, HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE);
    report("aql barrier=1 acq/rel none", true, HSA_FENCE_SCOPE_NONE, HSA_FENCE_SCOPE_NONE);
    report("aql barrier=0 acq/rel system", false, HSA_FENCE_SCOPE_SYSTEM, HSA_FENCE_SCOPE_SYSTEM);
    // ---- dispatch timeline from the CP's own timestamps (hsa_amd_profiling_get_dispatch_time): one 64-frame round per
    // configuration with a completion signal on every packet.  rocprofv3's kernel trace cannot show this: its queue
    // interception serialises the dispatches (max 1 kernel in flight, 6-8 us per frame under the profiler).
    // dedicated queues, profiling enabled before their first packet (toggling it on a used queue returned zero
    // timestamps on some boxes)
    hsa_queue_t *tq[4];
    for (int j = 0; j < 4; ++j) {
        HK(hsa_queue_create(f.agent, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &tq[j]));
        HK(hsa_amd_profiling_set_profiler_enabled(tq[j], 1));
    }
    auto timeline = [&](const char *name, int Q, bool barrier_bit) {
        hsa_queue_t **qs = tq;                                           // (shadows the benchmark queues)
        std::vector<hsa_signal_t> sig(N);
        for (int k = 0; k < N; ++k) HK(hsa_signal_create(1, 0, nullptr, &sig[k]));
        for (int rep = 0; rep < 3; ++rep) {                              // the last repetition is reported
            for (int k = 0; k < N; ++k) hsa_signal_store_relaxed(sig[k], 1);
            uint64_t idx[MAXQ]; int cnt[MAXQ], pos[MAXQ] = {0};
            for (int j = 0; j < Q; ++j) { cnt[j] = N / Q + (j < N % Q ? 1 : 0); idx[j] = hsa_queue_add_write_index_relaxed(qs[j], cnt[j]); }
            for (int k = 0; k < N; ++k) {
                const int j = k % Q;
                auto *rg = static_cast<hsa_kernel_dispatch_packet_t *>(qs[j]->base_address);
                hsa_kernel_dispatch_packet_t *pk = &rg[(idx[j] + pos[j]) & (qs[j]->size - 1)];
                pk->workgroup_size_x = d[k].block.x; pk->workgroup_size_y = d[k].block.y; pk->workgroup_size_z = d[k].block.z;
                pk->reserved0 = 0;
                pk->grid_size_x = d[k].grid.x * d[k].block.x; pk->grid_size_y = d[k].grid.y * d[k].block.y; pk->grid_size_z = d[k].grid.z * d[k].block.z;
                pk->private_segment_size = 0; pk->group_segment_size = 0;
                pk->kernel_object = f.kernel_object;
                pk->kernarg_address = dk + k * kstride;
                pk->reserved2 = 0;
                pk->completion_signal = sig[k];
                const int bar = (barrier_bit || pos[j] == 0) ? 1 : 0;
                const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (bar << HSA_PACKET_HEADER_BARRIER) |
                                        ((barrier_bit ? HSA_FENCE_SCOPE_AGENT : HSA_FENCE_SCOPE_NONE) << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                                        ((barrier_bit ? HSA_FENCE_SCOPE_AGENT : HSA_FENCE_SCOPE_NONE) << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
                __atomic_store_n(reinterpret_cast<uint32_t *>(pk), header | (uint32_t(3 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS) << 16), __ATOMIC_RELEASE);
                ++pos[j];
            }
            for (int j = 0; j < Q; ++j) hsa_signal_store_screlease(qs[j]->doorbell_signal, idx[j] + cnt[j] - 1);
            for (int k = 0; k < N; ++k)
                if (hsa_signal_wait_scacquire(sig[k], HSA_SIGNAL_CONDITION_LT, 1, 2000000000ull, HSA_WAIT_STATE_BLOCKED) >= 1) { printf("TIMEOUT (timeline)\n"); exit(2); }
        }
        uint64_t freq = 0;
        HK(hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &freq));
        std::vector<std::pair<uint64_t, int>> ev;
        uint64_t t0 = ~0ull, t1 = 0;
        double sum = 0;
        for (int k = 0; k < N; ++k) {
            hsa_amd_profiling_dispatch_time_t t{};
            HK(hsa_amd_profiling_get_dispatch_time(f.agent, sig[k], &t));
            ev.push_back({t.start, +1}); ev.push_back({t.end, -1});
            if (t.start < t0) t0 = t.start;
            if (t.end > t1) t1 = t.end;
            sum += double(t.end - t.start);
        }
        std::sort(ev.begin(), ev.end());
        int cur = 0, mx = 0; double area = 0; uint64_t last = t0;
        for (auto &e : ev) { area += double(e.first - last) * cur; last = e.first; cur += e.second; if (cur > mx) mx = cur; }
        const double us = 1e6 / double(freq);
        printf("timeline %-28s span %7.2f us = %5.2f us/frame; mean kernel %5.2f us; kernels in flight: mean %.2f, max %d\n", name,
               (t1 - t0) * us, (t1 - t0) * us / N, sum * us / N, area / double(t1 - t0), mx);
        fflush(stdout);
        for (int k = 0; k < N; ++k) hsa_signal_destroy(sig[k]);
    };
    timeline("1 queue, barrier bit", 1, true);
    timeline("1 queue, no barrier bit", 1, false);
    timeline("2 queues, no barrier bit", 2, false);
    timeline("4 queues, no barrier bit", 4, false);
    for (int j = 0; j < 4; ++j) hsa_queue_destroy(tq[j]);

    // ---- stream-ordered probe: can a HIP stream gate and await work on our queues?  HIP's "signal memory"
    // (hipExtMallocWithFlags(.., hipMallocSignalMemory)) is the value word of an HSA signal the runtime created; the
    // amd_signal_t it belongs to starts 8 bytes earlier (amd_hsa_signal.h), which makes it usable as an hsa_signal_t in
    // AQL barrier packets while hipStreamWriteValue64 / hipStreamWaitValue64 operate on the same word.
    {
        int can = 0;
        CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
        uint64_t *sv[1 + MAXQ] = {};
        bool ok = can != 0;
        const int Q = 4;
        for (int i = 0; ok && i < 1 + Q; ++i) {
            if (hipExtMallocWithFlags(reinterpret_cast<void **>(&sv[i]), 8, hipMallocSignalMemory) != hipSuccess) { ok = false; break; }
            const int64_t kind = reinterpret_cast<volatile int64_t *>(sv[i])[-1];
            if (kind != AMD_SIGNAL_KIND_USER || ((reinterpret_cast<uintptr_t>(sv[i]) - 8) & 63)) { printf("signal memory %d: kind %lld, not an amd_signal_t\n", i, (long long)kind); ok = false; }
        }
        printf("stream-ordered probe: CanUseStreamWaitValue=%d, signal memory %s\n", can, ok ? "looks like amd_signal_t.value" : "unusable");
        if (ok) {
            auto handle = [&](int i) { hsa_signal_t h; h.handle = reinterpret_cast<uint64_t>(sv[i]) - 8; return h; };
            hipStream_t st;
            CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
            for (int rep = 0; rep < 3; ++rep) {
                // producer on the HIP stream: clear, then regenerate the inputs; our kernels must not start before it is done
                CK(hipMemsetAsync(din, 0, ipx * 4 * N, st));
                CK(hipMemsetAsync(dout, 0, opx * 4 * N, st));
                for (int i = 0; i < 1 + Q; ++i) hsa_signal_store_screlease(handle(i), 1);
                // our queues: [barrier-AND on S] kernels (no barrier bit) [closing barrier-AND -> D_j]
                uint64_t idx[MAXQ]; int cnt[MAXQ], pos[MAXQ] = {0};
                for (int j = 0; j < Q; ++j) { cnt[j] = N / Q + (j < N % Q ? 1 : 0); idx[j] = hsa_queue_add_write_index_relaxed(qs[j], cnt[j] + 2); }
                for (int j = 0; j < Q; ++j) {
                    auto *rg = static_cast<hsa_kernel_dispatch_packet_t *>(qs[j]->base_address);
                    auto *bp = reinterpret_cast<hsa_barrier_and_packet_t *>(&rg[idx[j] & (qs[j]->size - 1)]);
                    memset(reinterpret_cast<uint8_t *>(bp) + 4, 0, sizeof *bp - 4);
                    bp->dep_signal[0] = handle(0);
                    const uint16_t bh = (HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                        (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE);
                    __atomic_store_n(reinterpret_cast<uint32_t *>(bp), uint32_t(bh), __ATOMIC_RELEASE);
                }
                for (int k = 0; k < N; ++k) {
                    const int j = k % Q;
                    auto *rg = static_cast<hsa_kernel_dispatch_packet_t *>(qs[j]->base_address);
                    hsa_kernel_dispatch_packet_t *pk = &rg[(idx[j] + 1 + pos[j]) & (qs[j]->size - 1)];
                    pk->workgroup_size_x = d[k].block.x; pk->workgroup_size_y = d[k].block.y; pk->workgroup_size_z = d[k].block.z;
                    pk->reserved0 = 0;
                    pk->grid_size_x = d[k].grid.x * d[k].block.x; pk->grid_size_y = d[k].grid.y * d[k].block.y; pk->grid_size_z = d[k].grid.z * d[k].block.z;
                    pk->private_segment_size = 0; pk->group_segment_size = 0;
                    pk->kernel_object = f.kernel_object;
                    pk->kernarg_address = dk + k * kstride;
                    pk->reserved2 = 0;
                    pk->completion_signal.handle = 0;
                    const int first = pos[j] == 0;
                    const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (first << HSA_PACKET_HEADER_BARRIER) |
                                            ((first ? HSA_FENCE_SCOPE_AGENT : HSA_FENCE_SCOPE_NONE) << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE);
                    __atomic_store_n(reinterpret_cast<uint32_t *>(pk), header | (uint32_t(3 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS) << 16), __ATOMIC_RELEASE);
                    ++pos[j];
                }
                for (int j = 0; j < Q; ++j) {
                    auto *rg = static_cast<hsa_kernel_dispatch_packet_t *>(qs[j]->base_address);
                    auto *bp = reinterpret_cast<hsa_barrier_and_packet_t *>(&rg[(idx[j] + 1 + cnt[j]) & (qs[j]->size - 1)]);
                    memset(reinterpret_cast<uint8_t *>(bp) + 4, 0, sizeof *bp - 4);
                    bp->completion_signal = handle(1 + j);
                    const uint16_t bh = (HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                        (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
                    __atomic_store_n(reinterpret_cast<uint32_t *>(bp), uint32_t(bh), __ATOMIC_RELEASE);
                }
                for (int j = 0; j < Q; ++j) hsa_signal_store_screlease(qs[j]->doorbell_signal, idx[j] + cnt[j] + 1);
                // the queues are armed and blocked on S; only now does the stream produce the inputs and open the gate
                csic_synth_frame_device(din, (int64_t)ipx * N, 0, 20250629u, st);
                CK(hipStreamWriteValue64(st, sv[0], 0, 0));
                for (int j = 0; j < Q; ++j) CK(hipStreamWaitValue64(st, sv[1 + j], 0, hipStreamWaitValueEq, ~0ull));
                uint64_t sum = 0;
                csic_checksum_device(dout, (int64_t)opx * N, &sum, st);      // ordered behind the waits; synchronises the stream
                printf("  rep %d: gate by hipStreamWriteValue64, await by hipStreamWaitValue64: output %s\n", rep,
                       sum == ref_sum ? "bit-exact vs HIP launch (ordering held both ways)" : "MISMATCH");
                fflush(stdout);
            }
            CK(hipStreamDestroy(st));
        }
        for (int i = 0; i < 1 + MAXQ; ++i) if (sv[i]) (void)hipFree(sv[i]);
    }

    for (int Q : {1, 2, 3, 4, 6, 8}) report_mq(Q, rounds);
    for (int Q : {1, 2, 4, 8}) report_mq(Q, 1);          // one 64-frame "graph launch" at a time, host waits in between
    for (int i = 0; i < MAXQ; ++i) { hsa_queue_destroy(qs[i]); hsa_signal_destroy(dones[i]); }
    hsa_signal_destroy(done);
    return 0;
}
