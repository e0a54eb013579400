// csic_graph.hip -- pre-recorded per-frame launches (BASELINE.json configs[4]: "hipGraph-captured per-frame launch")
// and the launch engine behind them.
//
// A frame of cfg 5 (3840x2160, sf=4) is 10.4 MB = 1.3 us of HBM time; a strong-scaling stripe of cfg 4 on 8 GPUs
// (8192x1024) is 25 MB = 3.1 us.  Launched one after the other on a HIP stream -- eagerly or from a captured
// hipGraph, it makes no difference -- every launch also pays the dependent-kernel boundary (~1.7 us: the AQL
// barrier bit drains the previous kernel before the next one may start) and the command processor's per-packet
// front end.  Measured on MI355X (profiles/r02_small_launch.md), per cfg 5 frame:
//     hipGraph, one chain                     3.65 us   35 % of the HBM roofline      <- what round 1 shipped
//     one hipGraph with parallel branches     3.1-4.2   ROCm replays it node by node from the host (3.1 us of host
//                                                       time per node): slower than the chain
//     3 chain graphs on 3 streams             2.9 us    44 %                           <- CSIC_FRAME_GRAPH_HIP
//     AQL packets, no barrier bit, 1 queue    2.8 us    46 %   (the CP front end is serial per queue)
//     AQL packets, no barrier bit, 4 queues   1.76 us   74 %                           <- CSIC_FRAME_GRAPH_DIRECT
//     one batched launch of all 64 frames     1.75 us   74 %   (needs contiguous frames)
//
// Three backends therefore (the third needs no overlap at all):
//  * CSIC_FRAME_GRAPH_FUSED: ONE launch over all frames -- the kernels read their frame base from a device-resident
//    pointer table indexed by grid z, so separate buffers cost nothing against the contiguous batched launch.  Not
//    "per-frame launches", but the fastest way to run N frames of one plan in stream order.
//  * CSIC_FRAME_GRAPH_HIP: `branches` hipGraph CHAINS (explicit kernel nodes from prepare_launch(), so a node is
//    bit for bit the eager launch), chain 0 replayed on the caller's stream and the others on internal streams,
//    forked and joined with events -- fully ordered with the caller's stream, works under stream capture rules of
//    plain HIP.
//  * CSIC_FRAME_GRAPH_DIRECT: the same launches as pre-built AQL kernel-dispatch packets, copied into `nq` user-mode
//    HSA queues owned by the library (frame k -> queue k % nq) WITHOUT the barrier bit, so consecutive frames
//    overlap like the workgroups of one big launch; the first packet of a submission on each queue carries the
//    barrier bit and an agent-scope acquire, a closing barrier-AND packet per queue carries the release and the
//    completion signal.  The kernels are the ones HIP loaded: their kernel objects are found among the process's
//    loaded executables (hsa_ven_amd_loader_iterate_executables) under the name hipKernelNameRefByPtr reports.
//    These queues are not HIP streams.  csic_frame_graph_submit / _wait order a submission by the host.
//    csic_frame_graph_launch(graph, stream) orders it with a HIP stream ON THE DEVICE: the signals are HIP "signal
//    memory" (the value word of an HSA signal): a one-wave gate kernel at the head of every queue spins on the gate
//    word, one hand-off kernel on the stream opens it and spins on the queues' done words (k_gate_wait / k_handoff
//    below; the command-processor form -- gate barrier packets, hipStreamWriteValue64 / hipStreamWaitValue64 -- stays
//    selectable) -- asynchronous, no host round trip (see `stream_ordered` below; without that runtime feature
//    launch() degrades to sync + submit + wait).
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <hsa/hsa_ven_amd_loader.h>
#include <hsa/amd_hsa_signal.h>

#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "csic_hip_common.h"

namespace csic {

#define HSA_TRY(expr)                                                                            \
    do {                                                                                         \
        hsa_status_t s_ = (expr);                                                                \
        if (s_ != HSA_STATUS_SUCCESS && s_ != HSA_STATUS_INFO_BREAK) {                           \
            const char *m_ = nullptr;                                                            \
            (void)hsa_status_string(s_, &m_);                                                    \
            return ::csic::set_error(CSIC_EHIP, "%s failed: %s", #expr, m_ ? m_ : "unknown HSA status"); \
        }                                                                                        \
    } while (0)

// ------------------------------------------------------------------------------------------------
// DirectEngine: `nq` user-mode AQL queues on one device, shared by every DIRECT graph of that device
// ------------------------------------------------------------------------------------------------
constexpr int MAX_QUEUES = 8;
constexpr uint32_t QUEUE_PACKETS = 4096;       // ring size per queue (packets of 64 B)
constexpr uint64_t WAIT_TICKS = 3000000000ull; // ~30 s at the 100 MHz system clock: a wait that long is an error, not a hang

struct KernelInfo {
    uint64_t object = 0;
    uint32_t kernarg_size = 0;
};

struct DirectEngine {
    int device = -1;
    hsa_agent_t agent{};
    int nq = 0;
    hsa_queue_t *q[MAX_QUEUES] = {};
    std::mutex mu;                              // serialises submissions (ring reservations stay contiguous per queue)
    std::map<const void *, KernelInfo> kernels; // host stub address -> kernel object
    int refs = 0;
    // A submission that failed AFTER some of its packets were published (a ring that stopped draining between two chunks)
    // leaves queues behind that nobody can complete: the engine is marked failed, every later submission on this device
    // returns its message instead of queueing behind the hole, and the engine (queues, and the device memory its packets
    // refer to) is leaked rather than destroyed under packets that may still run.
    bool failed = false;
    std::string failure;
};

static std::mutex g_engines_mu;
static std::map<int, DirectEngine *> g_engines;

struct AgentSearch {
    uint32_t want_bdf, want_domain;
    bool by_bdf;
    int want_ordinal, seen;
    hsa_agent_t agent;
    bool found;
};

static hsa_status_t agent_cb(hsa_agent_t a, void *data)
{
    AgentSearch *s = static_cast<AgentSearch *>(data);
    hsa_device_type_t t;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS || t != HSA_DEVICE_TYPE_GPU) return HSA_STATUS_SUCCESS;
    uint32_t bdf = 0, domain = 0;
    (void)hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf);
    (void)hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &domain);
    const bool hit = s->by_bdf ? (bdf == s->want_bdf && domain == s->want_domain) : (s->seen == s->want_ordinal);
    s->seen += 1;
    if (hit) { s->agent = a; s->found = true; return HSA_STATUS_INFO_BREAK; }
    return HSA_STATUS_SUCCESS;
}

// HIP device ordinal -> HSA agent, by PCI address (robust against *_VISIBLE_DEVICES reorderings); by GPU ordinal
// if the PCI attributes are not available.
static int find_agent(int device, hsa_agent_t *out)
{
    int bus = -1, dev = -1, dom = 0;
    AgentSearch s{};
    s.by_bdf = hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, device) == hipSuccess &&
               hipDeviceGetAttribute(&dev, hipDeviceAttributePciDeviceId, device) == hipSuccess && bus >= 0 && dev >= 0;
    if (hipDeviceGetAttribute(&dom, hipDeviceAttributePciDomainId, device) != hipSuccess) dom = 0;
    s.want_bdf = ((uint32_t)bus << 8) | ((uint32_t)dev << 3);
    s.want_domain = (uint32_t)dom;
    s.want_ordinal = device;
    HSA_TRY(hsa_iterate_agents(agent_cb, &s));
    if (!s.found && s.by_bdf) {                 // BDF layouts differ on some hosts: fall back to the ordinal
        s.by_bdf = false; s.seen = 0;
        HSA_TRY(hsa_iterate_agents(agent_cb, &s));
    }
    if (!s.found) return set_error(CSIC_ENODEVICE, "no HSA GPU agent matches HIP device %d", device);
    *out = s.agent;
    return CSIC_OK;
}

static void engine_free(DirectEngine *e)
{
    for (int i = 0; i < e->nq; ++i)
        if (e->q[i]) (void)hsa_queue_destroy(e->q[i]);
    delete e;
    (void)hsa_shut_down();                      // reference counted by the runtime; HIP keeps its own reference
}

// creates queues until the engine has `nq` of them (never destroys one while the engine lives)
static int engine_grow(DirectEngine *e, int nq)
{
    for (int i = e->nq; i < nq; ++i) {
        hsa_status_t s = hsa_queue_create(e->agent, QUEUE_PACKETS, HSA_QUEUE_TYPE_MULTI, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &e->q[i]);
        if (s != HSA_STATUS_SUCCESS) {
            const char *m = nullptr;
            (void)hsa_status_string(s, &m);
            return set_error(CSIC_EHIP, "hsa_queue_create (queue %d of %d) failed: %s", i, nq, m ? m : "?");
        }
        e->nq = i + 1;
    }
    return CSIC_OK;
}

// every packet the engine ever published has been taken off its ring
static bool engine_drained(const DirectEngine *e)
{
    for (int i = 0; i < e->nq; ++i)
        if (hsa_queue_load_read_index_scacquire(e->q[i]) != hsa_queue_load_write_index_relaxed(e->q[i])) return false;
    return true;
}


static int engine_acquire(int device, int nq, DirectEngine **out)
{
    std::lock_guard<std::mutex> lk(g_engines_mu);
    auto it = g_engines.find(device);
    if (it != g_engines.end() && it->second->failed) {
        // A failed engine stays in the table as the device's tombstone (engine_release never erases it): no second set of
        // queues is built beside queues that may still hold a hole.  It is only replaced once nothing uses it any more AND
        // every packet it published has been consumed -- then the hole has drained and a fresh engine cannot queue behind it.
        DirectEngine *dead = it->second;
        if (dead->refs > 0 || !engine_drained(dead))
            return set_error(CSIC_EHIP, "the direct-dispatch engine of device %d failed earlier: %s", device, dead->failure.c_str());
        g_engines.erase(it);
        engine_free(dead);
        it = g_engines.end();
    }
    if (it != g_engines.end()) {
        DirectEngine *e = it->second;
        {
            std::lock_guard<std::mutex> lk2(e->mu);
            int st = engine_grow(e, nq);
            if (st != CSIC_OK) return st;
        }
        e->refs += 1;
        *out = e;
        return CSIC_OK;
    }
    HSA_TRY(hsa_init());
    DirectEngine *e = new (std::nothrow) DirectEngine();
    if (!e) { (void)hsa_shut_down(); return set_error(CSIC_ENOMEM, "out of host memory"); }
    e->device = device;
    int st = find_agent(device, &e->agent);
    if (st == CSIC_OK) st = engine_grow(e, nq);
    if (st != CSIC_OK) { engine_free(e); return st; }
    e->refs = 1;
    g_engines[device] = e;
    *out = e;
    return CSIC_OK;
}

static void engine_release(DirectEngine *e)
{
    std::lock_guard<std::mutex> lk(g_engines_mu);
    if (--e->refs > 0) return;
    if (e->failed) return;                      // stays in g_engines as the device's tombstone: its queues may still hold packets
                                                // (see DirectEngine::failed; engine_acquire replaces it once they have drained)
    g_engines.erase(e->device);
    engine_free(e);
}

struct SymbolSearch {
    hsa_agent_t agent;
    std::string want;
    KernelInfo info;
    uint32_t group = 0, priv = 0;
    bool found = false;
};

static hsa_status_t symbol_cb(hsa_executable_t, hsa_agent_t, hsa_executable_symbol_t sym, void *data)
{
    SymbolSearch *s = static_cast<SymbolSearch *>(data);
    hsa_symbol_kind_t kind;
    if (hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_TYPE, &kind) != HSA_STATUS_SUCCESS || kind != HSA_SYMBOL_KIND_KERNEL)
        return HSA_STATUS_SUCCESS;
    uint32_t len = 0;
    if (hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_NAME_LENGTH, &len) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    std::string name(len, '\0');
    if (hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_NAME, &name[0]) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    if (name != s->want && name != s->want + ".kd") return HSA_STATUS_SUCCESS;
    (void)hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &s->info.object);
    (void)hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &s->info.kernarg_size);
    (void)hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &s->group);
    (void)hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &s->priv);
    s->found = true;
    return HSA_STATUS_INFO_BREAK;
}

static hsa_status_t executable_cb(hsa_executable_t ex, void *data)
{
    SymbolSearch *s = static_cast<SymbolSearch *>(data);
    (void)hsa_executable_iterate_agent_symbols(ex, s->agent, symbol_cb, data);
    return s->found ? HSA_STATUS_INFO_BREAK : HSA_STATUS_SUCCESS;
}

// ------------------------------------------------------------------------------------------------
// Device-polled hand-off between a HIP stream and the engine's queues.
// A cross-queue dependency expressed as an AQL barrier packet (or hipStreamWaitValue64) is resolved by the command
// processor POLLING the signal: about 10 us per hop on this part (tools/probe_stream_launch.py: a stream-ordered launch
// cost 17 + 10 * queues us more than the same submission ordered by the host).  A wave that polls the same word reacts
// within a microsecond.  So a stream-ordered launch is: a one-wave k_gate_wait at the head of every queue (dispatched when
// the host submits; spins until the gate word is 0), and ONE one-wave k_handoff on the launch stream, which opens the gate
// when the stream reaches it and then spins until every queue's closing packet has zeroed its done word.
// Both spins are bounded (s_memrealtime, 100 MHz): on a timeout the kernel sets the graph's error word and returns, so no
// wave outlives `timeout_ticks` whatever happens on the other side.
// ------------------------------------------------------------------------------------------------
struct GateArgs {
    const uint64_t *gate;
    uint32_t *err;
    uint64_t timeout_ticks;
};
struct HandoffArgs {
    uint64_t *gate;
    const uint64_t *done[MAX_QUEUES];
    uint32_t *err;
    uint64_t *passed;          // host-visible, one word per slot: ticket + 1 of the launch whose hand-off has finished (slot recycling)
    uint64_t seq;
    uint64_t timeout_ticks;
    int32_t nq;
};

__device__ __forceinline__ uint64_t poll64(const uint64_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void __launch_bounds__(64) k_gate_wait(GateArgs a)
{
    if (threadIdx.x != 0) return;
    const uint64_t t0 = wall_clock64();
    while (poll64(a.gate) != 0) {
        if (wall_clock64() - t0 > a.timeout_ticks) { __hip_atomic_fetch_or(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
        __builtin_amdgcn_s_sleep(4);
    }
}

__global__ void __launch_bounds__(64) k_handoff(HandoffArgs a)
{
    if (threadIdx.x != 0) return;
    // everything the stream did before this kernel is visible to the queues' kernels once they see the gate open
    __hip_atomic_store(a.gate, (uint64_t)0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    const uint64_t t0 = wall_clock64();
    for (int j = 0; j < a.nq; ++j) {
        while (poll64(a.done[j]) != 0) {
            if (wall_clock64() - t0 > a.timeout_ticks) {
                __hip_atomic_fetch_or(a.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(a.passed, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                return;
            }
            __builtin_amdgcn_s_sleep(4);
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);      // (the next packet on the stream carries its own acquire as well)
    // the slot's words have been read for the last time: the host may re-arm them (replaces a hipEventRecord per launch,
    // i.e. one more barrier packet on the stream between consecutive launches)
    __hip_atomic_store(a.passed, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Kernel object of the HIP kernel behind host stub `key` on the engine's device.
static int engine_kernel(DirectEngine *e, const void *key, size_t min_kernarg, KernelInfo *out)
{
    auto it = e->kernels.find(key);
    if (it != e->kernels.end()) { *out = it->second; return CSIC_OK; }
    hipFuncAttributes attr;
    HIP_TRY(hipFuncGetAttributes(&attr, key));              // forces HIP to load the code object on this device
    const char *name = hipKernelNameRefByPtr(key, nullptr);
    if (!name || !*name) return set_error(CSIC_EHIP, "hipKernelNameRefByPtr gave no name for the selected kernel");
    hsa_ven_amd_loader_1_03_pfn_t loader;
    std::memset(&loader, 0, sizeof loader);
    HSA_TRY(hsa_system_get_major_extension_table(HSA_EXTENSION_AMD_LOADER, 1, sizeof loader, &loader));
    if (!loader.hsa_ven_amd_loader_iterate_executables)
        return set_error(CSIC_EHIP, "this ROCm runtime has no hsa_ven_amd_loader_iterate_executables");
    SymbolSearch s;
    s.agent = e->agent;
    s.want = name;
    HSA_TRY(loader.hsa_ven_amd_loader_iterate_executables(executable_cb, &s));
    if (!s.found) return set_error(CSIC_EHIP, "kernel %s not found among the loaded executables", name);
    if (s.group != 0 || s.priv != 0)
        return set_error(CSIC_EHIP, "kernel %s uses LDS/scratch (%u/%u B): not dispatchable by the direct engine", name, s.group, s.priv);
    if (s.info.kernarg_size < min_kernarg)
        return set_error(CSIC_EHIP, "kernel %s: kernarg segment %u B smaller than its arguments (%zu B)", name, s.info.kernarg_size, min_kernarg);
    e->kernels[key] = s.info;
    *out = s.info;
    return CSIC_OK;
}

// ------------------------------------------------------------------------------------------------
// internal streams of the HIP backend: ONE pool per device, shared by every graph.  The device has 4 hardware
// queues (GPU_MAX_HW_QUEUES); if every graph brought its own streams, a ring of graphs would oversubscribe them
// (4 graphs x 3 own streams: 4.1 us per cfg 5 frame instead of 3.0 us).  Streams are created on first use and
// live until the process ends.
// ------------------------------------------------------------------------------------------------
static std::mutex g_streams_mu;
static std::map<int, std::vector<hipStream_t>> g_streams;

static int pooled_stream(int device, int idx, hipStream_t *out)
{
    std::lock_guard<std::mutex> lk(g_streams_mu);
    std::vector<hipStream_t> &v = g_streams[device];
    while ((int)v.size() <= idx) {
        hipStream_t s = nullptr;
        HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        v.push_back(s);
    }
    *out = v[idx];
    return CSIC_OK;
}

} // namespace csic

using namespace csic;

// ------------------------------------------------------------------------------------------------
// the graph object
// ------------------------------------------------------------------------------------------------
constexpr int DIRECT_SLOTS = 16;               // submissions that may be outstanding per DIRECT graph
constexpr int MAX_STREAM_ORDERED_QUEUES = 3;   // see csic_frame_graph::lpackets

struct csic_frame_graph {
    int device = 0;
    int32_t backend = CSIC_FRAME_GRAPH_HIP;
    int32_t nframes = 0, branches = 0;
    // HIP backend: one chain graph per branch; chain 0 runs on the caller's stream
    std::vector<hipGraph_t> graphs;
    std::vector<hipGraphExec_t> execs;
    std::vector<hipStream_t> streams;           // streams[i] serves chain i (streams[0] unused); borrowed from the device's pool
    std::vector<hipEvent_t> joins;
    hipEvent_t fork = nullptr;
    // FUSED backend: device-resident pointer tables + one launch descriptor per 65535 frames
    void *d_tables = nullptr;
    std::vector<LaunchDesc> fused;
    std::vector<PlanarLaunchDesc> fused_planar;       // the same for a CSIC_FMT_PLANAR plan (d_out[k] = frame k's planar buffer)
    // DIRECT backend
    DirectEngine *eng = nullptr;
    void *d_kernarg = nullptr;
    std::vector<hsa_kernel_dispatch_packet_t> packets[MAX_QUEUES];   // templates, header left INVALID: frame k on queue k % branches
    // The same frames dealt over the first `lbranches` = min(branches, MAX_STREAM_ORDERED_QUEUES) queues, built only when that
    // is fewer than `branches`: a stream-ordered launch keeps the launch stream's own hardware queue busy (its hand-off kernel
    // runs for as long as the frames do), and with 4 library queues beside it the scheduler time-slices (cfg 5, 64 frames:
    // 250 us instead of 126 us) -- so csic_frame_graph_launch never uses more than 3, whatever `branches` asked for.
    std::vector<hsa_kernel_dispatch_packet_t> lpackets[MAX_QUEUES];
    int32_t lbranches = 0;
    bool poisoned = false;                      // a submission of this graph failed half-published, or its final wait failed: leak, do not free
    double submit_timeout_s = 65.0;             // host-side bound on waiting for ring space (CSIC_DIRECT_SUBMIT_TIMEOUT_MS)
    hsa_signal_t done[DIRECT_SLOTS][MAX_QUEUES] = {};
    bool have_signals = false;
    int64_t next_ticket = 0, waited = 0;        // tickets < waited have completed and been observed
    // Stream-ordered launches.  HIP's "signal memory" (hipExtMallocWithFlags(.., hipMallocSignalMemory)) is the value
    // word of an HSA signal the runtime created; the amd_signal_t it belongs to starts 8 bytes earlier
    // (hsa/amd_hsa_signal.h).  Such a signal works on both sides: hipStreamWriteValue64 / hipStreamWaitValue64 on the
    // word, AQL barrier packets (and hsa_signal_wait) on the handle -- a HIP stream can open a gate in our queues and
    // wait for their closing packets without the host in between.  Verified at creation (kind == USER, 64-byte aligned);
    // if the runtime does not offer it the graph falls back to plain HSA signals and host-ordered launches.
    bool stream_ordered = false;
    uint64_t *sigmem[DIRECT_SLOTS][1 + MAX_QUEUES] = {};   // [slot][0] = gate, [slot][1 + j] = queue j done (HIP allocations)
    hsa_signal_t gate[DIRECT_SLOTS] = {};
    hipEvent_t consumed[DIRECT_SLOTS] = {};                 // recorded on the launch stream behind its waits
    bool slot_on_stream[DIRECT_SLOTS] = {};
    // device-polled hand-off (see k_gate_wait / k_handoff): default for stream-ordered launches; CSIC_DIRECT_HANDOFF=cp in the
    // environment keeps the command processor's barrier packets + hipStreamWriteValue64 / hipStreamWaitValue64 (A/B)
    bool kernel_handoff = false;
    KernelInfo gate_kernel;
    void *d_gateargs = nullptr;                             // DIRECT_SLOTS kernarg blocks for k_gate_wait, device memory
    uint32_t *err_word = nullptr;                           // pinned host memory: bit 0 = a gate wait timed out, bit 1 = a hand-off did
    uint64_t *passed_word = nullptr;                        // same allocation (+64): DIRECT_SLOTS words, see HandoffArgs::passed --
                                                            // per slot, so that launches of one graph on different streams may finish in any order
    int64_t slot_ticket[DIRECT_SLOTS] = {};                 // ticket of the stream-ordered launch that last used the slot
    int32_t slot_queues[DIRECT_SLOTS] = {};                 // queues the slot's submission went to
    bool slot_gated[DIRECT_SLOTS] = {};                     // kernel-gated: queue 0's closing packet does NOT depend on the other queues
    uint64_t timeout_ticks = 0;
};
constexpr size_t GATEARG_STRIDE = 512;

static void graph_free(csic_frame_graph *g)
{
    // HIP backend: the chains on the pooled streams must have finished before their execs and events go away (the
    // caller's launch stream is the caller's to synchronise first: csic.h)
    for (auto s : g->streams) if (s) (void)hipStreamSynchronize(s);        // pooled: not destroyed here
    for (auto ex : g->execs) if (ex) (void)hipGraphExecDestroy(ex);
    for (auto gr : g->graphs) if (gr) (void)hipGraphDestroy(gr);
    for (auto ev : g->joins) if (ev) (void)hipEventDestroy(ev);
    if (g->fork) (void)hipEventDestroy(g->fork);
    if (g->d_tables) (void)hipFree(g->d_tables);
    // DIRECT backend.  A poisoned graph (a submission failed half-published, or the final wait did not come back) may
    // still have packets queued and gate / hand-off waves spinning that refer to its kernargs, signal words and pinned
    // words: freeing them would be a use-after-free ON THE DEVICE (a memory fault takes every process on the GPU down),
    // so they are leaked -- a few hundred kilobytes, once, on a path that only a hung GPU reaches.
    const bool leak = g->poisoned || (g->eng && g->eng->failed);
    if (!leak) {
        if (g->d_kernarg) (void)hipFree(g->d_kernarg);
        if (g->d_gateargs) (void)hipFree(g->d_gateargs);
        if (g->err_word) (void)hipHostFree(g->err_word);
        if (g->have_signals && !g->stream_ordered)
            for (int s = 0; s < DIRECT_SLOTS; ++s)
                for (int j = 0; j < MAX_QUEUES; ++j)
                    if (g->done[s][j].handle) (void)hsa_signal_destroy(g->done[s][j]);
        for (int s = 0; s < DIRECT_SLOTS; ++s)
            for (int j = 0; j < 1 + MAX_QUEUES; ++j)
                if (g->sigmem[s][j]) (void)hipFree(g->sigmem[s][j]);
    }
    for (int s = 0; s < DIRECT_SLOTS; ++s)
        if (g->consumed[s]) (void)hipEventDestroy(g->consumed[s]);
    if (g->eng) engine_release(g->eng);
    delete g;
}

static int build_hip(csic_frame_graph *g, csic_plan *plan, const void *const *d_in, void *const *d_out)
{
    const int B = g->branches, n = g->nframes;
    try {
        g->graphs.assign(B, nullptr); g->execs.assign(B, nullptr); g->streams.assign(B, nullptr); g->joins.assign(B, nullptr);
    } catch (const std::bad_alloc &) { return set_error(CSIC_ENOMEM, "out of host memory"); }
    for (int i = 0; i < B; ++i) {
        HIP_TRY(hipGraphCreate(&g->graphs[i], 0));
        hipGraphNode_t prev{};
        bool have = false;
        for (int k = i; k < n; k += B) {
            LaunchDesc d;
            int st = prepare_launch(plan, d_in[k], d_out[k], 1, 0, 0, &d);
            if (st != CSIC_OK) return st;
            void *params[1] = {&d.args};                       // copied by hipGraphAddKernelNode
            hipKernelNodeParams np;
            std::memset(&np, 0, sizeof np);
            np.func = reinterpret_cast<void *>(d.fn);
            np.gridDim = d.grid;
            np.blockDim = d.block;
            np.kernelParams = params;
            hipGraphNode_t node;
            HIP_TRY(hipGraphAddKernelNode(&node, g->graphs[i], have ? &prev : nullptr, have ? 1 : 0, &np));
            prev = node;
            have = true;
        }
        HIP_TRY(hipGraphInstantiate(&g->execs[i], g->graphs[i], nullptr, nullptr, 0));
        if (i > 0) {
            int st = pooled_stream(g->device, i - 1, &g->streams[i]);
            if (st != CSIC_OK) return st;
            HIP_TRY(hipEventCreateWithFlags(&g->joins[i], hipEventDisableTiming));
        }
    }
    if (B > 1) HIP_TRY(hipEventCreateWithFlags(&g->fork, hipEventDisableTiming));
    return CSIC_OK;
}

// FUSED: the frames of ONE plan need not be contiguous to share a launch -- the kernels take their frame base from a
// device-resident pointer table indexed by grid z.  One ordinary kernel launch on the caller's stream: asynchronous, ordered,
// capturable, and as fast as the contiguous batched launch.
static int build_fused(csic_frame_graph *g, csic_plan *plan, const void *const *d_in, void *const *d_out)
{
    const int n = g->nframes;
    uintptr_t align_bits = 0;
    for (int k = 0; k < n; ++k) align_bits |= (uintptr_t)d_in[k] | (uintptr_t)d_out[k];
    const size_t bytes = (size_t)n * sizeof(void *);
    HIP_TRY(hipMalloc(&g->d_tables, 2 * bytes));
    HIP_TRY(hipMemcpy(g->d_tables, d_in, bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(static_cast<uint8_t *>(g->d_tables) + bytes, d_out, bytes, hipMemcpyHostToDevice));
    const void *const *tin = static_cast<const void *const *>(g->d_tables);
    void *const *tout = reinterpret_cast<void *const *>(static_cast<uint8_t *>(g->d_tables) + bytes);
    const bool planar = plan_params(plan).out_format == CSIC_FMT_PLANAR;
    if (planar)
        for (int k = 0; k < n; ++k)
            if ((uintptr_t)d_out[k] & 255u)
                return set_error(CSIC_EINVAL_SIZE, "frame %d: a planar frame buffer must be 256-byte aligned", k);
    for (int f0 = 0; f0 < n; f0 += 65535) {                       // grid z limit
        const int nz = (n - f0 < 65535) ? n - f0 : 65535;
        if (planar) {
            PlanarLaunchDesc d;
            const int st = planar_prepare_table(plan, tin + f0, tout + f0, align_bits, nz, &d);
            if (st != CSIC_OK) return st;
            try { g->fused_planar.push_back(d); } catch (const std::bad_alloc &) { return set_error(CSIC_ENOMEM, "out of host memory"); }
            continue;
        }
        LaunchDesc d;
        const int st = prepare_launch_table(plan, tin + f0, tout + f0, align_bits, nz, &d);
        if (st != CSIC_OK) return st;
        try { g->fused.push_back(d); } catch (const std::bad_alloc &) { return set_error(CSIC_ENOMEM, "out of host memory"); }
    }
    return CSIC_OK;
}

static int build_direct(csic_frame_graph *g, csic_plan *plan, const void *const *d_in, void *const *d_out)
{
    const int n = g->nframes;
    int st = engine_acquire(g->device, g->branches, &g->eng);
    if (st != CSIC_OK) return st;
    DirectEngine *e = g->eng;
    std::lock_guard<std::mutex> lk(e->mu);
    if (e->failed) return set_error(CSIC_EHIP, "the direct-dispatch engine of device %d failed earlier: %s", g->device, e->failure.c_str());
    g->lbranches = g->branches < MAX_STREAM_ORDERED_QUEUES ? g->branches : MAX_STREAM_ORDERED_QUEUES;
    {
        // Bounds.  Device side: the gate / hand-off spins give up after CSIC_DIRECT_TIMEOUT_MS (default 30 s).  Host side: a
        // submission waits for ring space at most CSIC_DIRECT_SUBMIT_TIMEOUT_MS (default: twice the device bound + 5 s, so
        // that a stalled stream shows up as the device-side timeout it is, not as a ring that does not drain).
        double ms = 30000.0;
        if (const char *t = std::getenv("CSIC_DIRECT_TIMEOUT_MS")) { const double v = std::atof(t); if (v >= 1.0) ms = v; }
        g->timeout_ticks = (uint64_t)(ms * 1.0e5);                               // s_memrealtime counts at 100 MHz
        g->submit_timeout_s = 2.0e-3 * ms + 5.0;
        if (const char *t = std::getenv("CSIC_DIRECT_SUBMIT_TIMEOUT_MS")) { const double v = std::atof(t); if (v >= 1.0) g->submit_timeout_s = 1.0e-3 * v; }
    }
    // resolve every node first (kernarg sizes may differ if some frames fall back to the 4-byte kernels)
    std::vector<LaunchDesc> descs;
    std::vector<KernelInfo> infos;
    try { descs.resize(n); infos.resize(n); } catch (const std::bad_alloc &) { return set_error(CSIC_ENOMEM, "out of host memory"); }
    size_t stride = 0;
    for (int k = 0; k < n; ++k) {
        st = prepare_launch(plan, d_in[k], d_out[k], 1, 0, 0, &descs[k]);
        if (st != CSIC_OK) return st;
        st = engine_kernel(e, reinterpret_cast<const void *>(descs[k].fn), sizeof(KArgs), &infos[k]);
        if (st != CSIC_OK) return st;
        const size_t need = (infos[k].kernarg_size + 255u) & ~size_t(255);
        if (need > stride) stride = need;
    }
    // kernarg blocks in device memory (what HIP does on this part too): the explicit KArgs, then the implicit
    // arguments of code-object v5 -- zero except block counts and group sizes (the shipped kernels read none)
    std::vector<uint8_t> host;
    try { host.assign(stride * n, 0); } catch (const std::bad_alloc &) { return set_error(CSIC_ENOMEM, "out of host memory"); }
    const size_t hidden = (sizeof(KArgs) + 7) & ~size_t(7);
    for (int k = 0; k < n; ++k) {
        uint8_t *b = host.data() + (size_t)k * stride;
        std::memcpy(b, &descs[k].args, sizeof(KArgs));
        if (hidden + 18 <= infos[k].kernarg_size) {
            const uint32_t bc[3] = {descs[k].grid.x, descs[k].grid.y, descs[k].grid.z};
            const uint16_t gs[3] = {(uint16_t)descs[k].block.x, (uint16_t)descs[k].block.y, (uint16_t)descs[k].block.z};
            std::memcpy(b + hidden, bc, 12);
            std::memcpy(b + hidden + 12, gs, 6);
        }
    }
    HIP_TRY(hipMalloc(&g->d_kernarg, host.size()));
    HIP_TRY(hipMemcpy(g->d_kernarg, host.data(), host.size(), hipMemcpyHostToDevice));
    for (int k = 0; k < n; ++k) {
        hsa_kernel_dispatch_packet_t p;
        std::memset(&p, 0, sizeof p);
        const LaunchDesc &d = descs[k];
        p.workgroup_size_x = (uint16_t)d.block.x; p.workgroup_size_y = (uint16_t)d.block.y; p.workgroup_size_z = (uint16_t)d.block.z;
        p.grid_size_x = d.grid.x * d.block.x; p.grid_size_y = d.grid.y * d.block.y; p.grid_size_z = d.grid.z * d.block.z;
        p.kernel_object = infos[k].object;
        p.kernarg_address = static_cast<uint8_t *>(g->d_kernarg) + (size_t)k * stride;
        try {
            g->packets[k % g->branches].push_back(p);
            if (g->lbranches < g->branches) g->lpackets[k % g->lbranches].push_back(p);
        } catch (const std::bad_alloc &) { return set_error(CSIC_ENOMEM, "out of host memory"); }
    }
    // signals: HIP signal memory when the runtime offers it (stream-ordered launches), plain HSA signals otherwise
    int can = 0;
    bool shared = hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, g->device) == hipSuccess && can != 0;
    for (int s = 0; shared && s < DIRECT_SLOTS; ++s)
        for (int j = 0; shared && j < 1 + g->branches; ++j) {
            void *p = nullptr;
            if (hipExtMallocWithFlags(&p, 8, hipMallocSignalMemory) != hipSuccess) { (void)hipGetLastError(); shared = false; break; }
            g->sigmem[s][j] = static_cast<uint64_t *>(p);
            const uintptr_t base = reinterpret_cast<uintptr_t>(p) - offsetof(amd_signal_t, value);
            if ((base & (AMD_SIGNAL_ALIGN_BYTES - 1)) || reinterpret_cast<const volatile amd_signal_t *>(base)->kind != AMD_SIGNAL_KIND_USER) shared = false;
        }
    if (shared) {
        for (int s = 0; s < DIRECT_SLOTS; ++s) {
            g->gate[s].handle = reinterpret_cast<uint64_t>(g->sigmem[s][0]) - offsetof(amd_signal_t, value);
            hsa_signal_store_relaxed(g->gate[s], 0);
            for (int j = 0; j < g->branches; ++j) {
                g->done[s][j].handle = reinterpret_cast<uint64_t>(g->sigmem[s][1 + j]) - offsetof(amd_signal_t, value);
                hsa_signal_store_relaxed(g->done[s][j], 0);
            }
            HIP_TRY(hipEventCreateWithFlags(&g->consumed[s], hipEventDisableTiming));
        }
        g->stream_ordered = true;
        g->have_signals = true;
        const char *mode = std::getenv("CSIC_DIRECT_HANDOFF");
        if (!(mode && std::strcmp(mode, "cp") == 0)) {
            // device-polled hand-off; any failure to set it up leaves the command-processor path in place
            KernelInfo gi;
            void *ew = nullptr;
            if (engine_kernel(e, reinterpret_cast<const void *>(k_gate_wait), sizeof(GateArgs), &gi) == CSIC_OK &&
                gi.kernarg_size <= GATEARG_STRIDE && hipHostMalloc(&ew, 64 + 8 * DIRECT_SLOTS, hipHostMallocDefault) == hipSuccess) {
                g->err_word = static_cast<uint32_t *>(ew);
                *g->err_word = 0;
                g->passed_word = reinterpret_cast<uint64_t *>(static_cast<uint8_t *>(ew) + 64);
                for (int s = 0; s < DIRECT_SLOTS; ++s) g->passed_word[s] = 0;
                std::vector<uint8_t> blocks;
                try { blocks.assign(GATEARG_STRIDE * DIRECT_SLOTS, 0); } catch (const std::bad_alloc &) { return set_error(CSIC_ENOMEM, "out of host memory"); }
                for (int s = 0; s < DIRECT_SLOTS; ++s) {
                    GateArgs ga{g->sigmem[s][0], g->err_word, g->timeout_ticks};
                    uint8_t *b = blocks.data() + (size_t)s * GATEARG_STRIDE;
                    std::memcpy(b, &ga, sizeof ga);
                    const size_t hid = (sizeof(GateArgs) + 7) & ~size_t(7);
                    if (hid + 18 <= gi.kernarg_size) {
                        const uint32_t bc[3] = {1, 1, 1};
                        const uint16_t gs[3] = {64, 1, 1};
                        std::memcpy(b + hid, bc, 12);
                        std::memcpy(b + hid + 12, gs, 6);
                    }
                }
                HIP_TRY(hipMalloc(&g->d_gateargs, blocks.size()));
                HIP_TRY(hipMemcpy(g->d_gateargs, blocks.data(), blocks.size(), hipMemcpyHostToDevice));
                g->gate_kernel = gi;
                g->kernel_handoff = true;
            } else {
                (void)hipGetLastError();
                clear_error();
            }
        }
        return CSIC_OK;
    }
    for (int s = 0; s < DIRECT_SLOTS; ++s)
        for (int j = 0; j < 1 + MAX_QUEUES; ++j)
            if (g->sigmem[s][j]) { (void)hipFree(g->sigmem[s][j]); g->sigmem[s][j] = nullptr; }
    for (int s = 0; s < DIRECT_SLOTS; ++s)
        for (int j = 0; j < g->branches; ++j) {
            g->have_signals = true;
            HSA_TRY(hsa_signal_create(0, 0, nullptr, &g->done[s][j]));
        }
    return CSIC_OK;
}

// header word = 16-bit AQL header | 16-bit setup, published last with release semantics
static inline void publish(void *slot, uint16_t header, uint16_t setup)
{
    __atomic_store_n(static_cast<uint32_t *>(slot), (uint32_t)header | ((uint32_t)setup << 16), __ATOMIC_RELEASE);
}

static int wait_slot(csic_frame_graph *g, int64_t ticket)
{
    const int slot = (int)(ticket % DIRECT_SLOTS);
    if (g->slot_on_stream[slot]) {              // a stream-ordered launch: done when the stream has passed its waits
        if (g->kernel_handoff) {
            // k_handoff publishes ticket + 1 when it has read the slot's words for the last time
            const uint64_t want = (uint64_t)g->slot_ticket[slot] + 1;
            timespec t0{}, now{};
            clock_gettime(CLOCK_MONOTONIC, &t0);
            for (uint64_t spins = 0; __atomic_load_n(&g->passed_word[slot], __ATOMIC_ACQUIRE) != want; ++spins) {
                if (spins < 2000) continue;                                     // ~ a few microseconds of pure spinning
                if ((spins & 63) == 0) {
                    clock_gettime(CLOCK_MONOTONIC, &now);
                    const double el = (double)(now.tv_sec - t0.tv_sec) + 1e-9 * (double)(now.tv_nsec - t0.tv_nsec);
                    if (el > 2.0e-8 * (double)g->timeout_ticks + 5.0) {        // both device spins have timed out by then
                        g->poisoned = true;                                     // nothing of this graph may be re-armed or freed any more
                        return set_error(CSIC_EHIP, "direct dispatch: launch %lld did not pass its hand-off in time", (long long)g->slot_ticket[slot]);
                    }
                    if (el > 2.0e-4) { const timespec nap{0, 20000}; nanosleep(&nap, nullptr); }
                }
            }
        } else {
            HIP_TRY(hipEventSynchronize(g->consumed[slot]));
        }
        if (g->err_word && *g->err_word) {
            // A gate or a hand-off gave up: the slot's queues may still be running (a kernel-gated submission's queue 0 does not
            // wait for the others), so the slot is neither marked idle nor ever waited for through the host-ordered path -- the
            // graph is poisoned: every later wait / launch / submit reports it, and destroy leaks its device words instead of
            // freeing them under packets that may still run.
            g->poisoned = true;
            return set_error(CSIC_EHIP, "direct dispatch: a stream-ordered launch timed out waiting for %s (flags %u)",
                             (*g->err_word & 1u) ? "its gate" : "its queues", *g->err_word);
        }
        g->slot_on_stream[slot] = false;
        return CSIC_OK;
    }
    // done[slot][0] is completed by queue 0's closing packet, which depends on every other queue's closing signal
    const hsa_signal_value_t v = hsa_signal_wait_scacquire(g->done[slot][0], HSA_SIGNAL_CONDITION_LT, 1, WAIT_TICKS, HSA_WAIT_STATE_BLOCKED);
    if (v >= 1) return set_error(CSIC_EHIP, "direct dispatch: submission %lld did not finish in time", (long long)ticket);
    return CSIC_OK;
}

static int direct_wait(csic_frame_graph *g, int64_t ticket)
{
    if (g->poisoned) return set_error(CSIC_EHIP, "direct dispatch: an earlier submission of this graph failed; the graph is unusable");
    if (ticket < 0 || ticket >= g->next_ticket) ticket = g->next_ticket - 1;
    while (g->waited <= ticket) {
        int st = wait_slot(g, g->waited);
        if (st != CSIC_OK) return st;
        g->waited += 1;
    }
    return CSIC_OK;
}

// Waits until `need` more packets fit in q's ring behind the read index.  Every writer of these queues holds e->mu, so the
// write index cannot move under the caller: once this returns true the reservation that follows cannot overrun the ring.
// Nothing is reserved while waiting -- a wait that gives up leaves the queue exactly as it found it.
static bool ring_wait(hsa_queue_t *q, uint64_t need, double timeout_s)
{
    const uint64_t w = hsa_queue_load_write_index_relaxed(q);
    timespec t0{}, now{};
    for (uint64_t spins = 0; w + need - hsa_queue_load_read_index_scacquire(q) > q->size; ++spins) {
        if (spins == 0) clock_gettime(CLOCK_MONOTONIC, &t0);
        if ((spins & 63) != 63) continue;
        clock_gettime(CLOCK_MONOTONIC, &now);
        const double el = (double)(now.tv_sec - t0.tv_sec) + 1e-9 * (double)(now.tv_nsec - t0.tv_nsec);
        if (el > timeout_s) return false;
        if (el > 2.0e-4) { const timespec nap{0, 20000}; nanosleep(&nap, nullptr); }
    }
    return true;
}

// A submission failed after some of its packets were published: nobody can complete it.  Open its gate (no wave may stay
// blocked on it), mark graph and engine failed (later submissions on the device get `why` back, nothing is freed under the
// packets -- see DirectEngine::failed) and report.  Called with e->mu held.
static int poison(DirectEngine *e, csic_frame_graph *g, int slot, bool gated, int queue)
{
    if (gated) {
        if (g->stream_ordered) __atomic_store_n(g->sigmem[slot][0], (uint64_t)0, __ATOMIC_RELEASE);
        if (!g->kernel_handoff && g->gate[slot].handle) hsa_signal_store_screlease(g->gate[slot], 0);
    }
    g->poisoned = true;
    e->failed = true;
    char buf[160];
    std::snprintf(buf, sizeof buf, "queue %d stopped draining in the middle of a submission (waited %.1f s)", queue, g->submit_timeout_s);
    e->failure = buf;
    return set_error(CSIC_EHIP, "direct dispatch: %s; the engine of device %d is disabled", buf, e->device);
}

// gated = stream-ordered: every queue starts with a gate on the slot's gate word -- a k_gate_wait dispatch (device-polled
// hand-off) or a barrier-AND packet on the gate signal -- which the launch stream opens once its earlier work is done.
// A gated submission uses the graph's first `lbranches` queues (lpackets), a host-ordered one all `branches` of them.
static int direct_submit(csic_frame_graph *g, int64_t *ticket, bool gated = false)
{
    DirectEngine *e = g->eng;
    if (g->poisoned) return set_error(CSIC_EHIP, "direct dispatch: an earlier submission of this graph failed; the graph is unusable");
    if (g->next_ticket - g->waited >= DIRECT_SLOTS) {           // recycle the oldest slot: the host waits for it
        int st = direct_wait(g, g->waited);
        if (st != CSIC_OK) return st;
    }
    const int64_t t = g->next_ticket;
    const int slot = (int)(t % DIRECT_SLOTS);
    const bool narrow = gated && g->lbranches < g->branches;
    const int nq = narrow ? g->lbranches : g->branches;         // this submission's queues: the engine's first `nq`
    const std::vector<hsa_kernel_dispatch_packet_t> *pk = narrow ? g->lpackets : g->packets;
    const bool kgate = gated && g->kernel_handoff;
    // Queues are filled in lock step, a chunk at a time, so that a graph larger than the rings flows through them.
    const uint32_t CHUNK = 512;
    // closing packets: queues 1.. end with one barrier-AND that completes their own signal; queue 0 ends with
    // barrier-AND packet(s) that additionally DEPEND on those signals (5 dependencies per packet) and complete
    // done[slot][0] -- the one signal the host waits for (k_handoff polls every queue's done word itself)
    auto nclose = [&](int j) -> uint32_t { return (j == 0 && nq > 6 && !kgate) ? 2u : 1u; };

    std::lock_guard<std::mutex> lk(e->mu);
    if (e->failed) return set_error(CSIC_EHIP, "the direct-dispatch engine of device %d failed earlier: %s", e->device, e->failure.c_str());
    // Phase 1 -- nothing published yet: room for the gate, the first chunk and (if that is all) the closing packets on
    // EVERY queue.  A queue that does not make room in time fails the call cleanly: no reservation, no armed signal.
    for (int j = 0; j < nq; ++j) {
        // (a gated submission must fit as a whole: its later chunks would wait behind its own closed gate; launch() has
        // checked that it can)
        const size_t n = pk[j].size();
        const uint64_t need = gated ? 1u + n + nclose(j) : (n < CHUNK ? n : CHUNK) + (n <= CHUNK ? nclose(j) : 0u);
        if (!ring_wait(e->q[j], need, g->submit_timeout_s))
            return set_error(CSIC_EHIP, "direct dispatch: queue %d is not draining (no room for %llu packets after %.1f s); nothing was submitted",
                             j, (unsigned long long)need, g->submit_timeout_s);
    }
    // Re-arm the slot's signals.  hsa_signal_store_* on an interrupt-capable signal also raises its event -- a KFD ioctl of
    // several microseconds, per signal -- which nobody listens for here (the slot is idle: no waiter, no packet refers to it
    // yet).  With HIP signal memory the value word is ours to write, so a plain store does (measured: 10 us per queue off a
    // stream-ordered launch).
    if (g->stream_ordered) {
        for (int j = 0; j < nq; ++j) __atomic_store_n(g->sigmem[slot][1 + j], (uint64_t)1, __ATOMIC_RELAXED);
        if (gated) __atomic_store_n(g->sigmem[slot][0], (uint64_t)1, __ATOMIC_RELEASE);
    } else {
        for (int j = 0; j < nq; ++j) hsa_signal_store_relaxed(g->done[slot][j], 1);
    }
    const uint16_t setup = 3 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
    const uint16_t h_first = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                             (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE);
    const uint16_t h_next = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE);      // no barrier bit, no fences
    const uint16_t h_close = (HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                             (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
    const uint16_t h_gate = (HSA_PACKET_TYPE_BARRIER_AND << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                            (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE);
    const uint16_t h_kgate = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE);     // no barrier bit: it only waits
    size_t pos[MAX_QUEUES] = {};
    bool closed[MAX_QUEUES] = {};
    int open = nq;
    bool first_round = true;
    while (open > 0) {
        for (int j = 0; j < nq; ++j) {
            if (closed[j]) continue;
            hsa_queue_t *q = e->q[j];
            const size_t left = pk[j].size() - pos[j];
            const uint32_t nk = (uint32_t)(left < CHUNK ? left : CHUNK);
            const bool last = (left == nk);
            const uint32_t ngate = (first_round && gated) ? 1u : 0u;
            const uint32_t ncl = last ? nclose(j) : 0u;
            const uint32_t total = ngate + nk + ncl;
            // later chunks wait for room again; by now packets of this submission are out, so giving up poisons the engine
            if (!first_round && !ring_wait(q, total, g->submit_timeout_s)) return poison(e, g, slot, gated, j);
            const uint64_t idx0 = hsa_queue_add_write_index_relaxed(q, total);
            auto *ring = static_cast<hsa_kernel_dispatch_packet_t *>(q->base_address);
            const uint64_t mask = q->size - 1;
            uint64_t idx = idx0;
            if (ngate) {
                void *slotp = &ring[idx & mask];
                if (kgate) {
                    // one wave that spins on the gate word; the first frame's barrier bit makes the queue wait for it
                    auto *kp = static_cast<hsa_kernel_dispatch_packet_t *>(slotp);
                    std::memset(reinterpret_cast<uint8_t *>(kp) + 4, 0, sizeof *kp - 4);
                    kp->workgroup_size_x = 64; kp->workgroup_size_y = 1; kp->workgroup_size_z = 1;
                    kp->grid_size_x = 64; kp->grid_size_y = 1; kp->grid_size_z = 1;
                    kp->kernel_object = g->gate_kernel.object;
                    kp->kernarg_address = static_cast<uint8_t *>(g->d_gateargs) + (size_t)slot * GATEARG_STRIDE;
                    publish(kp, h_kgate, 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS);
                } else {
                    auto *bp = static_cast<hsa_barrier_and_packet_t *>(slotp);
                    std::memset(reinterpret_cast<uint8_t *>(bp) + 4, 0, sizeof *bp - 4);
                    bp->dep_signal[0] = g->gate[slot];
                    publish(bp, h_gate, 0);
                }
                idx += 1;
            }
            for (uint32_t i = 0; i < nk; ++i) {
                hsa_kernel_dispatch_packet_t *dst = &ring[(idx + i) & mask];
                const hsa_kernel_dispatch_packet_t &src = pk[j][pos[j] + i];
                std::memcpy(reinterpret_cast<uint8_t *>(dst) + 4, reinterpret_cast<const uint8_t *>(&src) + 4, sizeof src - 4);
                publish(dst, (pos[j] + i == 0) ? h_first : h_next, setup);
            }
            if (last) {
                int dep = 1;                                            // next other-queue signal queue 0 still has to await
                for (uint32_t c = 0; c < ncl; ++c) {
                    auto *bp = reinterpret_cast<hsa_barrier_and_packet_t *>(&ring[(idx + nk + c) & mask]);
                    std::memset(reinterpret_cast<uint8_t *>(bp) + 4, 0, sizeof *bp - 4);
                    if (j == 0 && !kgate)                               // (k_handoff polls every queue's done word itself)
                        for (int k = 0; k < 5 && dep < nq; ++k) bp->dep_signal[k] = g->done[slot][dep++];
                    if (c + 1 == ncl) bp->completion_signal = g->done[slot][j];
                    publish(bp, h_close, 0);
                }
                closed[j] = true;
                open -= 1;
            }
            pos[j] += nk;
            hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)(idx0 + total - 1));
        }
        first_round = false;
    }
    g->next_ticket = t + 1;
    g->slot_queues[slot] = nq;
    g->slot_gated[slot] = kgate;
    if (ticket) *ticket = t;
    return CSIC_OK;
}

extern "C" {

int csic_frame_graph_create_ex(csic_plan *plan, const void *const *d_in, void *const *d_out, int32_t nframes,
                               int32_t branches, int32_t backend, csic_frame_graph **out)
{
    if (!out) return set_error(CSIC_EINVAL_NULL, "out is NULL");
    *out = nullptr;
    if (!plan || !d_in || !d_out) return set_error(CSIC_EINVAL_NULL, "argument is NULL");
    if (nframes < 1 || nframes > 65536) return set_error(CSIC_EINVAL_SIZE, "nframes must be in 1..65536. Got %d", nframes);
    // AUTO: every frame of a graph shares one plan, so ONE launch over a pointer table always applies and is the fastest
    // stream-ordered way through them at every frame size (profiles/r02_small_launch.md); the per-frame-launch backends are
    // for callers that ask for per-frame launches by name.
    if (backend == CSIC_FRAME_GRAPH_AUTO) backend = CSIC_FRAME_GRAPH_FUSED;
    if (plan_params(plan).out_format == CSIC_FMT_PLANAR && backend != CSIC_FRAME_GRAPH_FUSED)
        return set_error(CSIC_EINVAL_FORMAT, "a planar plan's frame graph is one fused launch (CSIC_FRAME_GRAPH_AUTO / _FUSED); the per-frame-launch "
                                              "backends take packed formats only");
    if (backend != CSIC_FRAME_GRAPH_HIP && backend != CSIC_FRAME_GRAPH_DIRECT && backend != CSIC_FRAME_GRAPH_FUSED)
        return set_error(CSIC_EINVAL_SIZE, "unknown frame-graph backend %d", backend);
    if (backend == CSIC_FRAME_GRAPH_FUSED) branches = 1;          // one launch: nothing to overlap with
    const int cap = backend == CSIC_FRAME_GRAPH_DIRECT ? MAX_QUEUES : 16;
    if (branches <= 0) {
        // Defaults from profiles/r02_small_launch.md, by the frame's data-movement floor at 8 TB/s: overlap pays the
        // more the smaller the launch; more than 4 queues/streams oversubscribe the 4 hardware queues.
        //   DIRECT  8192x8192 (25 us): 1 queue 31.3 us, 2 queues 32.1;  8192x2048 (6.3 us): 8.31 / 7.88 / 8.21 us with
        //           1 / 2 / 4;  8192x1024 (3.1 us): 4.69 / 3.87 / 3.99;  4K sf=4 (1.3 us): 2.78 / 2.12 / 1.78 / 1.68 with
        //           1 / 2 / 3 / 4 -- but never 4 by default: a stream-ordered launch keeps the launch stream's queue
        //           active as well, and a fifth active queue makes the hardware scheduler time-slice (csic.h)
        //   HIP     8192x4096: 17.1 / 15.9 us with 1 / 2 chains;  8192x1024: 5.88 / 5.15 / 4.84 with 1 / 2 / 4;  4K sf=4: 3.67 / 3.40 / 2.98
        const double floor_us = (double)plan_algorithmic_bytes(plan) / 8.0e6;
        if (backend == CSIC_FRAME_GRAPH_DIRECT) branches = floor_us >= 10.0 ? 1 : floor_us >= 2.5 ? 2 : CSIC_FRAME_GRAPH_DEFAULT_QUEUES;
        else                                    branches = floor_us >= 5.0 ? 2 : CSIC_FRAME_GRAPH_DEFAULT_BRANCHES;
    }
    if (branches > cap) branches = cap;
    if (branches > nframes) branches = nframes;
    for (int k = 0; k < nframes; ++k)
        if (!d_in[k] || !d_out[k]) return set_error(CSIC_EINVAL_NULL, "frame %d: device buffer is NULL", k);
    CSIC_DEVICE_SCOPE(plan_device(plan));

    csic_frame_graph *g = new (std::nothrow) csic_frame_graph();
    if (!g) return set_error(CSIC_ENOMEM, "out of host memory");
    g->device = plan_device(plan);
    g->backend = backend;
    g->nframes = nframes;
    g->branches = branches;
    const int st = backend == CSIC_FRAME_GRAPH_DIRECT ? build_direct(g, plan, d_in, d_out)
                 : backend == CSIC_FRAME_GRAPH_FUSED ? build_fused(g, plan, d_in, d_out) : build_hip(g, plan, d_in, d_out);
    if (st != CSIC_OK) { graph_free(g); return st; }
    *out = g;
    clear_error();
    return CSIC_OK;
}

int csic_frame_graph_create(csic_plan *plan, const void *const *d_in, void *const *d_out, int32_t nframes,
                            int32_t branches, csic_frame_graph **out)
{
    return csic_frame_graph_create_ex(plan, d_in, d_out, nframes, branches, CSIC_FRAME_GRAPH_AUTO, out);
}

int csic_frame_graph_launch(csic_frame_graph *g, void *hip_stream)
{
    if (!g) return set_error(CSIC_EINVAL_NULL, "graph is NULL");
    CSIC_DEVICE_SCOPE(g->device);
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    if (g->backend == CSIC_FRAME_GRAPH_FUSED) {
        for (const LaunchDesc &d : g->fused) {
            const int st = enqueue(d, stream);
            if (st != CSIC_OK) return st;
        }
        for (const PlanarLaunchDesc &d : g->fused_planar) {
            const int st = planar_enqueue(d, stream);
            if (st != CSIC_OK) return st;
        }
        clear_error();
        return CSIC_OK;
    }
    if (g->backend == CSIC_FRAME_GRAPH_DIRECT) {
        int64_t t = 0;
        // Not capturable: the packets go to the library's queues NOW, only the hand-off would be recorded -- a replay of the
        // captured graph would run a hand-off without a submission behind it (stale outputs, a slot that is never released),
        // and the host-ordered fallback synchronises the stream, which invalidates a capture.  HIP and FUSED graphs capture.
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
        if (cap != hipStreamCaptureStatusNone)
            return set_error(CSIC_ECAPTURE, "a CSIC_FRAME_GRAPH_DIRECT launch cannot be captured into a hipGraph (the stream is "
                                            "capturing); use the CSIC_FRAME_GRAPH_HIP or CSIC_FRAME_GRAPH_FUSED backend under capture");
        // A gated submission sits in the rings until the stream opens the gate -- which it is only asked to do after the
        // submission has been written.  One that does not fit in a ring (gate + packets + closing packets) can therefore
        // not be armed ahead of time; it takes the host-ordered path.
        const std::vector<hsa_kernel_dispatch_packet_t> *pk = g->lbranches < g->branches ? g->lpackets : g->packets;
        size_t longest = 0;
        for (int j = 0; j < g->lbranches; ++j) longest = pk[j].size() > longest ? pk[j].size() : longest;
        const bool fits = longest + 3 <= QUEUE_PACKETS;
        if (!g->stream_ordered || !fits) {
            // no shared signals on this runtime (or a graph larger than the rings): order by the host
            HIP_TRY(hipStreamSynchronize(stream));
            int st = direct_submit(g, &t);
            if (st == CSIC_OK) st = direct_wait(g, t);
            if (st == CSIC_OK) clear_error();
            return st;
        }
        // Asynchronous and ordered with `stream`: the queues are armed behind a gate; the stream opens it when its
        // earlier work is done and then waits for every queue's closing packet.
        if (g->err_word && *g->err_word)
            return set_error(CSIC_EHIP, "direct dispatch: an earlier stream-ordered launch timed out (flags %u)", *g->err_word);
        int st = direct_submit(g, &t, true);
        if (st != CSIC_OK) return st;
        const int slot = (int)(t % DIRECT_SLOTS);
        const int nq = g->slot_queues[slot];
        if (g->kernel_handoff) {
            HandoffArgs ha{};
            ha.gate = g->sigmem[slot][0];
            for (int j = 0; j < nq; ++j) ha.done[j] = g->sigmem[slot][1 + j];
            ha.err = g->err_word;
            ha.passed = &g->passed_word[slot];
            ha.seq = (uint64_t)t + 1;
            ha.timeout_ticks = g->timeout_ticks;
            ha.nq = nq;
            void *params[1] = {&ha};
            hipError_t e = hipLaunchKernel(reinterpret_cast<const void *>(k_handoff), dim3(1), dim3(64), params, 0, stream);
            if (e != hipSuccess) {
                // The queues are armed behind the gate and no hand-off will ever open it: open it from the host (the frames
                // then run unordered with the stream, which the error return tells the caller) and wait here for EVERY
                // queue's closing packet -- in this mode queue 0's does not depend on the others -- so that the ticket
                // direct_submit issued is complete before anybody recycles its slot or frees the graph.
                (void)hipGetLastError();
                __atomic_store_n(g->sigmem[slot][0], (uint64_t)0, __ATOMIC_RELEASE);
                bool drained = true;
                for (int j = 0; j < nq; ++j)
                    if (hsa_signal_wait_scacquire(g->done[slot][j], HSA_SIGNAL_CONDITION_LT, 1, WAIT_TICKS, HSA_WAIT_STATE_BLOCKED) >= 1) drained = false;
                if (!drained) g->poisoned = true;
                return set_error(CSIC_EHIP, "launching the hand-off kernel failed: %s%s", hipGetErrorString(e),
                                 drained ? " (the frames ran unordered with the stream)" : "; the queues did not drain either");
            }
            g->slot_ticket[slot] = t;
            g->slot_on_stream[slot] = true;
            clear_error();
            return CSIC_OK;
        }
        hipError_t e = hipStreamWriteValue64(stream, g->sigmem[slot][0], 0, 0);
        if (e != hipSuccess) {
            // the queues are armed behind the gate: never leave them blocked -- open it from the host (the work then
            // runs unordered with the stream, which the error return tells the caller); the ticket completes through
            // queue 0's closing packet as a host-ordered submission's would
            (void)hipGetLastError();
            hsa_signal_store_screlease(g->gate[slot], 0);
            return set_error(CSIC_EHIP, "hipStreamWriteValue64 failed: %s", hipGetErrorString(e));
        }
        HIP_TRY(hipStreamWaitValue64(stream, g->sigmem[slot][1], 0, hipStreamWaitValueEq, ~0ull));   // queue 0's closing packet = all queues done
        HIP_TRY(hipEventRecord(g->consumed[slot], stream));
        g->slot_on_stream[slot] = true;
        clear_error();
        return CSIC_OK;
    }
    const int B = g->branches;
    if (B > 1) {
        HIP_TRY(hipEventRecord(g->fork, stream));
        for (int i = 1; i < B; ++i) HIP_TRY(hipStreamWaitEvent(g->streams[i], g->fork, 0));
    }
    for (int i = 0; i < B; ++i) HIP_TRY(hipGraphLaunch(g->execs[i], i == 0 ? stream : g->streams[i]));
    for (int i = 1; i < B; ++i) {
        HIP_TRY(hipEventRecord(g->joins[i], g->streams[i]));
        HIP_TRY(hipStreamWaitEvent(stream, g->joins[i], 0));
    }
    clear_error();
    return CSIC_OK;
}

int csic_frame_graph_submit(csic_frame_graph *g, int64_t *ticket)
{
    if (!g) return set_error(CSIC_EINVAL_NULL, "graph is NULL");
    if (g->backend != CSIC_FRAME_GRAPH_DIRECT)
        return set_error(CSIC_EINVAL_SIZE, "csic_frame_graph_submit needs a CSIC_FRAME_GRAPH_DIRECT graph (use csic_frame_graph_launch)");
    CSIC_DEVICE_SCOPE(g->device);
    const int st = direct_submit(g, ticket);
    if (st == CSIC_OK) clear_error();
    return st;
}

int csic_frame_graph_wait(csic_frame_graph *g, int64_t ticket)
{
    if (!g) return set_error(CSIC_EINVAL_NULL, "graph is NULL");
    if (g->backend != CSIC_FRAME_GRAPH_DIRECT)
        return set_error(CSIC_EINVAL_SIZE, "csic_frame_graph_wait needs a CSIC_FRAME_GRAPH_DIRECT graph (synchronize the stream instead)");
    if (g->next_ticket == 0) { clear_error(); return CSIC_OK; }
    const int st = direct_wait(g, ticket);
    if (st == CSIC_OK) clear_error();
    return st;
}

int csic_frame_graph_count(const csic_frame_graph *g, int32_t *nframes, int32_t *branches)
{
    if (!g) return set_error(CSIC_EINVAL_NULL, "graph is NULL");
    if (nframes) *nframes = g->nframes;
    if (branches) *branches = g->branches;
    clear_error();
    return CSIC_OK;
}

int csic_frame_graph_launch_branches(const csic_frame_graph *g)
{
    if (!g) return set_error(CSIC_EINVAL_NULL, "graph is NULL");
    if (g->backend == CSIC_FRAME_GRAPH_FUSED) return 1;
    return g->backend == CSIC_FRAME_GRAPH_DIRECT ? g->lbranches : g->branches;
}

int csic_frame_graph_backend(const csic_frame_graph *g)
{
    if (!g) return set_error(CSIC_EINVAL_NULL, "graph is NULL");
    return g->backend;
}

int csic_frame_graph_stream_ordered(const csic_frame_graph *g)
{
    if (!g) return set_error(CSIC_EINVAL_NULL, "graph is NULL");
    return (g->backend != CSIC_FRAME_GRAPH_DIRECT || g->stream_ordered) ? 1 : 0;
}

int csic_frame_graph_destroy(csic_frame_graph *g)
{
    if (!g) return CSIC_OK;
    DeviceGuard guard(g->device);
    if (g->backend == CSIC_FRAME_GRAPH_DIRECT && g->eng && g->next_ticket > g->waited && !g->poisoned &&
        direct_wait(g, -1) != CSIC_OK)
        g->poisoned = true;                     // something of this graph may still be queued or spinning: leak its device memory
    graph_free(g);
    return CSIC_OK;
}

} // extern "C"
