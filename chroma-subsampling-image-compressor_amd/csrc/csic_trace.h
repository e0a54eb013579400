// csic_trace.h -- optional roctx ranges around the host-visible phases of the I/O paths (decode / H2D / kernel / D2H / GPU wait /
// encode), the analogue of the reference's WriteVcdAnnotation (ImageCompressorTopApp.scala:67): a timeline a user can look at.
// Off by default and free when off (one relaxed load per range); CSIC_ROCTX=1 in the environment turns it on, and the ranges
// show up in `rocprofv3 --marker-trace`.  The marker library (librocprofiler-sdk-roctx.so, ROCm >= 6.2; libroctx64.so before)
// is looked up at run time: libcsic_hip.so has no link-time dependency on a profiler.  Implementation: csic_host.cpp.
#pragma once

namespace csic {
namespace trace {

bool enabled();                 // CSIC_ROCTX set to something other than "0" AND the marker library found (checked once)
void push(const char *name);    // roctxRangePushA
void pop();                     // roctxRangePop

struct Range {
    explicit Range(const char *name) : on(enabled()) { if (on) push(name); }
    ~Range() { if (on) pop(); }
    Range(const Range &) = delete;
    Range &operator=(const Range &) = delete;
    bool on;
};

} // namespace trace

// CPU time this process may really use: min(CPUs in its affinity mask, its cgroup's cpu.max / cfs quota rounded up), >= 1.
// A GPU box shows 256 CPUs and grants a job the time of 16 (profiles/r03_host_io.json); worker pools size themselves by this.
int host_cpu_budget();

} // namespace csic
