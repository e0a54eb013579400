#!/usr/bin/env python3
"""tools/probe_pin.py -- what pinning a 35 MB frame slot costs, and whether transparent huge pages make it cheaper:
hipHostMalloc(mapped) against aligned_alloc + madvise(MADV_HUGEPAGE) + first touch + hipHostRegister(mapped), one thread and 16 at once."""
import ctypes as C
import threading
import time

hip = C.CDLL("libamdhip64.so")
libc = C.CDLL("libc.so.6", use_errno=True)
libc.aligned_alloc.restype = C.c_void_p
libc.aligned_alloc.argtypes = [C.c_size_t, C.c_size_t]
libc.free.argtypes = [C.c_void_p]
libc.madvise.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
libc.memset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipHostFree.argtypes = [C.c_void_p]
hip.hipHostRegister.argtypes = [C.c_void_p, C.c_size_t, C.c_uint]
hip.hipHostUnregister.argtypes = [C.c_void_p]
hip.hipHostGetDevicePointer.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint]
SIZE = 3840 * 2160 * 4 + 960 * 540 * 4
SIZE2M = (SIZE + (2 << 20) - 1) & ~((2 << 20) - 1)
MAPPED, REG_MAPPED, MADV_HUGEPAGE = 0x2, 0x2, 14
print("THP:", open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip(), "| defrag:", open("/sys/kernel/mm/transparent_hugepage/defrag").read().strip())
assert hip.hipSetDevice(0) == 0


def host_malloc(out):
    t0 = time.perf_counter()
    p = C.c_void_p()
    assert hip.hipHostMalloc(C.byref(p), SIZE, MAPPED) == 0
    t1 = time.perf_counter()
    libc.memset(p, 1, SIZE)
    t2 = time.perf_counter()
    assert hip.hipHostFree(p) == 0
    t3 = time.perf_counter()
    out.append((t1 - t0, t2 - t1, t3 - t2))


def register(out, huge=True):
    t0 = time.perf_counter()
    p = libc.aligned_alloc(2 << 20, SIZE2M)
    if huge:
        libc.madvise(p, SIZE2M, MADV_HUGEPAGE)
    libc.memset(p, 0, SIZE2M)                       # first touch: the pages exist before they are pinned
    t1 = time.perf_counter()
    rc = hip.hipHostRegister(p, SIZE2M, REG_MAPPED)
    assert rc == 0, rc
    d = C.c_void_p()
    assert hip.hipHostGetDevicePointer(C.byref(d), p, 0) == 0
    t2 = time.perf_counter()
    assert hip.hipHostUnregister(p) == 0
    libc.free(p)
    t3 = time.perf_counter()
    out.append((t1 - t0, t2 - t1, t3 - t2, d.value == p))


for name, fn in (("hipHostMalloc(mapped) / memset / hipHostFree", host_malloc), ("alloc+THP+touch / hipHostRegister / unregister+free", register),
                 ("alloc+touch (no THP) / hipHostRegister / unregister+free", lambda o: register(o, False))):
    for nthreads in (1, 16):
        res = []
        for rep in range(3):
            outs = [[] for _ in range(nthreads)]
            ths = [threading.Thread(target=fn, args=(o,)) for o in outs]
            t0 = time.perf_counter()
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            wall = time.perf_counter() - t0
            avg = [sum(o[0][k] for o in outs) / nthreads for k in range(3)]
            res.append((wall, avg, outs[0][0][3:] if len(outs[0][0]) > 3 else ()))
        wall, avg, extra = min(res)
        print(f"{name}: {nthreads:2d} thread(s): wall {1e3 * wall:7.2f} ms; per thread {1e3 * avg[0]:7.2f} / {1e3 * avg[1]:7.2f} / {1e3 * avg[2]:7.2f} ms {extra}", flush=True)
