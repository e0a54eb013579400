#!/usr/bin/env python3
"""Known-byte-count launches for calibrating FETCH_SIZE / WRITE_SIZE on gfx950 in OUR access patterns
(MI355X_MICROARCH.md: FETCH_SIZE reads 1/2 of a wide coalesced stream; other widths uncalibrated).
  k_synth     : 256 MiB of 4-byte-per-lane dense stores, no loads
  k_checksum  : 256 MiB of 4-byte-per-lane dense loads, no stores
  torch clone : 256 MiB read + 256 MiB written, 16 B per lane
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import csic_amd as csic

N = csic._native
lib = N.lib()
npix = 8192 * 8192
a = torch.empty(npix, dtype=torch.int32, device="cuda:0")
sh = C.c_void_p(torch.cuda.current_stream(0).cuda_stream)
for k in range(4):
    N.check(lib.csic_synth_frame_device(C.c_void_p(a.data_ptr()), npix, k * npix, 1, sh))
torch.cuda.synchronize()
s = C.c_uint64()
for k in range(4):
    N.check(lib.csic_checksum_device(C.c_void_p(a.data_ptr()), npix, C.byref(s), sh))
for k in range(4):
    b = a.clone()
torch.cuda.synchronize()
print("calibration launches done", s.value)
