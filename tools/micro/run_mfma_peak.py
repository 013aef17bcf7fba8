#!/usr/bin/env python
"""Sustained MFMA ceiling on the box: register-only, with fragment reads from LDS, and with a barrier per 24 MFMAs."""
import ctypes, os, torch
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libmfma_peak.so"))
out = torch.zeros(1024, device="cuda")
ms = ctypes.c_float()
for name, mode, blocks, threads, reads in [("regs only, 4 waves/CU", 0, 256, 256, 0), ("regs only, 8 waves/CU", 0, 256, 512, 0), ("regs only, 8 waves/CU x4 blocks", 0, 1024, 512, 0),
                                           ("12 ds_read_b128 / 24 mfma, 4 waves", 1, 256, 256, 12), ("12 ds_read, 8 waves", 1, 256, 512, 12),
                                           ("12 ds_read + barrier, 4 waves", 2, 256, 256, 12), ("12 ds_read + barrier, 8 waves", 2, 256, 512, 12),
                                           ("12 ds_read + barrier, 2x4 waves", 2, 512, 256, 12)]:
  for rnd in (0, 1):
    iters = 2000
    rc = lib.run_mfma(mode, blocks, threads, iters, reads, rnd, ctypes.c_void_p(out.data_ptr()), ctypes.byref(ms))
    waves = blocks * threads // 64
    flops = waves * iters * 24 * 32 * 32 * 16 * 2
    print(f"{name:40s} rnd={rnd} rc={rc} {ms.value:8.3f} ms  {flops / ms.value / 1e9:8.1f} TFLOP/s (bf16 dense)  = {flops / ms.value / 1e9 / 6:6.1f} TF-eq split", flush=True)
