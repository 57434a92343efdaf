#!/usr/bin/env python3
"""Weight-gradient (TN) GEMMs of the step: time per forced K split under the 256x224 tile, in-launch combine on / off, and the planner's own
choice.  One process (boxes differ by more than most changes).    python tools/tn_sweep.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaltopicsegmentation_amd import _lib as L, ops  # noqa: E402

dev = 'cuda'


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / n)
    return best


for M, N, K in ((5376, 1792, 16384), (1792, 1792, 16384), (2048, 1792, 16384), (2048, 1024, 32768), (2048, 512, 16384)):
    g = torch.Generator(device=dev).manual_seed(1)
    A = torch.randn(K, M, device=dev, generator=g).to(torch.bfloat16)
    B = torch.randn(K, N, device=dev, generator=g).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev)
    run = lambda: ops.gemm(L.TN, A, B, out, M=M, N=N, K=K)
    L.check(L.lib.mts_set_option(b'gemm_tile', 0)); L.check(L.lib.mts_set_option(b'gemm_splits', 0)); L.check(L.lib.mts_set_option(b'gemm_variant', 0))
    t0 = timeit(run)
    tile, sp = ctypes.c_int(0), ctypes.c_int(0)
    L.lib.mts_gemm_last_plan(ctypes.byref(tile), ctypes.byref(sp))
    print(f'TN {M}x{N}x{K}: planner -> tile {tile.value} splits {sp.value}: {t0:7.1f} us ({2.0 * M * N * K / t0 / 1e6:6.1f} TF/s)', flush=True)
    if N % 224:
        continue
    L.check(L.lib.mts_set_option(b'gemm_tile', 224))
    for splits in (1, 2, 3, 4, 5, 6, 8, 9, 12, 16):
        L.check(L.lib.mts_set_option(b'gemm_splits', splits))
        row = []
        for variant, combine in ((0, 1), (0, 0), (6, 1)):
            L.check(L.lib.mts_set_option(b'gemm_variant', variant)); L.check(L.lib.mts_set_option(b'gemm_combine', combine))
            t = timeit(run)
            L.lib.mts_gemm_last_plan(ctypes.byref(tile), ctypes.byref(sp))
            row.append(f'{t:7.1f}')
        print(f'   forced splits {splits:2d} (ran {sp.value:2d}): four-wave + combine {row[0]} | four-wave + reduce launch {row[1]} | eight-wave + reduce launch {row[2]} us', flush=True)
    L.check(L.lib.mts_set_option(b'gemm_combine', 1)); L.check(L.lib.mts_set_option(b'gemm_variant', 0))
L.check(L.lib.mts_set_option(b'gemm_tile', 0)); L.check(L.lib.mts_set_option(b'gemm_splits', 0))
