#!/usr/bin/env python3
"""A/B of whole library builds in ONE process on the BASELINE GEMM shapes with their real epilogues (box-to-box clock differences
are larger than most kernel changes, so builds are only ever compared inside one run, interleaved):

    python tools/gemm_ab_libs.py tools/ab_libs/libmts_r1.so multimodaltopicsegmentation_amd/libmts_hip.so [--blas]

Per shape and library: best of 4 repetitions of 20 back-to-back launches, and a sustained figure (300 launches)."""
import ctypes as C
import sys

import torch

vp, i32, f32, u32, sz = C.c_void_p, C.c_int, C.c_float, C.c_uint, C.c_size_t
libs = []
for spec in [a for a in sys.argv[1:] if not a.startswith('--')]:
    path, _, var = spec.partition('@')          # lib.so@5 = that library with mts_set_option("gemm_variant", 5); lib.so@k256: gemm_big_min_k
    lib = C.CDLL(path)
    lib.mts_set_option.argtypes = [C.c_char_p, i32]
    lib.mts_last_error.restype = C.c_char_p
    lib.mts_gemm.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, i32, vp, i32, vp, i32, vp, vp, i32, vp, i32, u32, f32, i32, vp, sz]
    libs.append((path.split('/')[-1] + ('@' + var if var else ''), lib, var or None))
blas = '--blas' in sys.argv
dev = 'cuda'
# (label, layout, M, N, K, epilogue flags)   flags: 1 bias, 2 residual, 8 column scale
SHAPES = [('fwd QKV', 0, 16384, 5376, 1792, 1 | 8), ('fwd out-proj', 0, 16384, 1792, 1792, 1 | 2),
          ('dgrad QKV', 1, 16384, 1792, 5376, 2), ('dgrad out-proj', 1, 16384, 1792, 1792, 0),
          ('wgrad QKV', 2, 5376, 1792, 16384, 0), ('wgrad out-proj', 2, 1792, 1792, 16384, 0),
          ('fwd FFN down', 0, 16384, 1792, 256, 1 | 2), ('dgrad FFN up', 1, 16384, 1792, 256, 2),
          ('fwd FFN up', 0, 16384, 256, 1792, 1), ('dgrad FFN down', 1, 16384, 256, 1792, 0),
          ('wgrad FFN up', 2, 256, 1792, 16384, 0), ('wgrad FFN down', 2, 1792, 256, 16384, 0)]
if '--ffn' in sys.argv:
    SHAPES = SHAPES[6:]
g = torch.Generator(device=dev).manual_seed(1)
ws = torch.empty(16 * 5376 * 1792 * 4, dtype=torch.uint8, device=dev)
for label, lay, M, N, K, epi in SHAPES:
    if lay == 0:
        A, B = torch.randn(M, K, device=dev, generator=g), torch.randn(N, K, device=dev, generator=g) * 0.02
    elif lay == 1:
        A, B = torch.randn(M, K, device=dev, generator=g), torch.randn(K, N, device=dev, generator=g) * 0.02
    else:
        A, B = torch.randn(K, M, device=dev, generator=g) * 0.05, torch.randn(K, N, device=dev, generator=g)
    A, B = A.to(torch.bfloat16), B.to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.float32 if lay == 2 else torch.bfloat16, device=dev)
    bias = torch.randn(N, device=dev, generator=g)
    res = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16)
    st = torch.cuda.current_stream().cuda_stream

    def make(lib, var):
        def run():
            if var is not None:
                if var.startswith('k'):
                    lib.mts_set_option(b'gemm_big_min_k', int(var[1:]))
                else:
                    lib.mts_set_option(b'gemm_big_min_k', 512)
                    lib.mts_set_option(b'gemm_variant', int(var))
            rc = lib.mts_gemm(st, 1, 0 if lay == 2 else 1, lay, M, N, K, A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), out.data_ptr(), N,
                              bias.data_ptr() if epi & 1 else None, res.data_ptr() if epi & 2 else None, N, None, 0, epi, 0.0668, 1792,
                              ws.data_ptr() if lay == 2 else None, ws.numel() if lay == 2 else 0)
            assert rc == 0, lib.mts_last_error()
        return run
    runs = [(n, make(l, v)) for n, l, v in libs]
    if blas:
        if lay == 0:
            runs.append(('hipBLASLt (torch.matmul, no epilogue)', lambda: torch.matmul(A, B.t(), out=out)))
        elif lay == 1:
            runs.append(('hipBLASLt (torch.matmul, no epilogue)', lambda: torch.matmul(A, B, out=out)))
        else:
            o16 = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            runs.append(('hipBLASLt (torch.matmul, bf16 out)', lambda: torch.matmul(A.t(), B, out=o16)))

    def timed(fn, n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fn()
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) * 1e3 / n
    best = {n: 1e30 for n, _ in runs}
    for rep in range(4):
        for n, fn in runs:
            fn(); fn()
            best[n] = min(best[n], timed(fn, 20))
    sus = {n: timed(fn, 300) for n, fn in runs}
    ref = None
    vals = []
    for n, fn in runs:                       # results must agree between builds
        fn()
        torch.cuda.synchronize()
        cur = out.float().clone()
        if ref is None:
            ref = cur
        elif 'BLAS' not in n:
            vals.append(float((cur - ref).abs().max()))
    print(f'{label:15s} M={M:6d} N={N:5d} K={K:6d}  ' + '  |  '.join(
        f'{n}: {best[n]:6.1f} us {2.0 * M * N * K / best[n] / 1e6:5.0f} TF/s (sustained {sus[n]:6.1f})' for n, _ in runs) +
        f'  max |diff| between builds {max(vals) if vals else 0:.3g}', flush=True)
