#!/usr/bin/env python3
"""Per-phase cycle sums of the 256x224 K loop (libraries from gemm224_phases.sh): mean over workgroups of each wave's cycles per
K-tile between the stamps of the mid-tile-barrier schedule."""
import ctypes as C
import sys

import torch

vp, i32, f32, u32, sz = C.c_void_p, C.c_int, C.c_float, C.c_uint, C.c_size_t
dev = 'cuda'
M, K = 16384, 1792
g = torch.Generator(device=dev).manual_seed(1)
A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
NAMES = ['b1xaA, b1xaB (24 MFMA) + reads landed', 'copies landed', 'barrier', 'b0xaA, b0xaB (32 MFMA, 8 copies)']
for path in [a for a in sys.argv[1:] if a.endswith('.so')]:
    lib = C.CDLL(path)
    lib.mts_last_error.restype = C.c_char_p
    lib.mts_gemm.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, i32, vp, i32, vp, i32, vp, vp, i32, vp, i32, u32, f32, i32, vp, sz]
    lib.mts_gemm_set_stamps.argtypes = [vp]
    for N, epi, label in ((5376, 1 | 8, 'fwd QKV'),):
        Bm = (torch.randn(N, K, device=dev, generator=g) * 0.02).to(torch.bfloat16)
        bias = torch.randn(N, device=dev, generator=g)
        res = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        nwg = (M // 256) * (N // 224)
        stamps = torch.zeros(nwg * 8 * 8, dtype=torch.int64, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        for rep in range(10):
            lib.mts_gemm_set_stamps(stamps.data_ptr() if rep == 9 else None)
            rc = lib.mts_gemm(st, 1, 1, 0, M, N, K, A.data_ptr(), K, Bm.data_ptr(), K, out.data_ptr(), N, bias.data_ptr(),
                              res.data_ptr() if epi & 2 else None, N, None, 0, epi, 0.0668, 1792, None, 0)
            assert rc == 0, lib.mts_last_error()
        torch.cuda.synchronize()
        lib.mts_gemm_set_stamps(None)
        t = stamps.view(nwg, 8, 8).double().cpu() / (K // 64)
        print('--- %s  %s: cycles per K-tile and wave %.0f' % (path.split('/')[-1], label, t[:, :, :4].sum(-1).mean().item()))
        for i, n in enumerate(NAMES):
            print('   %-24s %7.0f   (wave 0 %6.0f  1 %6.0f  4 %6.0f  7 %6.0f)' % (n, t[:, :, i].mean().item(), t[:, 0, i].mean().item(), t[:, 1, i].mean().item(),
                                                                             t[:, 4, i].mean().item(), t[:, 7, i].mean().item()), flush=True)
