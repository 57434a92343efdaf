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
    ws = torch.empty(16 * 5376 * 1792 * 4, dtype=torch.uint8, device=dev)
    # (label, layout, M, N, K, A shape, B shape, C dtype code)
    for label, lay, M_, N, K_ in (('fwd QKV (NT)', 0, 16384, 5376, 1792), ('dgrad QKV (NN)', 1, 16384, 1792, 5376)):
        if lay == 0:
            Am, Bm = torch.randn(M_, K_, device=dev, generator=g), torch.randn(N, K_, device=dev, generator=g) * 0.02
        elif lay == 1:
            Am, Bm = torch.randn(M_, K_, device=dev, generator=g), torch.randn(K_, N, device=dev, generator=g) * 0.02
        else:
            Am, Bm = torch.randn(K_, M_, device=dev, generator=g) * 0.05, torch.randn(K_, N, device=dev, generator=g)
        Am, Bm = Am.to(torch.bfloat16), Bm.to(torch.bfloat16)
        out = torch.empty(M_, N, dtype=torch.float32 if lay == 2 else torch.bfloat16, device=dev)
        nwg = 4096
        stamps = torch.zeros(nwg * 8 * 8, dtype=torch.int64, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        for rep in range(6):
            lib.mts_gemm_set_stamps(stamps.data_ptr() if rep == 5 else None)
            rc = lib.mts_gemm(st, 1, 0 if lay == 2 else 1, lay, M_, N, K_, Am.data_ptr(), Am.stride(0), Bm.data_ptr(), Bm.stride(0), out.data_ptr(), N,
                              None, None, N, None, 0, 0, 0.0668, 1792, ws.data_ptr() if lay == 2 else None, ws.numel() if lay == 2 else 0)
            assert rc == 0, lib.mts_last_error()
        torch.cuda.synchronize()
        lib.mts_gemm_set_stamps(None)
        t = stamps.view(nwg, 8, 8).double().cpu()
        used = t[:, 0, 3] > 0
        t = t[used]
        tiles = t[:, :, :4].sum(-1).mean().item()
        print('--- %s  %s: %d workgroups stamped; cycles per workgroup and wave %.0f' % (path.split('/')[-1], label, int(used.sum()), tiles))
        tot = t[:, :, :4].sum()
        for i, n in enumerate(NAMES):
            print('   %-44s %5.1f %%   (wave 0 %5.1f %%  4 %5.1f %%)' % (n, 100 * t[:, :, i].sum().item() / tot.item(), 100 * t[:, 0, i].sum().item() / t[:, 0, :4].sum().item(),
                                                                      100 * t[:, 4, i].sum().item() / t[:, 4, :4].sum().item()), flush=True)
