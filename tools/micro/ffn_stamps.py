#!/usr/bin/env python3
"""Where a step of the fused feed-forward kernel spends its cycles (library built with -DFFN_STAMPS by ffn_variants.sh):
    python tools/micro/ffn_stamps.py tools/ab_libs/libffn_STAMPS.so [--M 16384]"""
import ctypes as C
import sys

import torch

vp, i32 = C.c_void_p, C.c_int
lib = C.CDLL(sys.argv[1])
M = int(sys.argv[sys.argv.index('--M') + 1]) if '--M' in sys.argv else 16384
D, F, dev = 1792, 256, 'cuda'
Mp = (M + 63) // 64 * 64
bf = dict(dtype=torch.bfloat16, device=dev)
a1 = torch.randn(M, D, **bf)
w1, w2 = (torch.randn(F, D, device=dev) * D ** -0.5).to(torch.bfloat16), (torch.randn(D, F, device=dev) * F ** -0.5).to(torch.bfloat16)
b1, b2 = torch.randn(F, device=dev), torch.randn(D, device=dev)
u, f = torch.zeros(Mp, F, **bf), torch.zeros(Mp, F, **bf)
s2 = torch.zeros(Mp, D, **bf)
st = torch.cuda.current_stream().cuda_stream
lib.mts_ffn_fwd.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, vp, i32, vp, vp, vp]
lib.mts_ffn_bwd_data.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, i32, vp, vp]
lib.mts_ffn_set_stamps.argtypes = [vp]
nwg = Mp // 64
stamps = torch.zeros(nwg, 12, 16, dtype=torch.int64, device=dev)
NAMES = {0: 'A copies issued', 1: 'A operand reads', 2: 'A MFMA issue', 3: 'A wait copies', 4: 'A barrier', 5: 'A epilogue issue', 6: 'A drain+barrier',
         8: 'B copies (+epi loads) issued', 9: 'B operand reads', 10: 'B MFMA issue', 11: 'B (epilogue+) wait', 12: 'B barrier'}
for which in ('fwd', 'bwd'):
    for rep in range(20):
        lib.mts_ffn_set_stamps(stamps.data_ptr() if rep == 19 else None)
        if which == 'fwd':
            rc = lib.mts_ffn_fwd(st, M, D, F, a1.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), 0, u.data_ptr(), f.data_ptr(), s2.data_ptr())
        else:
            rc = lib.mts_ffn_bwd_data(st, M, D, F, a1.data_ptr(), w1.data_ptr(), w2.data_ptr(), u.data_ptr(), 0, f.data_ptr(), s2.data_ptr())
        assert rc == 0
    torch.cuda.synchronize()
    t = stamps.double().cpu()
    tot = t[:, :8].sum(-1).mean().item()
    print('--- %s  M=%d: mean cycles per wave %.0f (s_memtime ticks; 28 + 28 steps)' % (which, M, tot))
    for i, n in NAMES.items():
        steps = 28 if i not in (5, 6) else 1
        print('   %-30s total %8.0f  per step %7.0f   (wave 0: %7.0f  wave 7: %7.0f  copy wave 8: %7.0f  11: %7.0f)' % (n, t[:, :8, i].mean().item(), t[:, :8, i].mean().item() / steps,
                                                                                        t[:, 0, i].mean().item() / steps, t[:, 7, i].mean().item() / steps, t[:, 8, i].mean().item() / steps, t[:, 11, i].mean().item() / steps))
