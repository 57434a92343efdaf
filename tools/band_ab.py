#!/usr/bin/env python3
"""Band attention backward: one-pass kernel against the two-kernel form (mts_set_option("band_fused_bwd", 1 | 0)) in one process, at the
BASELINE shape (64 x 256 x 1792, 8 heads, radius 15) or MTS_B / MTS_L.  Prints us per call, interleaved best of 3 x 20."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaltopicsegmentation_amd import _lib as L, ops
B, Lq, D, heads, radius = int(os.environ.get('MTS_B', 64)), int(os.environ.get('MTS_L', 256)), 1792, 8, 15
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(0)
qkv = (torch.randn(B * Lq, 3 * D, device=dev, generator=g) * 0.5).to(torch.bfloat16)
dctx = torch.randn(B * Lq, D, device=dev, generator=g).to(torch.bfloat16)
slots = ops.band_slots(radius)
ctx = torch.empty(B * Lq, D, dtype=torch.bfloat16, device=dev)
probs = torch.empty(B * Lq, heads * slots, device=dev)
ops.band_attn_fwd(qkv, None, B, Lq, D, heads, radius, ctx, probs)
dqkv = torch.empty(B * Lq, 3 * D, dtype=torch.bfloat16, device=dev)
dsc = torch.empty_like(probs)
dbias = torch.empty(3 * D, device=dev)
def t(mode):
    L.check(L.lib.mts_set_option(b'band_fused_bwd', mode))
    fn = lambda: ops.band_attn_bwd(qkv, None, probs, dctx, B, Lq, D, heads, radius, dqkv, dsc, dbias=dbias)
    best = 1e9
    for _ in range(3):
        fn(); fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / 20)
    return best
res = {}
for rep in range(2):
    for mode in (1, 0):
        res.setdefault(mode, []).append(t(mode))
mb = B * Lq * (4 * D * 2 + heads * slots * 4 + 3 * D * 2) / 1e6
print(f'B={B} L={Lq}: fused {min(res[1]):.1f} us, two kernels {min(res[0]):.1f} us  (one pass over q,k,v,dctx,probs + dqkv = {mb:.0f} MB -> {mb / min(res[1]):.2f} TB/s)', flush=True)
