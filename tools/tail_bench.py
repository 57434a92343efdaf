#!/usr/bin/env python3
"""Time mts_layernorm_loss_tail (the last layer's LayerNorm + head + loss + backward in one pass) at the BASELINE shape, back to back."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaltopicsegmentation_amd import _lib as L, ops  # noqa: E402

dev = 'cuda'
B, Lq, D = 64, 256, 1792
rows = B * Lq
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(rows, D, device=dev, generator=g).to(torch.bfloat16)
gamma = torch.rand(D, device=dev, generator=g) + 0.5
beta = torch.randn(D, device=dev, generator=g) * 0.1
hw = torch.randn(1, D, device=dev, generator=g) * 0.05
hb = torch.zeros(1, device=dev)
y = (torch.rand(B, Lq, device=dev, generator=g) < 0.3).float()
lengths = torch.full((B,), Lq, dtype=torch.int32, device=dev)
scores = torch.empty(rows, 1, device=dev)
loss = torch.zeros(2, device=dev)
dx = torch.empty_like(x)
dg, db, dxs, dhw, dhb = (torch.empty(D, device=dev), torch.empty(D, device=dev), torch.empty(D, device=dev), torch.empty(1, D, device=dev), torch.empty(1, device=dev))


def run():
    ops.layernorm_loss_tail(L.LOSS_FOCAL, x, gamma, beta, 1e-5, hw, hb, y, lengths, 0.25, 2.0, 1.0, scores, loss, dx, dg, db, dxs, dhw, dhb, (B, Lq))


for _ in range(20):
    run()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50):
        run()
    e.record()
    torch.cuda.synchronize()
    ts.append(s.elapsed_time(e) * 1e3 / 50)
print(f'layernorm_loss_tail {rows} x {D} bf16: {sorted(ts)[2]:.1f} us per call (2 launches)  [{min(ts):.1f} .. {max(ts):.1f}]  loss {float(loss[0]):.6f}')
