#!/usr/bin/env python3
"""Time the embedding block's forward (input + position + type rows, LayerNorm; fp32 input -> bf16 pre + y) at the BASELINE shape, back to back."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaltopicsegmentation_amd import ops  # noqa: E402

dev = 'cuda'
B, Lq, D = 64, 256, 1792
rows = B * Lq
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(rows, D, device=dev, generator=g)
pos = torch.randn(Lq + 2, D, device=dev, generator=g) * 0.02
type0 = torch.randn(2, D, device=dev, generator=g) * 0.02
gamma = torch.rand(D, device=dev, generator=g) + 0.5
beta = torch.randn(D, device=dev, generator=g) * 0.1
y = torch.empty(rows, D, dtype=torch.bfloat16, device=dev)
pre = torch.empty_like(y)
mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
for name, xi in (('fp32 input', x), ('bf16 input', x.to(torch.bfloat16))):
    run = lambda: ops.embed_layernorm_fwd(xi.view(B, Lq, D), pos, 2, type0, gamma, beta, 1e-5, y, pre, mean, rstd)
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
    mb = (xi.numel() * xi.element_size() + 2 * y.numel() * 2) / 1e6      # MB / us = TB/s
    print(f'embed_layernorm_fwd {rows} x {D} {name}: {sorted(ts)[2]:.1f} us  ({mb / sorted(ts)[2]:.2f} TB/s for {mb:.0f} MB)  [{min(ts):.1f} .. {max(ts):.1f}]')
