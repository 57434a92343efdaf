#!/usr/bin/env python3
"""Every embedding width of the reference's table (train_fit.py:245-250) and their early-fusion sums, 8 heads, windows 30 and
120 (the default), bf16 vs fp32 on the same weights: runs, finite, and close."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaltopicsegmentation_amd import TextSegmenter  # noqa: E402

g = torch.Generator().manual_seed(0)
for D in (512, 768, 1024, 1536, 1792, 2304, 2560):
    for window in (30, 120):
        x = torch.randn(2, 300, D, generator=g).cuda()
        lengths = torch.tensor([300, 177])
        y = (torch.rand(2, 300, generator=g) < .2).float().cuda()
        out = {}
        sd = None
        for dt in ('fp32', 'bf16'):
            ts = TextSegmenter(2, D, 25, num_layers=1, architecture='Transformer', loss_fn='FocalLoss', nheads=8, attention_window=window,
                               compute_dtype=dt).cuda()
            if sd is None:
                sd = ts.model.state_dict()
            else:
                ts.model.load_state_dict(sd)
            loss = ts.model.loss(x, lengths, y)
            loss.backward()
            out[dt] = float(loss)
            assert all(torch.isfinite(p.grad).all() for p in ts.model.parameters())
        print(f'D={D:5d} hd={D // 8:4d} window={window:4d} loss fp32 {out["fp32"]:.6f} bf16 {out["bf16"]:.6f}', flush=True)
