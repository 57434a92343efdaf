#!/usr/bin/env python3
"""Longest real document (2437 sentences) and the 3600-sentence cap through the recurrent taggers: bf16 (CU-pair MFMA kernels)
against fp32 (generic kernels) on the same weights."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaltopicsegmentation_amd import TextSegmenter  # noqa: E402

g = torch.Generator().manual_seed(0)
for arch, dims in (('BiLSTM', 768), ('BiLSTMLateFusion', [1024, 768]), ('biLSTMCRF', 768)):
    for L in (2437, 3600):
        lengths = torch.tensor([L, 359, 84])
        x1 = torch.randn(3, L, dims if isinstance(dims, int) else dims[0], generator=g).cuda()
        x2 = torch.randn(3, L, dims[1], generator=g).cuda() if not isinstance(dims, int) else None
        y = (torch.rand(3, L, generator=g) < .05).float().cuda()
        for b, n in enumerate(lengths.tolist()):
            x1[b, n:] = 0
            y[b, n:] = -1 if arch != 'biLSTMCRF' else 0
        out, sd = {}, None
        for dt in ('fp32', 'bf16'):
            ts = TextSegmenter(2, dims, 256, num_layers=2, architecture=arch, loss_fn='FocalLoss', compute_dtype=dt).cuda()
            if sd is None:
                sd = ts.model.state_dict()
            else:
                ts.model.load_state_dict(sd)
            args = (x1, x2, lengths) if x2 is not None else (x1, lengths)
            loss = ts.model.loss(*args, y.long() if arch == 'biLSTMCRF' else y)
            loss.backward()
            assert all(torch.isfinite(p.grad).all() for p in ts.model.parameters())
            res = ts.model(*args)
            out[dt] = (float(loss), res[1])
        agree = sum(a == b for da, db in zip(out['fp32'][1], out['bf16'][1]) for a, b in zip(da, db)) / float(sum(lengths))
        print(f'{arch:18s} L={L} loss fp32 {out["fp32"][0]:.5f} bf16 {out["bf16"][0]:.5f}  decode agreement {agree:.4f}', flush=True)
