#!/usr/bin/env python3
"""A few launches of the BASELINE GEMMs (real epilogues) and of the vendor library on the same shapes, for
    rocprofv3 --pmc <counters> --output-format csv -d <dir> -- python3 tools/gemm_pmc.py
(stall / LDS / matrix-core counters; summarise with tools/pmc_summary.py).  No timing here."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaltopicsegmentation_amd import _lib as L, ops  # noqa: E402

dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(1)
M, D = 16384, 1792
x = torch.randn(M, D, device=dev, generator=g).to(torch.bfloat16)
res = torch.randn(M, D, device=dev, generator=g).to(torch.bfloat16)
wqkv = (torch.randn(3 * D, D, device=dev, generator=g) * 0.02).to(torch.bfloat16)
wo = (torch.randn(D, D, device=dev, generator=g) * 0.02).to(torch.bfloat16)
bq, bo = torch.randn(3 * D, device=dev, generator=g), torch.randn(D, device=dev, generator=g)
qkv = torch.empty(M, 3 * D, dtype=torch.bfloat16, device=dev)
y = torch.empty(M, D, dtype=torch.bfloat16, device=dev)
dq = torch.randn(M, 3 * D, device=dev, generator=g).to(torch.bfloat16)
gw = torch.empty(3 * D, D, device=dev)
blas = '--blas' in sys.argv
for _ in range(4):
    ops.linear_fwd(x, wqkv, bq, qkv, colscale=0.0668, ncols_scaled=D)        # NT 16384 x 5376 x 1792
    ops.linear_fwd(x, wo, bo, y, residual=res)                               # NT 16384 x 1792 x 1792
    ops.linear_dgrad(dq, wqkv, y, residual=res)                              # NN 16384 x 1792 x 5376
    ops.linear_wgrad(dq, x, gw)                                              # TN 5376 x 1792 x 16384 (+ split-K reduce)
    if blas:
        torch.matmul(x, wqkv.t(), out=qkv)
        torch.matmul(x, wo.t(), out=y)
        torch.matmul(dq, wqkv, out=y)
torch.cuda.synchronize()
