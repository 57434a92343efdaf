#!/usr/bin/env python3
"""Skinny FFN GEMMs with the step's epilogues, warm (same buffers) vs cold (rotating through > 256 MiB of operands)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaltopicsegmentation_amd import _lib as L, ops  # noqa: E402

dev = 'cuda'
M = 16384
NB = 8   # rotating buffer sets


def run(name, fn):
    for i in range(3):
        fn(i % NB)
    torch.cuda.synchronize()
    for mode, sel in (('warm', lambda i: 0), ('cold', lambda i: i % NB)):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for i in range(24):
            fn(sel(i))
        e.record()
        torch.cuda.synchronize()
        print(f'{name:34s} {mode}: {s.elapsed_time(e) * 1e3 / 24:7.1f} us', flush=True)


x1792 = [torch.randn(M, 1792, device=dev).to(torch.bfloat16) for _ in range(NB)]
x256 = [torch.randn(M, 256, device=dev).to(torch.bfloat16) for _ in range(NB)]
o1792 = [torch.empty(M, 1792, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
o256 = [torch.empty(M, 256, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
a256 = [torch.empty(M, 256, device=dev, dtype=torch.bfloat16) for _ in range(NB)]
wi = torch.randn(256, 1792, device=dev).to(torch.bfloat16)
wo = torch.randn(1792, 256, device=dev).to(torch.bfloat16)
bi = torch.randn(256, device=dev)
bo = torch.randn(1792, device=dev)
run('NT N=256 K=1792 plain', lambda i: ops.gemm(L.NT, x1792[i], wi, o256[i], M=M, N=256, K=1792))
run('NT N=256 K=1792 bias', lambda i: ops.gemm(L.NT, x1792[i], wi, o256[i], M=M, N=256, K=1792, bias=bi))
run('NT N=256 K=1792 bias+gelu+aux', lambda i: ops.gemm(L.NT, x1792[i], wi, o256[i], M=M, N=256, K=1792, bias=bi, gelu=True, aux=a256[i]))
run('NT N=1792 K=256 plain', lambda i: ops.gemm(L.NT, x256[i], wo, o1792[i], M=M, N=1792, K=256))
run('NT N=1792 K=256 bias+residual', lambda i: ops.gemm(L.NT, x256[i], wo, o1792[i], M=M, N=1792, K=256, bias=bo, residual=x1792[i]))
run('NN N=256 K=1792 plain', lambda i: ops.gemm(L.NN, x1792[i], wo, o256[i], M=M, N=256, K=1792))
run('NN N=1792 K=256 plain', lambda i: ops.gemm(L.NN, x256[i], wi, o1792[i], M=M, N=1792, K=256))
run('NN N=1792 K=256 +residual', lambda i: ops.gemm(L.NN, x256[i], wi, o1792[i], M=M, N=1792, K=256, residual=x1792[i]))
