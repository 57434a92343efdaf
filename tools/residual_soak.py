#!/usr/bin/env python3
"""Soak of the residual-through-the-LDS epilogues (gemm_bf16_224n_kernel, gemm_bf16_224d_kernel): the step's shapes, 300 launches each, every result compared
bit for bit with the eight-wave kernel's (gemm_variant 6), with a memory-bound kernel in between so that launch-to-launch timing varies."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaltopicsegmentation_amd import _lib as L, ops  # noqa: E402

dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(3)
junk = torch.empty(64 << 20, device=dev)
bad = 0
for lay, M, N, K, bias in (('NN', 16384, 1792, 5376, False), ('NT', 16384, 1792, 1792, True), ('NN', 16384, 1792, 1792, False), ('NT', 2048, 448, 256, True)):
    A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
    B = (torch.randn(K, N, device=dev, generator=g) if lay == 'NN' else torch.randn(N, K, device=dev, generator=g)).to(torch.bfloat16)
    R = torch.randn(M, N, device=dev, generator=g).to(torch.bfloat16)
    b = torch.randn(N, device=dev, generator=g) if bias else None
    ref = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    L.check(L.lib.mts_set_option(b'gemm_tile', 224))
    L.check(L.lib.mts_set_option(b'gemm_variant', 6))
    ops.gemm(getattr(L, lay), A, B, ref, M=M, N=N, K=K, bias=b, residual=R)
    L.check(L.lib.mts_set_option(b'gemm_variant', 0))
    out = torch.empty_like(ref)
    n_bad = 0
    for it in range(300):
        out.fill_(float('nan'))
        if it % 3 == 0:
            junk.mul_(1.0001)
        ops.gemm(getattr(L, lay), A, B, out, M=M, N=N, K=K, bias=b, residual=R)
        if not torch.equal(out.view(torch.int16), ref.view(torch.int16)):
            n_bad += 1
    L.check(L.lib.mts_set_option(b'gemm_tile', 0))
    print(f'{lay} {M} x {N} x {K} bias={int(bias)} residual=1: {300 - n_bad} / 300 launches bitwise equal to the eight-wave kernel', flush=True)
    bad += n_bad
sys.exit(1 if bad else 0)
