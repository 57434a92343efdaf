#!/usr/bin/env python3
"""The fused feed-forward block's weight-gradient shapes (TN, K = 16 384 tokens) on the 128-wide and the 224-wide tile at forced K splits, back to back
(what mts_wgrad_pair was sized from: one 224-tile launch at 16 slices fills half the chip)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaltopicsegmentation_amd import _lib as L, ops
from tools.blas_compare_util import timeit
dev='cuda'
g=torch.Generator(device=dev).manual_seed(1)
K=16384
for (M,N) in [(256,1792),(1792,256)]:
    A=torch.randn(K,M,device=dev,generator=g).to(torch.bfloat16)
    B=torch.randn(K,N,device=dev,generator=g).to(torch.bfloat16)
    out=torch.empty(M,N,device=dev)
    for tile,sp in [(0,0),(224,8),(224,16),(224,12),(128,16),(128,8)]:
        if tile==224 and N%224: continue
        L.check(L.lib.mts_set_option(b'gemm_tile',tile)); L.check(L.lib.mts_set_option(b'gemm_splits',sp))
        try:
            t=timeit(lambda: ops.gemm(L.TN,A,B,out,M=M,N=N,K=K))
        finally:
            L.check(L.lib.mts_set_option(b'gemm_tile',0)); L.check(L.lib.mts_set_option(b'gemm_splits',0))
        print(f'TN M={M} N={N} K={K} tile={tile} splits={sp}: {t:.1f} us (GEMM + reduce)', flush=True)
