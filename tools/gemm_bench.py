#!/usr/bin/env python3
"""Micro-benchmark of mts_gemm on the shapes of the BASELINE step (run on the GPU box; random data).
    python tools/gemm_bench.py [--reps 20] [--only NT|NN|TN]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaltopicsegmentation_amd import _lib as L, ops  # noqa: E402

SHAPES = [('NT', 16384, 5376, 1792), ('NT', 16384, 1792, 1792), ('NT', 16384, 256, 1792), ('NT', 16384, 1792, 256),
          ('NN', 16384, 1792, 5376), ('NN', 16384, 1792, 1792), ('NN', 16384, 256, 1792), ('NN', 16384, 1792, 256),
          ('TN', 5376, 1792, 16384), ('TN', 1792, 1792, 16384), ('TN', 1792, 256, 16384), ('TN', 256, 1792, 16384),
          ('NT', 8192, 8192, 8192), ('NT', 4096, 4096, 4096)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=20)
    ap.add_argument('--only', default=None)
    args = ap.parse_args()
    dev = 'cuda'
    for lay, M, N, K in SHAPES:
        if args.only and lay != args.only:
            continue
        g = torch.Generator(device=dev).manual_seed(1)
        if lay == 'NT':
            A, B = torch.randn(M, K, device=dev, generator=g), torch.randn(N, K, device=dev, generator=g)
        elif lay == 'NN':
            A, B = torch.randn(M, K, device=dev, generator=g), torch.randn(K, N, device=dev, generator=g)
        else:
            A, B = torch.randn(K, M, device=dev, generator=g), torch.randn(K, N, device=dev, generator=g)
        A, B = A.to(torch.bfloat16), B.to(torch.bfloat16)
        out = torch.empty(M, N, dtype=torch.float32 if lay == 'TN' else torch.bfloat16, device=dev)
        code = {'NT': L.NT, 'NN': L.NN, 'TN': L.TN}[lay]
        for _ in range(3):
            ops.gemm(code, A, B, out, M=M, N=N, K=K)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(args.reps):
            ops.gemm(code, A, B, out, M=M, N=N, K=K)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / args.reps
        print(f'{lay} M={M:6d} N={N:5d} K={K:6d}  {us:9.1f} us  {2.0 * M * N * K / us / 1e6:8.1f} TFLOP/s', flush=True)


if __name__ == '__main__':
    main()
