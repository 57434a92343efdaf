#!/usr/bin/env python3
"""In-kernel timeline of the 256x224 GEMM (diagnostic build, -DMTS_GEMM_STAMPS: the production library carries no stamps).
Builds gpurun_out/diag/libmts_diag.so on the GPU box, runs the BASELINE forward projections and prints, per tile round, the
median over workgroups of: K-loop cycles, cycles to issue the next tile's first DMAs, cycles to issue the stores, cycles spent in
the vmcnt(0) + barrier behind them, and the shader clock (s_memtime / s_memrealtime).

    python tools/gemm_stamps.py [variant ...]"""
import ctypes as C
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'multimodaltopicsegmentation_amd', 'csrc')
OUT = os.path.join(ROOT, 'gpurun_out', 'diag')
os.makedirs(OUT, exist_ok=True)
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wno-unused-result', '-DMTS_GEMM_STAMPS']
srcs = [f for f in sorted(os.listdir(CSRC)) if f.endswith('.hip')]


def cc(f):
    o = os.path.join(OUT, f.replace('.hip', '.o'))
    subprocess.run(['hipcc', *FLAGS, '-c', os.path.join(CSRC, f), '-o', o], check=True)
    return o


with ThreadPoolExecutor(8) as ex:
    objs = list(ex.map(cc, srcs))
lib_path = os.path.join(OUT, 'libmts_diag.so')
subprocess.run(['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib_path, *objs], check=True)
lib = C.CDLL(lib_path)
lib.mts_last_error.restype = C.c_char_p
vp, i32, f32, u32, sz = C.c_void_p, C.c_int, C.c_float, C.c_uint, C.c_size_t
lib.mts_gemm.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, i32, vp, i32, vp, i32, vp, vp, i32, vp, i32, u32, f32, i32, vp, sz]
lib.mts_gemm_set_stamps.argtypes = [vp]
lib.mts_set_option.argtypes = [C.c_char_p, i32]

dev = 'cuda'
variants = [int(v) for v in sys.argv[1:]] or [0]
M, K = 16384, 1792
g = torch.Generator(device=dev).manual_seed(1)
A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
res = torch.randn(M, 1792, device=dev, generator=g).to(torch.bfloat16)
for N, epi, label in ((5376, 1 | 8, 'QKV (bias + column scale)'), (1792, 1 | 2, 'attention output (bias + residual)')):
    Bm = (torch.randn(N, K, device=dev, generator=g) * 0.02).to(torch.bfloat16)
    bias = torch.randn(N, device=dev, generator=g)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    stamps = torch.zeros(2048 * 8 * 8, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for v in variants:
        lib.mts_set_option(b'gemm_variant', v)

        def run():
            rc = lib.mts_gemm(st, 1, 1, 0, M, N, K, A.data_ptr(), K, Bm.data_ptr(), K, out.data_ptr(), N, bias.data_ptr(),
                              res.data_ptr() if epi & 2 else None, 1792, None, 0, epi, 0.0668, 1792, None, 0)
            assert rc == 0, lib.mts_last_error()
        lib.mts_gemm_set_stamps(None)
        for _ in range(30):
            run()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            run()
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 1e3 / 20
        stamps.zero_()
        lib.mts_gemm_set_stamps(stamps.data_ptr())
        run()
        torch.cuda.synchronize()
        lib.mts_gemm_set_stamps(None)
        t = stamps.view(2048, 8, 8).cpu().double()
        ntiles = N // 224 * (M // 256)
        persistent = float(t[256:ntiles, 0, 0].abs().sum()) == 0.0
        print(f'--- {label}: N={N}, variant {v}: {us:.1f} us/launch = {2.0 * M * N * K / us / 1e6:.0f} TFLOP/s, '
              f'{"persistent, " + str(ntiles // 256) + " rounds" if persistent else "one tile per workgroup, " + str(ntiles) + " workgroups"}', flush=True)
        if persistent:
            rounds = ntiles // 256
            for r in range(rounds):
                x = t[:256, r]
                loop, pro, sto = x[:, 1] - x[:, 0], x[:, 2] - x[:, 1], x[:, 3] - x[:, 2]
                line = f'  round {r}: K loop {loop.median():8.0f} cyc  loads + prologue issue {pro.median():6.0f}  store phase {sto.median():7.0f}'
                if r + 1 < rounds:
                    wait = x[:, 4] - x[:, 3]
                    tot = x[:, 4] - x[:, 0]
                    clk = (tot / ((x[:, 7] - x[:, 6]).clamp(min=1) / 100.0)).median()      # cycles per microsecond = MHz
                    line += f'  wait+barrier {wait.median():7.0f}  tile total {tot.median():8.0f} cyc  clock {clk:6.0f} MHz'
                print(line, flush=True)
        else:
            x = t[:ntiles, 0]
            loop, pro, sto, tot = x[:, 1] - x[:, 0], x[:, 2] - x[:, 1], x[:, 3] - x[:, 2], x[:, 3] - x[:, 0]
            clk = (tot / ((x[:, 5] - x[:, 6]).clamp(min=1) / 100.0)).median()
            span_us = (x[:, 5].max() - x[:, 6].min()) / 100.0
            busy = (tot / clk).sum() / 256.0                  # us of stamped work per CU if perfectly packed
            print(f'  per workgroup: K loop {loop.median():8.0f} cyc  epilogue loads issued {pro.median():6.0f}  store phase {sto.median():7.0f}  '
                  f'loop start .. last store issued {tot.median():8.0f} cyc  clock {clk:6.0f} MHz', flush=True)
            print(f'  first loop start .. last store issue over the launch: {span_us:.1f} us; stamped work per CU {busy:.1f} us '
                  f'-> {span_us - busy:.1f} us per launch ({(span_us - busy) / (ntiles / 256):.2f} us per tile) outside the stamps '
                  f'(dispatch, tile origin, first DMAs landing, drain)', flush=True)
