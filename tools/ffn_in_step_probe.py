#!/usr/bin/env python3
"""Why does the fused feed-forward block take 73-75 us inside the training step and 51-58 us back to back?  The block is bound by the CU's copy
queue, i.e. by the shader clock, and the clock is what the step's MFMA-heavy neighbours pull down.  This probe times the SAME forward launch
(rotating buffer sets: nothing served from the Infinity Cache) (a) back to back, (b) alternating with the q|k|v projection GEMM of the step
(the launch in front of it there is a GEMM too), each launch bracketed by HIP events; and prints rocm-smi's shader clock in both phases."""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaltopicsegmentation_amd import ops  # noqa: E402
from multimodaltopicsegmentation_amd.flat import round_up  # noqa: E402

D, F, M, dev = 1792, 256, 16384, 'cuda'
bf = dict(dtype=torch.bfloat16, device=dev)
Mp = round_up(M, 64)
sets = [dict(a1=torch.randn(M, D, **bf), u=torch.empty(Mp, F, **bf)[:M], f=torch.empty(Mp, F, **bf)[:M], s2=torch.empty(Mp, D, **bf)[:M]) for _ in range(8)]
w1, w2 = (torch.randn(F, D, device=dev) * D ** -0.5).to(torch.bfloat16), (torch.randn(D, F, device=dev) * F ** -0.5).to(torch.bfloat16)
b1, b2 = torch.randn(F, device=dev), torch.randn(D, device=dev)
wq = (torch.randn(3 * D, D, device=dev) * D ** -0.5).to(torch.bfloat16)
bq = torch.randn(3 * D, device=dev)
qkv = torch.empty(M, 3 * D, **bf)
clocks = []


def poll(stop):
    while not stop.is_set():
        try:
            out = subprocess.run(['rocm-smi', '--showclocks'], capture_output=True, text=True, timeout=5).stdout
            for line in out.splitlines():
                if 'sclk' in line:
                    clocks.append((time.time(), line.strip().split(':')[-1].strip()))
                    break
        except Exception:  # noqa: BLE001
            pass
        time.sleep(0.2)


def phase(name, with_gemm, seconds=4.0):
    ev, i, t_end = [], 0, time.time() + seconds
    c0 = len(clocks)
    while time.time() < t_end:
        for _ in range(50):
            S = sets[i % 8]
            i += 1
            if with_gemm:
                ops.linear_fwd(S['a1'], wq, bq, qkv, colscale=0.066, ncols_scaled=D)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            ops.ffn_fwd(S['a1'], w1, b1, w2, b2, S['u'], S['f'], S['s2'])
            e.record()
            ev.append((s, e))
        torch.cuda.synchronize()
    us = sorted(s.elapsed_time(e) * 1e3 for s, e in ev[len(ev) // 4:])
    print(f'{name}: ffn_fused forward median {us[len(us) // 2]:.1f} us, p10 {us[len(us) // 10]:.1f}, p90 {us[9 * len(us) // 10]:.1f} over {len(us)} launches; '
          f'shader clock samples: {sorted(set(c for _, c in clocks[c0:]))}', flush=True)


stop = threading.Event()
th = threading.Thread(target=poll, args=(stop,), daemon=True)
th.start()
phase('back to back', False)
phase('alternating with the q|k|v projection GEMM', True)
phase('back to back again', False)
stop.set()
