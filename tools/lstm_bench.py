#!/usr/bin/env python3
"""Time mts_lstm_fwd / mts_lstm_bwd alone at the BASELINE shape (GPU box)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaltopicsegmentation_amd import ops
B, L, H = int(os.environ.get("MTS_B", 64)), int(os.environ.get("MTS_L", 256)), 256
dev = 'cuda'
DT = torch.float32 if os.environ.get('MTS_DT', 'bf16') == 'fp32' else torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
xproj = (torch.randn(B * L, 8 * H, device=dev, generator=g) * 0.5).to(DT)
w_hh = torch.randn(2, 4 * H, H, device=dev, generator=g) / 16
b_hh = torch.zeros(2, 4 * H, device=dev)
lengths = torch.full((B,), L, dtype=torch.int32, device=dev)
out = torch.empty(B * L, 2 * H, dtype=DT, device=dev)
gates = torch.empty(B * L, 8 * H, dtype=DT, device=dev)
cells = torch.empty(B * L, 2 * H, device=dev)
dout = torch.randn(B * L, 2 * H, device=dev, generator=g).to(DT)
dx = torch.empty(B * L, 8 * H, dtype=DT, device=dev)
dw = torch.empty(2, 4 * H, H, device=dev)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / reps
f = t(lambda: ops.lstm_fwd(xproj, w_hh, b_hh, lengths, B, L, H, 2, out, gates, cells))
b = t(lambda: ops.lstm_bwd(w_hh, lengths, out, gates, cells, dout, B, L, H, 2, dx, dw))
print(f'{DT} EXP={os.environ.get("MTS_LSTM_EXP","0")} fwd {f:.0f} us ({f/L:.2f} us/step)  bwd(+dW gemm) {b:.0f} us ({b/L:.2f} us/step)', flush=True)
