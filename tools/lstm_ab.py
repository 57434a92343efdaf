#!/usr/bin/env python3
"""CU-pair vs CU-quad form of the bf16 LSTM recurrences (mts_set_option("lstm_parts", 2 | 4)) in one process: results must be
bitwise equal (same MFMA accumulation order per gate column), timing per dependent step.  MTS_B / MTS_L / MTS_RAGGED env."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaltopicsegmentation_amd import _lib as Lb, ops
B, L, H = int(os.environ.get("MTS_B", 64)), int(os.environ.get("MTS_L", 256)), 256
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(0)
xproj = (torch.randn(B * L, 8 * H, device=dev, generator=g) * 0.5).to(torch.bfloat16)
w_hh = torch.randn(2, 4 * H, H, device=dev, generator=g) / 16
b_hh = torch.randn(2, 4 * H, device=dev, generator=g) * 0.1
lengths = torch.full((B,), L, dtype=torch.int32, device=dev)
if os.environ.get("MTS_RAGGED"):
    lengths = torch.randint(1, L + 1, (B,), generator=torch.Generator().manual_seed(3)).to(torch.int32).to(dev)
    lengths[0] = L
dout = torch.randn(B * L, 2 * H, device=dev, generator=g).to(torch.bfloat16)
def run(parts, bwd):
    Lb.check(Lb.lib.mts_set_option(b'lstm_parts', parts))
    out = torch.zeros(B * L, 2 * H, dtype=torch.bfloat16, device=dev)
    gates = torch.zeros(B * L, 8 * H, dtype=torch.bfloat16, device=dev)
    cells = torch.zeros(B * L, 2 * H, device=dev)
    dx = torch.zeros(B * L, 8 * H, dtype=torch.bfloat16, device=dev)
    dw = torch.zeros(2, 4 * H, H, device=dev)
    def t(fn, reps=5):
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) * 1e3 / reps
    f = t(lambda: ops.lstm_fwd(xproj, w_hh, b_hh, lengths, B, L, H, 2, out, gates, cells))
    b = t(lambda: ops.lstm_bwd(w_hh, lengths, out, gates, cells, dout, B, L, H, 2, dx, dw)) if bwd else 0.0
    Lb.check_async()
    return (out, gates, cells, dx, dw), f, b
bwd = os.environ.get("MTS_BWD", "1") != "0"
r2, f2, b2 = run(2, bwd)
r4, f4, b4 = run(4, bwd)
names = ['out', 'gates', 'cells', 'dxproj', 'dw_hh']
for n, a, c in zip(names, r2, r4):
    if n in ('gates', 'cells'):
        continue                      # saved state: private layouts (the quad form keeps step-major blocks)
    d = (a.float() - c.float()).abs().max().item()
    print(f'{n:7s} equal={torch.equal(a, c)} max|diff|={d:.3e} max|ref|={a.float().abs().max().item():.3e}')
print(f'pair: fwd {f2:.0f} us ({f2 / L:.2f} us/step) bwd {b2:.0f} us ({b2 / L:.2f} us/step)   quad: fwd {f4:.0f} us ({f4 / L:.2f} us/step) bwd {b4:.0f} us ({b4 / L:.2f} us/step)', flush=True)
