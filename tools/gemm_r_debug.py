#!/usr/bin/env python3
"""Where does gemm_variant 7 (gemm224r.hip) differ from the default 256x224 kernel?  python tools/gemm_r_debug.py NT 256 224 256"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from multimodaltopicsegmentation_amd import _lib as L, ops
lay, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
dev = 'cuda'
g = torch.Generator(device=dev).manual_seed(1)
A = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
B = (torch.randn(N, K, device=dev, generator=g) if lay == 'NT' else torch.randn(K, N, device=dev, generator=g)).to(torch.bfloat16)
code = {'NT': L.NT, 'NN': L.NN}[lay]
outs = {}
L.check(L.lib.mts_set_option(b'gemm_tile', 224))
VAR = int(sys.argv[5]) if len(sys.argv) > 5 else 7
for v in (0, VAR):
    L.check(L.lib.mts_set_option(b'gemm_variant', v))
    o = torch.full((M, N), float('nan'), dtype=torch.bfloat16, device=dev)
    ops.gemm(code, A, B, o, M=M, N=N, K=K)
    torch.cuda.synchronize()
    outs[v] = o.float().cpu()
d = (outs[0] - outs[VAR]).abs()
bad = d > 0
print('mismatching elements', int(bad.sum()), 'of', M * N, 'max diff', float(d.max()), 'nan', int(torch.isnan(outs[VAR]).sum()))
if bad.any():
    rows = bad.any(1).nonzero().flatten().tolist(); cols = bad.any(0).nonzero().flatten().tolist()
    print('rows', rows[:40], '...', len(rows)); print('cols', cols[:40], '...', len(cols))
    # which K-tile is wrong? compare against partial products
    Af, Bf = A.float().cpu(), (B.float().cpu() if lay == 'NT' else B.float().cpu().t())
    r, c = rows[0], cols[0]
    full = outs[0][r, c].item(); got = outs[VAR][r, c].item()
    parts = [(Af[r, k:k + 64] * Bf[c, k:k + 64]).sum().item() for k in range(0, K, 64)]
    print('elem', r, c, 'ref', full, 'got', got, 'diff', got - full, 'per-K-tile partials', [round(p, 3) for p in parts])
if bad.any():
    import itertools
    blk = [(int(bad[i:i + 16].sum())) for i in range(0, M, 16)]
    print('mismatches per 16-row block', blk)
    r = next(i for i in range(M) if bad[i].any()); c = int(bad[r].nonzero()[0])
    full, got = outs[0][r, c].item(), outs[VAR][r, c].item()
    parts = [(Af[r, k:k + 32] * Bf[c, k:k + 32]).sum().item() for k in range(0, K, 32)]
    print('elem', r, c, 'ref', full, 'got', got, '32-deep partials', [round(p, 3) for p in parts])
    n = len(parts)
    # other rows' A against this column: did the wave use a different A row block / K-step?
    cands = []
    for rr in range(r % 16, M, 16):
        for k in range(0, K, 32):
            cands.append(((rr, k), (Af[rr, k:k + 32] * Bf[c, k:k + 32]).sum().item()))
    base = sum(parts)
    for i in range(n):
        for (rr, k), v in cands:
            if abs(base - parts[i] + v - got) < 0.004 and (rr, k) != (r, 32 * i):
                print('  single substitution: k-step', i, '(k =', 32 * i, ') replaced by A row', rr, 'k', k, '->', base - parts[i] + v)
