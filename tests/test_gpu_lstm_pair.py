"""CU-pair LSTM recurrence kernels (bf16, H = 256) on awkward batches: a last group with a single document, documents of
length 1 and of full length, odd and even longest lengths per group, and repeated launches (the hand-off between the two
workgroups of a pair is tagged data polled through L2: results must be bitwise the same run after run)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'

from oracle import restatement as R  # noqa: E402


def _rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def _check(got, ref, rtol, atol, msg):
    err = (got.detach().float().cpu().double() - ref.detach().double()).abs()
    bad = err > atol + rtol * ref.detach().double().abs()
    assert not bad.any(), f'{msg}: {int(bad.sum())}/{bad.numel()} off, max err {float(err.max()):.3e}'


@pytest.mark.parametrize('B,Lq,lengths', [
    (33, 37, None),                                                       # three groups, the last holds one document
    (17, 8, [8] * 17),                                                    # every document full length, even count of steps
    (16, 9, [9, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1]),           # one long document among length-1 ones, odd steps
    (3, 64, [64, 63, 2]),
])
def test_pair_kernels_against_oracle_and_across_launches(B, Lq, lengths):
    from multimodaltopicsegmentation_amd import ops
    H = 256
    if lengths is None:
        g = torch.Generator().manual_seed(B)
        lengths = torch.randint(1, Lq + 1, (B,), generator=g).tolist()
        lengths[0], lengths[16], lengths[32] = Lq, 7, 5                   # group maxima: even / odd / odd
    N = B * Lq
    dt = torch.bfloat16
    xproj = _rnd(N, 8 * H, seed=1).to(dt)
    w_hh = _rnd(2, 4 * H, H, seed=2, scale=1 / math.sqrt(H))
    b_hh = _rnd(2, 4 * H, seed=3, scale=0.1)
    dout = _rnd(N, 2 * H, seed=4).to(dt)
    len_t = torch.tensor(lengths)
    li32 = len_t.to(torch.int32).to(DEV)
    xd, wd, bd, dd = xproj.to(DEV), w_hh.to(DEV), b_hh.to(DEV), dout.to(DEV)

    def run():
        out = torch.full((N, 2 * H), float('nan'), dtype=dt, device=DEV)
        gates = torch.empty(N, 8 * H, dtype=dt, device=DEV)
        cells = torch.empty(N, 2 * H, device=DEV)
        ops.lstm_fwd(xd, wd, bd, li32, B, Lq, H, 2, out, gates, cells)
        dxp = torch.full((N, 8 * H), float('nan'), dtype=dt, device=DEV)
        dwhh = torch.empty(2, 4 * H, H, device=DEV)
        ops.lstm_bwd(wd, li32, out, gates, cells, dd, B, Lq, H, 2, dxp, dwhh)
        torch.cuda.synchronize()
        return out, dxp, dwhh

    out, dxp, dwhh = run()
    xp = xproj.double().view(B, Lq, 8 * H).requires_grad_(True)
    whh = w_hh.double().requires_grad_(True)
    eye, zero = torch.eye(4 * H, dtype=torch.float64), torch.zeros(4 * H, dtype=torch.float64)
    ref = torch.cat([R.lstm_direction(xp[:, :, d * 4 * H:(d + 1) * 4 * H], len_t, eye, whh[d], zero, b_hh[d].double(), rev)
                     for d, rev in ((0, False), (1, True))], dim=2)
    _check(out, ref.view(N, 2 * H), 2e-2, 2e-2, 'out')
    o3, d3 = out.view(B, Lq, -1), dxp.view(B, Lq, -1)
    for b, n in enumerate(lengths):
        if n < Lq:
            assert float(o3[b, n:].abs().max()) == 0.0 and float(d3[b, n:].abs().max()) == 0.0   # padded rows exactly zero
    ref.backward(dout.double().view(B, Lq, 2 * H))
    _check(dxp, xp.grad.view(N, 8 * H), 5e-2, 3e-2, 'dxproj')
    _check(dwhh, whh.grad, 5e-2, 0.15 * max(1.0, math.sqrt(sum(lengths) / 100.0)), 'dw_hh')
    for _ in range(10):
        o2, x2, w2 = run()
        assert torch.equal(o2, out) and torch.equal(x2, dxp) and torch.equal(w2, dwhh)
