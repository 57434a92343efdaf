"""One-pass band attention backward (csrc/band_attn_mfma.hip band_mfma_bwd_fused_kernel; radius <= 15, head dim <= 224): against the fp64
oracle on bf16-rounded operands and against the two-kernel backward it replaces (mts_set_option("band_fused_bwd", 0)) --

  * single-tile documents (<= 256 rows: no halo), exactly 256 rows, one row;
  * long documents cut into 192-key tiles with 16-row query halos (tile seams at 192, 384, ...: a wrong halo shows up right there);
  * ragged padded batches, packed rows (row0), attention dropout, small windows (radius 2, 7), head dims 32 .. 224;
  * the fused q/k/v bias gradient = column sums of dqkv as stored.
Reference arithmetic: modeling_longformer.py:482-640 local attention (= RestrictedTransformerLayer.py:509-636), backward by autograd
of the oracle's statement (oracle/restatement.py band_attention)."""
import math

import pytest
import torch

from oracle import restatement as R

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _inputs(B, Lq, D, seed):
    g = torch.Generator().manual_seed(seed)
    qkv = (torch.randn(B * Lq, 3 * D, generator=g) * 0.7).to(torch.bfloat16)
    dctx = torch.randn(B * Lq, D, generator=g).to(torch.bfloat16)
    return qkv, dctx


def _run(ops, L, fused, qkv, dctx, li32, B, Lq, D, heads, radius, row0=None, drop=(0.0, 0), n_rows=None):
    slots = ops.band_slots(radius)
    n = qkv.shape[0]
    ctx = torch.empty(n, D, dtype=torch.bfloat16, device=DEV)
    probs = torch.empty(n, heads * slots, device=DEV)
    ops.band_attn_fwd(qkv, li32, B, Lq, D, heads, radius, ctx, probs, row0=row0, drop_p=drop[0], drop_seed=drop[1])
    dqkv = torch.full((n, 3 * D), float('nan'), dtype=torch.bfloat16, device=DEV)
    dsc = torch.empty_like(probs)
    dbias = torch.full((3 * D,), float('nan'), device=DEV)
    try:
        L.check(L.lib.mts_set_option(b'band_fused_bwd', int(fused)))
        ops.band_attn_bwd(qkv, li32, probs, dctx, B, Lq, D, heads, radius, dqkv, dsc, dbias=dbias, row0=row0, drop_p=drop[0], drop_seed=drop[1])
    finally:
        L.check(L.lib.mts_set_option(b'band_fused_bwd', 1))
    torch.cuda.synchronize()
    return dqkv, dbias


@pytest.mark.parametrize('B,Lq,D,heads,radius,lengths', [
    (2, 256, 448, 2, 15, None),               # the BASELINE tile: 256 rows, hd 224, one workgroup per (document, head)
    (3, 256, 448, 2, 15, [256, 1, 200]),      # ragged inside one tile
    (1, 40, 448, 2, 15, [17]),                # a short document: most waves idle
    (1, 700, 256, 2, 15, [650]),              # four 192-key tiles with halos, hd 128
    (2, 385, 64, 2, 15, [385, 193]),          # hd 32; documents ending one row after a tile seam
    (2, 500, 192, 2, 7, [500, 300]),          # radius 7 (15 of 32 slots), hd 96
    (2, 300, 320, 2, 2, [300, 5]),            # radius 2, hd 160
    (1, 2437, 128, 2, 15, None),              # the longest real document (SURVEY: 2 437 sentences), hd 64
    (2, 9, 64, 2, 1, [9, 4]),                 # radius 1, nine rows, hd 32
    (1, 257, 32, 1, 15, None),                # one head of 32; one row past the single-tile limit (two tiles, the second holds one key)
])
def test_fused_backward_against_oracle_and_the_two_kernel_form(B, Lq, D, heads, radius, lengths):
    from multimodaltopicsegmentation_amd import ops, _lib as L
    hd = D // heads
    qkv, dctx = _inputs(B, Lq, D, seed=Lq + D)
    len_t = torch.tensor(lengths if lengths is not None else [Lq] * B)
    li32 = len_t.to(torch.int32).to(DEV) if lengths is not None else None
    got, dbias = _run(ops, L, True, qkv.to(DEV), dctx.to(DEV), li32, B, Lq, D, heads, radius)
    two, dbias2 = _run(ops, L, False, qkv.to(DEV), dctx.to(DEV), li32, B, Lq, D, heads, radius)
    assert not torch.isnan(got.float()).any()
    # single-tile documents: the q gradient runs the same MFMA chain in both forms (bitwise); with halos a wave's key window is anchored
    # 16 rows off the two-kernel form's, and k / v always differ by the order of one fp32 accumulation
    gq, tq = got.float().cpu().view(B, Lq, 3, D), two.float().cpu().view(B, Lq, 3, D)
    if Lq <= 256:
        assert torch.equal(gq[:, :, 0], tq[:, :, 0])
    for which in (0, 1, 2):
        d = float((gq[:, :, which] - tq[:, :, which]).abs().max())
        assert d <= 2 ** -7 * max(1.0, float(tq[:, :, which].abs().max())), (which, d)
    # bias gradient = column sums of dqkv as stored
    cs = got.float().cpu().double().sum(0)
    assert float((dbias.cpu().double() - cs).abs().max()) <= 1e-5 * B * Lq
    # fp64 oracle on the bf16-rounded operands
    x64 = qkv.double().view(B, Lq, 3, heads, hd)
    q = x64[:, :, 0].clone().requires_grad_(True)
    k = x64[:, :, 1].clone().requires_grad_(True)
    v = x64[:, :, 2].clone().requires_grad_(True)
    ref = R.band_attention(q, k, v, len_t, radius)
    ref.backward(dctx.double().view(B, Lq, heads, hd))
    g5 = got.float().cpu().double().view(B, Lq, 3, heads, hd)
    for name, a, r in (('dq', g5[:, :, 0], q.grad / math.sqrt(hd)), ('dk', g5[:, :, 1], k.grad), ('dv', g5[:, :, 2], v.grad)):
        err = (a - r).abs()
        bad = err > 3e-2 + 3e-2 * r.abs()
        assert not bad.any(), (name, int(bad.sum()), float(err.max()), [int(x) for x in bad.nonzero()[0]])


def test_fused_backward_on_packed_rows_and_with_dropout():
    """Packed batches (row0: documents back to back, only their valid rows exist) and attention dropout (the mask is regenerated from
    (seed, packed row, head, slot) in part 1 for both dP and the probabilities that multiply dCtx)."""
    from multimodaltopicsegmentation_amd import ops, _lib as L
    B, Lq, D, heads, radius = 4, 420, 256, 2, 15
    lengths = [420, 3, 257, 192]
    n = sum(lengths)
    row0 = torch.tensor([0, 420, 423, 680], dtype=torch.int32, device=DEV)
    li32 = torch.tensor(lengths, dtype=torch.int32, device=DEV)
    qkv, dctx = _inputs(1, n, D, seed=9)
    for drop in ((0.0, 0), (0.3, 1234)):
        a, ba = _run(ops, L, True, qkv.to(DEV), dctx.to(DEV), li32, B, Lq, D, heads, radius, row0=row0, drop=drop)
        b, bb = _run(ops, L, False, qkv.to(DEV), dctx.to(DEV), li32, B, Lq, D, heads, radius, row0=row0, drop=drop)
        assert not torch.isnan(a.float()).any()
        d = float((a.float() - b.float()).abs().max())
        assert d <= 2 ** -7 * max(1.0, float(b.float().abs().max())), (drop, d)
        assert float((ba - bb).abs().max()) <= 2e-2 * max(1.0, float(bb.abs().max()))
    # packed = padded: the same documents in a padded [B, L] batch give the same rows
    pad_q = torch.zeros(B * Lq, 3 * D, dtype=torch.bfloat16)
    pad_d = torch.zeros(B * Lq, D, dtype=torch.bfloat16)
    for bi, (r0, nrow) in enumerate(zip([0, 420, 423, 680], lengths)):
        pad_q[bi * Lq:bi * Lq + nrow] = qkv[r0:r0 + nrow]
        pad_d[bi * Lq:bi * Lq + nrow] = dctx[r0:r0 + nrow]
    p, _ = _run(ops, L, True, pad_q.to(DEV), pad_d.to(DEV), li32, B, Lq, D, heads, radius)
    a, _ = _run(ops, L, True, qkv.to(DEV), dctx.to(DEV), li32, B, Lq, D, heads, radius, row0=row0)
    for bi, (r0, nrow) in enumerate(zip([0, 420, 423, 680], lengths)):
        assert torch.equal(p[bi * Lq:bi * Lq + nrow], a[r0:r0 + nrow]), bi
