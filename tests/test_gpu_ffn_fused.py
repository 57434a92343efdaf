"""Fused feed-forward block (csrc/ffn_fused.hip, mts_ffn_fwd / mts_ffn_bwd_data) on a real MI355X.

The kernel's contract is that its results are BITWISE those of the two-launch path it replaces (up-projection GEMM with the
activation epilogue, down-projection GEMM with the residual epilogue; backward: data-gradient GEMM, activation gradient in place,
data-gradient GEMM with the residual): same k order per output element, one rounding of the intermediate to bf16.  The cases cover
the smallest shape (one workgroup, one output chunk), a ragged row count (rows past M clamped on load, written into padding),
BASELINE's d = 1792 at a packed-batch row count and at the full padded batch, both activations, and the model-level switch.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'
F = 256


def _bits(t):
    return t.contiguous().view(torch.int16)


def _operands(M, D, seed):
    g = torch.Generator(device='cpu').manual_seed(seed)
    r = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale)
    a1 = r(M, D).to(DEV, torch.bfloat16)
    w1 = r(F, D, scale=D ** -0.5).to(DEV, torch.bfloat16)
    w2 = r(D, F, scale=F ** -0.5).to(DEV, torch.bfloat16)
    b1 = r(F, scale=0.5).to(DEV)
    b2 = r(D, scale=0.5).to(DEV)
    ds2 = r(M, D).to(DEV, torch.bfloat16)
    return a1, w1, w2, b1, b2, ds2


@pytest.mark.parametrize('M,D,relu', [(64, 256, False), (1, 256, True), (150, 512, True), (333, 768, False), (4999, 1792, False),
                                      (19584, 1792, False), (19584, 1792, True)])
def test_fused_block_is_bitwise_the_two_launch_path(M, D, relu):
    from multimodaltopicsegmentation_amd import ops
    from multimodaltopicsegmentation_amd.flat import round_up
    assert ops.ffn_supported(torch.bfloat16, M, D, F)
    a1, w1, w2, b1, b2, ds2 = _operands(M, D, 1000 + M + D)
    bf = dict(dtype=torch.bfloat16, device=DEV)
    # ---- two launches per direction (the path the model used before; modeling_longformer.py:1113-1131)
    u0, f0, s20 = torch.empty(M, F, **bf), torch.empty(M, F, **bf), torch.empty(M, D, **bf)
    ops.linear_fwd(a1, w1, b1, f0, gelu=not relu, relu=relu, aux=u0)
    ops.linear_fwd(f0, w2, b2, s20, residual=a1)
    du0, da0 = torch.empty(M, F, **bf), torch.empty(M, D, **bf)
    ops.linear_dgrad(ds2, w2, du0)
    (ops.relu_bwd if relu else ops.gelu_bwd)(u0, du0)
    ops.linear_dgrad(du0, w1, da0, residual=ds2)
    # ---- one launch per direction; outputs in buffers padded to whole 64-row tiles, poisoned so that an unwritten element shows
    Mp = round_up(M, 64)
    poison = lambda cols: torch.full((Mp, cols), float('nan'), **bf)
    U, Fo, S2, DU, DA = poison(F), poison(F), poison(D), poison(F), poison(D)
    a1_before, ds2_before = a1.clone(), ds2.clone()
    ops.ffn_fwd(a1, w1, b1, w2, b2, U[:M], Fo[:M], S2[:M], relu=relu)
    ops.ffn_bwd_data(ds2, w1, w2, u0, DU[:M], DA[:M], relu=relu)
    torch.cuda.synchronize()
    for name, got, want in (('u', U, u0), ('f', Fo, f0), ('s2', S2, s20), ('du', DU, du0), ('da1', DA, da0)):
        assert torch.equal(_bits(got[:M]), _bits(want)), '%s differs: max |d| = %g' % (name, float((got[:M].float() - want.float()).abs().max()))
    assert torch.equal(_bits(a1), _bits(a1_before)) and torch.equal(_bits(ds2), _bits(ds2_before))     # inputs untouched


def test_fused_block_rejects_what_it_does_not_cover():
    """F != 256, d not a multiple of 256 and fp32 are refused with MTS_ERR_UNSUPPORTED (no silent fallback inside the entry point)."""
    from multimodaltopicsegmentation_amd import ops
    assert not ops.ffn_supported(torch.float32, 128, 512, 256)
    assert not ops.ffn_supported(torch.bfloat16, 128, 512, 128)
    assert not ops.ffn_supported(torch.bfloat16, 128, 384, 256)
    bf = dict(dtype=torch.bfloat16, device=DEV)
    a1 = torch.zeros(128, 384, **bf)
    w1, w2 = torch.zeros(256, 384, **bf), torch.zeros(384, 256, **bf)
    b1, b2 = torch.zeros(256, device=DEV), torch.zeros(384, device=DEV)
    u, f, s2 = torch.zeros(128, 256, **bf), torch.zeros(128, 256, **bf), torch.zeros(128, 384, **bf)
    with pytest.raises(NotImplementedError):                     # MTS_ERR_UNSUPPORTED
        ops.ffn_fwd(a1, w1, b1, w2, b2, u, f, s2)


@pytest.mark.parametrize('packed', [False, True])
def test_transformer_step_is_bitwise_the_same_with_and_without_the_fused_block(packed, monkeypatch):
    """Model level: loss, scores and every gradient of a bf16 Transformer_segmenter (d = 512, ff = 256, ragged documents) agree
    bit for bit whether the layer runs the fused block or the GEMM pairs."""
    from multimodaltopicsegmentation_amd.taggers import Transformer_segmenter
    B, L, D = 5, 70, 512
    lengths = torch.tensor([70, 33, 51, 64, 7])
    g = torch.Generator().manual_seed(77)
    x = torch.randn(B, L, D, generator=g)
    y = torch.full((B, L), -1.0)
    for b, n in enumerate(lengths.tolist()):
        x[b, n:] = 0.0
        y[b, :n] = (torch.rand(n, generator=g) < 0.2).float()
    x, y = x.to(DEV), y.to(DEV)
    m = Transformer_segmenter(2, D, 256, num_layers=2, nheads=4, loss_fn='FocalLoss', window_size=6, compute_dtype='bf16', seed=5).to(DEV)
    m.pack_rows = packed
    m.fuse_ffn_min_rows = 0                      # (the model fuses from 12288 rows on, where 64-row workgroups fill the chip)
    from multimodaltopicsegmentation_amd import ops
    calls = {'n': 0}
    real = ops.ffn_fwd
    monkeypatch.setattr(ops, 'ffn_fwd', lambda *a, **k: (calls.__setitem__('n', calls['n'] + 1), real(*a, **k))[1])
    out = []
    for fuse in (True, False):
        m.fuse_ffn = fuse
        loss, sc = m.loss_and_grad(x, lengths, y, True)
        assert calls['n'] == (2 if fuse else 2)          # two layers through the fused kernel in the first pass, none added in the second
        out.append((loss.item(), sc.float().clone(), m.grad_flat().clone()))
    assert out[0][0] == out[1][0]
    assert torch.equal(out[0][1], out[1][1])
    assert torch.equal(out[0][2], out[1][2])
    assert float(out[0][2].abs().max()) > 0
