"""Dropout (reference: F.dropout in RNN.forward -- active even in eval; nn.Dropout of the HF layers): the kernel against
a host replica of its counter-based hash, and the taggers' gradients against directional finite differences of their own loss
with the masks held fixed."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _host_keep(n, p, seed):
    idx = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over='ignore'):
        z = np.uint64(seed) + np.uint64(0x9E3779B97F4A7C15) * idx
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    thr = min(4294967295.0, p * 4294967296.0)
    return (z >> np.uint64(32)).astype(np.float64) >= np.floor(thr)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_dropout_kernel_matches_host_hash(dtype):
    from multimodaltopicsegmentation_amd import ops
    n, p, seed = 40960, 0.3, 123456789
    x = torch.randn(n).to(dtype)
    r = torch.randn(n).to(dtype)
    y = torch.empty(n, dtype=dtype, device=DEV)
    mask = torch.empty(n, dtype=torch.uint8, device=DEV)
    ops.dropout_fwd(x.to(DEV), y, p, seed, mask=mask, residual=r.to(DEV))
    keep = _host_keep(n, p, seed)
    assert np.array_equal(mask.cpu().numpy().astype(bool), keep)
    assert abs(keep.mean() - (1 - p)) < 0.01
    ref = torch.where(torch.from_numpy(keep), x.float() / (1 - p), torch.zeros(n)) + r.float()
    tol = 1e-6 if dtype == torch.float32 else 2e-2
    assert float((y.cpu().float() - ref).abs().max()) <= tol * max(1.0, float(ref.abs().max()))
    dy = torch.randn(n).to(dtype).to(DEV)
    dx = torch.empty_like(dy)
    ops.dropout_bwd(dy, dx, mask, p)
    refg = torch.where(torch.from_numpy(keep), dy.cpu().float() / (1 - p), torch.zeros(n))
    assert float((dx.cpu().float() - refg).abs().max()) <= tol * max(1.0, float(refg.abs().max()))
    with pytest.raises(ValueError):
        ops.dropout_fwd(x.to(DEV), y, 1.0, seed)


def _directional_check(m, run_loss, n_dirs=3, eps=2e-3, tol=4e-2):
    """analytic g.d vs (L(w + eps d) - L(w - eps d)) / (2 eps) with identical dropout masks (the call counter is rewound)."""
    def loss_at():
        m._drop_calls = 0
        return run_loss(False)
    m._drop_calls = 0
    run_loss(True)
    g = m.grad_flat().clone()
    w0 = m.flat.clone()
    gen = torch.Generator(device=DEV).manual_seed(0)
    for _ in range(n_dirs):
        d = torch.randn(w0.numel(), device=DEV, generator=gen) * (w0 != 0)          # leave the inert padded units alone
        d /= d.norm()
        with torch.no_grad():
            m.flat.copy_(w0 + eps * d)
            m._wcopy_version = None
            lp = loss_at()
            m.flat.copy_(w0 - eps * d)
            m._wcopy_version = None
            lm = loss_at()
            m.flat.copy_(w0)
            m._wcopy_version = None
        num = (lp - lm) / (2 * eps)
        ana = float((g * d).sum())
        noise = 4e-7 * max(abs(lp), abs(lm), 1.0) / eps              # fp32 rounding of the two loss values, amplified by 1 / (2 eps)
        assert abs(num - ana) <= tol * max(abs(ana), abs(num), 1e-3) + noise, (num, ana, noise)


@pytest.mark.parametrize('arch', ['BiLSTM', 'BiLSTMLateFusion', 'biLSTMCRF'])
def test_recurrent_taggers_with_dropout(arch):
    from multimodaltopicsegmentation_amd import TextSegmenter
    B, L = 4, 17
    lengths = torch.tensor([17, 9, 2, 13])
    g = torch.Generator().manual_seed(7)
    dims = [40, 24] if arch == 'BiLSTMLateFusion' else 40
    m = TextSegmenter(2, dims, 24, num_layers=2, architecture=arch, loss_fn='FocalLoss', dropout_in=0.2, dropout_out=0.3, compute_dtype='fp32').to(DEV).model
    x1, x2 = torch.randn(B, L, 40, generator=g).to(DEV), torch.randn(B, L, 24, generator=g).to(DEV)
    y = (torch.rand(B, L, generator=g) < 0.3).float().to(DEV)
    tg = y.long() if arch == 'biLSTMCRF' else y
    args = (x1, x2, lengths) if arch == 'BiLSTMLateFusion' else (x1, lengths)
    x_before = x1.clone()
    _directional_check(m, lambda grad: float(m.loss_and_grad(*args, tg, grad)[0]))
    assert torch.equal(x1, x_before)                                  # input dropout must not touch the caller's tensor
    m.eval()                                                          # the reference drops in eval mode too (F.dropout without training=)
    a = m(*args)[0]
    b = m(*args)[0]
    assert not torch.equal(a, b)


def test_transformer_hidden_dropout():
    from multimodaltopicsegmentation_amd import TextSegmenter
    B, L, D = 3, 40, 64
    lengths = torch.tensor([40, 22, 5])
    g = torch.Generator().manual_seed(8)
    ts = TextSegmenter(2, D, 24, num_layers=2, architecture='Transformer', loss_fn='FocalLoss', nheads=4, attention_window=8, dropout_in=0.25,
                       dropout_out=0.2, compute_dtype='fp32').to(DEV)
    m = ts.model
    x = torch.randn(B, L, D, generator=g).to(DEV)
    y = (torch.rand(B, L, generator=g) < 0.3).float().to(DEV)
    m.train()
    for packed in (False, True):
        m.pack_rows = packed
        _directional_check(m, lambda grad: float(m.loss_and_grad(x, lengths, y, grad)[0]))
    m.eval()                                                          # nn.Dropout: inactive in eval mode
    ref = TextSegmenter(2, D, 24, num_layers=2, architecture='Transformer', loss_fn='FocalLoss', nheads=4, attention_window=8,
                        compute_dtype='fp32').to(DEV)
    ref.model.load_state_dict(m.state_dict())
    ref.model.eval()
    assert torch.equal(m(x, lengths)[0], ref.model(x, lengths)[0])
    with pytest.raises(ValueError):
        TextSegmenter(2, D, 24, architecture='Transformer', nheads=4, attention_window=8, dropout_out=1.5)


def test_attention_dropout_matrix_core_kernels_match_generic_ones():
    """bf16, head dim 32: the MFMA band kernels and the generic ones regenerate the same mask from (seed, row, head, slot)."""
    import math
    from multimodaltopicsegmentation_amd import _lib as L, ops
    B, Lq, D, heads, radius, p, seed = 2, 150, 128, 4, 15, 0.3, 99
    g = torch.Generator().manual_seed(5)
    qkv = (torch.randn(B * Lq, 3 * D, generator=g) * 0.7).to(torch.bfloat16).to(DEV)
    dctx = torch.randn(B * Lq, D, generator=g).to(torch.bfloat16).to(DEV)
    lens = torch.tensor([150, 77], dtype=torch.int32, device=DEV)
    slots = ops.band_slots(radius)
    out = {}
    try:
        for mode in (1, 0):
            L.check(L.lib.mts_set_option(b'band_mfma', mode))
            ctx = torch.empty(B * Lq, D, dtype=torch.bfloat16, device=DEV)
            probs = torch.empty(B * Lq, heads * slots, device=DEV)
            ops.band_attn_fwd(qkv, lens, B, Lq, D, heads, radius, ctx, probs, drop_p=p, drop_seed=seed)
            dqkv = torch.empty(B * Lq, 3 * D, dtype=torch.bfloat16, device=DEV)
            dsc = torch.empty_like(probs)
            ops.band_attn_bwd(qkv, lens, probs, dctx, B, Lq, D, heads, radius, dqkv, dsc, drop_p=p, drop_seed=seed)
            ctx0 = torch.empty_like(ctx)
            ops.band_attn_fwd(qkv, lens, B, Lq, D, heads, radius, ctx0, torch.empty_like(probs))
            out[mode] = (ctx.float().cpu(), dqkv.float().cpu(), probs.cpu(), ctx0.float().cpu())
    finally:
        L.check(L.lib.mts_set_option(b'band_mfma', 1))
    for a, b in zip(out[1][:3], out[0][:3]):
        assert float((a - b).abs().max()) <= 3e-2 * max(1.0, float(b.abs().max()))
    assert float((out[1][0] - out[1][3]).abs().max()) > 0.05        # dropout really changed the context rows
    rs = out[1][2].view(B, Lq, heads, slots).sum(-1)
    assert float((rs[0] - 1).abs().max()) < 1e-5                     # the saved probabilities are the un-dropped ones
