"""Optional side-stream overlap of the weight-gradient GEMMs: same gradients as the sequential backward, bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_side_stream_weight_gradients_are_identical(dtype):
    from multimodaltopicsegmentation_amd.taggers import Transformer_segmenter
    B, L, D = 6, 200, 448
    m = Transformer_segmenter(2, D, 64, num_layers=2, nheads=2, loss_fn='FocalLoss', window_size=30, compute_dtype=dtype, seed=3).to(DEV)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, L, D, generator=g).to(DEV)
    y = (torch.rand(B, L, generator=g) < 0.2).float().to(DEV)
    lengths = torch.full((B,), L)
    res = {}
    for mode in (False, True, True, False):
        m.overlap_wgrad = mode
        loss, _ = m.loss_and_grad(x, lengths, y, True)
        torch.cuda.synchronize()
        res.setdefault(mode, []).append((loss.item(), m.grad_flat().clone()))
    base = res[False][0]
    for mode in (False, True):
        for loss, gflat in res[mode]:
            assert loss == base[0] and torch.equal(gflat, base[1]), mode
