"""Late fusion runs its two encoders on two HIP streams: bitwise the same loss, scores and gradients as one after the other."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('dtype,H', [('fp32', 32), ('bf16', 256)])
def test_concurrent_encoders_are_bitwise_identical(dtype, H):
    from multimodaltopicsegmentation_amd import BiLSTMLateFusion
    B, L = 20, 50
    lengths = torch.randint(1, L + 1, (B,), generator=torch.Generator().manual_seed(0))
    lengths[0] = L
    g = torch.Generator().manual_seed(1)
    m = BiLSTMLateFusion(2, [96, 64], H, num_layers=2, loss_fn='FocalLoss', compute_dtype=dtype, seed=2).to(DEV)
    x1, x2 = torch.randn(B, L, 96, generator=g).to(DEV), torch.randn(B, L, 64, generator=g).to(DEV)
    y = (torch.rand(B, L, generator=g) < 0.2).float().to(DEV)
    res = {}
    for mode in (False, True, True, False):
        m.concurrent_encoders = mode
        loss, sc = m.loss_and_grad(x1, x2, lengths, y, True)
        torch.cuda.synchronize()
        res.setdefault(mode, []).append((loss.item(), sc.clone(), m.grad_flat().clone()))
    base = res[False][0]
    for mode in (False, True):
        for loss, sc, gf in res[mode]:
            assert loss == base[0] and torch.equal(sc, base[1]) and torch.equal(gf, base[2]), mode
