"""Split-K of the weight-gradient GEMM without slabs (slices add into C in turn): bitwise equal to the slab + reduce form, for
plain and accumulate-into outputs, across repetitions (the hand-off crosses XCDs)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


@pytest.mark.parametrize('M,N,K,splits', [(1792, 1792, 16384, 4), (5376, 1792, 16384, 2), (640, 264, 8192, 3), (256, 256, 4096, 8)])
def test_chained_split_k_equals_slab_reduce(M, N, K, splits):
    from multimodaltopicsegmentation_amd import _lib as L, ops
    g = torch.Generator().manual_seed(M + K)
    A = torch.randn(K, M, generator=g).to(torch.bfloat16).to(DEV)
    B = torch.randn(K, N, generator=g).to(torch.bfloat16).to(DEV)
    base = torch.randn(M, N, generator=g).to(DEV)
    try:
        L.check(L.lib.mts_set_option(b'gemm_tile', 128))
        L.check(L.lib.mts_set_option(b'gemm_splits', splits))
        res = {}
        for chain in (0, 1):
            L.check(L.lib.mts_set_option(b'gemm_chain', chain))
            out = torch.full((M, N), float('nan'), device=DEV)
            ops.gemm(L.TN, A, B, out, M=M, N=N, K=K)
            acc = base.clone()
            ops.gemm(L.TN, A, B, acc, M=M, N=N, K=K, accumulate=True)
            res[chain] = (out.clone(), acc.clone())
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
        ref = A.double().t().cpu() @ B.double().cpu()
        assert float((res[1][0].cpu().double() - ref).abs().max()) < 1e-3 * float(ref.abs().max())
        L.check(L.lib.mts_set_option(b'gemm_chain', 1))
        for _ in range(25):                                      # the in-place hand-off must be stable run after run
            out = torch.full((M, N), float('nan'), device=DEV)
            ops.gemm(L.TN, A, B, out, M=M, N=N, K=K)
            assert torch.equal(out, res[0][0])
    finally:
        for k in (b'gemm_tile', b'gemm_splits', b'gemm_chain'):
            L.check(L.lib.mts_set_option(k, 0))
