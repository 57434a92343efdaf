"""GEMM pipeline variants added late in round 1: the four-buffer copy pipeline of the 128x128 kernel ("gemm_deep") and the
weight-gradient (TN) plans on the 256-wide tiles.  Every variant must give the same numbers as the plain one."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda'


def _operands(layout, M, N, K, seed):
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(M, K, generator=g).to(torch.bfloat16)
    b = torch.randn(N, K, generator=g).to(torch.bfloat16)
    ref = a.double() @ b.double().t()
    if layout == 'NT':
        return a.to(DEV), b.to(DEV), ref
    if layout == 'NN':
        return a.to(DEV), b.t().contiguous().to(DEV), ref
    return a.t().contiguous().to(DEV), b.t().contiguous().to(DEV), ref


@pytest.mark.parametrize('layout', ['NT', 'NN', 'TN'])
@pytest.mark.parametrize('M,N,K', [(16384, 256, 1792), (2048, 256, 256), (1024, 384, 320), (640, 128, 448), (256, 256, 64 * 9)])
def test_deep_copy_pipeline_is_bitwise_the_same(layout, M, N, K):
    """Grids of <= 256 workgroups use four K-tile buffers with the copies three tiles ahead; K-loop lengths 4..28 K-tiles
    cover the pipeline's fill and drain (counted vmcnt waits) -- results must not depend on the pipeline depth."""
    from multimodaltopicsegmentation_amd import _lib as L, ops
    A, B, ref = _operands(layout, M, N, K, 7 * M + K)
    g = torch.Generator().manual_seed(3)
    bias = torch.randn(N, generator=g).to(DEV)
    res = torch.randn(M, N, generator=g).to(torch.bfloat16).to(DEV)
    code = getattr(L, layout)
    outs = {}
    try:
        L.check(L.lib.mts_set_option(b'gemm_tile', 128))
        for deep in (0, 1):
            L.check(L.lib.mts_set_option(b'gemm_deep', deep))
            o32 = torch.full((M, N), float('nan'), device=DEV)
            ops.gemm(code, A, B, o32, M=M, N=N, K=K)
            o16 = None
            if layout != 'TN':
                o16 = torch.full((M, N), float('nan'), dtype=torch.bfloat16, device=DEV)
                ops.gemm(code, A, B, o16, M=M, N=N, K=K, bias=bias, residual=res)
            outs[deep] = (o32.clone(), None if o16 is None else o16.clone())
    finally:
        L.check(L.lib.mts_set_option(b'gemm_deep', 1))
        L.check(L.lib.mts_set_option(b'gemm_tile', 0))
    assert torch.equal(outs[0][0], outs[1][0])
    if layout != 'TN':
        assert torch.equal(outs[0][1], outs[1][1])
    err = float((outs[1][0].cpu().double() - ref).abs().max())
    assert err < 1e-4 * (K ** 0.5) * 4, err


@pytest.mark.parametrize('M,N', [(5376, 1792), (1792, 1792), (1792, 256)])
def test_weight_gradient_plans_at_the_baseline_shape(M, N):
    """dW = dY^T X with K = 16 384 tokens: whatever tile / split the cost model picks (256-wide tiles for the wide projections
    since the transposed LDS reads became inline asm) agrees with the forced 128x128 plan and with fp64, and is reproducible."""
    from multimodaltopicsegmentation_amd import _lib as L, ops
    K = 16384
    A, B, ref = _operands('TN', M, N, K, M + N)
    out = torch.full((M, N), float('nan'), device=DEV)
    ops.gemm(L.TN, A, B, out, M=M, N=N, K=K)
    tile, splits = ctypes.c_int(0), ctypes.c_int(0)
    L.check(L.lib.mts_gemm_last_plan(ctypes.byref(tile), ctypes.byref(splits)))
    if N % 224 == 0:
        assert tile.value in (224, 256) and splits.value >= 2, (tile.value, splits.value)
    first = out.clone()
    for _ in range(4):
        out.fill_(float('nan'))
        ops.gemm(L.TN, A, B, out, M=M, N=N, K=K)
        assert torch.equal(out, first)                    # fixed-order slab reduce: bitwise reproducible
    try:
        L.check(L.lib.mts_set_option(b'gemm_tile', 128))
        small = torch.full((M, N), float('nan'), device=DEV)
        ops.gemm(L.TN, A, B, small, M=M, N=N, K=K)
    finally:
        L.check(L.lib.mts_set_option(b'gemm_tile', 0))
    scale = float(ref.abs().max())
    assert float((first.cpu().double() - ref).abs().max()) < 2e-5 * scale
    assert float((first - small).abs().max()) < 2e-5 * scale


def test_accumulating_weight_gradient_on_big_tiles():
    """MTS_EPI_ACCUM (gradient accumulation into the flat buffer) through the split-K big-tile plan."""
    from multimodaltopicsegmentation_amd import _lib as L, ops
    M, N, K = 1792, 1792, 8192
    A, B, ref = _operands('TN', M, N, K, 99)
    base = torch.randn(M, N, generator=torch.Generator().manual_seed(5))
    acc = base.clone().to(DEV)
    ops.gemm(L.TN, A, B, acc, M=M, N=N, K=K, accumulate=True)
    want = base.double() + ref
    assert float((acc.cpu().double() - want).abs().max()) < 2e-5 * float(want.abs().max())
