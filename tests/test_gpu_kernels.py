"""Kernel-level parity on a real MI355X: every C-ABI entry point against the CPU oracle / plain torch fp32 math.

Tolerances: fp32 kernels ~1e-5 relative (different summation order only); bf16 kernels are compared against the
same math evaluated in fp64 on the bf16-ROUNDED inputs, so the only error left is fp32 accumulation + one bf16
rounding of the output (<= 2^-8 relative).
"""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import restatement as R  # noqa: E402


@pytest.fixture(scope='module')
def ops():
    from multimodaltopicsegmentation_amd import ops as o
    return o


DEV = 'cuda'


def _rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def _close(got, ref, rtol, atol, msg=''):
    got = got.detach().float().cpu().double()
    ref = ref.detach().double()
    err = (got - ref).abs()
    lim = atol + rtol * ref.abs()
    bad = err > lim
    assert not bad.any(), f'{msg}: {int(bad.sum())}/{bad.numel()} off, max err {float(err.max()):.3e} (ref max {float(ref.abs().max()):.3e})'


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('layout', ['NT', 'NN', 'TN'])
@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (200, 136, 72), (64, 32, 256), (520, 264, 1032), (16, 8, 8)])
def test_gemm_layouts(ops, dtype, layout, M, N, K):
    from multimodaltopicsegmentation_amd import _lib as L
    a = _rnd(M, K, seed=1).to(dtype)
    b = _rnd(N, K, seed=2).to(dtype)          # logical B^T: C = a @ b^T
    ref = a.double() @ b.double().t()
    if layout == 'NT':
        A, Bm, code = a, b, L.NT
    elif layout == 'NN':
        A, Bm, code = a, b.t().contiguous(), L.NN
    else:
        A, Bm, code = a.t().contiguous(), b.t().contiguous(), L.TN
    out = torch.full((M, N), float('nan'), dtype=torch.float32, device=DEV)
    ops.gemm(code, A.to(DEV), Bm.to(DEV), out, M=M, N=N, K=K)
    torch.cuda.synchronize()
    _close(out, ref, 2e-5 if dtype == torch.float32 else 1e-4, 1e-4 * math.sqrt(K), f'{layout} {dtype}')


@pytest.mark.parametrize('tile', [128, 256])
@pytest.mark.parametrize('layout', ['NT', 'NN', 'TN'])
@pytest.mark.parametrize('M,N,K', [(512, 768, 1024), (264, 520, 640), (256, 256, 4096), (1000, 264, 2048)])
def test_gemm_bf16_tile_variants(ops, tile, layout, M, N, K):
    """Both MFMA kernels (128x128 and 256x256 LDS-DMA pipelines) on K % 64 == 0 shapes incl. M/N tails and split-K."""
    from multimodaltopicsegmentation_amd import _lib as L
    a = _rnd(M, K, seed=51).to(torch.bfloat16)
    b = _rnd(N, K, seed=52).to(torch.bfloat16)
    ref = a.double() @ b.double().t()
    if layout == 'NT':
        A, Bm, code = a, b, L.NT
    elif layout == 'NN':
        A, Bm, code = a, b.t().contiguous(), L.NN
    else:
        A, Bm, code = a.t().contiguous(), b.t().contiguous(), L.TN
    try:
        L.check(L.lib.mts_set_option(b'gemm_tile', tile))
        for out_dtype in ([torch.float32] if layout == 'TN' else [torch.bfloat16, torch.float32]):
            out = torch.full((M, N), float('nan'), dtype=out_dtype, device=DEV)
            ops.gemm(code, A.to(DEV), Bm.to(DEV), out, M=M, N=N, K=K)
            tol = (1e-4, 1e-4 * math.sqrt(K)) if out_dtype == torch.float32 else (1e-2, 2e-2 * math.sqrt(K) / 8)
            _close(out, ref, tol[0], tol[1], f'{layout} tile {tile} {out_dtype}')
    finally:
        L.check(L.lib.mts_set_option(b'gemm_tile', 0))


@pytest.mark.parametrize('layout', ['NT', 'NN', 'TN', 'TT'])
@pytest.mark.parametrize('M,N,K', [(512, 448, 1024), (304, 224, 640), (520, 1792, 256)])
def test_gemm_bf16_tile_224(ops, layout, M, N, K):
    """256x224 MFMA kernel (N a multiple of 224 = the d=1792 projections), M tails, bias + residual epilogue."""
    from multimodaltopicsegmentation_amd import _lib as L
    a = _rnd(M, K, seed=53).to(torch.bfloat16)
    b = _rnd(N, K, seed=54).to(torch.bfloat16)
    bias = _rnd(N, seed=55)
    res = _rnd(M, N, seed=56).to(torch.bfloat16)
    ref = a.double() @ b.double().t()
    if layout == 'NT':
        A, Bm, code = a, b, L.NT
    elif layout == 'NN':
        A, Bm, code = a, b.t().contiguous(), L.NN
    elif layout == 'TT':
        A, Bm, code = a.t().contiguous(), b, L.TT
    else:
        A, Bm, code = a.t().contiguous(), b.t().contiguous(), L.TN
    try:
        L.check(L.lib.mts_set_option(b'gemm_tile', 224))
        for out_dtype in ([torch.float32] if layout in ('TN', 'TT') else [torch.bfloat16, torch.float32]):
            out = torch.full((M, N), float('nan'), dtype=out_dtype, device=DEV)
            ops.gemm(code, A.to(DEV), Bm.to(DEV), out, M=M, N=N, K=K)
            tol = (1e-4, 1e-4 * math.sqrt(K)) if out_dtype == torch.float32 else (1e-2, 2e-2 * math.sqrt(K) / 8)
            _close(out, ref, tol[0], tol[1], f'{layout} tile 224 {out_dtype}')
        if layout not in ('TN', 'TT'):
            out = torch.full((M, N), float('nan'), dtype=torch.bfloat16, device=DEV)
            ops.gemm(code, A.to(DEV), Bm.to(DEV), out, M=M, N=N, K=K, bias=bias.to(DEV), residual=res.to(DEV))
            _close(out, ref + bias.double() + res.double(), 1e-2, 2e-2 * math.sqrt(K) / 8, f'{layout} tile 224 bias+residual')
    finally:
        L.check(L.lib.mts_set_option(b'gemm_tile', 0))


@pytest.mark.parametrize('layout', ['NT', 'NN', 'TT'])
@pytest.mark.parametrize('M,N,K', [(512, 448, 64), (512, 448, 128), (512, 448, 192), (256, 224, 256), (768, 1792, 1792), (1000, 224, 704), (1024, 5376, 1792),
                                   (2048, 1792, 5376)])
def test_gemm_224_barrier_schedules_agree_bitwise(ops, layout, M, N, K):
    """The two K-loop schedules of the 256x224 kernel (barrier in the middle of the K-tile: default for NT / NN / TT; at its end:
    gemm_variant 5) accumulate every output element in the same order -- results must be identical bit for bit, with 1, 2 and many
    K-tiles (prologue-only, no steady state, steady state), M tails, bf16 and fp32 C, bias + residual epilogue."""
    from multimodaltopicsegmentation_amd import _lib as L
    a = _rnd(M, K, seed=61).to(torch.bfloat16)
    b = _rnd(N, K, seed=62).to(torch.bfloat16)
    bias, res = _rnd(N, seed=63).to(DEV), _rnd(M, N, seed=64).to(torch.bfloat16).to(DEV)
    A, Bm, code = {'NT': (a, b, L.NT), 'NN': (a, b.t().contiguous(), L.NN), 'TT': (a.t().contiguous(), b, L.TT)}[layout]
    A, Bm = A.to(DEV), Bm.to(DEV)
    outs = {}
    try:
        L.check(L.lib.mts_set_option(b'gemm_tile', 224))
        for variant in (0, 5, 6, 9, 12):          # 0: the defaults (NT, bf16 C: the four-wave kernels -- persistent, or one tile per workgroup with a residual; NN: four-wave); 12: NT with a
                                                  # residual on the persistent kernel; 5: the eight-wave
                                                  # kernel's end-of-tile barrier schedule; 6: the eight-wave kernel everywhere; 9: the four-wave kernel with one tile per workgroup
            L.check(L.lib.mts_set_option(b'gemm_variant', variant))
            o16 = torch.full((M, N), float('nan'), dtype=torch.bfloat16, device=DEV)
            ops.gemm(code, A, Bm, o16, M=M, N=N, K=K, bias=bias, residual=res)
            o32 = torch.full((M, N), float('nan'), dtype=torch.float32, device=DEV)
            ops.gemm(code, A, Bm, o32, M=M, N=N, K=K)
            outs[variant] = (o16.clone(), o32.clone())
    finally:
        L.check(L.lib.mts_set_option(b'gemm_variant', 0))
        L.check(L.lib.mts_set_option(b'gemm_tile', 0))
    for v in (5, 6, 9, 12):
        assert torch.equal(outs[0][0].view(torch.int16), outs[v][0].view(torch.int16)), v
        assert torch.equal(outs[0][1].view(torch.int32), outs[v][1].view(torch.int32)), v
    assert not torch.isnan(outs[0][1]).any()
    ref = a.double() @ b.double().t()
    _close(outs[0][1], ref, 1e-4, 1e-4 * math.sqrt(K), f'{layout} mid-tile barrier fp32 C')


@pytest.mark.parametrize('M,N,K,splits', [(256, 224, 128, 1), (256, 224, 192, 1), (512, 448, 256, 1), (768, 224, 1088, 1), (512, 448, 2304, 3), (256, 1792, 2048, 16),
                                          (1792, 1792, 4096, 4), (5376, 1792, 2048, 0), (2048, 1792, 16384, 0)])
def test_gemm_224t_matches_the_eight_wave_kernel_bitwise(ops, M, N, K, splits):
    """Weight gradients (TN, fp32 C): the four-wave unit-pipelined kernel (gemm224t.hip, default) against the eight-wave kernel it replaces
    (gemm_variant 6) -- same accumulation order per output element, same slices, same fixed-order reduce: identical bits.  Covers the
    minimum of four units, odd K-tile counts, slices of unequal length, forced and planned K splits, plain and accumulating stores."""
    from multimodaltopicsegmentation_amd import _lib as L
    a = _rnd(K, M, seed=71).to(torch.bfloat16).to(DEV)
    b = _rnd(K, N, seed=72).to(torch.bfloat16).to(DEV)
    base = _rnd(M, N, seed=73).to(DEV)
    outs = {}
    try:
        L.check(L.lib.mts_set_option(b'gemm_tile', 224))
        L.check(L.lib.mts_set_option(b'gemm_splits', splits))
        for variant, combine in ((0, 1), (0, 0), (6, 1)):      # in-launch combine by the last-arriving slice | reduce launch | the eight-wave kernel
            L.check(L.lib.mts_set_option(b'gemm_variant', variant))
            L.check(L.lib.mts_set_option(b'gemm_combine', combine))
            for rep in range(2):                               # twice: the arrival tickets are re-zeroed by every call
                o = torch.full((M, N), float('nan'), dtype=torch.float32, device=DEV)
                ops.gemm(L.TN, a, b, o, M=M, N=N, K=K)
                oa = base.clone()
                ops.gemm(L.TN, a, b, oa, M=M, N=N, K=K, accumulate=True)
            outs[variant if combine else 1] = (o.clone(), oa.clone())
    finally:
        L.check(L.lib.mts_set_option(b'gemm_variant', 0))
        L.check(L.lib.mts_set_option(b'gemm_combine', 1))
        L.check(L.lib.mts_set_option(b'gemm_splits', 0))
        L.check(L.lib.mts_set_option(b'gemm_tile', 0))
    torch.cuda.synchronize()
    assert not torch.isnan(outs[0][0]).any()
    ref = a.double().t().cpu() @ b.double().cpu()
    _close(outs[0][0], ref, 1e-4, 1e-4 * math.sqrt(K), 'TN four-wave fp32 C')
    for i in range(2):
        assert torch.equal(outs[0][i].view(torch.int32), outs[6][i].view(torch.int32)), ('plain', 'accumulate')[i]
        assert torch.equal(outs[0][i].view(torch.int32), outs[1][i].view(torch.int32)), ('plain', 'accumulate')[i]


@pytest.mark.parametrize('M,N,K', [(256, 224, 128), (256, 224, 192), (512, 448, 256), (768, 224, 1088), (512, 1792, 1792), (1024, 1792, 5376)])
@pytest.mark.parametrize('epi', ['plain', 'residual', 'bias+residual'])
def test_gemm_224n_matches_the_eight_wave_kernel_bitwise(ops, M, N, K, epi):
    """Data gradients (NN, bf16 C): the four-wave kernel with a k-strided B (gemm224n.hip, the default; the residual tile goes through the LDS stages) against
    the eight-wave kernel (gemm_variant 6) -- same accumulation order per output element, same epilogue arithmetic: identical bits.  Covers the minimum of two
    K-tiles, odd K-tile counts, both step shapes' K, and the residual / bias epilogues; B is a column window of a wider matrix (ldb > N)."""
    from multimodaltopicsegmentation_amd import _lib as L
    a = _rnd(M, K, seed=81).to(torch.bfloat16).to(DEV)
    bw = _rnd(K, N + 224, seed=82, scale=0.2).to(torch.bfloat16).to(DEV)
    b = bw[:, 224:]
    res = _rnd(M, N, seed=83).to(torch.bfloat16).to(DEV) if 'residual' in epi else None
    bias = _rnd(N, seed=84).to(DEV) if 'bias' in epi else None
    outs = {}
    try:
        L.check(L.lib.mts_set_option(b'gemm_tile', 224))
        for variant in (0, 10, 6):            # default (four-wave) | the same, forced | eight-wave
            L.check(L.lib.mts_set_option(b'gemm_variant', variant))
            o = torch.full((M, N), float('nan'), dtype=torch.bfloat16, device=DEV)
            ops.gemm(L.NN, a, b, o, M=M, N=N, K=K, ldb=N + 224, bias=bias, residual=res)
            outs[variant] = o.clone()
    finally:
        L.check(L.lib.mts_set_option(b'gemm_variant', 0))
        L.check(L.lib.mts_set_option(b'gemm_tile', 0))
    torch.cuda.synchronize()
    assert not torch.isnan(outs[0].float()).any()
    ref = a.double().cpu() @ b.double().cpu()
    if res is not None:
        ref = ref + res.double().cpu()
    if bias is not None:
        ref = ref + bias.double().cpu()
    _close(outs[0].float(), ref, 1e-2, 1e-2 * math.sqrt(K) * 0.2 + 2e-2, 'NN four-wave bf16 C')
    assert torch.equal(outs[0].view(torch.int16), outs[6].view(torch.int16))
    assert torch.equal(outs[10].view(torch.int16), outs[6].view(torch.int16))


@pytest.mark.parametrize('M,N,K', [(256, 224, 256), (256, 224, 1088), (512, 448, 4096), (256, 1792, 16384), (256, 1792, 6000 // 64 * 64)])
def test_wgrad_pair_matches_two_launches(ops, M, N, K):
    """mts_wgrad_pair (the fused feed-forward block's two weight gradients as ONE launch of the four-wave weight-gradient kernel + two fixed-order
    reduces, the second result stored transposed) against the two mts_gemm calls it replaces and against fp64: same products, another grouping of
    the K slices -> equal to fp32 summation noise; bitwise reproducible from call to call; plain and accumulating stores; padded output rows."""
    from multimodaltopicsegmentation_amd import _lib as L
    a1 = _rnd(K, M, seed=91).to(torch.bfloat16).to(DEV)
    b1 = _rnd(K, N, seed=92).to(torch.bfloat16).to(DEV)
    a2 = _rnd(K, M, seed=93).to(torch.bfloat16).to(DEV)
    b2 = _rnd(K, N, seed=94).to(torch.bfloat16).to(DEV)
    assert ops.wgrad_pair_supported(a1, b1)
    base1, base2 = _rnd(M, N, seed=95).to(DEV), _rnd(N, M + 8, seed=96).to(DEV)
    outs = []
    for rep in range(2):
        o1 = torch.full((M, N), float('nan'), device=DEV)
        o2 = torch.full((N, M + 8), float('nan'), device=DEV)          # the transposed result lands in a wider buffer (padded storage)
        ops.wgrad_pair(a1, b1, o1, a2, b2, o2)
        acc1, acc2 = base1.clone(), base2.clone()
        ops.wgrad_pair(a1, b1, acc1, a2, b2, acc2, accumulate=True)
        outs.append((o1.clone(), o2.clone(), acc1.clone(), acc2.clone()))
    torch.cuda.synchronize()
    for x, y in zip(outs[0], outs[1]):
        assert torch.equal(x.view(torch.int32), y.view(torch.int32))
    o1, o2, acc1, acc2 = outs[0]
    assert torch.isnan(o2[:, M:]).all() and not torch.isnan(o2[:, :M]).any() and not torch.isnan(o1).any()
    r1 = a1.double().t().cpu() @ b1.double().cpu()
    r2 = (a2.double().t().cpu() @ b2.double().cpu()).t()
    _close(o1, r1, 1e-4, 1e-4 * math.sqrt(K), 'pair problem 1')
    _close(o2[:, :M], r2, 1e-4, 1e-4 * math.sqrt(K), 'pair problem 2 (transposed)')
    _close(acc1, r1 + base1.double().cpu(), 1e-4, 1e-4 * math.sqrt(K), 'pair problem 1, accumulate')
    _close(acc2[:, :M], r2 + base2[:, :M].double().cpu(), 1e-4, 1e-4 * math.sqrt(K), 'pair problem 2, accumulate')
    assert torch.equal(acc2[:, M:], base2[:, M:])
    # the two-launch path
    g1 = torch.empty(M, N, device=DEV)
    g2 = torch.empty(N, M, device=DEV)
    ops.gemm(L.TN, a1, b1, g1, M=M, N=N, K=K)
    ops.gemm(L.TN, b2, a2, g2, M=N, N=M, K=K)
    _close(o1, g1.double().cpu(), 1e-5, 2e-5 * math.sqrt(K), 'pair vs mts_gemm, problem 1')
    _close(o2[:, :M], g2.double().cpu(), 1e-5, 2e-5 * math.sqrt(K), 'pair vs mts_gemm, problem 2')


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_gemm_epilogues(ops, dtype):
    M, N, K = 192, 256, 128
    a, w = _rnd(M, K, seed=3).to(dtype), _rnd(N, K, seed=4, scale=0.2).to(dtype)
    bias = _rnd(N, seed=5)
    res = _rnd(M, N, seed=6).to(dtype)
    base = a.double() @ w.double().t() + bias.double()
    tol = dict(rtol=2e-5, atol=1e-4) if dtype == torch.float32 else dict(rtol=1e-2, atol=2e-2)
    ad, wd, bd, rd = a.to(DEV), w.to(DEV), bias.to(DEV), res.to(DEV)
    # bias + residual, activation-dtype output
    out = torch.empty(M, N, dtype=dtype, device=DEV)
    ops.linear_fwd(ad, wd, bd, out, residual=rd)
    _close(out, base + res.double(), msg='bias+residual', **tol)
    # bias + column scale on the first 64 columns (q / sqrt(hd))
    out2 = torch.empty(M, N, dtype=dtype, device=DEV)
    ops.linear_fwd(ad, wd, bd, out2, colscale=0.25, ncols_scaled=64)
    ref2 = base.clone()
    ref2[:, :64] *= 0.25
    _close(out2, ref2, msg='colscale', **tol)
    # bias + GELU with the pre-activation kept
    out3 = torch.empty(M, N, dtype=dtype, device=DEV)
    aux = torch.empty(M, N, dtype=dtype, device=DEV)
    ops.linear_fwd(ad, wd, bd, out3, gelu=True, aux=aux)
    _close(aux, base, msg='aux', **tol)
    _close(out3, R.gelu_erf(base), msg='gelu', **tol)
    # fp32 accumulate-into (weight-gradient style) + split-K path: long K, few tiles
    K2 = 4096
    dy, x = _rnd(K2, 128, seed=7).to(dtype), _rnd(K2, 72, seed=8).to(dtype)
    g = torch.ones(128, 72, dtype=torch.float32, device=DEV)
    ops.linear_wgrad(dy.to(DEV), x.to(DEV), g, accumulate=True)
    _close(g, dy.double().t() @ x.double() + 1.0, rtol=2e-4, atol=2e-3, msg='wgrad accumulate')
    g2 = torch.full((128, 72), 7.0, dtype=torch.float32, device=DEV)
    ops.linear_wgrad(dy.to(DEV), x.to(DEV), g2)
    _close(g2, dy.double().t() @ x.double(), rtol=2e-4, atol=2e-3, msg='wgrad overwrite')


def test_gemm_rejects_bad_arguments(ops):
    from multimodaltopicsegmentation_amd import _lib as L
    a = torch.zeros(8, 8, device=DEV)
    with pytest.raises(ValueError):
        ops.gemm(7, a, a, a, M=8, N=8, K=8)
    with pytest.raises(NotImplementedError):      # bf16 NT needs K % 8 == 0
        ab = torch.zeros(8, 12, dtype=torch.bfloat16, device=DEV)
        ops.gemm(L.NT, ab, ab, torch.zeros(8, 8, dtype=torch.bfloat16, device=DEV), M=8, N=8, K=4, lda=12, ldb=12)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_colsum_cast(ops, dtype):
    x = _rnd(1000, 264, seed=9).to(dtype)
    out = torch.empty(264, device=DEV)
    ops.colsum(x.to(DEV), out)
    _close(out, x.double().sum(0), 1e-5, 1e-3, 'colsum')
    src = _rnd(1027, seed=10).to(DEV)
    dst = torch.empty(1027, dtype=dtype, device=DEV)
    ops.cast(src, dst)
    assert torch.equal(dst.cpu(), src.cpu().to(dtype))


# ------------------------------------------------------------------------------------------------ LayerNorm family
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('D', [64, 96, 1792, 2304])
def test_layernorm_fwd_bwd(ops, dtype, D):
    rows, eps = 301, 1e-12
    x = (_rnd(rows, D, seed=11) * 1.7 + 0.3).to(dtype)
    gam, bet = 1 + 0.1 * _rnd(D, seed=12), 0.1 * _rnd(D, seed=13)
    hw, hb = _rnd(2, D, seed=14, scale=0.1), _rnd(2, seed=15)
    xd = x.to(DEV)
    y = torch.empty(rows, D, dtype=dtype, device=DEV)
    mean, rstd = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    scores = torch.empty(rows, 2, device=DEV)
    ops.layernorm_fwd(xd, gam.to(DEV), bet.to(DEV), eps, y, mean, rstd, head_w=hw.to(DEV), head_b=hb.to(DEV), scores=scores)
    x64 = x.double().requires_grad_(True)
    g64 = gam.double().requires_grad_(True)
    b64 = bet.double().requires_grad_(True)
    yref = R.layer_norm(x64, g64, b64, eps)
    tol = dict(rtol=1e-5, atol=2e-5) if dtype == torch.float32 else dict(rtol=8e-3, atol=8e-3)
    _close(y, yref, msg='ln y', **tol)
    _close(mean, x.double().mean(-1), 1e-5, 1e-5, 'mean')
    _close(scores, y.cpu().double() @ hw.double().t() + hb.double(), 1e-4, 1e-3, 'fused head')
    # backward with a fused head gradient plus an explicit dy
    dy = _rnd(rows, D, seed=16).to(dtype)
    dl = _rnd(rows, 2, seed=17)
    dx = torch.empty(rows, D, dtype=dtype, device=DEV)
    dg, db, dxs = (torch.empty(D, device=DEV) for _ in range(3))
    ops.layernorm_bwd(xd, dy.to(DEV), gam.to(DEV), mean, rstd, dx, dg, db, dxsum=dxs, dlogit=dl.to(DEV), head_w=hw.to(DEV))
    gin = dy.double() + dl.double() @ hw.double()
    yref.backward(gin)
    tolb = dict(rtol=1e-4, atol=2e-4) if dtype == torch.float32 else dict(rtol=2e-2, atol=3e-2)
    _close(dx, x64.grad, msg='ln dx', **tolb)
    _close(dg, g64.grad, rtol=1e-3 if dtype == torch.float32 else 2e-2, atol=5e-3 if dtype == torch.float32 else 0.3, msg='dgamma')
    _close(db, b64.grad, rtol=1e-3 if dtype == torch.float32 else 2e-2, atol=5e-3 if dtype == torch.float32 else 0.3, msg='dbeta')
    _close(dxs, dx.cpu().double().sum(0), 1e-4, 1e-2, 'dxsum = colsum(dx)')


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_embed_layernorm_and_bwd(ops, dtype):
    B, Lq, D, eps = 3, 37, 96, 1e-12
    x = _rnd(B, Lq, D, seed=18)
    pos = _rnd(64, D, seed=19, scale=0.5)
    typ = _rnd(2, D, seed=20, scale=0.5)
    gam, bet = 1 + 0.1 * _rnd(D, seed=21), 0.1 * _rnd(D, seed=22)
    y = torch.empty(B * Lq, D, dtype=dtype, device=DEV)
    pre = torch.empty(B * Lq, D, dtype=dtype, device=DEV)
    mean, rstd = torch.empty(B * Lq, device=DEV), torch.empty(B * Lq, device=DEV)
    ops.embed_layernorm_fwd(x.to(DEV), pos.to(DEV), 2, typ[0].to(DEV).contiguous(), gam.to(DEV), bet.to(DEV), eps, y, pre, mean, rstd)
    s = x.double() + pos.double()[2:2 + Lq].unsqueeze(0) + typ.double()[0]
    tol = dict(rtol=1e-5, atol=2e-5) if dtype == torch.float32 else dict(rtol=1e-2, atol=2e-2)
    _close(pre, s.view(-1, D), msg='pre', **tol)
    _close(y, R.layer_norm(s, gam.double(), bet.double(), eps).view(-1, D), msg='embed ln', **tol)
    dpre = _rnd(B * Lq, D, seed=23).to(dtype)
    dpos = torch.zeros(64, D, device=DEV)
    ops.embed_bwd(dpre.to(DEV), B, Lq, dpos, 2)
    ref = dpre.double().view(B, Lq, D).sum(0)
    _close(dpos[2:2 + Lq], ref, 1e-5, 1e-4, 'dpos')
    assert float(dpos[:2].abs().max()) == 0 and float(dpos[2 + Lq:].abs().max()) == 0


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_gelu_bwd_and_head(ops, dtype):
    u = _rnd(77, 64, seed=24, scale=2).to(dtype)
    dy = _rnd(77, 64, seed=25).to(dtype)
    dyd = dy.to(DEV).clone()
    ops.gelu_bwd(u.to(DEV), dyd)
    u64 = u.double().requires_grad_(True)
    R.gelu_erf(u64).backward(dy.double())
    _close(dyd, u64.grad, 1e-4 if dtype == torch.float32 else 1e-2, 1e-5 if dtype == torch.float32 else 1e-2, 'gelu bwd')
    # stand-alone head
    x = _rnd(130, 72, seed=26).to(dtype)
    w, b = _rnd(4, 72, seed=27, scale=0.2), _rnd(4, seed=28)
    sc = torch.empty(130, 4, device=DEV)
    ops.head_fwd(x.to(DEV), w.to(DEV), b.to(DEV), sc)
    _close(sc, x.double() @ w.double().t() + b.double(), 1e-5, 1e-4, 'head fwd')
    ds = _rnd(130, 4, seed=29)
    dw, dbb = torch.empty(4, 72, device=DEV), torch.empty(4, device=DEV)
    ops.head_bwd_params(x.to(DEV), ds.to(DEV), dw, dbb)
    _close(dw, ds.double().t() @ x.double(), 1e-5, 1e-4, 'head dw')
    _close(dbb, ds.double().sum(0), 1e-5, 1e-4, 'head db')
    dx = torch.empty(130, 72, dtype=dtype, device=DEV)
    ops.head_bwd_data(ds.to(DEV), w.to(DEV), dx)
    _close(dx, ds.double() @ w.double(), 1e-5 if dtype == torch.float32 else 1e-2, 1e-5 if dtype == torch.float32 else 1e-2, 'head dx')


# ------------------------------------------------------------------------------------------------ band attention
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,Lq,D,heads,radius,lengths', [
    (3, 21, 64, 4, 2, [21, 13, 5]),          # hd 16, tiny window
    (2, 50, 96, 4, 15, [50, 8]),             # hd 24, window wider than a short doc
    (2, 70, 64, 2, 30, [70, 33]),            # two slot blocks (61 slots)
    (1, 40, 448, 2, 15, [1]),                # hd 224 (the BASELINE head dim), a length-1 document
    (2, 33, 64, 4, 4, None),                 # no lengths given
    # head dims that are multiples of 32: bf16 takes the matrix-core kernels (band_attn_mfma.hip)
    (2, 300, 448, 2, 15, [300, 131]),        # hd 224, three 128-row tiles, ragged
    (2, 150, 128, 4, 30, [150, 77]),         # hd 32, 61 slots -> 6 key blocks
    (1, 200, 128, 2, 63, [170]),             # hd 64, the reference's default window 127 -> 10 key blocks
    (2, 256, 512, 2, 40, None),              # hd 256, 96 slots
    (3, 128, 256, 2, 15, [128, 1, 127]),     # hd 128, exactly one tile
    (2, 140, 192, 2, 15, [140, 60]),         # hd 96  = 768 / 8 heads (RoBERTa / wav2vec widths)
    (1, 130, 384, 2, 60, [130]),             # hd 192 = 1536 / 8 heads, the default attention_window 120
    (2, 70, 576, 2, 15, [70, 9]),            # hd 288 = 2304 / 8 heads (768 + 1536): above the matrix-core kernels' 256 -> generic kernels
])
def test_band_attention_fwd_bwd(ops, dtype, B, Lq, D, heads, radius, lengths):
    hd = D // heads
    qkv = _rnd(B * Lq, 3 * D, seed=30, scale=0.7).to(dtype)
    dctx = _rnd(B * Lq, D, seed=31).to(dtype)
    len_t = torch.tensor(lengths if lengths is not None else [Lq] * B)
    li32 = len_t.to(torch.int32).to(DEV) if lengths is not None else None
    slots = ops.band_slots(radius)
    assert slots % 32 == 0 and slots >= 2 * radius + 1
    ctx = torch.full((B * Lq, D), float('nan'), dtype=dtype, device=DEV)
    probs = torch.full((B * Lq, heads * slots), float('nan'), device=DEV)
    qd = qkv.to(DEV)
    ops.band_attn_fwd(qd, li32, B, Lq, D, heads, radius, ctx, probs)
    x64 = qkv.double().view(B, Lq, 3, heads, hd)
    q = x64[:, :, 0].clone().requires_grad_(True)
    k = x64[:, :, 1].clone().requires_grad_(True)
    v = x64[:, :, 2].clone().requires_grad_(True)
    ref = R.band_attention(q, k, v, len_t, radius)
    tol = dict(rtol=2e-5, atol=2e-5) if dtype == torch.float32 else dict(rtol=1e-2, atol=1e-2)
    _close(ctx, ref.reshape(B * Lq, D), msg='ctx', **tol)
    pr = probs.cpu().view(B, Lq, heads, slots)
    rowsum = pr.sum(-1)
    valid = (torch.arange(Lq).view(1, Lq) < len_t.view(B, 1))
    assert torch.allclose(rowsum[valid], torch.ones_like(rowsum[valid]), atol=1e-5)
    assert float(rowsum[~valid].abs().max()) == 0 if (~valid).any() else True
    # backward
    dqkv = torch.full((B * Lq, 3 * D), float('nan'), dtype=dtype, device=DEV)
    dsc = torch.empty_like(probs)
    dbias = torch.full((3 * D,), float('nan'), device=DEV)
    ops.band_attn_bwd(qd, li32, probs, dctx.to(DEV), B, Lq, D, heads, radius, dqkv, dsc, dbias=dbias)
    # fused q/k/v bias gradient = column sums of dqkv exactly as stored
    _close(dbias, dqkv.cpu().double().sum(0), rtol=1e-5, atol=1e-5 * B * Lq, msg='dbias')
    ref.backward(dctx.double().view(B, Lq, heads, hd))
    scale = 1.0 / math.sqrt(hd)
    got = dqkv.cpu().double().view(B, Lq, 3, heads, hd)
    tolb = dict(rtol=1e-4, atol=1e-4) if dtype == torch.float32 else dict(rtol=3e-2, atol=3e-2)
    _close(got[:, :, 0], q.grad * scale, msg='dq (x q_scale)', **tolb)
    _close(got[:, :, 1], k.grad, msg='dk', **tolb)
    _close(got[:, :, 2], v.grad, msg='dv', **tolb)


# ------------------------------------------------------------------------------------------------ loss / decode
@pytest.mark.parametrize('kind,name', [(2, 'FocalLoss'), (1, 'BinaryCrossEntropy'), (0, 'CrossEntropy')])
def test_tagger_loss_and_decode(ops, kind, name):
    B, Lq = 5, 23
    lengths = torch.tensor([23, 17, 1, 9, 23])
    n_out = 2 if kind == 0 else 1
    sc = _rnd(B, Lq, n_out, seed=32, scale=3)
    sc[0, 0] = 0.0
    sc[0, 1] = 40.0
    sc[0, 2] = -40.0
    g = torch.Generator().manual_seed(33)
    tg = torch.full((B, Lq), -1.0)
    for b, n in enumerate(lengths.tolist()):
        tg[b, :n] = (torch.rand(n, generator=g) < 0.3).float()
    s64 = sc.double().requires_grad_(True)
    ref = R.tagger_loss(s64, lengths, tg.double(), name)
    ref.backward()
    out = torch.empty(2, device=DEV)
    ds = torch.empty(B, Lq, n_out, device=DEV)
    ops.tagger_loss(kind, sc.to(DEV), tg.to(DEV), lengths.to(torch.int32).to(DEV), 0.9, 2.0, out, ds)
    assert abs(float(out[0]) - ref.item()) < 2e-6 * max(1, abs(ref.item()))
    assert int(out[1]) == int(lengths.sum())
    _close(ds, s64.grad, 2e-4, 1e-8, 'dscores')
    for th in (0.4, 0.5):
        tags = torch.empty(B, Lq, dtype=torch.uint8, device=DEV)
        ops.greedy_decode(sc.to(DEV), lengths.to(torch.int32).to(DEV), th, tags)
        want = R.greedy_decode(sc, lengths, th, bce=(kind != 0))
        got = [tags[b, :n].cpu().bool().tolist() for b, n in enumerate(lengths.tolist())]
        assert got == want
        assert int(tags.cpu()[2, 1:].sum()) == 0          # positions past the length are 0


def test_tagger_loss_rejects_unknown_kind(ops):
    with pytest.raises(ValueError, match='Choose one of CrossEntropy or BinaryCrossEntropy'):
        ops.tagger_loss(9, torch.zeros(1, 1, 1, device=DEV), torch.zeros(1, 1, device=DEV), None, 0.9, 2.0, torch.zeros(2, device=DEV), None)


# ------------------------------------------------------------------------------------------------ LSTM
@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
@pytest.mark.parametrize('B,Lq,H,lengths', [(5, 19, 32, [19, 11, 1, 7, 19]), (9, 12, 24, [12, 3, 5, 12, 1, 2, 8, 9, 4]),
                                            (20, 10, 256, [10, 3, 5, 10, 1, 2, 8, 9, 4, 7, 10, 6, 2, 9, 1, 10, 3, 10, 5, 8])])   # H=256: MFMA path in bf16
def test_lstm_fwd_bwd(ops, dtype, B, Lq, H, lengths):
    N = B * Lq
    xproj = _rnd(N, 8 * H, seed=34).to(dtype)
    w_hh = _rnd(2, 4 * H, H, seed=35, scale=1 / math.sqrt(H))
    b_hh = _rnd(2, 4 * H, seed=36, scale=0.1)
    dout = _rnd(N, 2 * H, seed=37).to(dtype)
    len_t = torch.tensor(lengths)
    li32 = len_t.to(torch.int32).to(DEV)
    out = torch.full((N, 2 * H), float('nan'), dtype=dtype, device=DEV)
    gates = torch.empty(N, 8 * H, dtype=dtype, device=DEV)
    cells = torch.empty(N, 2 * H, device=DEV)
    ops.lstm_fwd(xproj.to(DEV), w_hh.to(DEV), b_hh.to(DEV), li32, B, Lq, H, 2, out, gates, cells)
    # oracle: identity input projection trick -> feed xproj as "x" with W_ih = I
    xp = xproj.double().view(B, Lq, 8 * H).requires_grad_(True)
    whh = w_hh.double().requires_grad_(True)
    eye = torch.eye(4 * H, dtype=torch.float64)
    zero = torch.zeros(4 * H, dtype=torch.float64)
    outs = []
    for d, rev in ((0, False), (1, True)):
        outs.append(R.lstm_direction(xp[:, :, d * 4 * H:(d + 1) * 4 * H], len_t, eye, whh[d], zero, b_hh[d].double(), rev))
    ref = torch.cat(outs, dim=2)
    tol = dict(rtol=1e-5, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    _close(out, ref.view(N, 2 * H), msg='lstm out', **tol)
    for b, n in enumerate(lengths):
        assert float(out.view(B, Lq, -1)[b, n:].abs().max() if n < Lq else 0.0) == 0.0
    # backward
    dxp = torch.full((N, 8 * H), float('nan'), dtype=dtype, device=DEV)
    dwhh = torch.empty(2, 4 * H, H, device=DEV)
    ops.lstm_bwd(w_hh.to(DEV), li32, out, gates, cells, dout.to(DEV), B, Lq, H, 2, dxp, dwhh)
    ref.backward(dout.double().view(B, Lq, 2 * H))
    tolb = dict(rtol=1e-4, atol=1e-5) if dtype == torch.float32 else dict(rtol=5e-2, atol=3e-2)
    _close(dxp, xp.grad.view(N, 8 * H), msg='dxproj', **tolb)
    _close(dwhh, whh.grad, rtol=1e-4 if dtype == torch.float32 else 5e-2, atol=1e-4 if dtype == torch.float32 else 0.15, msg='dw_hh')


# ------------------------------------------------------------------------------------------------ CRF
def test_crf_nll_viterbi(ops):
    B, Lq, C = 4, 17, 4
    lengths = torch.tensor([17, 9, 1, 17])
    feats = _rnd(B, Lq, C, seed=38)
    trans = _rnd(C, C, seed=39)
    trans[C - 2, :] = R.IMPOSSIBLE
    trans[:, C - 1] = R.IMPOSSIBLE
    g = torch.Generator().manual_seed(40)
    tags = torch.zeros(B, Lq)
    for b, n in enumerate(lengths.tolist()):
        tags[b, :n] = (torch.rand(n, generator=g) < 0.3).float()
    f64 = feats.double().requires_grad_(True)
    t64 = trans.double().requires_grad_(True)
    mask = R.create_mask(Lq, lengths).double()
    ref = (R.crf_forward_score(f64, mask, t64) - R.crf_gold_score(f64, tags.long(), mask, t64)).mean()
    ref.backward()
    out = torch.empty(2, device=DEV)
    df = torch.empty(B, Lq, C, device=DEV)
    dt = torch.empty(C, C, device=DEV)
    li32 = lengths.to(torch.int32).to(DEV)
    ops.crf_nll(feats.to(DEV), tags.to(DEV), li32, trans.to(DEV), out, df, dt)
    assert abs(float(out[0]) - ref.item()) < 1e-4
    _close(df, f64.grad, 1e-3, 1e-6, 'dfeats')
    _close(dt, t64.grad, 1e-3, 1e-5, 'dtrans')
    score = torch.empty(B, device=DEV)
    paths = torch.empty(B, Lq, dtype=torch.int32, device=DEV)
    ops.crf_viterbi(feats.to(DEV), li32, trans.to(DEV), score, paths)
    eye = torch.eye(C, dtype=torch.float64)
    rs, rp = R.crf_viterbi(feats.double(), mask, eye, torch.zeros(C, dtype=torch.float64), trans.double())
    _close(score, rs, 1e-5, 1e-4, 'viterbi score')
    assert [paths[b, :n].cpu().tolist() for b, n in enumerate(lengths.tolist())] == rp


@pytest.mark.parametrize('B,Lq,scale', [(20, 2437, 1.0), (3, 4000, 4.0), (1, 2437, 1.0)])
def test_crf_long_documents(ops, B, Lq, scale):
    """CRF NLL in the scaled probability domain and the 2-bit back-pointer Viterbi at predict.py's document lengths (RadioNews: up to 2 437
    sentences; one document per call) and beyond the first version's 64-KB limit: loss, emission / transition gradients and every Viterbi
    tag against the fp64 oracle.  scale 4: emission rows whose maximum sits on the START tag (nobody can move there) and spreads of +-15."""
    g = torch.Generator().manual_seed(77)
    C = 4
    lengths = torch.randint(Lq // 2, Lq + 1, (B,), generator=g)
    lengths[0] = Lq
    if B > 2:
        lengths[1] = 1
    feats = torch.randn(B, Lq, C, generator=g) * scale
    if scale > 1:
        feats[:, ::3, C - 2] += 12.0                   # START carries the row maximum on every third step
    trans = torch.randn(C, C, generator=g)
    trans[C - 2, :] = R.IMPOSSIBLE
    trans[:, C - 1] = R.IMPOSSIBLE
    tags = torch.zeros(B, Lq)
    for b, n in enumerate(lengths.tolist()):
        tags[b, :n] = (torch.rand(n, generator=g) < 0.3).float()
    f64 = feats.double().requires_grad_(True)
    t64 = trans.double().requires_grad_(True)
    mask = R.create_mask(Lq, lengths).double()
    ref = (R.crf_forward_score(f64, mask, t64) - R.crf_gold_score(f64, tags.long(), mask, t64)).mean()
    ref.backward()
    out = torch.empty(2, device=DEV)
    df = torch.full((B, Lq, C), float('nan'), device=DEV)
    dt = torch.empty(C, C, device=DEV)
    li32 = lengths.to(torch.int32).to(DEV)
    ops.crf_nll(feats.to(DEV), tags.to(DEV), li32, trans.to(DEV), out, df, dt)
    assert abs(float(out[0]) - ref.item()) < 2e-6 * abs(ref.item()), (float(out[0]), ref.item())
    assert not torch.isnan(df).any()
    _close(df, f64.grad, 1e-3, 2e-5 / B, 'dfeats')              # (the reference's gradient carries the 1 / B of the batch mean)
    _close(dt, t64.grad, 1e-3, 1e-4, 'dtrans')
    score = torch.empty(B, device=DEV)
    paths = torch.empty(B, Lq, dtype=torch.int32, device=DEV)
    ops.crf_viterbi(feats.to(DEV), li32, trans.to(DEV), score, paths)
    eye = torch.eye(C, dtype=torch.float64)
    rs, rp = R.crf_viterbi(feats.double(), mask, eye, torch.zeros(C, dtype=torch.float64), trans.double())
    _close(score, rs, 5e-5, 1e-3, 'viterbi score')                 # (an fp32 running sum over thousands of steps)
    got = [paths[b, :n].cpu().tolist() for b, n in enumerate(lengths.tolist())]
    assert got == rp
    assert bool((paths.cpu()[0, lengths[0]:] == -1).all()) and (B < 3 or bool((paths.cpu()[1, 1:] == -1).all()))


# ------------------------------------------------------------------------------------------------ optimizers
def test_adam_and_sgd_match_torch(ops):
    n = 10007
    p0, g = _rnd(n, seed=41), _rnd(n, seed=42, scale=0.1)
    pt = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([pt], lr=1e-3, eps=1e-7)
    p, m, v = p0.to(DEV).clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    mirror = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    for step in range(1, 4):
        pt.grad = g.clone() * step
        opt.step()
        ops.adam_step(p, (g * step * 4).to(DEV), m, v, 1e-3, 0.9, 0.999, 1e-7, step, grad_scale=0.25, bf16_copy=mirror)
    _close(p, pt.detach(), 1e-6, 1e-7, 'adam')
    assert torch.equal(mirror.cpu(), p.cpu().to(torch.bfloat16))
    ps = torch.nn.Parameter(p0.clone())
    sgd = torch.optim.SGD([ps], lr=0.01, weight_decay=1e-4, momentum=0.9)
    p2, buf = p0.to(DEV).clone(), torch.zeros(n, device=DEV)
    for step in range(1, 4):
        ps.grad = g.clone()
        sgd.step()
        ops.sgd_step(p2, g.to(DEV), buf, 0.01, 0.9, 1e-4, step == 1)
    _close(p2, ps.detach(), 1e-6, 1e-7, 'sgd')


@pytest.mark.parametrize('layout', ['NT', 'NN', 'TN', 'TT'])
@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (200, 136, 72), (130, 260, 1030), (16, 8, 8), (1000, 264, 2048)])
def test_gemm_f32_matrix_core_kernel_matches_the_valu_kernel(ops, layout, M, N, K):
    """Parity mode's GEMM runs on v_mfma_f32_16x16x4_f32 (exact fp32): against fp64 math to 2e-6 * sqrt(K) relative, and against the
    VALU kernel ("gemm_f32_mfma" = 0, the same k-ordered fma chain per output element) BIT FOR BIT."""
    from multimodaltopicsegmentation_amd import _lib as L
    g = torch.Generator().manual_seed(M + N + K)
    shp = {'NT': ((M, K), (N, K)), 'NN': ((M, K), (K, N)), 'TN': ((K, M), (K, N)), 'TT': ((K, M), (N, K))}[layout]
    A, B = torch.randn(*shp[0], generator=g), torch.randn(*shp[1], generator=g)
    bias, res = torch.randn(N, generator=g), torch.randn(M, N, generator=g)
    a64 = A.double() if layout in ('NT', 'NN') else A.double().t()
    b64 = B.double().t() if layout in ('NT', 'TT') else B.double()
    ref = a64 @ b64 + bias.double() + res.double()
    outs = {}
    try:
        for mode in (1, 0):
            L.check(L.lib.mts_set_option(b'gemm_f32_mfma', mode))
            out = torch.empty(M, N, device=DEV)
            ops.gemm(getattr(L, layout), A.to(DEV), B.to(DEV), out, M=M, N=N, K=K, bias=bias.to(DEV), residual=res.to(DEV))
            outs[mode] = out.cpu()
    finally:
        L.lib.mts_set_option(b'gemm_f32_mfma', 1)
    scale = float(ref.abs().max())
    assert float((outs[1].double() - ref).abs().max()) < 2e-6 * (K ** 0.5) * scale
    # measured on MI355X: the instruction chains its four products in k order, so the two kernels agree bit for bit
    assert torch.equal(outs[1], outs[0])


@pytest.mark.parametrize('layout', ['TN', 'TT'])
@pytest.mark.parametrize('M,N,K', [(1024, 256, 16384), (256, 1792, 4104), (130, 68, 3000)])
def test_gemm_f32_split_k_weight_gradient_shapes(ops, layout, M, N, K):
    """fp32 weight gradients have a handful of 128 x 128 output tiles and K = every token of the batch: the matrix-core fp32 kernel splits
    K into up to 16 slices (partial tiles -> workspace, summed in slice order by splitk_reduce_kernel).  Against fp64 math, plain and
    accumulating into a non-zero C, a ragged last slice (K not a multiple of 16 x slices), and the same bits run after run."""
    from multimodaltopicsegmentation_amd import _lib as L
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(K, M, generator=g)
    B = torch.randn(K, N, generator=g) if layout == 'TN' else torch.randn(N, K, generator=g)
    ref = A.double().t() @ (B.double() if layout == 'TN' else B.double().t())
    scale = float(ref.abs().max())
    Ad, Bd = A.to(DEV), B.to(DEV)
    out = torch.full((M, N), float('nan'), device=DEV)
    ops.gemm(getattr(L, layout), Ad, Bd, out, M=M, N=N, K=K)
    splits = ctypes_int()
    L.lib.mts_gemm_last_plan(None, splits.ref)
    assert splits.value > 1, splits.value
    assert float((out.cpu().double() - ref).abs().max()) < 2e-6 * (K ** 0.5) * scale
    acc = torch.full((M, N), 3.0, device=DEV)
    ops.gemm(getattr(L, layout), Ad, Bd, acc, M=M, N=N, K=K, accumulate=True)
    assert float((acc.cpu().double() - ref - 3.0).abs().max()) < 2e-6 * (K ** 0.5) * scale
    for _ in range(3):
        again = torch.empty(M, N, device=DEV)
        ops.gemm(getattr(L, layout), Ad, Bd, again, M=M, N=N, K=K)
        assert torch.equal(again, out)


class ctypes_int:
    def __init__(self):
        import ctypes
        self._c = ctypes.c_int(0)
        self.ref = ctypes.byref(self._c)

    @property
    def value(self):
        return self._c.value
