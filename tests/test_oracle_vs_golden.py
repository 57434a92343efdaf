"""Pin the CPU oracle (oracle/restatement.py) against fixtures recorded from the reference itself.

CPU-only (runs under ``-m "not gpu"``).  Fixtures: tests/golden/*.npz, produced by
tests/golden/make_golden.py which imports /root/reference and records its outputs.
"""
import numpy as np
import pytest
import torch

from oracle import restatement as R
from tests import helpers as H

torch.set_num_threads(8)


def _t(a, dtype=torch.float64):
    return torch.from_numpy(np.asarray(a)).to(dtype)


def _grads(loss, params):
    names = [n for n, p in params.items() if p.requires_grad]
    gs = torch.autograd.grad(loss, [params[n] for n in names], allow_unused=True)
    return {n: g for n, g in zip(names, gs) if g is not None}


# ------------------------------------------------------------------------------------------ BiLSTM (a4, a6)
@pytest.mark.parametrize('tag,loss_fn', [('Fo', 'FocalLoss'), ('Bi', 'BinaryCrossEntropy'), ('Cr', 'CrossEntropy')])
@pytest.mark.parametrize('batched', [False, True])
def test_bilstm_small(tag, loss_fn, batched):
    g = H.load('g1_bilstm_small')
    D, Hd, NL = [int(v) for v in g['cfg']]
    p = {k[len(tag) + 3:]: _t(v).requires_grad_(True) for k, v in g.items() if k.startswith(tag + '.w.')}
    x, lengths, tags = _t(g['x']), torch.from_numpy(g['lengths']), _t(g['tags'])
    scores = R.bilstm_scores(x, lengths, p, NL, batched=batched)
    np.testing.assert_allclose(scores.detach().numpy(), g[f'{tag}.scores'], rtol=0, atol=2e-6)
    loss = R.tagger_loss(scores, lengths, tags, loss_fn)
    assert abs(loss.item() - float(g[f'{tag}.loss'])) < 1e-6
    gr = _grads(loss, p)
    for n, gv in gr.items():
        np.testing.assert_allclose(gv.numpy(), g[f'{tag}.g.{n}'], rtol=1e-4, atol=2e-7, err_msg=n)
    for th in (0.4, 0.5):
        got = R.greedy_decode(scores.detach().float(), lengths, th, bce=(loss_fn != 'CrossEntropy'))
        assert got == H.split_tags(g[f'{tag}.tags{th}'], g['lengths'])
    got = R.greedy_decode(scores.detach().float(), lengths, None, bce=(loss_fn != 'CrossEntropy'))
    assert got == H.split_tags(g[f'{tag}.tagsdefault'], g['lengths'])


def test_padded_rows_are_zero_and_output_is_maxlen():
    g = H.load('g1_bilstm_small')
    D, Hd, NL = [int(v) for v in g['cfg']]
    p = {k[5:]: _t(v) for k, v in g.items() if k.startswith('Fo.w.')}
    lengths = torch.tensor([5, 3, 1, 2, 4])
    h = R.rnn_forward(_t(g['x']), lengths, p, 'model.', NL)
    assert h.shape[1] == 5                                  # pad_packed_sequence -> max(lengths)
    for b, n in enumerate(lengths.tolist()):
        assert torch.all(h[b, n:] == 0)


def test_late_fusion_small():
    g = H.load('g7_latefusion_small')
    D1, D2, Hd, NL = [int(v) for v in g['cfg']]
    p = {k[2:]: _t(v).requires_grad_(True) for k, v in g.items() if k.startswith('w.')}
    lengths = torch.from_numpy(g['lengths'])
    scores = R.late_fusion_scores(_t(g['x1']), _t(g['x2']), lengths, p, NL)
    np.testing.assert_allclose(scores.detach().numpy(), g['scores'], atol=2e-6, rtol=0)
    loss = R.tagger_loss(scores, lengths, _t(g['tags']), 'FocalLoss')
    assert abs(loss.item() - float(g['loss'])) < 1e-6
    for n, gv in _grads(loss, p).items():
        np.testing.assert_allclose(gv.numpy(), g['g.' + n], rtol=1e-4, atol=2e-7, err_msg=n)
    assert R.greedy_decode(scores.detach().float(), lengths, 0.5, True) == H.split_tags(g['tags0.5'], g['lengths'])


def test_bilstm_1792_seeded():
    g = H.load('g2_bilstm_1792')
    D, Hd, NL = [int(v) for v in g['cfg']]
    shapes = H.bilstm_param_shapes(D, Hd, NL, 1)
    shapes['classification.weight'] = (1, 2 * Hd)
    shapes['classification.bias'] = (1,)
    p = H.seeded_params(shapes, int(g['seed']), torch.float64, True)
    lengths = torch.from_numpy(g['lengths'])
    x = _t(g['x'].astype(np.float32))
    scores = R.bilstm_scores(x, lengths, p, NL, batched=True)
    # fp64 oracle vs the reference's fp32 run at K=1792: the gap is the reference's own rounding
    np.testing.assert_allclose(scores.detach().numpy(), g['scores'], atol=5e-5, rtol=0)
    loss = R.tagger_loss(scores, lengths, _t(g['tags']), 'FocalLoss')
    assert abs(loss.item() - float(g['loss'])) < 2e-6
    assert R.greedy_decode(scores.detach().float(), lengths, 0.5, True) == H.split_tags(g['tags0.5'], g['lengths'])
    for n, gv in _grads(loss, p).items():
        np.testing.assert_allclose(gv.numpy().ravel()[:32], g['ghead.' + n], rtol=2e-3, atol=1e-7, err_msg=n)
        cs = H.checksum(gv.numpy())
        np.testing.assert_allclose(cs[1:], g['gsum.' + n][1:], rtol=1e-3, atol=1e-9, err_msg=n)


# ------------------------------------------------------------------------------------------ band encoder (a9, a10)
@pytest.mark.parametrize('name,loss_fn', [('g3a_transformer_w4x2', 'FocalLoss'), ('g3b_transformer_w30', 'BinaryCrossEntropy'),
                                          ('g3c_transformer_ce', 'CrossEntropy')])
def test_band_encoder_small(name, loss_fn):
    g = H.load(name)
    D, heads, ff, NL, window = [int(v) for v in g['cfg']]
    n_out = 2 if loss_fn == 'CrossEntropy' else 1
    p = H.seeded_params(H.band_param_shapes(D, ff, NL, n_out), int(g['seed']), torch.float64, True)
    radii = R.pyramidal_radii(NL, window)
    x, lengths, tags = _t(g['x']), torch.from_numpy(g['lengths']), _t(g['tags'])
    hidden = R.band_encoder(x, lengths, p, heads, radii)
    np.testing.assert_allclose(hidden.detach().numpy(), g['hidden'], atol=3e-6, rtol=0)   # incl. padded rows
    scores = hidden @ p['classification.weight'].t() + p['classification.bias']
    np.testing.assert_allclose(scores.detach().numpy(), g['scores'], atol=3e-6, rtol=0)
    loss = R.tagger_loss(scores, lengths, tags, loss_fn)
    assert abs(loss.item() - float(g['loss'])) < 1e-6
    L = x.shape[1]
    for n, gv in _grads(loss, p).items():
        ref = g['g.' + n]
        got = gv.numpy()
        if 'position_embeddings' in n:
            got = got[: ref.shape[0]]
            assert np.all(got[L + 2:] == 0) and np.all(got[:2] == 0)
        np.testing.assert_allclose(got, ref, rtol=2e-4, atol=5e-8, err_msg=n)
    for th in (0.4, 0.5):
        assert R.greedy_decode(scores.detach().float(), lengths, th, loss_fn != 'CrossEntropy') == \
            H.split_tags(g[f'tags{th}'], g['lengths'])


def test_band_encoder_1792_seeded():
    g = H.load('g4_transformer_1792')
    D, heads, ff, NL, window = [int(v) for v in g['cfg']]
    p = H.seeded_params(H.band_param_shapes(D, ff, NL, 1, max_pos=300), int(g['seed']), torch.float32, True)
    # position table rows beyond L+2 are never read; the recipe is per-tensor so regenerate at full size
    full = H.seeded_param('model.model.embeddings.position_embeddings.weight', (4096, D), int(g['seed']))
    p['model.model.embeddings.position_embeddings.weight'] = torch.from_numpy(full[:300].copy()).requires_grad_(True)
    lengths = torch.from_numpy(g['lengths'])
    x = _t(g['x'].astype(np.float32), torch.float32)
    scores = R.transformer_scores(x, lengths, p, heads, R.pyramidal_radii(NL, window))
    np.testing.assert_allclose(scores.detach().numpy(), g['scores'], atol=2e-5, rtol=0)
    loss = R.tagger_loss(scores, lengths, _t(g['tags'], torch.float32), 'FocalLoss')
    assert abs(loss.item() - float(g['loss'])) < 2e-6
    assert R.greedy_decode(scores.detach(), lengths, 0.5, True) == H.split_tags(g['tags0.5'], g['lengths'])
    L = x.shape[1]
    for n, gv in _grads(loss, p).items():
        got = gv.numpy()
        if 'position_embeddings' in n:
            got = got[: L + 2]
        np.testing.assert_allclose(got.ravel()[:32], g['ghead.' + n], rtol=5e-3, atol=2e-7, err_msg=n)
        if 'key.bias' in n:
            continue   # analytically zero (softmax shift invariance): pure rounding noise on both sides
        np.testing.assert_allclose(H.checksum(got)[1:], g['gsum.' + n][1:], rtol=1e-3, atol=1e-9, err_msg=n)


def test_legacy_layer_matches_band_attention():
    """a11: the reference's own second statement of the band-attention math."""
    g = H.load('g10_legacy_layer')
    d, h, ff, w = [int(v) for v in g['cfg']]
    p = {k[2:]: _t(v) for k, v in g.items() if k.startswith('w.')}
    y = R.legacy_restricted_layer(_t(g['x']), p, h, w)
    np.testing.assert_allclose(y.numpy(), g['y'], atol=3e-6, rtol=0)


# ------------------------------------------------------------------------------------------ losses, CRF, layout, metrics
def test_focal_edge_cases():
    g = H.load('g6_focal')
    for alpha, gamma in ((0.9, 2.0), (0.25, 2.0), (-1.0, 2.0), (0.9, 0.0), (0.5, 3.0)):
        x = _t(g['x']).requires_grad_(True)
        l = R.sigmoid_focal_loss(x, _t(g['y']), alpha, gamma)
        assert abs(l.item() - float(g[f'loss_a{alpha}_g{gamma}'])) < 3e-7 * max(1.0, abs(l.item()))  # ref is fp32
        (gx,) = torch.autograd.grad(l, x)
        np.testing.assert_allclose(gx.numpy(), g[f'grad_a{alpha}_g{gamma}'], rtol=2e-4, atol=1e-8)


def test_crf_nll_and_viterbi():
    g = H.load('g5_crf')
    lengths = torch.from_numpy(g['lengths'])
    feats = _t(g['features']).requires_grad_(True)
    w, b, tr = (_t(g[k]).requires_grad_(True) for k in ('fc.weight', 'fc.bias', 'transitions'))
    mask = R.create_mask(feats.shape[1], lengths)
    loss = R.crf_nll(feats, _t(g['tags']), mask, w, b, tr)
    assert abs(loss.item() - float(g['loss'])) < 1e-5
    gf, gw, gb, gt = torch.autograd.grad(loss, [feats, w, b, tr])
    np.testing.assert_allclose(gf.numpy(), g['g.features'], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(gw.numpy(), g['g.fc.weight'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(gb.numpy(), g['g.fc.bias'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(gt.numpy(), g['g.transitions'], rtol=1e-4, atol=1e-6)
    score, paths = R.crf_viterbi(feats.detach(), mask, w.detach(), b.detach(), tr.detach())
    np.testing.assert_allclose(score.numpy(), g['viterbi_score'], rtol=1e-5)
    assert [v for p_ in paths for v in p_] == g['viterbi_paths'].tolist()
    assert [len(p_) for p_ in paths] == g['lengths'].tolist()


def test_rnn_crf_composition():
    g = H.load('g5_crf')
    D, Hd, NL = [int(v) for v in g['c.cfg']]
    lengths = torch.from_numpy(g['lengths'])
    p = {k[4:]: _t(v).requires_grad_(True) for k, v in g.items() if k.startswith('c.w.')}
    x = _t(g['c.x'])
    h = R.rnn_forward(x, lengths, p, 'model.', NL)
    mask = R.create_mask(x.shape[1], lengths)
    loss = R.crf_nll(h, _t(g['tags']), mask, p['crf.fc.weight'], p['crf.fc.bias'], p['crf.transitions'])
    assert abs(loss.item() - float(g['c.loss'])) < 1e-5
    for n, gv in _grads(loss, p).items():
        np.testing.assert_allclose(gv.numpy(), g['c.g.' + n], rtol=2e-4, atol=2e-7, err_msg=n)
    score, paths = R.crf_viterbi(h.detach(), mask, p['crf.fc.weight'].detach(), p['crf.fc.bias'].detach(),
                                 p['crf.transitions'].detach())
    np.testing.assert_allclose(score.numpy(), g['c.viterbi_score'], rtol=1e-5)
    assert [v for p_ in paths for v in p_] == g['c.viterbi_paths'].tolist()


def test_collater_layout():
    g = H.load('g8_collater')
    lens = g['lens'].tolist()
    samples = [{'id': i, 'target': g[f'tgt{i}'].tolist(), 'embeddings': torch.from_numpy(g[f'emb{i}']),
                'embeddings2': torch.from_numpy(g[f'emb2_{i}']), 'domain': None} for i in range(len(lens))]
    for crf in (True, False):
        for trunc, tv in ((False, 100), (True, 5), (True, 16)):
            b = R.collate(samples, crf, trunc, tv, has_second=True)
            key = f'crf{int(crf)}_tr{int(trunc)}_{tv}.'
            for f in ('src_tokens', 'src_tokens2', 'tgt_tokens', 'src_lengths', 'id'):
                got = b[f].numpy()
                assert got.shape == g[key + f].shape and got.dtype == g[key + f].dtype, (key, f)
                np.testing.assert_array_equal(got, g[key + f])
    assert R.collate([], True, False, 1) == {}


def test_boundaries_and_metrics():
    g = H.load('g11_boundaries')
    for i in range(6):
        assert R.get_boundaries(g[f'b{i}'].tolist()) == g[f'm{i}'].tolist()
    # textbook values of P_k / WindowDiff (no segeval here: definitions, not a pinned third-party run)
    ref = [0, 0, 1, 0, 0, 0, 1, 0, 0, 0]
    assert R.compute_pk(ref, ref) == 0.0 and R.compute_window_diff(ref, ref) == 0.0
    hyp = [0, 0, 0, 0, 1, 0, 0, 0, 0, 0]
    assert 0.0 < R.compute_pk(hyp, ref) <= 1.0
    assert R.compute_window_diff(hyp, ref) >= R.compute_pk(hyp, ref) - 1e-12


def test_lstm_init_statistics_fixture():
    """a5: what the Keras-style init must satisfy (used by the host-side init test as well)."""
    g = H.load('g9_lstm_init')
    for k, v in g.items():
        if k.startswith('orth.'):
            assert np.abs(v).max() < 1e-5
        if k.startswith('bias.') and 'bias_ih' in k:
            n = v.shape[0]
            assert np.all(v[n // 4:n // 2] == 1) and np.all(v[:n // 4] == 0) and np.all(v[n // 2:] == 0)
        if k.startswith('bias.') and 'bias_hh' in k:
            assert np.all(v == 0)
        if k.startswith('xavier.'):
            assert v[0] <= v[1] + 1e-6


def test_blocked_band_attention_equals_the_shifted_statement():
    """bench.py's cpu_baseline uses the block-wise matmul form; it must be the same function (values and gradients), ragged
    lengths, a document shorter than the radius, L not a multiple of the block."""
    g = torch.Generator().manual_seed(0)
    for (B, L, Hh, hd, w, block) in ((3, 50, 2, 8, 15, 16), (2, 130, 1, 4, 4, 64), (2, 7, 2, 4, 15, 64)):
        lengths = torch.randint(1, L + 1, (B,), generator=g)
        lengths[0] = L
        q, k, v = (torch.randn(B, L, Hh, hd, generator=g, dtype=torch.float64).requires_grad_(True) for _ in range(3))
        a = R.band_attention(q, k, v, lengths, w)
        b = R.band_attention_blocked(q, k, v, lengths, w, block)
        assert float((a - b).abs().max()) < 1e-12
        c = torch.randn(a.shape, generator=g, dtype=torch.float64)
        ga = torch.autograd.grad((a * c).sum(), [q, k, v])
        gb = torch.autograd.grad((b * c).sum(), [q, k, v])
        for x, y in zip(ga, gb):
            assert float((x - y).abs().max()) < 1e-12
    # and through the whole model on fixture g3b
    gg = H.load('g3b_transformer_w30')
    D, heads, ff, NL, window = [int(v) for v in gg['cfg']]
    p = {k_: _t(H.seeded_param(k_, shp, int(gg['seed']))) for k_, shp in H.band_param_shapes(D, ff, NL, 1).items()}
    sc = R.transformer_scores(_t(gg['x']), torch.from_numpy(gg['lengths']), p, heads, R.pyramidal_radii(NL, window),
                              attention=R.band_attention_blocked)
    np.testing.assert_allclose(sc.numpy(), gg['scores'], atol=2e-5, rtol=0)
