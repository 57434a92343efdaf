#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE itself on CPU.

Run in the build container only (needs /root/reference, which never travels to the GPU box):

    python tests/golden/make_golden.py

The reference is imported read-only with three stub modules: its own missing
``models/longformer_noffn.py`` (only touched when just_mha=True), ``pytorch_lightning`` (not installed:
``LightningModule`` -> nn.Module with no-op log/log_dict) and ``segeval`` (not installed, not called here).
Only inputs, weights (small cases) / a seed recipe (large cases) and the reference's outputs are stored --
never reference source.  Fixtures are data: ``*.npz`` with float32/int64 arrays.
"""
import os
import sys
import types
import zlib

import numpy as np
import torch
import torch.nn as nn

REF = os.environ.get('MTS_REFERENCE', '/root/reference')
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

_m = types.ModuleType('models.longformer_noffn')
_m.LongformerLayer = type('LongformerLayer', (nn.Module,), {})
sys.modules['models.longformer_noffn'] = _m
_pl = types.ModuleType('pytorch_lightning')


class _LM(nn.Module):
    def log(self, *a, **k):
        pass

    def log_dict(self, *a, **k):
        pass


_pl.LightningModule = _LM
sys.modules['pytorch_lightning'] = _pl
sys.modules['segeval'] = types.ModuleType('segeval')

from models.lightning_model import TextSegmenter, get_boundaries, WinPR  # noqa: E402
from models.CRF import CRF  # noqa: E402
from models.NeuralArchitectures import RNN, create_mask  # noqa: E402
from models.focal_loss import sigmoid_focal_loss  # noqa: E402
from models.RestrictedTransformerLayer import RestrictedTransformerEncoderLayer  # noqa: E402
import EncoderDataset  # noqa: E402

torch.set_num_threads(8)
DEAD = ('word_embeddings', 'query_global', 'key_global', 'value_global', 'pooler')


def used(name):
    return not any(d in name for d in DEAD)


def seeded_param(name, shape, seed):
    """The documented weight recipe shared with tests/ (tests/helpers.py::seeded_param)."""
    rng = np.random.default_rng((zlib.crc32(name.encode()) + seed) & 0xFFFFFFFF)
    u = rng.uniform(-1.0, 1.0, size=shape).astype(np.float32)
    if 'LayerNorm.weight' in name or name.endswith('norm1.weight') or name.endswith('norm2.weight'):
        return 1.0 + 0.1 * u
    if len(shape) >= 2:
        return u / np.sqrt(shape[-1]).astype(np.float32)
    return 0.1 * u


def reseed_model(model, seed):
    with torch.no_grad():
        for n, p in model.named_parameters():
            if used(n):
                p.copy_(torch.from_numpy(seeded_param(n, tuple(p.shape), seed)))


def grads_of(model):
    return {n: p.grad.detach().numpy().copy() for n, p in model.named_parameters() if used(n) and p.grad is not None}


def checksum(a):
    a = np.asarray(a, dtype=np.float64).ravel()
    return np.array([a.sum(), np.abs(a).sum(), (a * a).sum()], dtype=np.float64)


def make_targets(rng, lengths, L, pad):
    y = np.full((len(lengths), L), pad, dtype=np.float32)
    for b, n in enumerate(lengths):
        t = (rng.random(n) < 0.25).astype(np.float32)
        t[-1] = 0.0  # utils/load_datasets_precomputed.py:172
        y[b, :n] = t
    return y


def save(name, **arrays):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrays)
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


def pack_sd(prefix, sd):
    return {prefix + k: v for k, v in sd.items()}


# ------------------------------------------------------------------------------------------ G1 / G7
def g1_bilstm():
    rng = np.random.default_rng(101)
    D, H, NL, B, L = 64, 32, 2, 5, 23
    lengths = [23, 17, 1, 9, 23]
    x = rng.standard_normal((B, L, D)).astype(np.float32)
    for b, n in enumerate(lengths):
        x[b, n:] = 0
    y = make_targets(rng, lengths, L, -1)
    out = {'x': x, 'lengths': np.array(lengths, dtype=np.int64), 'tags': y,
           'cfg': np.array([D, H, NL], dtype=np.int64)}
    for loss_fn, tagset in (('FocalLoss', 2), ('BinaryCrossEntropy', 2), ('CrossEntropy', 2)):
        torch.manual_seed(7)
        ts = TextSegmenter(tagset, D, H, num_layers=NL, architecture='BiLSTM', loss_fn=loss_fn)
        m = ts.model
        m.device = 'cpu'
        xt, lt, yt = torch.from_numpy(x), torch.tensor(lengths), torch.from_numpy(y)
        loss = m.loss(xt, lt, yt)
        loss.backward()
        tag = loss_fn[:2]
        out.update(pack_sd(f'{tag}.w.', {n: p.detach().numpy().copy() for n, p in m.named_parameters()}))
        out.update(pack_sd(f'{tag}.g.', grads_of(m)))
        out[f'{tag}.loss'] = np.array(loss.item(), dtype=np.float64)
        for th in (0.4, 0.5):
            m.th = th
            with torch.no_grad():
                scores, tags = m(xt, lt)
            out[f'{tag}.scores'] = scores.numpy().copy()
            out[f'{tag}.tags{th}'] = np.concatenate([np.array(t, dtype=np.int64) for t in tags])
        m.th = None
        with torch.no_grad():
            _, tags = m(xt, lt)  # default threshold 0.4 (CRF.py:358)
        out[f'{tag}.tagsdefault'] = np.concatenate([np.array(t, dtype=np.int64) for t in tags])
    save('g1_bilstm_small', **out)


def g7_latefusion():
    rng = np.random.default_rng(107)
    D1, D2, H, NL, B, L = 40, 24, 16, 2, 4, 19
    lengths = [19, 11, 19, 3]
    x1 = rng.standard_normal((B, L, D1)).astype(np.float32)
    x2 = rng.standard_normal((B, L, D2)).astype(np.float32)
    y = make_targets(rng, lengths, L, -1)
    torch.manual_seed(8)
    ts = TextSegmenter(2, [D1, D2], H, num_layers=NL, architecture='BiLSTMLateFusion', loss_fn='FocalLoss')
    m = ts.model
    m.device = 'cpu'
    lt = torch.tensor(lengths)
    loss = m.loss(torch.from_numpy(x1), torch.from_numpy(x2), lt, torch.from_numpy(y))
    loss.backward()
    m.th = 0.5
    with torch.no_grad():
        scores, tags = m(torch.from_numpy(x1), torch.from_numpy(x2), lt)
    out = {'x1': x1, 'x2': x2, 'lengths': np.array(lengths, dtype=np.int64), 'tags': y,
           'cfg': np.array([D1, D2, H, NL], dtype=np.int64), 'loss': np.array(loss.item()),
           'scores': scores.numpy().copy(),
           'tags0.5': np.concatenate([np.array(t, dtype=np.int64) for t in tags])}
    out.update(pack_sd('w.', {n: p.detach().numpy().copy() for n, p in m.named_parameters()}))
    out.update(pack_sd('g.', grads_of(m)))
    save('g7_latefusion_small', **out)


# ------------------------------------------------------------------------------------------ G3
def g3_transformer(name, D, heads, ff, NL, window, B, L, lengths, loss_fn, seed):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, L, D)).astype(np.float32)
    y = make_targets(rng, lengths, L, -1)
    torch.manual_seed(seed)
    ts = TextSegmenter(2, D, ff, num_layers=NL, architecture='Transformer', loss_fn=loss_fn, nheads=heads,
                       attention_window=window)
    m = ts.model
    m.device = 'cpu'
    reseed_model(m, seed)   # HF init leaves biases at 0; use the seeded recipe so every term is exercised
    m.eval()
    xt, lt, yt = torch.from_numpy(x), torch.tensor(lengths), torch.from_numpy(y)
    loss = m.loss(xt, lt, yt)
    loss.backward()
    out = {'x': x, 'lengths': np.array(lengths, dtype=np.int64), 'tags': y,
           'cfg': np.array([D, heads, ff, NL, window], dtype=np.int64), 'seed': np.array(seed),
           'loss': np.array(loss.item(), dtype=np.float64)}
    with torch.no_grad():
        hidden = m.model(xt, lt)
        out['hidden'] = hidden.numpy().copy()
        for th in (0.4, 0.5):
            m.th = th
            scores, tags = m(xt, lt)
            out[f'tags{th}'] = np.concatenate([np.array(t, dtype=np.int64) for t in tags])
        out['scores'] = scores.numpy().copy()
    g = grads_of(m)
    # position embeddings: only rows 2..L+1 carry gradient; store that slice
    pe = 'model.model.embeddings.position_embeddings.weight'
    g[pe] = g[pe][: L + 2 + 64]
    out.update(pack_sd('g.', g))
    save(name, **out)


# ------------------------------------------------------------------------------------------ G2 / G4 (large, seeded)
def g2_bilstm_large():
    rng = np.random.default_rng(202)
    D, H, NL, B, L = 1792, 256, 2, 4, 64
    lengths = [64, 40, 64, 21]
    x = rng.standard_normal((B, L, D)).astype(np.float32)
    y = make_targets(rng, lengths, L, -1)
    ts = TextSegmenter(2, D, H, num_layers=NL, architecture='BiLSTM', loss_fn='FocalLoss')
    m = ts.model
    m.device = 'cpu'
    reseed_model(m, 202)
    xt, lt, yt = torch.from_numpy(x), torch.tensor(lengths), torch.from_numpy(y)
    loss = m.loss(xt, lt, yt)
    loss.backward()
    m.th = 0.5
    with torch.no_grad():
        scores, tags = m(xt, lt)
    out = {'x': x.astype(np.float16), 'lengths': np.array(lengths, dtype=np.int64), 'tags': y,
           'cfg': np.array([D, H, NL], dtype=np.int64), 'seed': np.array(202),
           'loss': np.array(loss.item(), dtype=np.float64), 'scores': scores.numpy().copy(),
           'tags0.5': np.concatenate([np.array(t, dtype=np.int64) for t in tags])}
    for n, gv in grads_of(m).items():
        out['gsum.' + n] = checksum(gv)
        out['ghead.' + n] = gv.ravel()[:32].copy()
    save('g2_bilstm_1792', **out)


def g4_transformer_large():
    D, heads, ff, NL, window, B, L = 1792, 8, 256, 1, 30, 2, 256
    lengths = [256, 173]
    rng = np.random.default_rng(404)
    x = rng.standard_normal((B, L, D)).astype(np.float16)   # stored as fp16; both sides widen the SAME values
    y = make_targets(rng, lengths, L, -1)
    ts = TextSegmenter(2, D, ff, num_layers=NL, architecture='Transformer', loss_fn='FocalLoss', nheads=heads,
                       attention_window=window)
    m = ts.model
    m.device = 'cpu'
    reseed_model(m, 404)
    m.eval()
    xt, lt, yt = torch.from_numpy(x.astype(np.float32)), torch.tensor(lengths), torch.from_numpy(y)
    loss = m.loss(xt, lt, yt)
    loss.backward()
    m.th = 0.5
    with torch.no_grad():
        scores, tags = m(xt, lt)
    out = {'x': x, 'lengths': np.array(lengths, dtype=np.int64), 'tags': y,
           'cfg': np.array([D, heads, ff, NL, window], dtype=np.int64), 'seed': np.array(404),
           'loss': np.array(loss.item(), dtype=np.float64), 'scores': scores.numpy().copy(),
           'tags0.5': np.concatenate([np.array(t, dtype=np.int64) for t in tags])}
    for n, gv in grads_of(m).items():
        if 'position_embeddings' in n:
            gv = gv[: L + 2]
        out['gsum.' + n] = checksum(gv)
        out['ghead.' + n] = gv.ravel()[:32].copy()
    save('g4_transformer_1792', **out)


# ------------------------------------------------------------------------------------------ G5 CRF
def g5_crf():
    rng = np.random.default_rng(505)
    B, L, F, T = 4, 17, 24, 2
    lengths = [17, 9, 1, 17]
    feats = rng.standard_normal((B, L, F)).astype(np.float32)
    tags = np.zeros((B, L), dtype=np.float32)   # CRF collater pads with 0 (EncoderDataset.py:23)
    for b, n in enumerate(lengths):
        tags[b, :n] = (rng.random(n) < 0.3)
    torch.manual_seed(5)
    crf = CRF(F, T)
    ft = torch.from_numpy(feats).requires_grad_(True)
    mask = create_mask(ft, torch.tensor(lengths))
    loss = crf.loss(ft, torch.from_numpy(tags), mask)
    loss.backward()
    with torch.no_grad():
        score, paths = crf(ft, mask)
    out = {'features': feats, 'tags': tags, 'lengths': np.array(lengths, dtype=np.int64),
           'fc.weight': crf.fc.weight.detach().numpy().copy(), 'fc.bias': crf.fc.bias.detach().numpy().copy(),
           'transitions': crf.transitions.detach().numpy().copy(), 'loss': np.array(loss.item(), dtype=np.float64),
           'g.features': ft.grad.numpy().copy(), 'g.fc.weight': crf.fc.weight.grad.numpy().copy(),
           'g.fc.bias': crf.fc.bias.grad.numpy().copy(), 'g.transitions': crf.transitions.grad.numpy().copy(),
           'viterbi_score': score.numpy().copy(),
           'viterbi_paths': np.concatenate([np.array(p, dtype=np.int64) for p in paths])}
    # RNN -> CRF composition (the reference's BiRnnCrf wrapper is broken): built from the
    # reference's own RNN and CRF classes
    D, H, NL = 20, 12, 2
    x = rng.standard_normal((B, L, D)).astype(np.float32)
    torch.manual_seed(6)
    rnn = RNN(D, H, NL, 2, True, 0.0, 0.0, batch_first=True, LSTM=True)
    crf2 = CRF(2 * H, T)
    xt = torch.from_numpy(x)
    lt = torch.tensor(lengths)
    h = rnn(xt, lt)
    mask2 = create_mask(xt, lt)
    loss2 = crf2.loss(h, torch.from_numpy(tags), mask2)
    loss2.backward()
    with torch.no_grad():
        score2, paths2 = crf2(rnn(xt, lt), mask2)
    out.update({'c.x': x, 'c.cfg': np.array([D, H, NL], dtype=np.int64), 'c.loss': np.array(loss2.item()),
                'c.viterbi_score': score2.numpy().copy(),
                'c.viterbi_paths': np.concatenate([np.array(p, dtype=np.int64) for p in paths2])})
    out.update(pack_sd('c.w.model.', {n: p.detach().numpy().copy() for n, p in rnn.named_parameters()}))
    out.update(pack_sd('c.w.crf.', {n: p.detach().numpy().copy() for n, p in crf2.named_parameters()}))
    out.update(pack_sd('c.g.model.', {n: p.grad.numpy().copy() for n, p in rnn.named_parameters()}))
    out.update(pack_sd('c.g.crf.', {n: p.grad.numpy().copy() for n, p in crf2.named_parameters()}))
    save('g5_crf', **out)


# ------------------------------------------------------------------------------------------ G6 focal
def g6_focal():
    rng = np.random.default_rng(606)
    x = np.concatenate([rng.standard_normal(64) * 3, np.array([-80., -30., -10., 0., 10., 30., 80., 1e-4])]).astype(np.float32)
    y = (rng.random(x.shape[0]) < 0.4).astype(np.float32)
    out = {'x': x, 'y': y}
    for alpha, gamma in ((0.9, 2.0), (0.25, 2.0), (-1.0, 2.0), (0.9, 0.0), (0.5, 3.0)):
        xt = torch.from_numpy(x).requires_grad_(True)
        fl = sigmoid_focal_loss(alpha=alpha, gamma=gamma, reduction='mean')
        l = fl(xt, torch.from_numpy(y))
        l.backward()
        out[f'loss_a{alpha}_g{gamma}'] = np.array(l.item(), dtype=np.float64)
        out[f'grad_a{alpha}_g{gamma}'] = xt.grad.numpy().copy()
    save('g6_focal', **out)


# ------------------------------------------------------------------------------------------ G8 collater
def g8_collater():
    rng = np.random.default_rng(808)
    lens = [7, 3, 12, 1]
    D = 6
    lines = [(torch.from_numpy(rng.standard_normal((n, D)).astype(np.float32)),
              (rng.random(n) < 0.3).astype(np.int64).tolist(), f'doc{n}') for n in lens]
    lines2 = [(torch.from_numpy(rng.standard_normal((n, 4)).astype(np.float32)), None, f'doc{n}') for n in lens]
    out = {'lens': np.array(lens, dtype=np.int64)}
    for i, (e, t, _) in enumerate(lines):
        out[f'emb{i}'] = e.numpy()
        out[f'tgt{i}'] = np.array(t, dtype=np.int64)
        out[f'emb2_{i}'] = lines2[i][0].numpy()
    for crf in (True, False):
        for trunc, tv in ((False, 100), (True, 5), (True, 16)):
            ds = EncoderDataset.AudioPortionDataset(lines, {'0': 0, '1': 1}, CRF=crf, truncate=trunc, truncate_value=tv,
                                                    second_input=lines2)
            batch = ds.collater([ds[i] for i in range(len(ds))])
            key = f'crf{int(crf)}_tr{int(trunc)}_{tv}.'
            out[key + 'src_tokens'] = batch['src_tokens'].numpy()
            out[key + 'src_tokens2'] = batch['src_tokens2'].numpy()
            out[key + 'tgt_tokens'] = batch['tgt_tokens'].numpy()
            out[key + 'src_lengths'] = batch['src_lengths'].numpy()
            out[key + 'id'] = batch['id'].numpy()
    save('g8_collater', **out)


# ------------------------------------------------------------------------------------------ G9 init / G10 legacy
def g9_init():
    torch.manual_seed(9)
    rnn = RNN(48, 32, 2, 2, True, 0.0, 0.0, batch_first=True, LSTM=True)
    out = {}
    for n, p in rnn.named_parameters():
        a = p.detach().numpy()
        if 'weight_hh' in n:
            out['orth.' + n] = (a.T @ a - np.eye(a.shape[1])).astype(np.float32)   # W_hh [4H,H]: columns orthonormal
        if 'bias' in n:
            out['bias.' + n] = a.copy()
        if 'weight_ih' in n:
            bound = np.sqrt(6.0 / (a.shape[0] + a.shape[1]))
            out['xavier.' + n] = np.array([np.abs(a).max(), bound, a.std()], dtype=np.float64)
    save('g9_lstm_init', **out)


def g10_legacy():
    rng = np.random.default_rng(1010)
    d, h, ff, w, B, L = 32, 4, 48, 5, 2, 21
    torch.manual_seed(10)
    layer = RestrictedTransformerEncoderLayer(d, h, dim_feedforward=ff, window_size=w, dropout=0.0, batch_first=True)
    layer.eval()
    with torch.no_grad():
        for n, p in layer.named_parameters():
            p.copy_(torch.from_numpy(seeded_param(n, tuple(p.shape), 1010)))
    x = rng.standard_normal((B, L, d)).astype(np.float32)
    with torch.no_grad():
        y = layer(torch.from_numpy(x))
    out = {'x': x, 'y': y.numpy().copy(), 'cfg': np.array([d, h, ff, w], dtype=np.int64)}
    out.update(pack_sd('w.', {n: p.detach().numpy().copy() for n, p in layer.named_parameters()}))
    save('g10_legacy_layer', **out)


def g11_boundaries():
    rng = np.random.default_rng(1111)
    out = {}
    for i in range(6):
        b = (rng.random(int(rng.integers(3, 40))) < 0.3).astype(np.int64)
        out[f'b{i}'] = b
        out[f'm{i}'] = np.array(get_boundaries(b.tolist()), dtype=np.int64)
    save('g11_boundaries', **out)


def g12_loader():
    """On-disk formats (utils/load_datasets_precomputed.py:103-224): a synthetic corpus written to a temp dir, loaded by the
    reference with a standard split (+ timing features), without a split (5 folds) and for inference.  The fixture holds the
    corpus itself (inputs) and, per returned item, its file name, labels and embedding matrix (outputs)."""
    import json
    import pickle
    import tempfile
    from utils.load_datasets_precomputed import load_dataset_for_inference, load_dataset_from_precomputed
    rng = np.random.default_rng(1212)
    names = [f'doc{i:02d}.npy' for i in range(11)]
    lens = [int(v) for v in rng.integers(3, 12, len(names))]
    out = {'names': np.array(names), 'lens': np.array(lens)}
    with tempfile.TemporaryDirectory() as tmp:
        da, db = os.path.join(tmp, 'text'), os.path.join(tmp, 'audio')
        os.makedirs(da), os.makedirs(db)
        labs, times = {}, {}
        for i, (n, ln) in enumerate(zip(names, lens)):
            a = rng.standard_normal((ln, 6)).astype(np.float32)
            b = rng.standard_normal((1, ln, 4) if i == 2 else (ln, 4)).astype(np.float32)   # doc02: stray leading unit dim
            np.save(os.path.join(da, n), a), np.save(os.path.join(db, n), b)
            lab = (rng.random(ln) < 0.35).astype(int).tolist()
            if i == 4:
                lab = []                      # "has no data": skipped
            if i == 5:
                lab = [0] * ln                # no positive boundary: kept with a warning
            lab = list(lab)
            if i == 6 and ln:
                lab[-1] = 1                   # last label is forced to 0 by the loader
            labs[n[:-4]] = lab
            times[n[:-4]] = rng.random((ln, 2)).astype(np.float32).tolist()
            out[f'in.text.{n}'], out[f'in.audio.{n}'] = a, b
            out[f'in.labs.{n}'] = np.array(lab, dtype=np.int64)
            out[f'in.times.{n}'] = np.array(times[n[:-4]], dtype=np.float32)
        lab_file, time_file, split_file = os.path.join(tmp, 'labs_dict.pkl'), os.path.join(tmp, 'times.pkl'), os.path.join(tmp, 'split.json')
        with open(lab_file, 'wb') as f:
            pickle.dump(labs, f)
        with open(time_file, 'wb') as f:
            pickle.dump(times, f)
        split = {'train': names[:6], 'test': names[6:9], 'validation': names[9:]}
        with open(split_file, 'w') as f:
            json.dump(split, f)
        out['split.train'], out['split.test'], out['split.validation'] = (np.array(split[k]) for k in ('train', 'test', 'validation'))

        def pack(prefix, items):
            out[prefix + '.names'] = np.array([it[2] for it in items])
            for j, it in enumerate(items):
                out[f'{prefix}.{j}.embs'] = it[0].numpy()
                out[f'{prefix}.{j}.labs'] = np.array(it[1], dtype=np.int64)

        for tag, kw in (('split', {}), ('split_times', {'timing_info': time_file})):
            with open(lab_file, 'wb') as f:       # the loader mutates the label lists it unpickles: start from a fresh pickle
                pickle.dump(labs, f)
            res = load_dataset_from_precomputed(da + '+' + db, lab_file, split=split_file, **kw)
            assert len(res) == 1 and len(res[0]) == 3
            for part, items in zip(('train', 'test', 'validation'), res[0]):
                pack(f'out.{tag}.{part}', items)
        folds = load_dataset_from_precomputed(da + '+' + db, lab_file, k_folds=5)
        out['out.folds.n'] = np.array(len(folds))
        for i, (tr, te) in enumerate(folds):
            out[f'out.folds.{i}.train'] = np.array(sorted(it[2] for it in tr))
            out[f'out.folds.{i}.test_size'] = np.array(len(te))
        data, files = load_dataset_for_inference(da)
        out['out.inference.files'] = np.array(sorted(files))
        out['out.inference.shapes'] = np.array([list(data[files.index(f)].shape) for f in sorted(files)])
    save('g12_loader', **out)


def g13_inference_collater():
    """``AudioPortionDatasetInference.collater`` (EncoderDataset.py:198-232): no targets; truncate=True reports every length as
    truncate_value (:221-222)."""
    rng = np.random.default_rng(1313)
    lens = [9, 2, 14, 1, 6]
    D = 5
    embs = [torch.from_numpy(rng.standard_normal((n, D)).astype(np.float32)) for n in lens]
    out = {'lens': np.array(lens, dtype=np.int64)}
    for i, e in enumerate(embs):
        out[f'emb{i}'] = e.numpy()
    for trunc, tv in ((False, 100), (True, 4), (True, 20)):
        ds = EncoderDataset.AudioPortionDatasetInference(embs, truncate=trunc, truncate_value=tv)
        batch = ds.collater([ds[i] for i in range(len(ds))])
        key = f'tr{int(trunc)}_{tv}.'
        assert sorted(batch.keys()) == ['id', 'src_lengths', 'src_tokens'], batch.keys()
        out[key + 'src_tokens'] = batch['src_tokens'].numpy()
        out[key + 'src_lengths'] = batch['src_lengths'].numpy()
        out[key + 'id'] = batch['id'].numpy()
    ds = EncoderDataset.AudioPortionDatasetInference(embs)
    out['empty_is_dict'] = np.array(int(ds.collater([]) == {}))
    save('g13_inference_collater', **out)


def config0_corpus(names, seed=1414, D=768, lo=12, hi=60):
    """Synthetic stand-in for the NonNews-SBBC corpus (the Zenodo data is not in the tree): one [n_sent, 768] fp32 matrix per file
    name of NonNews-SBBC/NonNews_split.json, labels ~ Bernoulli(0.2).  Seed recipe shared with tests/ (the matrices are
    regenerated there, not stored)."""
    docs = {}
    for name in names:
        rng = np.random.default_rng((zlib.crc32(name.encode()) + seed) & 0xFFFFFFFF)
        n = int(rng.integers(lo, hi))
        emb = rng.standard_normal((n, D)).astype(np.float32)
        lab = (rng.random(n) < 0.2).astype(int).tolist()
        docs[name] = (emb, lab)
    return docs


def g14_config0():
    """BASELINE.json configs[0]: the reference's own CPU plumbing run -- BiLSTM tagger (H=256, 2 layers, focal loss, Adam lr 1e-3,
    run_nonnews_unimodal.sh) on precomputed 768-d text-only embeddings, NonNews-SBBC standard split (37/9/8), batch 8:
    load_dataset_from_precomputed -> AudioPortionDataset.collater -> TextSegmenter.training_step / validation_step / model().
    Records the split's file names, per-document lengths and labels, the order the loader returns documents in, the loss of every
    training step of one epoch, the validation losses after it and the boundaries predicted for the test documents."""
    import json
    import pickle
    import tempfile
    from torch.utils.data import DataLoader
    from utils.load_datasets_precomputed import load_dataset_from_precomputed
    with open(os.path.join(REF, 'NonNews-SBBC', 'NonNews_split.json')) as f:
        split = json.load(f)
    names = split['train'] + split['validation'] + split['test']
    docs = config0_corpus(names)
    out = {'split.train': np.array(split['train']), 'split.validation': np.array(split['validation']),
           'split.test': np.array(split['test']), 'cfg': np.array([768, 256, 2, 8], dtype=np.int64), 'seed': np.array(1414)}
    for n in names:
        out[f'len.{n}'] = np.array(docs[n][0].shape[0])
        out[f'lab.{n}'] = np.array(docs[n][1], dtype=np.int64)
        out[f'embsum.{n}'] = checksum(docs[n][0])
    with tempfile.TemporaryDirectory() as tmp:
        d = os.path.join(tmp, 'roberta')
        os.makedirs(d)
        for n in names:
            np.save(os.path.join(d, n), docs[n][0])
        lab_file, split_file = os.path.join(tmp, 'labs_dict.pkl'), os.path.join(tmp, 'split.json')
        with open(lab_file, 'wb') as f:
            pickle.dump({n[:-4]: list(docs[n][1]) for n in names}, f)
        with open(split_file, 'w') as f:
            json.dump(split, f)
        folds = load_dataset_from_precomputed(d, lab_file, split=split_file)
    train, test, valid = folds[0]
    out['order.train'], out['order.test'], out['order.validation'] = (np.array([it[2] for it in part]) for part in (train, test, valid))
    tag_to_ix = {'0': 0, '1': 1}
    mk = lambda part: EncoderDataset.AudioPortionDataset(part, tag_to_ix, encoder='roberta', CRF=False, truncate=False, truncate_value=100)
    tr_ds, va_ds, te_ds = mk(train), mk(valid), mk(test)
    bs = 8
    tr = DataLoader(tr_ds, batch_size=min(bs, len(tr_ds)), collate_fn=tr_ds.collater)          # train_fit.py:141
    va = DataLoader(va_ds, batch_size=min(bs, len(va_ds)), collate_fn=va_ds.collater)
    te = DataLoader(te_ds, batch_size=1, collate_fn=te_ds.collater)                              # train_fit.py:154
    torch.manual_seed(14)
    ts = TextSegmenter(2, 768, 256, num_layers=2, architecture='BiLSTM', loss_fn='FocalLoss', lr=1e-3, optimizer='Adam',
                       threshold=0.4)
    ts.model.device = 'cpu'
    reseed_model(ts.model, 1414)
    opt = ts.configure_optimizers()['optimizer']
    losses = []
    for bi, batch in enumerate(tr):
        opt.zero_grad()
        loss = ts.training_step(batch, bi)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    out['train_losses'] = np.array(losses, dtype=np.float64)
    with torch.no_grad():
        out['val_losses'] = np.array([ts.validation_step(b, i).item() for i, b in enumerate(va)], dtype=np.float64)
        tags, scores = [], []
        ts.model.th = 0.4
        for b in te:
            sc, tg = ts.model(b['src_tokens'], b['src_lengths'])
            tags.append(np.array(tg[0], dtype=np.int64))
            scores.append(sc[0, :, 0].numpy().copy())
    out['test_tags'] = np.concatenate(tags)
    out['test_scores'] = np.concatenate(scores)
    save('g14_config0_plumbing', **out)


def g15_adjacent_encoders():
    """SURVEY.md §8(f4) evidence: what the reference itself does when its adjacent-encoder options are selected on a tagger call
    (GRU: NeuralArchitectures.py:46-50,118-121; unidirectional: :134-145; cosine auxiliary loss: lightning_model.py:276-277;
    BiLSTMRestrictedMHA: CRF.py:636-684 needs models/longformer_noffn.py, which the reference tree does not contain).
    Stored: exception type and message per option (empty type = the call succeeded)."""
    x = torch.randn(3, 7, 16, generator=torch.Generator().manual_seed(15))
    l = torch.tensor([7, 4, 2])
    y = (torch.rand(3, 7, generator=torch.Generator().manual_seed(16)) < .3).float()
    out = {}

    def record(tag, fn):
        try:
            fn()
            out[tag + '.type'], out[tag + '.msg'] = np.array(''), np.array('')
        except Exception as e:  # noqa: BLE001
            out[tag + '.type'], out[tag + '.msg'] = np.array(type(e).__name__), np.array(str(e)[:200])
        print(tag, out[tag + '.type'], out[tag + '.msg'])

    def tagger(**kw):
        ts = TextSegmenter(2, 16, 8, num_layers=1, architecture='BiLSTM', loss_fn='FocalLoss', **kw)
        ts.model.device = 'cpu'
        return ts

    record('gru.loss', lambda: tagger(LSTM=False).model.loss(x, l, y))
    record('gru.forward', lambda: tagger(LSTM=False).model(x, l))
    record('unidirectional.loss', lambda: tagger(bidirectional=False).model.loss(x, l, y))
    record('unidirectional.forward', lambda: tagger(bidirectional=False).model(x, l))
    record('cosine.training_step', lambda: tagger(cosine_loss=True).training_step({'src_tokens': x, 'tgt_tokens': y, 'src_lengths': l}, 0))
    # BiLSTMRestrictedMHA (CRF.py:636-684) builds its attention from models/longformer_noffn.py, which the reference tree does not
    # ship: `import models.CRF` itself fails with ModuleNotFoundError on an unmodified checkout.  THAT is the record (this
    # generator can only import models.CRF at all because it registers an empty stand-in module, so whatever a constructor call
    # raised here would describe the stand-in, not the reference -- round 2 stored exactly that artefact; dropped).
    out['longformer_noffn_source_present'] = np.array(int(os.path.exists(os.path.join(REF, 'models', 'longformer_noffn.py'))))
    importers = sorted(f for f in os.listdir(os.path.join(REF, 'models')) if f.endswith('.py')
                       and any(ln.startswith(('from models.longformer_noffn import', 'import models.longformer_noffn'))
                               for ln in open(os.path.join(REF, 'models', f)).read().splitlines()))
    out['longformer_noffn_imported_at_module_level_by'] = np.array(','.join(importers))
    print('longformer_noffn present:', out['longformer_noffn_source_present'], 'imported by:', importers)
    save('g15_adjacent_encoders', **out)


def g16_winpr():
    """WinPR of the reference itself (lightning_model.py:57-124; pure Python, no third-party call): random segmentations at several
    window sizes, k larger than the document (every start index negative: python's wrap-around slicing feeds the 'previous span'
    test), k = 1, an empty hypothesis (ZeroDivisionError caught upstream -> (0, 0, 0)), all-boundary inputs, and the two inputs on
    which the reference RAISES ZeroDivisionError out of the function (recall with TP = FN = 0; f1 with precision = recall = 0).
    Stored flat: case c uses ref/hyp[off[c]:off[c+1]], k[c]; result prf[c] (nan where it raised), raised[c] = exception name."""
    rng = np.random.default_rng(1616)
    cases = []
    for N, k, pr, ph in [(40, 10, .15, .15), (40, 3, .3, .1), (25, 1, .2, .2), (7, 10, .3, .3), (12, 12, .25, .4), (60, 5, .05, .3),
                         (33, 7, .5, .5), (90, 10, .1, .1), (5, 2, .4, .4), (1, 10, 1., 1.), (18, 4, .2, .6), (64, 16, .1, .2)]:
        cases.append(((rng.random(N) < pr).astype(int).tolist(), (rng.random(N) < ph).astype(int).tolist(), k))
    r = (rng.random(30) < .2).astype(int).tolist()
    cases.append((r, [0] * 30, 10))                          # empty hypothesis
    cases.append(([1] * 15, [1] * 15, 4))                    # every position a boundary
    cases.append((r, list(r), 10))                           # perfect hypothesis
    cases.append(([0] * 20, [0] * 20, 10))                   # nothing anywhere
    cases.append(([0] * 20, [0, 0, 1] + [0] * 17, 10))       # raises: recall = 0 / 0
    cases.append(([0, 0, 0, 0, 0, 0, 0, 0, 0, 1], [1, 0, 0, 0, 0, 0, 0, 0, 0, 0], 1))   # may raise: precision = recall = 0
    cases.append(([0, 1, 0, 0, 1, 0], [0, 0, 1, 0, 0, 1], 2))     # near misses
    refs, hyps, off, ks, prf, raised = [], [], [0], [], [], []
    for ref, hyp, k in cases:
        try:
            res = WinPR(list(ref), list(hyp), k)
            prf.append([float(v) for v in res])
            raised.append('')
        except Exception as e:  # noqa: BLE001
            prf.append([np.nan] * 3)
            raised.append(type(e).__name__)
        refs += ref
        hyps += hyp
        off.append(len(refs))
        ks.append(k)
        print('winpr', len(ref), k, prf[-1], raised[-1])
    save('g16_winpr', ref=np.array(refs, dtype=np.int64), hyp=np.array(hyps, dtype=np.int64), off=np.array(off, dtype=np.int64),
         k=np.array(ks, dtype=np.int64), prf=np.array(prf, dtype=np.float64), raised=np.array(raised))


if __name__ == '__main__':
    which = sys.argv[1:] or ['g1', 'g7', 'g3a', 'g3b', 'g3c', 'g2', 'g4', 'g5', 'g6', 'g8', 'g9', 'g10', 'g11', 'g12', 'g13', 'g14', 'g15', 'g16']
    for w in which:
        if w == 'g1':
            g1_bilstm()
        elif w == 'g7':
            g7_latefusion()
        elif w == 'g3a':   # 2 layers, pyramidal windows [8, 4] -> radii [4, 2]; L=21 not a multiple of 8
            g3_transformer('g3a_transformer_w4x2', 64, 4, 32, 2, 4, 3, 21, [21, 13, 5], 'FocalLoss', 301)
        elif w == 'g3b':   # 1 layer, window 30 -> radius 15 (BASELINE "win=15"); L=50, docs shorter than the window
            g3_transformer('g3b_transformer_w30', 64, 4, 32, 1, 30, 4, 50, [50, 37, 8, 1], 'BinaryCrossEntropy', 302)
        elif w == 'g3c':   # CrossEntropy head (2 logits), hd=24 (not a power of two)
            g3_transformer('g3c_transformer_ce', 96, 4, 40, 1, 6, 3, 18, [18, 18, 7], 'CrossEntropy', 303)
        elif w == 'g2':
            g2_bilstm_large()
        elif w == 'g4':
            g4_transformer_large()
        elif w == 'g5':
            g5_crf()
        elif w == 'g6':
            g6_focal()
        elif w == 'g8':
            g8_collater()
        elif w == 'g9':
            g9_init()
        elif w == 'g12':
            g12_loader()
        elif w == 'g10':
            g10_legacy()
        elif w == 'g11':
            g11_boundaries()
        elif w == 'g13':
            g13_inference_collater()
        elif w == 'g14':
            g14_config0()
        elif w == 'g15':
            g15_adjacent_encoders()
        elif w == 'g16':
            g16_winpr()
