"""Shared test helpers: fixture loading and the documented seeded-weight recipe."""
import os
import zlib

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
DEAD = ('word_embeddings', 'query_global', 'key_global', 'value_global', 'pooler')


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + '.npz')))


def seeded_param(name, shape, seed):
    """Same recipe as tests/golden/make_golden.py::seeded_param (weights are regenerated, not stored)."""
    rng = np.random.default_rng((zlib.crc32(name.encode()) + seed) & 0xFFFFFFFF)
    u = rng.uniform(-1.0, 1.0, size=shape).astype(np.float32)
    if 'LayerNorm.weight' in name or name.endswith('norm1.weight') or name.endswith('norm2.weight'):
        return 1.0 + 0.1 * u
    if len(shape) >= 2:
        return u / np.sqrt(shape[-1]).astype(np.float32)
    return 0.1 * u


def band_param_shapes(D, ff, n_layers, n_out, max_pos=4096):
    """Names/shapes of the LIVE parameters of the reference's Transformer_segmenter (state_dict keys)."""
    s = {}
    e = 'model.model.embeddings.'
    s[e + 'token_type_embeddings.weight'] = (2, D)
    s[e + 'LayerNorm.weight'] = (D,)
    s[e + 'LayerNorm.bias'] = (D,)
    s[e + 'position_embeddings.weight'] = (max_pos, D)
    for i in range(n_layers):
        l = f'model.model.encoder.layer.{i}.'
        for n in ('query', 'key', 'value'):
            s[l + f'attention.self.{n}.weight'] = (D, D)
            s[l + f'attention.self.{n}.bias'] = (D,)
        s[l + 'attention.output.dense.weight'] = (D, D)
        s[l + 'attention.output.dense.bias'] = (D,)
        s[l + 'attention.output.LayerNorm.weight'] = (D,)
        s[l + 'attention.output.LayerNorm.bias'] = (D,)
        s[l + 'intermediate.dense.weight'] = (ff, D)
        s[l + 'intermediate.dense.bias'] = (ff,)
        s[l + 'output.dense.weight'] = (D, ff)
        s[l + 'output.dense.bias'] = (D,)
        s[l + 'output.LayerNorm.weight'] = (D,)
        s[l + 'output.LayerNorm.bias'] = (D,)
    s['classification.weight'] = (n_out, D)
    s['classification.bias'] = (n_out,)
    return s


def bilstm_param_shapes(D, H, n_layers, n_out, prefix='model.'):
    s = {}
    for k in range(n_layers):
        din = D if k == 0 else 2 * H
        for sfx in ('', '_reverse'):
            s[f'{prefix}rnn.weight_ih_l{k}{sfx}'] = (4 * H, din)
            s[f'{prefix}rnn.weight_hh_l{k}{sfx}'] = (4 * H, H)
            s[f'{prefix}rnn.bias_ih_l{k}{sfx}'] = (4 * H,)
            s[f'{prefix}rnn.bias_hh_l{k}{sfx}'] = (4 * H,)
    return s


def seeded_params(shapes, seed, dtype=torch.float32, requires_grad=False):
    out = {}
    for n, shp in shapes.items():
        t = torch.from_numpy(seeded_param(n, shp, seed)).to(dtype)
        out[n] = t.requires_grad_(requires_grad)
    return out


def split_tags(flat, lengths):
    out, o = [], 0
    for n in lengths:
        out.append([bool(v) for v in flat[o:o + int(n)]])
        o += int(n)
    return out


def checksum(a):
    a = np.asarray(a, dtype=np.float64).ravel()
    return np.array([a.sum(), np.abs(a).sum(), (a * a).sum()], dtype=np.float64)
