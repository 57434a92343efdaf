"""On-disk formats: multimodaltopicsegmentation_amd/datasets.py against what the reference's loader returned
for the same synthetic corpus (tests/golden/g12_loader.npz, recorded by tests/golden/make_golden.py::g12_loader)."""
import json
import os
import pickle

import numpy as np
import pytest
import torch

from tests import helpers as H


def _write_corpus(g, tmp):
    da, db = os.path.join(tmp, 'text'), os.path.join(tmp, 'audio')
    os.makedirs(da), os.makedirs(db)
    labs, times = {}, {}
    for n in g['names'].tolist():
        np.save(os.path.join(da, n), g[f'in.text.{n}'])
        np.save(os.path.join(db, n), g[f'in.audio.{n}'])
        labs[n[:-4]] = g[f'in.labs.{n}'].tolist()
        times[n[:-4]] = g[f'in.times.{n}'].tolist()
    lab_file, time_file, split_file = os.path.join(tmp, 'labs_dict.pkl'), os.path.join(tmp, 'times.pkl'), os.path.join(tmp, 'split.json')
    with open(time_file, 'wb') as f:
        pickle.dump(times, f)
    with open(split_file, 'w') as f:
        json.dump({k: g[f'split.{k}'].tolist() for k in ('train', 'test', 'validation')}, f)

    def fresh_labs():
        with open(lab_file, 'wb') as f:
            pickle.dump(labs, f)
        return lab_file
    return da, db, fresh_labs, time_file, split_file


def test_standard_split_matches_reference_loader(tmp_path):
    from multimodaltopicsegmentation_amd.datasets import load_dataset_from_precomputed
    g = H.load('g12_loader')
    da, db, fresh_labs, time_file, split_file = _write_corpus(g, str(tmp_path))
    for tag, kw in (('split', {}), ('split_times', {'timing_info': time_file})):
        res = load_dataset_from_precomputed(da + '+' + db, fresh_labs(), split=split_file, **kw)
        assert len(res) == 1 and len(res[0]) == 3
        for part, items in zip(('train', 'test', 'validation'), res[0]):
            names = g[f'out.{tag}.{part}.names'].tolist()
            assert [it[2] for it in items] == names, (tag, part)                 # same documents in the same (popped) order
            for j, (embs, lab, _) in enumerate(items):
                ref = g[f'out.{tag}.{part}.{j}.embs']
                assert tuple(embs.shape) == ref.shape and embs.shape[1] == (12 if tag == 'split_times' else 10)
                assert torch.equal(embs, torch.from_numpy(ref))                  # early fusion concat (+ timing columns), bit-exact
                assert list(lab) == g[f'out.{tag}.{part}.{j}.labs'].tolist()
                assert lab[-1] == 0                                               # last label forced to 0
    all_names = [n for part in ('train', 'test', 'validation') for n in g[f'out.split.{part}.names'].tolist()]
    assert 'doc04.npy' not in all_names                                           # the document without labels was skipped


def test_cross_validation_folds_and_inference(tmp_path):
    from multimodaltopicsegmentation_amd.datasets import cross_validation_split, load_dataset_for_inference, load_dataset_from_precomputed
    g = H.load('g12_loader')
    da, db, fresh_labs, _, _ = _write_corpus(g, str(tmp_path))
    folds = load_dataset_from_precomputed(da + '+' + db, fresh_labs(), k_folds=5)
    assert len(folds) == int(g['out.folds.n'])
    n_docs = len(g['names']) - 1                                                  # doc04 skipped
    for i, (tr, te) in enumerate(folds):
        assert len(te) == int(g[f'out.folds.{i}.test_size']) and len(tr) == len(g[f'out.folds.{i}.train'])
        assert len({it[2] for it in tr} | {it[2] for it in te}) == len(tr) + len(te) <= n_docs     # disjoint
    # fold membership follows directory order (os.listdir), which is not portable: check the rule itself
    data = list(range(11))
    f = cross_validation_split(data, 5)
    assert [x[1] for x in f] == [[0, 1], [2, 3], [4, 5], [6, 7], [8, 9]] and f[0][0] == list(range(2, 11))
    data_i, files = load_dataset_for_inference(da)
    assert sorted(files) == g['out.inference.files'].tolist()
    shapes = [list(data_i[files.index(n)].shape) for n in sorted(files)]
    assert shapes == g['out.inference.shapes'].tolist()


def test_loader_feeds_the_collater(tmp_path):
    from multimodaltopicsegmentation_amd.datasets import load_dataset_from_precomputed
    from multimodaltopicsegmentation_amd.encoder_dataset import AudioPortionDataset
    g = H.load('g12_loader')
    da, db, fresh_labs, _, split_file = _write_corpus(g, str(tmp_path))
    train = load_dataset_from_precomputed(da + '+' + db, fresh_labs(), split=split_file)[0][0]
    ds = AudioPortionDataset(train, {0: 0, 1: 1}, truncate=False)
    batch = ds.collater([ds[i] for i in range(len(ds))])
    lens = batch['src_lengths'].tolist()
    assert batch['src_tokens'].shape == (len(train), max(lens), 10)
    assert sorted(lens) == sorted(t[0].shape[0] for t in train)


def test_split_modalities_keeps_the_two_directories_apart(tmp_path):
    """K-split loading: same documents, labels and order as the concatenating loader (pinned by g12), but the text and audio
    matrices stay separate; the collater emits them as src_tokens / src_tokens2 and their concat equals the fused batch."""
    from multimodaltopicsegmentation_amd import AudioPortionDataset, load_dataset_from_precomputed, second_input_of
    g = H.load('g12_loader')
    da, db, fresh_labs, time_file, split_file = _write_corpus(g, str(tmp_path))
    fused = load_dataset_from_precomputed(da + '+' + db, fresh_labs(), split=split_file)[0]
    parts = load_dataset_from_precomputed(da + '+' + db, fresh_labs(), split=split_file, split_modalities=True)[0]
    for fa, pa in zip(fused, parts):
        assert [it[2] for it in fa] == [it[2] for it in pa]
        for f, p in zip(fa, pa):
            assert len(p) == 4 and p[1] == f[1]
            assert torch.equal(torch.cat((p[0], p[3]), dim=-1), f[0])
    train = parts[0]
    ds = AudioPortionDataset(train, {'0': 0, '1': 1}, CRF=False, truncate=False, second_input=second_input_of(train))
    b = ds.collater([ds[i] for i in range(len(ds))])
    dsf = AudioPortionDataset(fused[0], {'0': 0, '1': 1}, CRF=False, truncate=False)
    bf = dsf.collater([dsf[i] for i in range(len(dsf))])
    assert torch.equal(torch.cat((b['src_tokens'], b['src_tokens2']), dim=-1), bf['src_tokens'])
    assert torch.equal(b['tgt_tokens'], bf['tgt_tokens']) and torch.equal(b['src_lengths'], bf['src_lengths'])
    with pytest.raises(ValueError):
        load_dataset_from_precomputed(da, fresh_labs(), split=split_file, split_modalities=True)
