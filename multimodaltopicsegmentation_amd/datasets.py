"""On-disk formats of the precomputed-embedding pipeline (host side).

Mirror of ``utils/load_datasets_precomputed.py:56-224`` in the reference:

  * one ``.npy`` per document, ``[n_sentences, dim]`` (a stray leading/trailing unit dimension is squeezed, ``:159``);
  * several embedding directories joined with ``+`` are concatenated along the feature axis (early fusion, ``:136,:157-160``),
    file names come from the FIRST directory;
  * ``labs_dict.pkl``: ``{file stem: [0/1 per sentence]}``; the last label of every document is forced to 0 (``:172``),
    documents without labels are skipped with a warning (``:168-171``);
  * optional ``timing_info`` pickle ``{stem: [n_sentences, 2]}`` appended as two extra features (``:164-166``);
  * optional split JSON ``{"train": [...], "test": [...], "validation": [...]}``: documents are taken by popping from the
    END of the train list, then test, then validation, once per directory entry (``:143-154``) -> returns
    ``[[train, test, validation]]``; without a split: ``k_folds`` contiguous cross-validation folds ``[[train, test], …]``
    (``cross_validation_split`` ``:56-71``, no augmentation);
  * ``load_dataset_for_inference``: every file of a directory, ``(list of tensors, list of file names)`` (``:212-224``).

Items are ``(embeddings float tensor [n, dim], labels list[int], file name)`` tuples, what ``AudioPortionDataset``
(encoder_dataset.py) consumes.

K-split loading (extension, ``split_modalities=True``): with exactly two ``+``-joined directories the feature-axis concat is NOT
made; an item is ``(embeddings of the first directory, labels, file name, embeddings of the second directory)`` -- the first three
fields as above, so every consumer of the reference's items keeps working -- and ``second_input_of(items)`` turns the fourth into
the ``second_input=`` list of ``AudioPortionDataset``, whose collater then emits ``src_tokens`` and ``src_tokens2`` side by side.
``TextSegmenter(..., ksplit=True)`` hands that pair to an early-fusion model as one input of width D1 + D2 (the kernels read both
parts: mts_embed_layernorm_fwd2 / mts_cast_concat), so the concatenated batch exists neither on the host nor in HBM.
"""
import json
import os
import pickle

import numpy as np
import torch

_SKIP_STEMS = ("24580", "25539", "25684", "26071", "26214", "26321", "26427")     # over-long Podcast documents, reference :140


def cross_validation_split(dataset, num_folds=5, n_test_folds=1):
    """Contiguous folds: fold i tests on ``dataset[i*u : i*u + u*n_test_folds]`` with ``u = len // num_folds`` (reference :56-71)."""
    unit = len(dataset) // num_folds
    test_size = unit * n_test_folds
    folds = []
    for i in range(num_folds):
        a, b = i * unit, i * unit + test_size
        test = dataset[a:b]
        if i == num_folds + 1 - n_test_folds:
            test = test + dataset[:test_size // n_test_folds]
            train = dataset[test_size // n_test_folds: -test_size // n_test_folds]
        else:
            train = dataset[:a] + dataset[b:]
        folds.append([train, test])
    return folds


def _load_doc(directories, file, split_modalities=False):
    parts = [torch.from_numpy(np.load(os.path.join(root, file)).squeeze()) for root in directories]
    if split_modalities:
        return parts[0], parts[1]
    return torch.cat(parts, dim=-1), None


def second_input_of(items):
    """``second_input=`` list for AudioPortionDataset from items loaded with split_modalities=True."""
    return [(it[3], None, it[2]) for it in items]


def load_dataset_from_precomputed(embedding_directory, lab_file, delete_last_sentence=False, compute_confidence_intervals=False,
                                  inverse_augmentation=False, umap_project=False, k_folds=5, mask_inner_sentences=False,
                                  mask_probability=0.9, split=None, timing_info=None, split_modalities=False):
    if inverse_augmentation or umap_project:
        raise NotImplementedError('inverse_augmentation / umap_project are outside the hot path (SURVEY.md §8f)')
    standard_split = split is not None
    if standard_split:
        with open(split) as f:
            split = json.load(f)
        split = {k: list(v) for k, v in split.items()}
        data = [[], [], []]
    else:
        data = []
    original = []
    with open(lab_file, 'rb') as f:
        labs = pickle.load(f)
    assert isinstance(labs, dict)
    times = None
    if timing_info is not None:
        with open(timing_info, 'rb') as f:
            times = pickle.load(f)
    directories = embedding_directory.split('+')
    if split_modalities and (len(directories) != 2 or timing_info is not None or mask_inner_sentences):
        raise ValueError('split_modalities=True needs exactly two "+"-joined embedding directories and neither timing_info nor '
                         'mask_inner_sentences')
    for file in os.listdir(directories[0]):
        if file[-16:] == ':Zone.Identifier' or file[:-4] in _SKIP_STEMS:
            continue
        bucket = None
        if standard_split:
            if len(split['train']):
                file, bucket = split['train'].pop(), 0
            elif len(split['test']):
                file, bucket = split['test'].pop(), 1
            else:
                file, bucket = split['validation'].pop(), 2
        embs, embs2 = _load_doc(directories, file, split_modalities)
        stem = file[:-4]
        if times is not None:
            embs = torch.cat((embs, torch.tensor(times[stem])), dim=-1)
        if len(labs[stem]) < 1:
            print('Warning: {} has no data'.format(stem))
            continue
        labs[stem][-1] = 0
        if mask_inner_sentences:          # drop non-boundary sentences with probability 1 - mask_probability (reference :174-185)
            original.append((embs, labs[stem].copy(), file))
            np.random.seed(1)
            keep, kept_labs = [], []
            for i in range(embs.shape[0]):
                if np.random.rand() > mask_probability and not labs[stem][i]:
                    continue
                keep.append(i)
                kept_labs.append(labs[stem][i])
            embs = embs[keep]
            labs[stem] = kept_labs
        if sum(labs[stem]) < 1:
            print('Warning: {} has no positive topic boundaries'.format(stem))
        item = (embs, labs[stem], file) if embs2 is None else (embs, labs[stem], file, embs2)
        if standard_split:
            data[bucket].append(item)
        else:
            data.append(item)
    if standard_split:
        return [data]
    folds = cross_validation_split(data, num_folds=k_folds)
    if mask_inner_sentences:
        for i in range(len(folds)):
            folds[i][1] = [original[i]]
    return folds


def load_dataset_for_inference(embedding_directory):
    files = os.listdir(embedding_directory)
    data = [torch.from_numpy(np.load(os.path.join(embedding_directory, f)).squeeze()) for f in files]
    return data, files
