"""Batch layout of the reference's ``AudioPortionDataset`` / ``AudioPortionDatasetInference``
(EncoderDataset.py:18-152, :154-232): ragged per-document embeddings -> zero-padded batch dict

    {'id', 'src_tokens' [B, Lmax, D] fp32, 'src_lengths' [B] int64, 'tgt_tokens' [B, Lmax] fp32 (pad -1, or 0 for CRF),
     'src_tokens2' or None, 'domain' or None}

with ``truncate=True`` meaning "pad/truncate to exactly truncate_value" (3600 for the Transformer, train_fit.py:104-106).
Host-side only; the PCA/UMAP projection of the reference (:49-69, buggy re-slicing) is out of scope.
"""
import torch
from torch.utils.data import Dataset


def _merge(values, truncate, tv):
    if len(values[0].shape) < 2:
        return torch.stack(values)
    max_length = tv if truncate else max(v.size(0) for v in values)
    result = torch.zeros((len(values), max_length, values[0].shape[1]))
    for i, v in enumerate(values):
        n = min(tv, len(v)) if truncate else len(v)
        result[i, :n] = v[:n]
    return result


def _merge_tags(tags, truncate, tv, minus):
    max_length = tv if truncate else max(v.size(0) for v in tags)
    result = torch.zeros((len(tags), max_length)) - minus
    for i, v in enumerate(tags):
        n = min(tv, len(v)) if truncate else len(v)
        result[i, :n] = v[:n]
    return result


class AudioPortionDataset(Dataset):
    def __init__(self, lines, tag_to_ix, encoder='x-vectors', CRF=True, truncate=True, truncate_value=100, umap_project=False,
                 umap_project_value=100, umap_class=None, second_input=None, domain_adapt=False):
        if umap_project:
            raise NotImplementedError('PCA/UMAP projection (EncoderDataset.py:49-69) is outside the hot path')
        self.minus = 0 if CRF else 1                                   # EncoderDataset.py:23
        self.embeddings = [line[0] for line in lines]
        self.tgt_dataset = [line[1] for line in lines]
        self.embeddings2 = [line[0] for line in second_input] if second_input is not None else []
        self.truncate, self.tv = truncate, truncate_value
        self.encoder_name = encoder
        self.da = bool(domain_adapt)
        if self.da:
            self.domain = []
            for line in lines:
                try:
                    int(line[2][0])                                       # RadioNews files start with a digit (:41-45)
                    self.domain.append(1)
                except ValueError:
                    self.domain.append(0)
        else:
            self.domain = [None for _ in lines]

    def __getitem__(self, index):
        item = {'id': torch.tensor(index), 'target': self.tgt_dataset[index], 'embeddings': self.embeddings[index],
                'domain': self.domain[index]}
        if self.embeddings2:
            item['embeddings2'] = self.embeddings2[index]
        return item

    def __len__(self):
        return len(self.embeddings)

    def collater(self, samples):
        """Merge a list of samples to form a mini-batch (EncoderDataset.py:91-152)."""
        if len(samples) == 0:
            return {}
        src_tokens = _merge([s['embeddings'] for s in samples], self.truncate, self.tv)
        src_tokens2 = _merge([s['embeddings2'] for s in samples], self.truncate, self.tv) if self.embeddings2 else None
        tgt_tokens = _merge_tags([torch.as_tensor(s['target']) for s in samples], self.truncate, self.tv, self.minus)
        if self.truncate:
            src_lengths = torch.LongTensor([min(self.tv, len(s['embeddings'])) for s in samples])
        else:
            src_lengths = torch.LongTensor([len(s['embeddings']) for s in samples])
        return {'id': torch.tensor([int(s['id']) for s in samples]), 'src_tokens': src_tokens, 'src_lengths': src_lengths,
                'tgt_tokens': tgt_tokens, 'src_tokens2': src_tokens2,
                'domain': [s['domain'] for s in samples] if self.da else None}


class AudioPortionDatasetInference(Dataset):
    """EncoderDataset.py:154-232 (no targets; with truncate=True every length is reported as truncate_value, :221-222)."""

    def __init__(self, lines, encoder='x-vectors', CRF=True, truncate=False, truncate_value=100, umap_project=False,
                 umap_project_value=100, umap_class=None):
        if umap_project:
            raise NotImplementedError('PCA/UMAP projection is outside the hot path')
        self.minus = 0 if CRF else 1
        self.embeddings = lines
        self.truncate, self.tv = truncate, truncate_value
        self.encoder_name = encoder

    def __getitem__(self, index):
        return {'id': torch.tensor(index), 'embeddings': self.embeddings[index]}

    def __len__(self):
        return len(self.embeddings)

    def collater(self, samples):
        if len(samples) == 0:
            return {}
        src_tokens = _merge([s['embeddings'] for s in samples], self.truncate, self.tv)
        if self.truncate:
            src_lengths = torch.LongTensor([self.tv for _ in samples])
        else:
            src_lengths = torch.LongTensor([len(s['embeddings']) for s in samples])
        return {'id': torch.tensor([int(s['id']) for s in samples]), 'src_tokens': src_tokens, 'src_lengths': src_lengths}
