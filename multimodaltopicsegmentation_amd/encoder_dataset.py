"""Batch layout of the reference's ``AudioPortionDataset`` / ``AudioPortionDatasetInference``
(EncoderDataset.py:18-152, :154-232): ragged per-document embeddings -> zero-padded batch dict

    {'id', 'src_tokens' [B, Lmax, D] fp32, 'src_lengths' [B] int64, 'tgt_tokens' [B, Lmax] fp32 (pad -1, or 0 for CRF),
     'src_tokens2' or None, 'domain' or None}

with ``truncate=True`` meaning "pad/truncate to exactly truncate_value" (3600 for the Transformer, train_fit.py:104-106).
Host-side only; the PCA/UMAP projection of the reference (:49-69, buggy re-slicing) is out of scope.

The default collater is the reference's, byte for byte (fixtures g8 / g13).  It is also the slowest stage of a training step: a fresh
pageable fp32 ``torch.zeros`` batch filled document by document on one thread -- 117 MB per step at BASELINE configs[1], an order of
magnitude longer than the 2 ms the GPU needs for it.  ``AudioPortionDataset(..., pin_memory=True, wire_dtype='fp32' | 'bf16')`` makes the
dataset the fast producer instead (same dict, same values -- in bf16 the values ``src_tokens.to(torch.bfloat16)`` would have):

  * the embeddings are padded straight into a slot of a small ring of PINNED buffers by ``mts_collate_pad`` (include/mts.h: one pass,
    split over host threads), so ``prefetch.DevicePrefetcher`` -- or ``.to(device, non_blocking=True)`` -- starts the host-to-device copy
    from the collater's own output, no staging copy;
  * ``wire_dtype='bf16'``: the documents are converted ONCE, when the dataset is built, and held in bf16 on the host; the batch is bf16
    (half the PCIe bytes; the recurrent taggers round their input to bf16 first thing anyway -- bit-identical for them,
    tests/test_gpu_prefetch.py -- the transformer reads bf16 embeddings into its fp32 LayerNorm: an approximation, opt-in);
  * ``__getitems__`` (the batched fetch torch's DataLoader uses when a dataset offers it) hands the collater the INDICES of a batch instead
    of 64 per-document dicts: the batch is then built from per-document pointer tables with two native calls and ~50 us of Python -- the
    per-sample path costs ~0.8 ms of interpreter time per batch, under the GIL the training loop itself needs (bench.py --h2d collater);
  * a slot is reused ``pin_slots`` batches later; whoever copies a batch out of it asynchronously registers the copy's event with
    ``release_after(batch, event)`` (DevicePrefetcher does) and the collater waits for that event before it overwrites the slot.
"""
import ctypes

import numpy as np
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


class _PinnedRing:
    """``slots`` reusable host buffers per field (pinned when a GPU runtime is there to pin them), handed out round-robin."""

    def __init__(self, slots):
        self.slots = max(2, int(slots))
        self.bufs = [dict() for _ in range(self.slots)]
        self.busy = [None] * self.slots            # slot -> event of the asynchronous copy that last read it (release_after)
        self.turn = 0
        self.pinned = torch.cuda.is_available()

    def next_slot(self):
        slot = self.turn % self.slots
        self.turn += 1
        ev = self.busy[slot]
        if ev is not None:
            ev.synchronize()                       # the copy out of this slot has finished: it may be overwritten
            self.busy[slot] = None
        return slot

    def get(self, slot, field, shape, dtype):
        n = 1
        for d in shape:
            n *= int(d)
        buf = self.bufs[slot].get(field)
        if buf is None or buf.numel() < n or buf.dtype != dtype:
            buf = torch.empty(max(n, 1), dtype=dtype)
            if self.pinned:
                buf = buf.pin_memory()
            self.bufs[slot][field] = buf
        out = buf[:n].view(*shape)
        out._mts_ring = (self, slot)               # lets release_after() find the slot
        return out


def release_after(batch, event):
    """``event``: recorded behind an asynchronous copy out of ``batch``'s ring-slot tensors (a torch.cuda.Event); the dataset's collater
    waits for it before that slot is written again.  No-op for batches that do not live in a ring."""
    for v in batch.values():
        ring = getattr(v, '_mts_ring', None) if isinstance(v, torch.Tensor) else None
        if ring is not None:
            ring[0].busy[ring[1]] = event


_DT = {torch.float32: 0, torch.bfloat16: 1}          # MTS_F32, MTS_BF16


class _IndexBatch(list):
    """what ``AudioPortionDataset.__getitems__`` returns in ring mode: the document indices of one batch (the collater's fast path)"""


def _pad_native(ptrs, rows, D, src_dt, truncate, tv, ring, slot, field, dtype, threads, pad=0.0):
    """ptrs / rows: int64 numpy arrays (data pointer and row count of every document of the batch) -> padded batch in ring slot ``slot``"""
    from ._lib import check, lib
    B = int(ptrs.shape[0])
    max_length = int(tv if truncate else rows.max())
    out = ring.get(slot, field, (B, max_length) if D == 0 else (B, max_length, D), dtype)
    check(lib.mts_collate_pad(_DT[src_dt], _DT[dtype], B, max_length, max(D, 1), ptrs.ctypes.data, rows.ctypes.data, out.data_ptr(), float(pad),
                              int(threads)))
    return out


def _merge_native(values, truncate, tv, ring, slot, field, dtype, threads, pad=0.0):
    """the `merge` of the reference in one native pass: [B, Lmax, D] (or [B, Lmax] for 1-d values) in ``dtype`` inside ring slot ``slot``"""
    from ._lib import check, lib
    vals = [v if (v.dtype in _DT and v.is_contiguous()) else v.to(torch.float32).contiguous() for v in values]
    src_dt = vals[0].dtype
    if any(v.dtype != src_dt for v in vals):
        vals = [v.to(torch.float32) for v in vals]
        src_dt = torch.float32
    flat = vals[0].dim() == 1
    B, D = len(vals), 1 if flat else int(vals[0].shape[1])
    max_length = tv if truncate else max(v.size(0) for v in vals)
    out = ring.get(slot, field, (B, max_length) if flat else (B, max_length, D), dtype)
    ptrs = (ctypes.c_void_p * B)(*[v.data_ptr() for v in vals])
    rows = (ctypes.c_int64 * B)(*[int(v.size(0)) for v in vals])
    check(lib.mts_collate_pad(_DT[src_dt], _DT[dtype], B, int(max_length), D, ptrs, rows, out.data_ptr(), float(pad), int(threads)))
    return out


class AudioPortionDataset(Dataset):
    def __init__(self, lines, tag_to_ix, encoder='x-vectors', CRF=True, truncate=True, truncate_value=100, umap_project=False,
                 umap_project_value=100, umap_class=None, second_input=None, domain_adapt=False, pin_memory=False, wire_dtype='fp32',
                 pin_slots=4, collate_threads=8):
        if umap_project:
            raise NotImplementedError('PCA/UMAP projection (EncoderDataset.py:49-69) is outside the hot path')
        if wire_dtype not in ('fp32', 'bf16'):
            raise ValueError("wire_dtype must be 'fp32' or 'bf16'")
        self.minus = 0 if CRF else 1                                   # EncoderDataset.py:23
        self.embeddings = [line[0] for line in lines]
        self.tgt_dataset = [line[1] for line in lines]
        self.embeddings2 = [line[0] for line in second_input] if second_input is not None else []
        # the fast producer (module docstring); off by default: the reference's collater, byte for byte
        self.wire = torch.bfloat16 if wire_dtype == 'bf16' else torch.float32
        self._ring = _PinnedRing(pin_slots) if (pin_memory or wire_dtype == 'bf16') else None
        self.collate_threads = int(collate_threads)
        self._tgt_cache = {}
        self._tab = None
        if wire_dtype == 'bf16':                                       # converted once, held in bf16 on the host
            self.embeddings = [torch.as_tensor(e).to(torch.bfloat16).contiguous() for e in self.embeddings]
            self.embeddings2 = [torch.as_tensor(e).to(torch.bfloat16).contiguous() for e in self.embeddings2]
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

    def __getstate__(self):
        """a DataLoader worker gets a COPY of the dataset: pointer tables and ring buffers are rebuilt there, never shipped (their addresses
        belong to this process)"""
        d = dict(self.__dict__)
        d['_tab'] = None
        d['_tgt_cache'] = {}
        if d.get('_ring') is not None:
            d['_ring'] = _PinnedRing(d['_ring'].slots)
        return d

    def __getitems__(self, indices):
        """batched fetch (torch's DataLoader calls this when it exists): ring mode -> the indices themselves, for the collater's fast path"""
        if self._ring is None:
            return [self[i] for i in indices]
        return _IndexBatch(int(i) for i in indices)

    def _tables(self):
        """per-document pointer / row-count tables of the fast path (built once; the tensors they point into are kept alive here)"""
        if self._tab is None:
            def table(items):
                ts = [t if (isinstance(t, torch.Tensor) and t.dtype in _DT and t.is_contiguous()) else torch.as_tensor(t).to(torch.float32).contiguous()
                      for t in items]
                if len({t.dtype for t in ts}) > 1:
                    ts = [t.to(torch.float32) for t in ts]
                return (ts, np.array([t.data_ptr() for t in ts], dtype=np.int64), np.array([t.shape[0] for t in ts], dtype=np.int64))
            self._tab = {'e': table(self.embeddings), 'e2': table(self.embeddings2) if self.embeddings2 else None,
                         't': table([torch.as_tensor(t, dtype=torch.float32) for t in self.tgt_dataset])}
        return self._tab

    def _collate_indices(self, idx):
        tab, ring = self._tables(), self._ring
        slot = ring.next_slot()
        ii = np.asarray(idx, dtype=np.int64)
        out = {}
        for key, field in (('e', 'src_tokens'), ('e2', 'src_tokens2')):
            if tab[key] is None:
                out[field] = None
                continue
            ts, ptrs, rows = tab[key]
            out[field] = _pad_native(ptrs[ii], rows[ii], int(ts[0].shape[1]), ts[0].dtype, self.truncate, self.tv, ring, slot, field, self.wire,
                                     self.collate_threads)
        ts, ptrs, rows = tab['t']
        tgt = _pad_native(ptrs[ii], rows[ii], 0, torch.float32, self.truncate, self.tv, ring, slot, 'tgt_tokens', torch.float32, 1, pad=-float(self.minus))
        n = tab['e'][2][ii]
        lengths = torch.from_numpy(np.minimum(n, self.tv) if self.truncate else n.copy())
        return {'id': torch.from_numpy(ii.copy()), 'src_tokens': out['src_tokens'], 'src_lengths': lengths, 'tgt_tokens': tgt,
                'src_tokens2': out['src_tokens2'], 'domain': [self.domain[i] for i in idx] if self.da else None}

    def __len__(self):
        return len(self.embeddings)

    def collater(self, samples):
        """Merge a list of samples to form a mini-batch (EncoderDataset.py:91-152)."""
        if len(samples) == 0:
            return {}
        if isinstance(samples, _IndexBatch):
            return self._collate_indices(samples)
        if self._ring is not None and torch.as_tensor(samples[0]['embeddings']).dim() == 2:
            ring = self._ring
            slot = ring.next_slot()
            src_tokens = _merge_native([torch.as_tensor(s['embeddings']) for s in samples], self.truncate, self.tv, ring, slot, 'src_tokens',
                                       self.wire, self.collate_threads)
            src_tokens2 = _merge_native([torch.as_tensor(s['embeddings2']) for s in samples], self.truncate, self.tv, ring, slot,
                                        'src_tokens2', self.wire, self.collate_threads) if self.embeddings2 else None
            # targets: cached as fp32 tensors the first time a document is collated (a Python list of 256 floats costs ~20 us to convert,
            # every step, under the GIL the training loop needs); pad -1 (0 for the CRF head), EncoderDataset.py:23
            tg = []
            for s_ in samples:
                i = int(s_['id'])
                t = self._tgt_cache.get(i)
                if t is None or s_['target'] is not self.tgt_dataset[i]:
                    t = torch.as_tensor(s_['target']).to(torch.float32).contiguous()
                    if s_['target'] is self.tgt_dataset[i]:
                        self._tgt_cache[i] = t
                tg.append(t)
            tgt_tokens = _merge_native(tg, self.truncate, self.tv, ring, slot, 'tgt_tokens', torch.float32, 1, pad=-float(self.minus))
        else:
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
