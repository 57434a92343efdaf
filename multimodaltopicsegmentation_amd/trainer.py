"""Native training step: forward + backward + (RCCL gradient all-reduce) + fused optimizer, no autograd.

This is the MI355X-first replacement for what Lightning does around ``TextSegmenter.training_step``
(train_fit.py:335 -> loss.backward(); optimizer.step(); with ``gpus>1`` DDP's bucketed NCCL all-reduce):

  * one process per GPU, documents sharded across ranks (rank r takes documents r::world of the global batch);
  * gradients live in ONE flat fp32 buffer (flat.py), so the data-parallel exchange is a couple of large
    ``all_reduce`` calls over RCCL/xGMI (per-link-bound ring: few large messages, not one per tensor), issued on
    a side stream as soon as the span they cover is final so they overlap the rest of the backward;
  * Adam(eps 1e-7) / SGD(momentum .9, wd 1e-4) (lightning_model.py:759-765) is one streaming kernel over the
    flat buffer that also emits the bf16 weight mirror for the next step's GEMMs.
"""
import torch
import torch.distributed as dist

from . import ops


def shard_batch(batch, rank, world):
    """Document-sharded view of a collated batch (EncoderDataset.py batch dict): rank r keeps documents r::world."""
    if world == 1:
        return batch
    out = {}
    for k, v in batch.items():
        if isinstance(v, torch.Tensor) and v.dim() >= 1:
            out[k] = v[rank::world]
        elif isinstance(v, list):
            out[k] = v[rank::world]
        else:
            out[k] = v
    return out


class NativeTrainer:
    def __init__(self, model, lr=1e-3, optimizer='Adam', process_group=None, token_weighted=False):
        """model: a tagger from taggers.py / rnn_taggers.py (or a TextSegmenter, whose .model is used)."""
        # a TextSegmenter wraps the tagger in .model; a bare tagger may itself own a parameter container called "model"
        self.model = model.model if hasattr(model, 'training_step') else model
        self.lr, self.kind = float(lr), optimizer
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.step_count = 0
        self._m = self._v = None
        self._comm_stream = None
        self.token_weighted = token_weighted

    def _state(self):
        flat = self.model.flat
        if self._m is None or self._m.device != flat.device:
            self._m = torch.zeros_like(flat)
            self._v = torch.zeros_like(flat)
        return self._m, self._v

    def _reduce_spans(self):
        """Spans of the flat gradient that can be non-zero (the position table only in the rows a batch touched)."""
        m = self.model
        n = m.flat.numel()
        pos = 'model.model.embeddings.position_embeddings.weight'
        if pos in m._layout.entries and getattr(self, '_last_L', None) is not None:
            off, shape = m._layout.entries[pos]
            D = shape[1]
            end = off + shape[0] * D
            return [(off + 2 * D, off + (self._last_L + 2) * D), (end, n)] if off == 0 else [(0, n)]
        return [(0, n)]

    def _adam_spans(self):
        """Adam without weight decay leaves a parameter whose gradient and both moments are exactly 0 untouched, so the
        rows of the position table above the longest batch seen so far (never read, gradient always 0) are skipped:
        bitwise the same result as stepping the whole buffer, a third less optimizer traffic at max_position 4096."""
        m = self.model
        n = m.flat.numel()
        pos = 'model.model.embeddings.position_embeddings.weight'
        if pos not in m._layout.entries or getattr(self, '_last_L', None) is None:
            return [(0, n)]
        off, shape = m._layout.entries[pos]
        D = shape[1]
        self._pos_rows = max(getattr(self, '_pos_rows', 0), min(self._last_L + 2, shape[0]))
        lo, hi = off + self._pos_rows * D, off + shape[0] * D
        return [(a, b) for a, b in ((0, lo), (hi, n)) if b > a]

    def allreduce_grads(self):
        """Blocking exchange of every span (models without gradient-ready hooks, e.g. the recurrent taggers)."""
        if self.world == 1:
            return
        g = self.model.grad_flat()
        for a, b in self._reduce_spans():
            if b > a:
                dist.all_reduce(g[a:b], op=dist.ReduceOp.SUM, group=self.pg)

    def _on_grads_ready(self, a, b):
        """Called from inside the backward as soon as flat-gradient span [a, b) is final: the all-reduce is enqueued
        asynchronously (RCCL runs it on its own stream behind the kernels issued so far), so it overlaps the rest of
        the backward; few large messages because ring collectives over xGMI are per-link-bound."""
        g = self.model.grad_flat()
        self._pending.append(dist.all_reduce(g[a:b], op=dist.ReduceOp.SUM, group=self.pg, async_op=True))

    def step(self, batch):
        """One optimizer step on this rank's shard; returns the local loss (0-d tensor)."""
        m = self.model
        x, lengths, tags = batch['src_tokens'], batch['src_lengths'], batch['tgt_tokens']
        self._last_L = x.shape[1]
        overlapped = self.world > 1 and getattr(m, 'grad_hooks_cover_all', False)
        self._pending = []
        m._grad_hook = self._on_grads_ready if overlapped else None
        if batch.get('src_tokens2') is not None and hasattr(m, '_rnn2'):
            loss, _ = m.loss_and_grad(x, batch['src_tokens2'], lengths, tags, True)
        else:
            loss, _ = m.loss_and_grad(x, lengths, tags, True)
        if overlapped:
            for h in self._pending:
                h.wait()                      # stream-level wait: the optimizer kernel is ordered behind the collectives
            self._pending = []
        else:
            self.allreduce_grads()
        self.apply_optimizer()
        return loss

    def apply_optimizer(self):
        m = self.model
        self.step_count += 1
        mirror = m._wcopy if (m.compute_dtype == torch.bfloat16 and m._wcopy is not None) else None
        gscale = 1.0 / self.world
        if self.kind == 'SGD':
            buf, _ = self._state()
            ops.sgd_step(m.flat, m.grad_flat(), buf, self.lr, 0.9, 1e-4, self.step_count == 1, gscale, mirror)
        else:
            mm, vv = self._state()
            g = m.grad_flat()
            for a, b in self._adam_spans():
                ops.adam_step(m.flat[a:b], g[a:b], mm[a:b], vv[a:b], self.lr, 0.9, 0.999, 1e-7, self.step_count, gscale,
                              mirror[a:b] if mirror is not None else None)
        # the kernel wrote `flat` through a raw pointer, which does not bump torch's version counter: keep the
        # bf16-mirror cache of the model coherent by hand
        if mirror is not None:
            m.mark_weights_synced()
        else:
            m._wcopy_version = None
