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
import os

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


def local_loss_count(model, batch):
    """What the tagger's loss averages over on THIS rank: valid sentences for BCE / focal / CE (models/CRF.py:348-352: mean over
    the concatenated un-padded rows; CE: rows whose target is not the pad -1, :298,354), documents for the CRF NLL (:145)."""
    lengths = batch['src_lengths']
    Lq = batch['src_tokens'].shape[1]
    if hasattr(model, 'num_tags'):
        return int(lengths.numel())
    return int(lengths.clamp(max=Lq).sum())


class NativeTrainer:
    def __init__(self, model, lr=1e-3, optimizer='Adam', process_group=None, token_weighted=False, grad_exchange_dtype='fp32',
                 always_hook=False, exchange_schedule=None):
        """model: a tagger from taggers.py / rnn_taggers.py (or a TextSegmenter, whose .model is used).

        token_weighted: the reference's loss is a mean over the LOCAL batch's valid sentences (models/CRF.py:352), so plain data
        parallelism (DDP included) averages shard means; with ragged shards that is not the global mean.  True weights each
        rank's gradient by n_r / sum_r n_r (one scalar all-reduce per step), which makes the exchanged gradient EXACTLY the
        single-process gradient of the whole batch.  Identical to False when every shard holds the same number of sentences.

        grad_exchange_dtype: 'fp32' | 'bf16'.  bf16 halves the bytes on the xGMI ring (84.5 -> 42.3 MB per step for the 1-layer
        band encoder); each rank's contribution is rounded to bf16 (relative 2^-9) and summed in bf16 by the collective, so the
        exchanged sum is within 2^-8 * sum_r |g_r| of the fp32 one (tests/test_distributed_cpu.py).

        exchange_schedule: 'allreduce' (default) | 'rs_ag' (env MTS_DP_SCHEDULE when None).  'rs_ag' exchanges every span as a
        reduce-scatter into 1/world shards followed by an all-gather of the reduced shards (the remainder of a span that does not
        divide by world goes through a small all-reduce): SURVEY.md 8(e)'s schedule -- on a fully connected xGMI node each rank then
        talks to its 7 peers at once instead of passing whole buffers round one ring.  Same sum on every rank; for world = 2 the same
        bits as 'allreduce'.  Behind a switch until an 8-GPU node has timed the two.

        always_hook: take the overlapped exchange path (gradient-ready hooks -> asynchronous all-reduce per span) even in a
        process group of ONE rank.  Measurement aid: the N > 1 step path -- hook order, per-projection release, collective
        launches, stream waits -- on a single GPU (bench.py MTS_BENCH_SINGLE_RANK_DP=1); needs an initialised process group."""
        # a TextSegmenter wraps the tagger in .model; a bare tagger may itself own a parameter container called "model"
        self.model = model.model if hasattr(model, 'training_step') else model
        self.lr, self.kind = float(lr), optimizer
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (dist.is_available() and dist.is_initialized()) else 1
        self.step_count = 0
        self._m = self._v = None
        self._comm_stream = None
        self.token_weighted = bool(token_weighted)
        self.always_hook = bool(always_hook) and dist.is_available() and dist.is_initialized()
        if grad_exchange_dtype not in ('fp32', 'bf16'):
            raise ValueError("grad_exchange_dtype must be 'fp32' or 'bf16'")
        self.exchange_bf16 = grad_exchange_dtype == 'bf16'
        self._xbuf = None
        self.last_global_count = None
        sched = exchange_schedule if exchange_schedule is not None else os.environ.get('MTS_DP_SCHEDULE', 'allreduce')
        if sched not in ('allreduce', 'rs_ag'):
            raise ValueError("exchange_schedule must be 'allreduce' or 'rs_ag'")
        self.exchange_schedule = sched
        self._inflight_shards = []

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
            return [(0, off + (self._last_L + 2) * D), (end, n)]      # (what sits in front of the table, and its rows a batch of this length touches)
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

    def exchange_weight(self, batch):
        """Factor this rank's loss gradient is multiplied by BEFORE the SUM all-reduce (the optimizer then applies 1/world):
        1 for mean-of-shard-means (the reference's / DDP's behaviour), world * n_r / sum_r n_r with token_weighted."""
        if not self.token_weighted or self.world == 1:
            return 1.0
        n_local = local_loss_count(self.model, batch)
        dev = self.model.flat.device
        cnt = torch.tensor([float(n_local)], dtype=torch.float64, device=dev if dev.type == 'cuda' else 'cpu')
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM, group=self.pg)
        total = float(cnt.item())                     # one scalar per step; lengths are host data, nothing else waits on this
        self.last_global_count = total
        return self.world * n_local / total if total > 0 else 1.0

    def _sum_in_place(self, buf, async_op):
        """SUM of `buf` over the ranks, in place, by the configured schedule -> list of work handles (async_op) or []."""
        world = self.world
        if self.exchange_schedule != 'rs_ag' or (world == 1 and not self.always_hook) or buf.numel() < world:
            h = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=async_op)
            return [h] if async_op else []
        n = buf.numel()
        per = n // world
        body = per * world
        # one shard buffer per in-flight span: the all-gather reads it after this call has returned
        shard = torch.empty(per, dtype=buf.dtype, device=buf.device)
        hs = [dist.reduce_scatter_tensor(shard, buf[:body], op=dist.ReduceOp.SUM, group=self.pg, async_op=async_op)]
        if async_op and dist.get_backend(self.pg) != 'nccl':
            hs[0].wait()          # gloo runs asynchronous work on a thread pool: the gather below must not overtake the scatter.  (RCCL
                                  # orders the collectives of one communicator on its stream -- no host or stream wait needed there.)
        hs.append(dist.all_gather_into_tensor(buf[:body], shard, group=self.pg, async_op=async_op))
        if body < n:
            hs.append(dist.all_reduce(buf[body:], op=dist.ReduceOp.SUM, group=self.pg, async_op=async_op))
        if async_op and shard.device.type == 'cuda':
            self._inflight_shards.append(shard)       # keep the caching allocator from recycling it under the collectives
        return hs if async_op else []

    def _exchange(self, a, b, async_op):
        """SUM of flat-gradient span [a, b) over the ranks in the exchange dtype -> (work handles, span to copy back | None)."""
        g = self.model.grad_flat()
        if not self.exchange_bf16:
            return self._sum_in_place(g[a:b], async_op), None
        if self._xbuf is None or self._xbuf.numel() < g.numel() or self._xbuf.device != g.device:
            self._xbuf = torch.empty(g.numel(), dtype=torch.bfloat16, device=g.device)
        buf = self._xbuf[a:b]
        buf.copy_(g[a:b])                             # fp32 -> bf16 (round to nearest even), behind the kernels that wrote the span
        return self._sum_in_place(buf, async_op), (a, b)

    def _finish(self, handles, span):
        for handle in handles or ():
            handle.wait()                             # stream-level wait: what follows is ordered behind the collective
        if span is not None:
            a, b = span
            self.model.grad_flat()[a:b].copy_(self._xbuf[a:b])

    def allreduce_grads(self):
        """Blocking exchange of every span (models without gradient-ready hooks)."""
        if self.world == 1:
            return
        for a, b in self._reduce_spans():
            if b > a:
                _, span = self._exchange(a, b, False)
                self._finish(None, span)

    def _on_grads_ready(self, a, b):
        """Called from inside the backward as soon as flat-gradient span [a, b) is final: the all-reduce is enqueued
        asynchronously (RCCL runs it on its own stream behind the kernels issued so far), so it overlaps the rest of
        the backward; few large messages because ring collectives over xGMI are per-link-bound."""
        self._pending.append(self._exchange(a, b, True))

    def step(self, batch):
        """One optimizer step on this rank's shard; returns the local loss (0-d tensor)."""
        dev = self.model.flat.device
        if dev.type == 'cuda' and torch.cuda.current_device() != dev.index:
            with torch.cuda.device(dev):             # kernels go to the CURRENT device's current stream (taggers._on_model_device)
                return self._step(batch)
        return self._step(batch)

    def _step(self, batch):
        m = self.model
        x, lengths, tags = batch['src_tokens'], batch['src_lengths'], batch['tgt_tokens']
        if self.world > 1:
            # every rank must issue collectives of the same sizes: the position-table span depends on the collated length
            self._check_same_length(x.shape[1])
        self._last_L = x.shape[1]
        m.loss_grad_scale = self.exchange_weight(batch)
        overlapped = (self.world > 1 or self.always_hook) and getattr(m, 'grad_hooks_cover_all', False)
        self._pending = []
        self._inflight_shards = []
        m._grad_hook = self._on_grads_ready if overlapped else None
        if batch.get('src_tokens2') is not None and hasattr(m, '_rnn2'):
            loss, _ = m.loss_and_grad(x, batch['src_tokens2'], lengths, tags, True)
        else:
            loss, _ = m.loss_and_grad(x, lengths, tags, True)
        if overlapped:
            for h, span in self._pending:
                self._finish(h, span)         # stream-level wait: the optimizer kernel is ordered behind the collectives
            self._pending = []
            self._inflight_shards = []
        else:
            self.allreduce_grads()
        self.apply_optimizer()
        return loss

    def _check_same_length(self, Lq):
        """Shards of one global batch are collated together (same padded length).  Ranks that collate separately may differ:
        catch that once per distinct length instead of hanging in mismatched collectives."""
        seen = getattr(self, '_len_checked', None)
        if seen == Lq:
            return
        dev = self.model.flat.device
        t = torch.tensor([float(Lq), -float(Lq)], dtype=torch.float64, device=dev if dev.type == 'cuda' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.pg)
        if int(t[0].item()) != Lq or int(-t[1].item()) != Lq:
            raise ValueError(f'data-parallel ranks collated to different lengths (this rank {Lq}, max {int(t[0].item())}, '
                             f'min {int(-t[1].item())}): shard one collated batch with shard_batch() or pad to a common length')
        self._len_checked = Lq

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
