"""Keep the training step fed: host batches -> device, one or more batches AHEAD of the step, on a side HIP stream.

The reference moves every batch synchronously in the step that consumes it (Lightning's default transfer of the collated dict,
train_fit.py:145-154 with ``num_workers=0``; EncoderDataset.py:103-109 builds a fresh pageable fp32 ``src_tokens``).  At
BASELINE configs[1] that is 64 x 256 x 1792 fp32 = 117 MB per step: ~1.9 ms of PCIe at ~63 GB/s against a 2.05 ms step -- a
drop-in user who hands the tagger host batches would get half the resident-input throughput.

``DevicePrefetcher(batches, device)`` iterates over the same batch dicts with the tensors already on the device:

  * a producer thread pulls collated batches from ``batches`` (so the Python collater overlaps the GPU too), stages the
    floating-point fields in PINNED host buffers (a ring of ``depth + 1`` slots, grown on demand; a field that is already pinned is
    sent as it is) and enqueues the host-to-device copies on a side stream, ``depth`` batches ahead of the consumer;
  * the consumer's current stream waits on the copy's event (no host synchronisation), and the device tensors are tied to the
    consuming stream (``record_stream``) so the caching allocator cannot hand them out again while the step still reads them;
  * ``wire_dtype='bf16'`` converts ``src_tokens`` / ``src_tokens2`` to bf16 on the HOST while staging (half the PCIe bytes; the
    recurrent taggers round their input to bf16 first thing anyway, so for them the result is bit-identical to fp32 transport --
    tests/test_gpu_prefetch.py; the transformer reads fp32 embeddings into an fp32 LayerNorm, so there it is an approximation and
    stays opt-in).  ``src_lengths``, ``id`` and ``domain`` stay on the host, where the taggers read them.

Nothing here touches the arithmetic of the path; bench.py --h2d reports the step time with this loader in the loop as a separate
line, never mixed into ``value`` (which is quoted with inputs resident in HBM).
"""
import queue
import threading

import torch

from .encoder_dataset import release_after

_FLOAT_FIELDS = ('src_tokens', 'src_tokens2', 'tgt_tokens')
_WIRE_FIELDS = ('src_tokens', 'src_tokens2')


class _Stop:
    pass


class DevicePrefetcher:
    def __init__(self, batches, device, depth=2, wire_dtype='fp32'):
        if wire_dtype not in ('fp32', 'bf16'):
            raise ValueError("wire_dtype must be 'fp32' or 'bf16'")
        if depth < 1:
            raise ValueError('depth must be >= 1')
        self.batches, self.device, self.depth = batches, torch.device(device), int(depth)
        if self.device.type == 'cuda' and self.device.index is None:          # 'cuda' -> the current device, with its index
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.wire = torch.bfloat16 if wire_dtype == 'bf16' else None
        self._cuda = self.device.type == 'cuda'
        self._stream = torch.cuda.Stream(device=self.device) if self._cuda else None
        self._ring = [dict() for _ in range(self.depth + 1)]      # slot -> {field: pinned staging tensor}
        self._slot_done = [None] * (self.depth + 1)               # slot -> event of the last copy that read its staging buffers
        self.bytes_sent = 0

    # ---- producer side -------------------------------------------------------------------------------------------------
    def _stage(self, slot, field, t):
        """-> a pinned host tensor holding t (converted to the wire dtype where that applies)"""
        want = self.wire if (self.wire is not None and field in _WIRE_FIELDS and t.is_floating_point()) else t.dtype
        if not self._cuda:
            return t.to(want)
        if t.is_pinned() and t.dtype == want and t.is_contiguous():
            return t
        buf = self._ring[slot].get(field)
        if buf is None or buf.numel() < t.numel() or buf.dtype != want:
            buf = self._ring[slot][field] = torch.empty(t.numel(), dtype=want).pin_memory()
        out = buf[:t.numel()].view(t.shape)
        out.copy_(t)                                              # host memcpy (+ fp32 -> bf16 conversion): releases the GIL
        return out

    def _send(self, slot, batch):
        if not isinstance(batch, dict):
            raise TypeError('DevicePrefetcher expects the collater\'s batch dicts')
        if self._cuda and self._slot_done[slot] is not None:
            self._slot_done[slot].synchronize()                   # the copy that last read this slot's staging buffers has finished
        out, ev = dict(batch), None
        if self._cuda:
            with torch.cuda.stream(self._stream):
                for f in _FLOAT_FIELDS:
                    t = batch.get(f)
                    if isinstance(t, torch.Tensor) and t.device.type == 'cpu':
                        h = self._stage(slot, f, t)
                        out[f] = h.to(self.device, non_blocking=True)
                        self.bytes_sent += h.numel() * h.element_size()
                ev = torch.cuda.Event()
                ev.record(self._stream)
            self._slot_done[slot] = ev
            release_after(batch, ev)                                  # a collater that pads into its own pinned ring (encoder_dataset.py) reuses the slot after this
        else:
            for f in _FLOAT_FIELDS:
                t = batch.get(f)
                if isinstance(t, torch.Tensor):
                    out[f] = self._stage(slot, f, t)
        return out, ev

    def _produce(self, q, stop):
        try:
            if self._cuda:
                torch.cuda.set_device(self.device)
            for i, batch in enumerate(self.batches):
                if stop.is_set():
                    return
                item = self._send(i % (self.depth + 1), batch)
                while not stop.is_set():
                    try:
                        q.put(item, timeout=0.1)
                        break
                    except queue.Full:
                        continue
            q.put(_Stop)
        except BaseException as e:  # noqa: BLE001  (handed to the consumer, which re-raises it)
            q.put(e)

    # ---- consumer side -------------------------------------------------------------------------------------------------
    def __iter__(self):
        # at most `depth` batches are in flight beyond the one being consumed: queue of depth - 1 + the one the producer holds
        q = queue.Queue(maxsize=max(1, self.depth - 1))
        stop = threading.Event()
        th = threading.Thread(target=self._produce, args=(q, stop), daemon=True, name='mts-prefetch')
        th.start()
        try:
            while True:
                item = q.get()
                if item is _Stop:
                    return
                if isinstance(item, BaseException):
                    raise item
                batch, ev = item
                if ev is not None:
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(ev)
                    for f in _FLOAT_FIELDS:
                        t = batch.get(f)
                        if isinstance(t, torch.Tensor) and t.device.type == 'cuda':
                            t.record_stream(cur)
                yield batch
        finally:
            stop.set()
            while th.is_alive():                                  # unblock a producer waiting on a full queue, then let it end
                try:
                    q.get_nowait()
                except queue.Empty:
                    pass
                th.join(timeout=0.05)
