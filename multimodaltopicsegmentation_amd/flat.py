"""Flat fp32 parameter / gradient storage.

All parameters of a tagger live in ONE contiguous fp32 device buffer (and their gradients in another), laid out
for MI355X: fused-GEMM operands that the reference keeps as separate tensors (query/key/value weights, the two
directions of an LSTM layer) are placed back to back so one [3D, D] / [8H, D] GEMM reads them in place, the
optimizer is a single streaming kernel over the buffer, and data-parallel training all-reduces the gradient
buffer with one RCCL call (or a few large buckets) instead of one call per tensor.  Each tensor is still exposed
as an ``nn.Parameter`` *view* under the reference's own ``state_dict`` key, so torch optimizers, ``state_dict``
and reference checkpoints keep working.
"""
from collections import OrderedDict

import torch
import torch.nn as nn

ALIGN = 64  # elements (256 B): every group starts on a 256-byte boundary -> 16-byte vector access everywhere


class FlatLayout:
    def __init__(self, groups):
        """groups: list of lists of (name, shape); tensors of a group are contiguous, groups are ALIGN-aligned."""
        self.entries = OrderedDict()   # name -> (offset, shape)
        self.group_spans = []          # (offset, numel) per group
        off = 0
        for grp in groups:
            off = (off + ALIGN - 1) // ALIGN * ALIGN
            start = off
            for name, shape in grp:
                n = 1
                for s in shape:
                    n *= int(s)
                self.entries[name] = (off, tuple(int(s) for s in shape))
                off += n
            self.group_spans.append((start, off - start))
        self.numel = (off + ALIGN - 1) // ALIGN * ALIGN

    def view(self, flat, name):
        off, shape = self.entries[name]
        n = 1
        for s in shape:
            n *= s
        return flat[off:off + n].view(shape)

    def span(self, first, last):
        """[offset, offset+numel) covering tensors first..last (must be adjacent in the layout)."""
        o0, _ = self.entries[first]
        o1, s1 = self.entries[last]
        n1 = 1
        for s in s1:
            n1 *= s
        return o0, o1 + n1 - o0


def register_nested(root, dotted, param):
    """Register ``param`` under a dotted reference state_dict key, creating container modules on the way."""
    mod = root
    parts = dotted.split('.')
    for p in parts[:-1]:
        if p not in mod._modules:
            mod.add_module(p, nn.Module())
        mod = mod._modules[p]
    mod.register_parameter(parts[-1], param)


class FlatModule(nn.Module):
    """nn.Module whose parameters are views into one flat buffer; survives .to()/.cuda() by re-flattening."""

    def _init_flat(self, layout, init_values):
        self._layout = layout
        flat = torch.zeros(layout.numel, dtype=torch.float32)
        for name, val in init_values.items():
            layout.view(flat, name).copy_(val)
        self._flat = flat
        self._grad_flat = None
        self._flat_params = OrderedDict()
        for name in layout.entries:
            p = nn.Parameter(layout.view(flat, name))
            self._flat_params[name] = p
            register_nested(self, name, p)

    def _apply(self, fn, recurse=True):
        out = super()._apply(fn, recurse)
        self._reflatten()
        return out

    def _reflatten(self):
        """Gather the (possibly moved) parameters back into one flat buffer on their current device."""
        params = self._flat_params
        first = next(iter(params.values()))
        dev = first.device
        flat = torch.empty(self._layout.numel, dtype=torch.float32, device=dev)
        flat.zero_()
        with torch.no_grad():
            for name, p in params.items():
                v = self._layout.view(flat, name)
                v.copy_(p.data.to(torch.float32))
                p.data = v
                p.grad = None
        self._flat = flat
        self._grad_flat = None
        self._on_reflatten()

    def _on_reflatten(self):
        pass

    @property
    def flat(self):
        return self._flat

    def grad_flat(self):
        if self._grad_flat is None or self._grad_flat.device != self._flat.device:
            self._grad_flat = torch.zeros_like(self._flat)
        return self._grad_flat

    def grad_views(self):
        g = self.grad_flat()
        return OrderedDict((name, self._layout.view(g, name)) for name in self._layout.entries)

    def attach_grads(self):
        """Point every parameter's .grad at its slice of the flat gradient buffer (native training path)."""
        g = self.grad_flat()
        for name, p in self._flat_params.items():
            p.grad = self._layout.view(g, name)
