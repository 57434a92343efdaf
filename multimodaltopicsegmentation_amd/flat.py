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


def round_up(n, m):
    return (int(n) + m - 1) // m * m


# ---- padded storage ---------------------------------------------------------------------------------
# The kernels want feature dimensions that are multiples of 8 (16-byte bf16 vectors, MFMA k-steps); the reference's
# DEFAULT hidden size is 25 (train_fit.py:689).  Such a dimension is stored padded: a pad spec is a list of
# (axis, blocks, n, n_pad) -- along `axis` the logical tensor is `blocks` blocks of n entries (e.g. the 4 LSTM gates of H
# units), each stored as n_pad entries with zeros behind the n real ones.  Padded units are inert by construction (zero
# weights and biases -> zero activations and zero gradients, so Adam/SGD leave them at zero); parameters are the padded
# storage tensors, and state_dict()/load_state_dict() convert to/from the reference's logical shapes.
def pad_blocks(t, spec):
    for axis, blocks, n, n_pad in spec:
        shp = list(t.shape)
        assert shp[axis] == blocks * n, (shp, spec)
        t = t.reshape(shp[:axis] + [blocks, n] + shp[axis + 1:])
        pad_shape = list(t.shape)
        pad_shape[axis + 1] = n_pad - n
        t = torch.cat([t, t.new_zeros(pad_shape)], dim=axis + 1)
        t = t.reshape(shp[:axis] + [blocks * n_pad] + shp[axis + 1:])
    return t


def unpad_blocks(t, spec):
    for axis, blocks, n, n_pad in spec:
        shp = list(t.shape)
        assert shp[axis] == blocks * n_pad, (shp, spec)
        t = t.reshape(shp[:axis] + [blocks, n_pad] + shp[axis + 1:])
        t = t.narrow(axis + 1, 0, n)
        t = t.reshape(shp[:axis] + [blocks * n] + shp[axis + 1:])
    return t


class FlatLayout:
    def __init__(self, groups, pads=None):
        """groups: list of lists of (name, STORAGE shape); tensors of a group are contiguous, groups are ALIGN-aligned.
        pads: {name: pad spec} for tensors whose storage shape is a padded form of the reference's shape."""
        self.pads = dict(pads or {})
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
        for name, val in init_values.items():                      # init values come in the reference's (logical) shapes
            layout.view(flat, name).copy_(pad_blocks(val, layout.pads[name]) if name in layout.pads else val)
        self._flat = flat
        self._grad_flat = None
        self._flat_params = OrderedDict()
        for name in layout.entries:
            p = nn.Parameter(layout.view(flat, name))
            self._flat_params[name] = p
            register_nested(self, name, p)
        if layout.pads:
            self._register_state_dict_hook(FlatModule._unpad_state_dict)
            self._register_load_state_dict_pre_hook(self._pad_incoming_state_dict)

    @staticmethod
    def _unpad_state_dict(module, state_dict, prefix, local_metadata):
        for name, spec in module._layout.pads.items():
            k = prefix + name
            if k in state_dict:
                state_dict[k] = unpad_blocks(state_dict[k], spec).clone()
        return state_dict

    def _pad_incoming_state_dict(self, state_dict, prefix, *args):
        for name, spec in self._layout.pads.items():
            k = prefix + name
            if k in state_dict and tuple(state_dict[k].shape) != self._layout.entries[name][1]:
                state_dict[k] = pad_blocks(state_dict[k], spec)

    def logical_view(self, tensor_by_name, name):
        """The reference-shaped view of a (padded) parameter / gradient tensor."""
        t = tensor_by_name[name]
        return unpad_blocks(t, self._layout.pads[name]) if name in self._layout.pads else t

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
