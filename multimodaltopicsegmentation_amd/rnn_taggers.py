"""Recurrent taggers on the HIP kernels: BiLSTM, BiLSTMLateFusion, BiRnnCrf (see taggers.py for the contract).

Reference: models/NeuralArchitectures.py:23-145 (RNN), models/CRF.py:274-369 (BiLSTM), :371-479 (BiLSTMLateFusion),
:243-272 + :98-240 (BiRnnCrf + CRF; the reference wrapper unpacks a single tensor and crashes, SURVEY.md Q2 -- this
one composes RNN -> CRF the way the reference evidently intended).
"""
import math
import os

import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .flat import FlatLayout, round_up
from .taggers import _TaggerBase, _linear_init, _xavier_uniform, LOSS_KINDS

IMPOSSIBLE = -1e4   # models/CRF.py:95


def _rnn_groups(prefix, D, H, num_layers, gen):
    """Parameter groups + Keras-style init of ``RNN._reinitialize`` (NeuralArchitectures.py:58-79): xavier_uniform
    W_ih, orthogonal W_hh, zero biases with the forget-gate slice of bias_ih set to 1.  The two directions of a
    layer are adjacent so that one [8H, Din] GEMM projects both.

    Returns (groups of STORAGE shapes, init values in the reference's shapes, pad specs): hidden size and input width are
    stored rounded up to a multiple of 8 (the reference's default hidden size is 25); see flat.py "padded storage"."""
    Hp, Dp = round_up(H, 8), round_up(D, 8)
    groups, init, pads = [], {}, {}
    gate_rows = [(0, 4, H, Hp)] if Hp != H else []
    for k in range(num_layers):
        din, din_p = (D, Dp) if k == 0 else (2 * H, 2 * Hp)
        col = ([(1, 1, D, Dp)] if Dp != D else []) if k == 0 else ([(1, 2, H, Hp)] if Hp != H else [])
        g = []
        for sfx in ('', '_reverse'):
            n = f'{prefix}rnn.weight_ih_l{k}{sfx}'
            g.append((n, (4 * Hp, din_p)))
            init[n] = _xavier_uniform((4 * H, din), gen)
            if gate_rows or col:
                pads[n] = gate_rows + col
        groups.append(g)
        g = []
        for sfx in ('', '_reverse'):
            n = f'{prefix}rnn.weight_hh_l{k}{sfx}'
            g.append((n, (4 * Hp, Hp)))
            w = torch.empty(4 * H, H)
            nn.init.orthogonal_(w, generator=gen)
            init[n] = w
            if gate_rows:
                pads[n] = gate_rows + [(1, 1, H, Hp)]
        groups.append(g)
        for kind in ('bias_ih', 'bias_hh'):
            g = []
            for sfx in ('', '_reverse'):
                n = f'{prefix}rnn.{kind}_l{k}{sfx}'
                g.append((n, (4 * Hp,)))
                b = torch.zeros(4 * H)
                if kind == 'bias_ih':
                    b[H:2 * H] = 1.0
                init[n] = b
                if gate_rows:
                    pads[n] = gate_rows
            groups.append(g)
    return groups, init, pads


def _head_pad(H, nblocks):
    """Pad spec of a head weight [n_out, nblocks*H] reading nblocks concatenated hidden vectors."""
    Hp = round_up(H, 8)
    return [(1, nblocks, H, Hp)] if Hp != H else None


class _RnnStack:
    """Native forward/backward of one ``RNN`` (bidirectional multi-layer LSTM) living inside a flat-parameter model."""

    def __init__(self, owner, prefix, D, H, num_layers, tag):
        """D, H: the reference's sizes; the kernels run on the padded ones (multiples of 8, padded units inert)."""
        self.o, self.prefix, self.D, self.H, self.nl, self.tag = owner, prefix, round_up(D, 8), round_up(H, 8), num_layers, tag

    def _names(self, kind, k):
        return f'{self.prefix}rnn.{kind}_l{k}', f'{self.prefix}rnn.{kind}_l{k}_reverse'

    def forward(self, x2d, lengths_i32, B, Lq):
        """x2d: [B*Lq, D] in compute dtype.  Returns (out [N, 2H], saved state)."""
        o, H = self.o, self.H
        dt, dev = o.compute_dtype, x2d.device
        N = B * Lq
        wf, pf = o._weights(), o._flat
        saved = []
        h = x2d
        for k in range(self.nl):
            din = h.shape[1]
            w_ih = o._wspan(wf, *self._names('weight_ih', k), 8 * H, din)
            b_ih = o._wspan(pf, *self._names('bias_ih', k), 1, 8 * H).view(-1)
            b_hh = o._wspan(pf, *self._names('bias_hh', k), 1, 8 * H).view(-1)
            w_hh = o._wspan(pf, *self._names('weight_hh', k), 8 * H, H)
            xproj = o._ws.get(f'{self.tag}xproj{k}', N, 8 * H, dt, dev)
            ops.linear_fwd(h, w_ih, b_ih, xproj)
            out = o._ws.get(f'{self.tag}out{k}', N, 2 * H, dt, dev)
            gates = o._ws.get(f'{self.tag}gates{k}', N, 8 * H, dt, dev)
            cells = o._ws.get(f'{self.tag}cells{k}', N, 2 * H, torch.float32, dev)
            ops.lstm_fwd(xproj, w_hh, b_hh, lengths_i32, B, Lq, H, 2, out, gates, cells)
            saved.append(dict(hin=h, out=out, gates=gates, cells=cells))
            h = out
        return h, saved

    overlap_wgrad = True     # every layer's parameter gradients on side streams, off the chain of dependent launches: the recurrence occupies
                             # 32 of the 256 CUs and is pure dependent-step latency, and what follows it on the critical path is ONE data-
                             # gradient GEMM -- h_{t-1} + dW_hh (two split-K GEMMs and their reduces, ~80 us per layer at 64 x 256, H = 256) and
                             # the bias sums + dW_ih (~140 us) wait for nobody until the optimizer.  Round 3 kept dW_hh behind every recurrence
                             # on the main stream (inside mts_lstm_bwd) and ran layer 0's gradients, all of them, after its recurrence, one
                             # after the other (mts_lstm_bwd_recurrence / _whh split the call; which arrangement: overlap_mode below)

    # Measured, one box per table (profiles/r04_rnn_overlap_modes.txt; ms per step, BiLSTM 64 x 256 | late fusion 64 x 512):
    #   0 everything on the issuing stream                                            3.087 | 7.24
    #   1 both chains (h_{t-1} + dW_hh, then bias sums + dW_ih) on ONE side stream    3.014 | 7.07      <- default
    #   2 the two chains on two side streams                                          3.020 | 8.0 (its two extra streams crowd the hardware queues)
    #   3 dW_hh on the issuing stream behind the data gradient, the rest on a side stream   3.068 | 7.21
    #   4 as 3, layers above the first only; 5 round 3 call for call (h_{t-1} in front of the recurrence, dW_hh behind it, on the issuing stream)   3.087 | 7.24, 7.20
    # The chains do not come for free: next to a recurrence the 128-tile dW_hh GEMMs take 61 us instead of 25 and the recurrence 663 us instead
    # of 639 (rocprofv3 of modes 0 / 2) -- both poll / stream through the same L2s.
    overlap_mode = None if os.environ.get('MTS_RNN_OVERLAP') is None else int(os.environ['MTS_RNN_OVERLAP'])

    def _wg_streams(self, dev, n):
        """n (1 | 2) of the small shared pool of side streams for this encoder: the dW_hh chain's and the bias + dW_ih chain's (late fusion: pool
        stream 0 is the second encoder's own).  Only the streams a mode USES are created: HIP hands its hardware queues out in creation order, and
        two unused streams of the pool in front of the ones that matter moved late fusion's encoders onto one queue (8.2 against 7.2 ms)."""
        base = 2 if self.tag == 'r2' else 1
        return tuple(ops.side_stream(dev, base + 2 * i) for i in range(n))

    def backward(self, saved, dout, lengths_i32, B, Lq):
        o, H = self.o, self.H
        dt, dev = o.compute_dtype, dout.device
        N = B * Lq
        wf, pf, lay = o._weights(), o._flat, o._layout
        g = o.grad_flat()
        cur = torch.cuda.current_stream(dev) if dev.type == 'cuda' else None
        # not under a data-parallel hook: with RCCL's streams and late fusion's second encoder stream in the process two more streams
        # made the step slower, not faster (one-rank RCCL run at 64 x 512: 8.69 ms against 7.75 without them; 7.56 / 7.74 without a hook)
        mode = self.overlap_mode if self.overlap_mode is not None else 1
        if not (self.overlap_wgrad and o._grad_hook is None and cur is not None) or (mode == 4 and self.nl < 2):
            mode = 0
        sides = self._wg_streams(dev, 2 if mode == 2 else 1) if mode else None
        for k in range(self.nl - 1, -1, -1):
            S = saved[k]
            din = S['hin'].shape[1]
            w_hh = o._wspan(pf, *self._names('weight_hh', k), 8 * H, H)
            # one buffer PER LAYER: layer k's weight gradient and bias sums read its dxproj on the side streams, and nothing on the main
            # stream waits for them before the loop ends -- a buffer shared by layers k and k - 2 (two buffers, as this was) is rewritten by
            # layer k - 2's recurrence with only "a recurrence outlasts a GEMM" in between (silent corruption at >= 3 layers)
            dxproj = o._ws.get(f'{self.tag}dxproj{k}', N, 8 * H, dt, dev)
            # ... and one recurrence workspace per layer: mts_lstm_bwd_whh finds (or builds) h_{t-1} in it and takes its split-K slabs from it
            # while the next layer's recurrence is already running
            ws = ops.lstm_workspace(dt, B, Lq, H, 2, dev, tag=f'{self.tag}lstm{k}')
            off_hh, n_hh = lay.span(*self._names('weight_hh', k))
            if mode == 5:                              # round 3, call for call: h_{t-1}, recurrence, dW_hh in ONE call on the issuing stream
                ops.lstm_bwd(w_hh, lengths_i32, S['out'], S['gates'], S['cells'], dout, B, Lq, H, 2, dxproj, g[off_hh:off_hh + n_hh], ws=ws)
            else:
                ops.lstm_bwd_recurrence(w_hh, lengths_i32, S['out'], S['gates'], S['cells'], dout, B, Lq, H, 2, dxproj, ws)
            ev = None
            if sides is not None:
                ev = torch.cuda.Event()
                ev.record(cur)                         # dxproj of layer k is complete on the issuing stream
            if k > 0:                                  # the critical path first: what the next recurrence needs
                dprev = o._ws.get(f'{self.tag}dprev{k & 1}', N, din, dt, dev)
                ops.linear_dgrad(dxproj, o._wspan(wf, *self._names('weight_ih', k), 8 * H, din), dprev)
                dout = dprev

            def whh_of_layer():
                ops.lstm_bwd_whh(lengths_i32, S['out'], dxproj, B, Lq, H, 2, g[off_hh:off_hh + n_hh], ws)

            def params_of_layer():
                off, n = lay.span(*self._names('bias_ih', k))
                ops.colsum(dxproj, g[off:off + n])
                off2, n2 = lay.span(*self._names('bias_hh', k))
                g[off2:off2 + n2].copy_(g[off:off + n])
                off, n = lay.span(*self._names('weight_ih', k))
                ops.linear_wgrad(dxproj, S['hin'], g[off:off + n].view(8 * H, din))

            if mode == 2:                              # two chains, two streams
                for side, fn in zip(sides, (whh_of_layer, params_of_layer)):
                    side.wait_event(ev)
                    with torch.cuda.stream(side):
                        fn()
            elif mode == 1:                            # one side stream, both chains one after the other
                sides[0].wait_event(ev)
                with torch.cuda.stream(sides[0]):
                    whh_of_layer()
                    params_of_layer()
            elif mode == 5:
                if k > 0:
                    sides[0].wait_event(ev)
                    with torch.cuda.stream(sides[0]):
                        params_of_layer()
                else:
                    params_of_layer()
            elif mode == 3 or (mode == 4 and k > 0):   # dW_hh behind the data gradient on the main stream, the rest on a side stream
                whh_of_layer()                         # (4: only for the layers above the first -- round 3's arrangement)
                sides[0].wait_event(ev)
                with torch.cuda.stream(sides[0]):
                    params_of_layer()
            else:
                whh_of_layer()
                params_of_layer()
                # layer k's parameters (both directions: W_ih, W_hh, b_ih, b_hh are adjacent groups) are final: a data-parallel
                # trainer may start reducing them while the lower layers' recurrences still run (announced from the stream that
                # produced them: the collective is ordered behind it)
                a0 = lay.entries[self._names('weight_ih', k)[0]][0]
                b0, bn = lay.span(*self._names('bias_hh', k))
                o._grads_ready(a0, b0 + bn)
        if sides is not None:
            for side in sides:
                cur.wait_stream(side)                  # every gradient has landed before the caller (optimizer, exchange waits) goes on


class _RnnTaggerBase(_TaggerBase):
    grad_hooks_cover_all = True      # every parameter span is announced through _grads_ready (trainer.NativeTrainer overlaps the exchange)

    def _span_of(self, first, last):
        a, _ = self._layout.entries[first]
        b, n = self._layout.span(last, last)
        return a, b + n

    def _check_rnn_args(self, dropout_in, dropout_out, LSTM, bidirectional):
        for pr in (dropout_in, dropout_out):
            if not 0.0 <= float(pr) < 1.0:
                raise ValueError(f'dropout probability has to be between 0 and 1, but got {pr}')      # F.dropout's own check
        # RNN.forward calls F.dropout(x, p) WITHOUT training=..., so the reference drops in eval mode too; same here
        self.dropout_in, self.dropout_out = float(dropout_in), float(dropout_out)
        self._drop_calls = 0
        if not LSTM:
            # dead upstream (fixture tests/golden/g15_adjacent_encoders.npz): RNN.forward hands nn.GRU an (h0, c0) tuple
            # (NeuralArchitectures.py:104-113) -> AttributeError: 'tuple' object has no attribute 'index_select' on every call
            raise NotImplementedError('GRU (LSTM=False, NeuralArchitectures.py:46-50): the reference itself raises AttributeError on '
                                      'every tagger call with this option; there is no behaviour to reproduce')
        if not bidirectional:
            # dead upstream (g15): :134-145 returns the PackedSequence un-padded -> TypeError in the tagger's nn.Linear
            raise NotImplementedError('unidirectional RNN (bidirectional=False, NeuralArchitectures.py:134-145): the reference itself '
                                      'raises TypeError (PackedSequence handed to nn.Linear) on every tagger call with this option')

    def _prep_input(self, xs, lengths):
        """pad_packed_sequence semantics: the output covers max(lengths) positions (NeuralArchitectures.py:115).
        ``xs`` may be a K-split pair (taggers._split_input): both parts are trimmed, the pair is returned."""
        x1, x2, B, Lin, _ = self._split_input(xs)
        maxlen = int(lengths.max()) if lengths is not None else Lin
        maxlen = min(maxlen, Lin)
        if x2 is not None:
            return (x1[:, :maxlen].contiguous(), x2[:, :maxlen].contiguous()), maxlen
        return x1[:, :maxlen].contiguous(), maxlen

    def _drop_seed(self):
        """A fresh 64-bit seed per dropout call, derived from torch's seed (torch.manual_seed makes runs repeatable)."""
        self._drop_calls += 1
        return (torch.initial_seed() * 1000003 + self._drop_calls * 7919) & 0x7FFFFFFFFFFFFFFF

    def _drop_in(self, xa, tag):
        """F.dropout on the RNN input (NeuralArchitectures.py:94); no mask kept: nothing upstream needs a gradient."""
        if not self.dropout_in:
            return xa
        out = self._ws.get('xdrop_' + tag, xa.shape[0], xa.shape[1], xa.dtype, xa.device)
        ops.dropout_fwd(xa, out, self.dropout_in, self._drop_seed())
        return out

    def _drop_out(self, h, tag):
        """F.dropout on the RNN output (NeuralArchitectures.py:119) -> (dropped copy, mask); `h` itself stays intact for the
        recurrence backward."""
        if not self.dropout_out:
            return h, None
        out = self._ws.get('hdrop_' + tag, h.shape[0], h.shape[1], h.dtype, h.device)
        mask = self._ws.get('hmask_' + tag, h.shape[0], h.shape[1], torch.uint8, h.device)
        ops.dropout_fwd(h, out, self.dropout_out, self._drop_seed(), mask=mask)
        return out, mask

    def _to_act(self, x):
        if isinstance(x, tuple):                                       # K-split pair: concat + cast in one pass over the two fp32 parts
            a, b = (t.reshape(-1, t.shape[-1]).to(torch.float32).contiguous() for t in x)
            D = a.shape[1] + b.shape[1]
            if D % 8 == 0:
                out = self._ws.get('xin_' + str(D), a.shape[0], D, self.compute_dtype, a.device)
                return ops.cast_concat(a, b, out)
            x = torch.cat((a, b), dim=1)                               # odd widths (timing features): the padded path below
        x2 = x.reshape(-1, x.shape[-1])
        if x2.dtype == torch.bfloat16 and self.compute_dtype == torch.bfloat16 and x2.shape[1] % 8 == 0:
            return x2.contiguous()                                     # already in the act dtype (prefetch.DevicePrefetcher wire_dtype='bf16'):
                                                                       # the same bits the cast below would produce from the fp32 batch
        if x2.shape[1] % 8:                                           # e.g. 768 + 2 timing features: zero columns up to a multiple of 8
            x2 = torch.nn.functional.pad(x2.to(torch.float32), (0, round_up(x2.shape[1], 8) - x2.shape[1]))
        if self.compute_dtype == torch.float32:
            return x2.to(torch.float32).contiguous()
        out = self._ws.get('xin_' + str(x2.shape[1]), x2.shape[0], x2.shape[1], torch.bfloat16, x.device)
        ops.cast(x2.to(torch.float32).contiguous(), out)
        return out


class BiLSTM(_RnnTaggerBase):
    """models/CRF.py:274-369."""

    def __init__(self, tagset_size, embedding_dim, hidden_dim, num_layers=1, bidirectional=True, dropout_in=0.0, dropout_out=0.0,
                 batch_first=True, LSTM=True, loss_fn='CrossEntropy', threshold=None, device=None, alpha=0.9, gamma=2,
                 compute_dtype=None, seed=None):
        super().__init__()
        self._init_common(loss_fn, threshold, alpha, gamma, compute_dtype)
        self._check_rnn_args(dropout_in, dropout_out, LSTM, bidirectional)
        self.embedding_dim, self.hidden_dim, self.tagset_size, self.num_layers = embedding_dim, hidden_dim, tagset_size, num_layers
        self.n_out = tagset_size if loss_fn == 'CrossEntropy' else 1
        gen = torch.Generator().manual_seed(torch.initial_seed() if seed is None else seed)
        groups, init, pads = _rnn_groups('model.', embedding_dim, hidden_dim, num_layers, gen)
        self._hp = round_up(hidden_dim, 8)
        cw, cb = _linear_init(self.n_out, 2 * hidden_dim, gen)
        groups.append([('classification.weight', (self.n_out, 2 * self._hp)), ('classification.bias', (self.n_out,))])
        init['classification.weight'], init['classification.bias'] = cw, cb
        if _head_pad(hidden_dim, 2):
            pads['classification.weight'] = _head_pad(hidden_dim, 2)
        self._init_flat(FlatLayout(groups, pads), init)
        self._rnn = _RnnStack(self, 'model.', embedding_dim, hidden_dim, num_layers, 'r')

    def _fwd(self, xs, lengths):
        x, Lq = self._prep_input(xs, lengths)
        first = x[0] if isinstance(x, tuple) else x
        B, dev = first.shape[0], first.device
        li32 = self._prep_lengths(lengths, B, Lq, dev)
        h, saved = self._rnn.forward(self._drop_in(self._to_act(x), 'r'), li32, B, Lq)
        h, hmask = self._drop_out(h, 'r')
        scores = self._ws.get('scores', B * Lq, self.n_out, torch.float32, dev)
        ops.head_fwd(h, self._w(self._flat, 'classification.weight'), self._w(self._flat, 'classification.bias'), scores)
        return dict(B=B, L=Lq, li32=li32, h=h, hmask=hmask, saved=saved, scores=scores.view(B, Lq, self.n_out))

    def loss_and_grad(self, xs, lengths, tags, want_grad=True):
        L.require_gpu()
        st = self._fwd(xs, lengths)
        dev, B, Lq = st['scores'].device, st['B'], st['L']
        tg = tags.to(device=dev, dtype=torch.float32).contiguous()
        if self.loss_kind == L.LOSS_CE and tg.shape[1] != Lq:
            # the reference reshapes x [B*maxlen, 2] against tags [B*L] (CRF.py:354) and fails on a size mismatch
            raise ValueError(f'Expected input batch_size ({B * Lq}) to match target batch_size ({tg.numel()}).')
        loss_out = torch.empty(2, dtype=torch.float32, device=dev)
        dsc = self._ws.get('dscores', B * Lq, self.n_out, torch.float32, dev) if want_grad else None
        ops.tagger_loss(self.loss_kind, st['scores'], tg, st['li32'], self.alpha, self.gamma, loss_out, dsc)
        if want_grad:
            ops.scale_(dsc, self.loss_grad_scale)
            g, lay = self.grad_flat(), self._layout
            ops.head_bwd_params(st['h'], dsc, lay.view(g, 'classification.weight'), lay.view(g, 'classification.bias'))
            self._grads_ready(*self._span_of('classification.weight', 'classification.bias'))
            dout = self._ws.get('dout', B * Lq, 2 * self._hp, self.compute_dtype, dev)
            ops.head_bwd_data(dsc, self._w(self._flat, 'classification.weight'), dout)
            if st['hmask'] is not None:
                ops.dropout_bwd(dout, dout, st['hmask'], self.dropout_out)
            self._rnn.backward(st['saved'], dout, st['li32'], B, Lq)
        return loss_out[0], st['scores']

    def loss(self, xs, lengths, tags, segments=None):
        """models/CRF.py:319-356 (segments / cosine auxiliary loss: SURVEY.md §8f 'next')."""
        if segments is not None:
            raise NotImplementedError('cosine auxiliary loss (models/CRF.py:23-92): no collater of the reference produces '
                                      "batch['src_segments'] (TextSegmenter.training_step raises KeyError upstream, fixture g15)")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self._flat_params.values()):
            return self._autograd_loss(lambda: self.loss_and_grad(xs, lengths, tags, True)[0])
        return self.loss_and_grad(xs, lengths, tags, False)[0].clone()

    def forward(self, xs, lenghts, threshold=0.4):
        """models/CRF.py:358-369."""
        L.require_gpu()
        with torch.no_grad():
            st = self._fwd(xs, lenghts)
            scores = st['scores'].clone()
            tags = self._decode(scores, st['li32'], lenghts, threshold)
        return scores, tags


class BiLSTMLateFusion(_RnnTaggerBase):
    """models/CRF.py:371-479: two independent RNNs, plain concat (there is no gate in the reference), one head."""
    concurrent_encoders = True       # model1 / model2 on two HIP streams (bitwise the same results).  Under a data-parallel hook too:
                                     # model2's spans are announced from inside `with stream(side)`, so the collective is issued
                                     # with the side stream current and waits on exactly the kernels that wrote the span (RCCL and
                                     # gloo both order a collective behind the stream that is current when it is issued); the host
                                     # order of the announcements -- model2's layers, then model1's -- is the same on every rank

    def _side_stream(self, dev):
        return ops.side_stream(dev, 0)

    def __init__(self, tagset_size, embedding_dim, hidden_dim, num_layers=1, bidirectional=True, dropout_in=0.0, dropout_out=0.0,
                 batch_first=True, LSTM=True, loss_fn='CrossEntropy', threshold=None, device=None, alpha=0.9, gamma=2,
                 compute_dtype=None, seed=None):
        super().__init__()
        self._init_common(loss_fn, threshold, alpha, gamma, compute_dtype)
        self._check_rnn_args(dropout_in, dropout_out, LSTM, bidirectional)
        self.embedding_dim, self.hidden_dim, self.tagset_size, self.num_layers = embedding_dim, hidden_dim, tagset_size, num_layers
        self.n_out = tagset_size if loss_fn == 'CrossEntropy' else 1
        gen = torch.Generator().manual_seed(torch.initial_seed() if seed is None else seed)
        g1, i1, p1 = _rnn_groups('model1.', embedding_dim[0], hidden_dim, num_layers, gen)
        g2, i2, p2 = _rnn_groups('model2.', embedding_dim[1], hidden_dim, num_layers, gen)
        self._hp = round_up(hidden_dim, 8)
        cw, cb = _linear_init(self.n_out, 4 * hidden_dim, gen)
        groups = g1 + g2 + [[('classification.weight', (self.n_out, 4 * self._hp)), ('classification.bias', (self.n_out,))]]
        init = {**i1, **i2, 'classification.weight': cw, 'classification.bias': cb}
        pads = {**p1, **p2}
        if _head_pad(hidden_dim, 4):
            pads['classification.weight'] = _head_pad(hidden_dim, 4)
        self._init_flat(FlatLayout(groups, pads), init)
        self._rnn1 = _RnnStack(self, 'model1.', embedding_dim[0], hidden_dim, num_layers, 'r1')
        self._rnn2 = _RnnStack(self, 'model2.', embedding_dim[1], hidden_dim, num_layers, 'r2')

    def _fwd(self, x1, x2, lengths):
        xa, Lq = self._prep_input(x1, lengths)
        xb, _ = self._prep_input(x2, lengths)
        B, H = xa.shape[0], self._hp
        li32 = self._prep_lengths(lengths, B, Lq, x1.device)
        # the two encoders are independent and each recurrence occupies a few dozen CUs: run the second one on a side stream
        side = self._side_stream(x1.device) if self.concurrent_encoders else None
        if side is not None:
            main = torch.cuda.current_stream(x1.device)
            self._weights()                            # a stale bf16 mirror is re-cast HERE, on the main stream, ahead of both encoders
            side.wait_stream(main)
            with torch.cuda.stream(side):
                h2, s2 = self._rnn2.forward(self._drop_in(self._to_act(xb), 'r2'), li32, B, Lq)
            h1, s1 = self._rnn1.forward(self._drop_in(self._to_act(xa), 'r1'), li32, B, Lq)
            main.wait_stream(side)
        else:
            h1, s1 = self._rnn1.forward(self._drop_in(self._to_act(xa), 'r1'), li32, B, Lq)
            h2, s2 = self._rnn2.forward(self._drop_in(self._to_act(xb), 'r2'), li32, B, Lq)
        h1, m1 = self._drop_out(h1, 'r1')
        h2, m2 = self._drop_out(h2, 'r2')
        cat = self._ws.get('cat', B * Lq, 4 * H, self.compute_dtype, x1.device)
        cat[:, :2 * H].copy_(h1)                       # torch.cat((x1, x2), axis=2), models/CRF.py:425
        cat[:, 2 * H:].copy_(h2)
        scores = self._ws.get('scores', B * Lq, self.n_out, torch.float32, x1.device)
        ops.head_fwd(cat, self._w(self._flat, 'classification.weight'), self._w(self._flat, 'classification.bias'), scores)
        return dict(B=B, L=Lq, li32=li32, cat=cat, s1=s1, s2=s2, m1=m1, m2=m2, scores=scores.view(B, Lq, self.n_out))

    def loss_and_grad(self, x1, x2, lengths, tags, want_grad=True):
        L.require_gpu()
        st = self._fwd(x1, x2, lengths)
        dev, B, Lq, H = x1.device, st['B'], st['L'], self._hp
        tg = tags.to(device=dev, dtype=torch.float32).contiguous()
        loss_out = torch.empty(2, dtype=torch.float32, device=dev)
        dsc = self._ws.get('dscores', B * Lq, self.n_out, torch.float32, dev) if want_grad else None
        ops.tagger_loss(self.loss_kind, st['scores'], tg, st['li32'], self.alpha, self.gamma, loss_out, dsc)
        if want_grad:
            ops.scale_(dsc, self.loss_grad_scale)
            g, lay = self.grad_flat(), self._layout
            ops.head_bwd_params(st['cat'], dsc, lay.view(g, 'classification.weight'), lay.view(g, 'classification.bias'))
            self._grads_ready(*self._span_of('classification.weight', 'classification.bias'))
            dcat = self._ws.get('dcat', B * Lq, 4 * H, self.compute_dtype, dev)
            ops.head_bwd_data(dsc, self._w(self._flat, 'classification.weight'), dcat)
            d1 = self._ws.get('dout1', B * Lq, 2 * H, self.compute_dtype, dev)
            d2 = self._ws.get('dout2', B * Lq, 2 * H, self.compute_dtype, dev)
            d1.copy_(dcat[:, :2 * H])
            d2.copy_(dcat[:, 2 * H:])
            if st['m1'] is not None:
                ops.dropout_bwd(d1, d1, st['m1'], self.dropout_out)
                ops.dropout_bwd(d2, d2, st['m2'], self.dropout_out)
            side = self._side_stream(dev) if self.concurrent_encoders else None
            if side is not None:
                main = torch.cuda.current_stream(dev)
                self._weights()                        # see _fwd: never let the side stream be the one that refreshes the mirror
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    self._rnn2.backward(st['s2'], d2, st['li32'], B, Lq)
                self._rnn1.backward(st['s1'], d1, st['li32'], B, Lq)
                main.wait_stream(side)
            else:
                self._rnn1.backward(st['s1'], d1, st['li32'], B, Lq)
                self._rnn2.backward(st['s2'], d2, st['li32'], B, Lq)
        return loss_out[0], st['scores']

    def loss(self, x1, x2, lengths, tags, segments=None):
        """models/CRF.py:420-461."""
        if segments is not None:
            raise NotImplementedError('cosine auxiliary loss (models/CRF.py:23-92): no collater of the reference produces '
                                      "batch['src_segments'] (TextSegmenter.training_step raises KeyError upstream, fixture g15)")
        if torch.is_grad_enabled() and any(p.requires_grad for p in self._flat_params.values()):
            return self._autograd_loss(lambda: self.loss_and_grad(x1, x2, lengths, tags, True)[0])
        return self.loss_and_grad(x1, x2, lengths, tags, False)[0].clone()

    def forward(self, x1, x2, lenghts, threshold=0.4):
        """models/CRF.py:463-479."""
        L.require_gpu()
        with torch.no_grad():
            st = self._fwd(x1, x2, lenghts)
            scores = st['scores'].clone()
            tags = self._decode(scores, st['li32'], lenghts, threshold)
        return scores, tags


class BiRnnCrf(_RnnTaggerBase):
    """models/CRF.py:243-272 composed with CRF (:98-240): RNN -> fc(2H -> tags+2) -> CRF NLL / Viterbi."""

    def __init__(self, tagset_size, embedding_dim, hidden_dim, num_layers=1, bidirectional=True, dropout_in=0.0, dropout_out=0.0,
                 batch_first=True, LSTM=True, architecture='rnn', compute_dtype=None, seed=None):
        super().__init__()
        self._init_common('CrossEntropy', None, 0.9, 2, compute_dtype)
        self._check_rnn_args(dropout_in, dropout_out, LSTM, bidirectional)
        self.embedding_dim, self.hidden_dim, self.tagset_size, self.num_layers = embedding_dim, hidden_dim, tagset_size, num_layers
        self.num_tags = tagset_size + 2                       # models/CRF.py:108-110
        self.start_idx, self.stop_idx = self.num_tags - 2, self.num_tags - 1
        if self.num_tags > 4:
            raise NotImplementedError('CRF head supports tagset_size <= 2 (fused head kernels cover <= 4 outputs)')
        gen = torch.Generator().manual_seed(torch.initial_seed() if seed is None else seed)
        groups, init, pads = _rnn_groups('model.', embedding_dim, hidden_dim, num_layers, gen)
        self._hp = round_up(hidden_dim, 8)
        C = self.num_tags
        fw, fb = _linear_init(C, 2 * hidden_dim, gen)
        trans = torch.randn(C, C, generator=gen)
        trans[self.start_idx, :] = IMPOSSIBLE                 # models/CRF.py:115-117
        trans[:, self.stop_idx] = IMPOSSIBLE
        groups.append([('crf.fc.weight', (C, 2 * self._hp)), ('crf.fc.bias', (C,))])
        groups.append([('crf.transitions', (C, C))])
        init.update({'crf.fc.weight': fw, 'crf.fc.bias': fb, 'crf.transitions': trans})
        if _head_pad(hidden_dim, 2):
            pads['crf.fc.weight'] = _head_pad(hidden_dim, 2)
        self._init_flat(FlatLayout(groups, pads), init)
        self._rnn = _RnnStack(self, 'model.', embedding_dim, hidden_dim, num_layers, 'r')

    def _fwd(self, xs, lengths):
        x, Lq = self._prep_input(xs, lengths)
        first = x[0] if isinstance(x, tuple) else x
        B, dev = first.shape[0], first.device
        li32 = self._prep_lengths(lengths, B, Lq, dev)
        h, saved = self._rnn.forward(self._drop_in(self._to_act(x), 'r'), li32, B, Lq)
        h, hmask = self._drop_out(h, 'r')
        feats = self._ws.get('feats', B * Lq, self.num_tags, torch.float32, dev)
        ops.head_fwd(h, self._w(self._flat, 'crf.fc.weight'), self._w(self._flat, 'crf.fc.bias'), feats)
        return dict(B=B, L=Lq, li32=li32, h=h, hmask=hmask, saved=saved, feats=feats.view(B, Lq, self.num_tags))

    def loss_and_grad(self, xs, lengths, tags, want_grad=True):
        L.require_gpu()
        st = self._fwd(xs, lengths)
        dev, B, Lq, C = st['feats'].device, st['B'], st['L'], self.num_tags
        tg = tags.to(device=dev, dtype=torch.float32).contiguous()
        loss_out = torch.empty(2, dtype=torch.float32, device=dev)
        g, lay = self.grad_flat(), self._layout
        dfe = self._ws.get('dfeats', B * Lq, C, torch.float32, dev) if want_grad else None
        ops.crf_nll(st['feats'], tg, st['li32'], self._w(self._flat, 'crf.transitions'), loss_out,
                    dfe.view(B, Lq, C) if want_grad else None, lay.view(g, 'crf.transitions') if want_grad else None)
        if want_grad:
            ops.scale_(dfe, self.loss_grad_scale)
            ops.scale_(lay.view(g, 'crf.transitions'), self.loss_grad_scale)
            ops.head_bwd_params(st['h'], dfe, lay.view(g, 'crf.fc.weight'), lay.view(g, 'crf.fc.bias'))
            self._grads_ready(*self._span_of('crf.fc.weight', 'crf.transitions'))
            dout = self._ws.get('dout', B * Lq, 2 * self._hp, self.compute_dtype, dev)
            ops.head_bwd_data(dfe, self._w(self._flat, 'crf.fc.weight'), dout)
            if st['hmask'] is not None:
                ops.dropout_bwd(dout, dout, st['hmask'], self.dropout_out)
            self._rnn.backward(st['saved'], dout, st['li32'], B, Lq)
        return loss_out[0], st['feats']

    def loss(self, xs, lengths, tags):
        """models/CRF.py:261-265."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self._flat_params.values()):
            return self._autograd_loss(lambda: self.loss_and_grad(xs, lengths, tags, True)[0])
        return self.loss_and_grad(xs, lengths, tags, False)[0].clone()

    def forward(self, xs, lenghts):
        """models/CRF.py:267-272 -> (best_score [B], best_paths list of int lists)."""
        L.require_gpu()
        with torch.no_grad():
            st = self._fwd(xs, lenghts)
            B, Lq = st['B'], st['L']
            score = torch.empty(B, dtype=torch.float32, device=st['feats'].device)
            paths = torch.empty(B, Lq, dtype=torch.int32, device=st['feats'].device)
            ops.crf_viterbi(st['feats'], st['li32'], self._w(self._flat, 'crf.transitions'), score, paths)
            ph = paths.cpu().numpy()
            L.check_async()            # synchronised by the copy: report a CU-pair LSTM timeout of this forward instead of its paths
            lens = [int(v) for v in (lenghts.tolist() if lenghts is not None else [Lq] * B)]
        return score, [ph[i, :min(lens[i], Lq)].tolist() for i in range(B)]
