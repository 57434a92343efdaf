"""Host-side mirror of the reference's tagger classes (models/CRF.py) on top of the HIP kernels.

Same class names, constructor arguments, ``.loss(...)`` / ``.forward(...)`` contracts, ``.th`` attribute,
error behaviour and ``state_dict`` keys as the reference:

    Transformer_segmenter   models/CRF.py:508-610  (+ Longformer_Local_Attention, RestrictedTransformerLayer.py:65-133)
    BiLSTM                  models/CRF.py:274-369  (+ RNN, NeuralArchitectures.py:23-145)
    BiLSTMLateFusion        models/CRF.py:371-479
    BiRnnCrf / CRF          models/CRF.py:243-272, :98-240   (the reference wrapper crashes; this one works)

The arithmetic runs ONLY in libmts_hip.so (see include/mts.h); there is no PyTorch fallback.  Every model keeps
its parameters in one flat fp32 buffer (flat.py) and has two front-ends over the same kernels:

  * ``.loss()`` returns a 0-d tensor wired into autograd through a single custom Function (drop-in for
    Lightning / torch optimizers: train_fit.py:300-335 works unchanged);
  * ``.loss_and_grad()`` writes all gradients straight into the flat gradient buffer without autograd
    (native trainer, bench.py, RCCL data parallel).
"""
import functools
import math
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .flat import FlatLayout, FlatModule, round_up

LOSS_KINDS = {'CrossEntropy': L.LOSS_CE, 'BinaryCrossEntropy': L.LOSS_BCE, 'FocalLoss': L.LOSS_FOCAL}
DEAD_HF_KEYS = ('word_embeddings', 'query_global', 'key_global', 'value_global', 'pooler', 'position_ids', 'token_type_ids')


def default_compute_dtype():
    v = os.environ.get('MTS_COMPUTE_DTYPE', 'fp32').lower()
    return torch.bfloat16 if v in ('bf16', 'bfloat16') else torch.float32


def _as_dtype(d):
    if d is None:
        return default_compute_dtype()
    if isinstance(d, str):
        return torch.bfloat16 if d.lower() in ('bf16', 'bfloat16') else torch.float32
    return d


def _xavier_uniform(shape, gen):
    bound = math.sqrt(6.0 / (shape[0] + shape[1]))
    return (torch.rand(shape, generator=gen) * 2 - 1) * bound


def _linear_init(out_f, in_f, gen):
    """nn.Linear default init (kaiming_uniform(a=sqrt(5)) -> U(-1/sqrt(in), 1/sqrt(in)) for weight and bias)."""
    k = 1.0 / math.sqrt(in_f)
    return (torch.rand(out_f, in_f, generator=gen) * 2 - 1) * k, (torch.rand(out_f, generator=gen) * 2 - 1) * k


_LENGTHS_CACHE = {}


class _Workspace:
    """Row-capacity-based activation buffers: allocated once for the largest B*L seen, sliced per batch."""

    def __init__(self):
        self.bufs = {}
        self.generation = 0              # bumped whenever a buffer is (re)allocated: captured graphs hold the old addresses

    def get(self, name, rows, cols, dtype, device):
        key = name
        t = self.bufs.get(key)
        need = rows * cols
        if t is None or t.numel() < need or t.dtype != dtype or t.device != device:
            t = torch.empty(max(need, 1), dtype=dtype, device=device)
            self.bufs[key] = t
            self.generation += 1
        return t[:need].view(rows, cols)


def _on_model_device(fn):
    """Run a public entry point with the model's GPU as the current device: the kernels are launched on torch's
    CURRENT stream (`_lib.stream_ptr`) of the CURRENT device, so a model living on cuda:1 called while cuda:0 is current would
    otherwise enqueue on the wrong device's stream."""
    @functools.wraps(fn)
    def guarded(self, *args, **kwargs):
        dev = self._flat.device
        if dev.type != 'cuda' or torch.cuda.current_device() == dev.index:
            return fn(self, *args, **kwargs)
        with torch.cuda.device(dev):
            return fn(self, *args, **kwargs)
    guarded._mts_device_guard = True
    return guarded


class _TaggerBase(FlatModule):
    """Shared tail: loss kinds, decode, autograd bridge, bf16 weight mirror."""

    _GUARDED_ENTRY_POINTS = ('loss_and_grad', 'loss', 'encode', 'forward')

    def __init_subclass__(cls, **kw):
        super().__init_subclass__(**kw)
        for name in cls._GUARDED_ENTRY_POINTS:
            fn = cls.__dict__.get(name)
            if fn is not None and not getattr(fn, '_mts_device_guard', False):
                setattr(cls, name, _on_model_device(fn))


    def _init_common(self, loss_fn, threshold, alpha, gamma, compute_dtype):
        if loss_fn not in LOSS_KINDS:
            raise ValueError('Choose one of CrossEntropy or BinaryCrossEntropy as loss function')   # models/CRF.py:312
        self.loss_kind = LOSS_KINDS[loss_fn]
        self.bce = loss_fn != 'CrossEntropy'
        self.fl = loss_fn == 'FocalLoss'
        self.alpha, self.gamma = float(alpha), float(gamma)
        self.th = threshold
        self.compute_dtype = _as_dtype(compute_dtype)
        self._ws = _Workspace()
        self._wcopy = None
        self._wcopy_version = None
        self.device = 'cuda' if torch.cuda.is_available() else 'cpu'   # models/CRF.py:283-286 (kept for API parity)

    def _on_reflatten(self):
        self._wcopy = None
        self._wcopy_version = None

    # ---- weights in compute precision ----------------------------------------------------------
    def _weights_version(self):
        """Staleness signal of the bf16 mirror.  The flat buffer's own version counter is NOT enough: after ``_reflatten``
        (every .to()/.cuda()) each parameter is re-pointed with ``p.data = view`` and keeps a version counter of its own, so an
        in-place write through the parameter (torch optimizers, load_state_dict) no longer bumps ``_flat._version``.  Writes
        through raw pointers (the fused optimizer kernels) bump nothing: NativeTrainer calls mark_weights_synced()."""
        return (self._flat._version, sum(p._version for p in self._flat_params.values()))

    def _weights(self):
        """flat buffer in the compute dtype (bf16 mirror refreshed when the fp32 master changed)."""
        if self.compute_dtype == torch.float32:
            return self._flat
        ver = self._weights_version()
        if self._wcopy is None or self._wcopy.device != self._flat.device:
            self._wcopy = torch.empty(self._flat.numel(), dtype=torch.bfloat16, device=self._flat.device)
            self._wcopy_version = None
        if self._wcopy_version != ver:
            ops.cast(self._flat, self._wcopy)
            self._wcopy_version = ver
        return self._wcopy

    loss_grad_scale = 1.0            # multiplies d loss / d scores (trainer.NativeTrainer: token-weighted data parallelism)
    _grad_hook = None
    grad_hooks_cover_all = False     # subclasses that announce every parameter span through _grads_ready set this

    def _grads_ready(self, a, b):
        """Flat-gradient span [a, b) is final for this step (hook installed by trainer.NativeTrainer for RCCL overlap)."""
        if self._grad_hook is not None and b > a:
            self._grad_hook(a, b)

    def mark_weights_synced(self):
        """The fused optimizer wrote the bf16 mirror itself: skip the next cast."""
        self._wcopy_version = self._weights_version()

    def _w(self, wflat, name):
        return self._layout.view(wflat, name)

    def _wspan(self, wflat, first, last, rows, cols):
        off, n = self._layout.span(first, last)
        return wflat[off:off + n].view(rows, cols)

    @staticmethod
    def _split_input(xs):
        """K-split input: ``xs`` may be a pair (text [B, L, D1], audio [B, L, D2]) of separate fp32 tensors standing for their
        concatenation along the feature axis -- the early-fusion torch.cat of utils/load_datasets_precomputed.py:158-161 -- which
        is then never materialised: the kernels read both parts.  -> (x1, x2 or None, B, L, D1 + D2)"""
        if isinstance(xs, (tuple, list)):
            x1, x2 = xs
            if x1.shape[:2] != x2.shape[:2]:
                raise ValueError(f'K-split input parts disagree: {tuple(x1.shape)} vs {tuple(x2.shape)}')
            if x1.shape[2] % 4 or x2.shape[2] % 4:
                raise ValueError('K-split input parts must have widths that are multiples of 4')
            return x1, x2, x1.shape[0], x1.shape[1], x1.shape[2] + x2.shape[2]
        return xs, None, xs.shape[0], xs.shape[1], xs.shape[2]

    @staticmethod
    def _prep_lengths(lengths, B, L, device):
        if lengths is None:
            return torch.full((B,), L, dtype=torch.int32, device=device)
        if lengths.device.type == 'cpu' and lengths.numel() <= 4096:
            # a host tensor (the collater's): one small upload per distinct length vector, not one per step
            key = (tuple(lengths.tolist()), str(device))
            hit = _LENGTHS_CACHE.get(key)
            if hit is None:
                if len(_LENGTHS_CACHE) > 64:
                    _LENGTHS_CACHE.clear()
                hit = _LENGTHS_CACHE[key] = lengths.to(device=device, dtype=torch.int32).contiguous()
            return hit
        return lengths.to(device=device, dtype=torch.int32).contiguous()

    # ---- decode (models/CRF.py:358-369) -----------------------------------------------------------
    def _decode(self, scores, lengths_i32, lengths, threshold):
        if self.th is not None:
            threshold = self.th
        B, Lq, _ = scores.shape
        tags = torch.empty(B, Lq, dtype=torch.uint8, device=scores.device)
        ops.greedy_decode(scores, lengths_i32, threshold, tags)
        tags_h = tags.cpu().numpy().astype(bool)
        L.check_async()                # the copy synchronised: a device-side error of this forward (CU-pair LSTM timeout) is visible now
        lens = [int(v) for v in (lengths.tolist() if lengths is not None else [Lq] * B)]
        return [tags_h[i, :lens[i]].tolist() for i in range(B)]

    # ---- one restricted-window encoder layer (shared by Transformer_segmenter and the legacy layer) ---------------------
    # names: {'wqkv', 'bqkv', 'wo', 'bo', 'ln1w', 'ln1b', 'w1', 'b1', 'w2', 'b2', 'ln2w', 'ln2b'} -> (first, last) layout names
    # (a span of adjacent tensors; q/k/v are three tensors in the HF layout and one packed in_proj in the legacy one).
    # Needs self.embedding_dim, self.nheads, self._ffp (FFN width as stored), self.ln_eps, self.ffn_act ('gelu' | 'relu').
    ffn_act = 'gelu'
    qkv_release = os.environ.get('MTS_DP_QKV_RELEASE', 'block')    # under a data-parallel hook: 'projection' | 'block' (see _band_layer_bwd)
    fuse_ffn = os.environ.get('MTS_FUSE_FFN', '1') != '0'    # one launch per direction for the feed-forward block where mts_ffn_* covers it
                                                              # (bf16, F = 256, d a multiple of 256, no hidden dropout); bitwise the same results
    fuse_tail = os.environ.get('MTS_FUSE_TAIL', '1') != '0'  # training: the last layer's LayerNorm + head + loss + their backward in one pass over s2 where
                                                              # mts_layernorm_loss_tail covers it (n_out <= 2, D in {256, 512, 1024, 1792, 2048}); bitwise the
                                                              # same scores and gradients as the four launches it replaces
    fuse_ffn_min_rows = 12288        # ... and where its 64-row workgroups fill the chip: below ~192 workgroups the 128x128 GEMM pair is
                                     # 1-2 % faster end to end (8192 rows: 0.432 vs 0.438 ms per inference call; 2437 rows: 0.217 vs 0.221)

    def _lt(self, flat, names, key, rows, cols):
        first, last = names[key]
        return self._wspan(flat, first, last, rows, cols)

    def _band_layer_fwd(self, names, tag, h, lengths_i32, B, Lq, N, radius, row0, pdrop, pattn, head=None, store_out=True, skip_ln2=False):
        """post-LN layer: a = LN(dropout(ctx Wo^T + bo) + h), out = LN(dropout(act(a W1^T + b1) W2^T + b2) + a), ctx = band
        attention over q|k|v = h Wqkv^T + b (q scaled by 1/sqrt(hd)).  modeling_longformer.py:482-640,1061-1172 /
        RestrictedTransformerLayer.py:269-310.  head = (w, b, scores): tagger head fused into the last LayerNorm."""
        dt, dev = self.compute_dtype, h.device
        D, F, H = self.embedding_dim, self._ffp, self.nheads
        ws = self._ws
        wf, pf = self._weights(), self._flat
        slots = ops.band_slots(radius)
        qkv = ws.get(f'qkv{tag}', N, 3 * D, dt, dev)
        ops.linear_fwd(h, self._lt(wf, names, 'wqkv', 3 * D, D), self._lt(pf, names, 'bqkv', 1, 3 * D).view(-1), qkv,
                       colscale=1.0 / math.sqrt(D // H), ncols_scaled=D)
        ctx = ws.get(f'ctx{tag}', N, D, dt, dev)
        probs = ws.get(f'probs{tag}', N, H * slots, torch.float32, dev)
        aseed = self._drop_seed() if pattn else 0                  # attention_probs_dropout_prob, modeling_longformer.py:590
        ops.band_attn_fwd(qkv, lengths_i32, B, Lq, D, H, radius, ctx, probs, row0=row0, drop_p=pattn, drop_seed=aseed)
        s1 = ws.get(f's1_{tag}', N, D, dt, dev)
        m1 = m2 = None
        wo, bo = self._lt(wf, names, 'wo', D, D), self._lt(pf, names, 'bo', 1, D).view(-1)
        if pdrop:                                                  # dense -> dropout -> (+ input) -> LayerNorm, :1069-1072
            tmp = ws.get('droptmp', N, D, dt, dev)
            m1 = ws.get(f'dropmask1_{tag}', N, D, torch.uint8, dev)
            ops.linear_fwd(ctx, wo, bo, tmp)
            ops.dropout_fwd(tmp, s1, pdrop, self._drop_seed(), mask=m1, residual=h)
        else:
            ops.linear_fwd(ctx, wo, bo, s1, residual=h)
        a1 = ws.get(f'a1_{tag}', N, D, dt, dev)
        mean1 = ws.get(f'mean1_{tag}', N, 1, torch.float32, dev)
        rstd1 = ws.get(f'rstd1_{tag}', N, 1, torch.float32, dev)
        ops.layernorm_fwd(s1, self._lt(pf, names, 'ln1w', 1, D).view(-1), self._lt(pf, names, 'ln1b', 1, D).view(-1), self.ln_eps,
                          a1, mean1, rstd1)
        # the fused block writes whole 64-row tiles: its outputs live in buffers with ceil(N / 64) * 64 rows (rows past N are scratch)
        Np = round_up(N, 64)
        u = ws.get(f'u{tag}', Np, F, dt, dev)[:N]
        f = ws.get(f'f{tag}', Np, F, dt, dev)[:N]
        relu = self.ffn_act == 'relu'
        s2 = ws.get(f's2_{tag}', Np, D, dt, dev)[:N]
        w2, b2 = self._lt(wf, names, 'w2', D, F), self._lt(pf, names, 'b2', 1, D).view(-1)
        fused = self.fuse_ffn and not pdrop and N >= self.fuse_ffn_min_rows and ops.ffn_supported(dt, N, D, F)
        w1, b1 = self._lt(wf, names, 'w1', F, D), self._lt(pf, names, 'b1', 1, F).view(-1)
        if fused:                                                  # up-projection, activation, down-projection, residual: one launch
            ops.ffn_fwd(a1, w1, b1, w2, b2, u, f, s2, relu=relu)
        else:
            ops.linear_fwd(a1, w1, b1, f, gelu=not relu, relu=relu, aux=u)
        if fused:
            pass
        elif pdrop:                                                # :1128-1131
            tmp = ws.get('droptmp', N, D, dt, dev)
            m2 = ws.get(f'dropmask2_{tag}', N, D, torch.uint8, dev)
            ops.linear_fwd(f, w2, b2, tmp)
            ops.dropout_fwd(tmp, s2, pdrop, self._drop_seed(), mask=m2, residual=a1)
        else:
            ops.linear_fwd(f, w2, b2, s2, residual=a1)
        # store_out = False (with a fused head): the layer's output is fed to the head and NOT written -- the native training step never
        # reads it again (the LayerNorm backward recomputes it for the head's weight gradient), forward() only wants the scores
        if skip_ln2:
            # the caller takes s2 through the last LayerNorm, the head, the loss and their backward in ONE pass (ops.layernorm_loss_tail)
            hout = mean2 = rstd2 = None
        else:
            hout = ws.get(f'hout{tag}', N, D, dt, dev) if (store_out or head is None) else None
            mean2 = ws.get(f'mean2_{tag}', N, 1, torch.float32, dev)
            rstd2 = ws.get(f'rstd2_{tag}', N, 1, torch.float32, dev)
            ops.layernorm_fwd(s2, self._lt(pf, names, 'ln2w', 1, D).view(-1), self._lt(pf, names, 'ln2b', 1, D).view(-1), self.ln_eps,
                              hout, mean2, rstd2, head_w=head[0] if head else None, head_b=head[1] if head else None,
                              scores=head[2] if head else None)
        return dict(hin=h, qkv=qkv, ctx=ctx, probs=probs, s1=s1, a1=a1, mean1=mean1, rstd1=rstd1, u=u, f=f, s2=s2, hout=hout,
                    mean2=mean2, rstd2=rstd2, radius=radius, slots=slots, m1=m1, m2=m2, pattn=pattn, aseed=aseed, fused_ffn=fused)

    pair_ffn_wgrads = os.environ.get('MTS_PAIR_FFN_WGRADS', '1') == '1'

    def _band_layer_bwd(self, names, S, dh, lengths_i32, B, Lq, N, pdrop, row0, wgrad, tail_end, head=None, slot=0, head_grads=None, tail_done=False,
                        wgrad_direct=True):
        """Gradients of one layer into grad_flat; returns d(layer input).  dh: gradient wrt the layer output (None when only the
        fused head contributes); head = (dscores, head_w): the tagger head's data gradient is formed inside the LayerNorm backward.
        wgrad(dy, x, gview): the caller's weight-gradient launcher (may run on a side stream).  tail_end: end offset of the span
        [wo .. ) that is final once the attention-output weight gradient has been issued."""
        dt = self.compute_dtype
        D, F, H = self.embedding_dim, self._ffp, self.nheads
        ws, lay = self._ws, self._layout
        dev = S['s2'].device
        wf, pf = self._weights(), self._flat
        g = self.grad_flat()
        Gv = lambda key, rows, cols: self._lt(g, names, key, rows, cols)
        ds2 = ws.get('ds2', N, D, dt, dev)
        # head_grads = (dW_head, db_head): the head's parameter gradients come out of this pass too (last layer, dh is None)
        # tail_done: ops.layernorm_loss_tail has produced ds2 and the LayerNorm / head parameter gradients already (loss_and_grad)
        if not tail_done:
            ops.layernorm_bwd(S['s2'], dh, self._lt(pf, names, 'ln2w', 1, D).view(-1), S['mean2'], S['rstd2'], ds2,
                              Gv('ln2w', 1, D).view(-1), Gv('ln2b', 1, D).view(-1), dxsum=Gv('b2', 1, D).view(-1),
                              dlogit=head[0] if head else None, head_w=head[1] if head else None,
                              beta=self._lt(pf, names, 'ln2b', 1, D).view(-1) if head_grads else None,
                              dhead_w=head_grads[0] if head_grads else None, dhead_b=head_grads[1] if head_grads else None)
        # FFN down:  s2 = dropout(f W2^T + b2) + a1
        ds2d = ds2
        if pdrop:
            ds2d = ws.get('ds2d', N, D, dt, dev)               # gradient of the dense branch; the residual branch keeps ds2
            ops.dropout_bwd(ds2, ds2d, S['m2'], pdrop)
            ops.colsum(ds2d, Gv('b2', 1, D).view(-1))
        Np = round_up(N, 64)
        du = ws.get('du', Np, F, dt, dev)[:N]
        da1 = ws.get('da1', Np, D, dt, dev)[:N]
        if S.get('fused_ffn') and not pdrop:
            # du = (ds2 W2) * act'(u) and da1 = du W1 + ds2 in one launch (the intermediate stays in LDS); weight gradients as before
            ops.ffn_bwd_data(ds2, self._lt(wf, names, 'w1', F, D), self._lt(wf, names, 'w2', D, F), S['u'], du, da1, relu=self.ffn_act == 'relu')
            if self.pair_ffn_wgrads and wgrad_direct and ops.wgrad_pair_supported(du, S['a1']):
                # dW1 = du^T a1 and dW2 = ds2^T f (computed as f^T ds2, stored transposed) are two problems of ONE shape, 8 output tiles each:
                # one launch fills the chip, two fill half of it twice (63 -> ~40 us per step at the BASELINE shape)
                ops.colsum(du, Gv('b1', 1, F).view(-1))
                ops.wgrad_pair(du, S['a1'], Gv('w1', F, D), S['f'], ds2, Gv('w2', D, F))
            else:
                wgrad(ds2, S['f'], Gv('w2', D, F))
                ops.colsum(du, Gv('b1', 1, F).view(-1))
                wgrad(du, S['a1'], Gv('w1', F, D))
        else:
            wgrad(ds2d, S['f'], Gv('w2', D, F))
            ops.linear_dgrad(ds2d, self._lt(wf, names, 'w2', D, F), du)
            if self.ffn_act == 'relu':
                ops.relu_bwd(S['u'], du)
            else:
                ops.gelu_bwd(S['u'], du)
            ops.colsum(du, Gv('b1', 1, F).view(-1))
            # FFN up:  u = a1 W1^T + b1 ;  da1 = ds2 (residual) + du W1
            wgrad(du, S['a1'], Gv('w1', F, D))
            ops.linear_dgrad(du, self._lt(wf, names, 'w1', F, D), da1, residual=ds2)
        ds1 = ws.get('ds1', N, D, dt, dev)
        ops.layernorm_bwd(S['s1'], da1, self._lt(pf, names, 'ln1w', 1, D).view(-1), S['mean1'], S['rstd1'], ds1,
                          Gv('ln1w', 1, D).view(-1), Gv('ln1b', 1, D).view(-1), dxsum=Gv('bo', 1, D).view(-1))
        # attention output projection: s1 = dropout(ctx Wo^T + bo) + hin
        ds1d = ds1
        if pdrop:
            ds1d = ws.get('ds1d', N, D, dt, dev)
            ops.dropout_bwd(ds1, ds1d, S['m1'], pdrop)
            ops.colsum(ds1d, Gv('bo', 1, D).view(-1))
        wgrad(ds1d, S['ctx'], Gv('wo', D, D))
        # everything of this layer behind the q/k/v block (and the head, for the last layer) is final: let a
        # data-parallel trainer start reducing it while attention backward and the QKV GEMMs still run
        o_wo = lay.entries[names['wo'][0]][0]
        self._grads_ready(o_wo, tail_end)
        dctx = ws.get('dctx', N, D, dt, dev)
        ops.linear_dgrad(ds1d, self._lt(wf, names, 'wo', D, D), dctx)
        dqkv = ws.get('dqkv', N, 3 * D, dt, dev)
        dsc = ws.get('dsc', N, H * S['slots'], torch.float32, dev)
        ops.band_attn_bwd(S['qkv'], lengths_i32, S['probs'], dctx, B, Lq, D, H, S['radius'], dqkv, dsc,
                          dbias=Gv('bqkv', 1, 3 * D).view(-1), row0=row0, drop_p=S['pattn'], drop_seed=S['aseed'])
        o_qkv = lay.entries[names['wqkv'][0]][0]
        if self._grad_hook is not None and self.qkv_release == 'projection':
            # data parallel: one weight-gradient GEMM per projection, each third handed to the exchange as soon as it is final --
            # what is still in flight when the backward ends is the last 12.8 MB instead of the whole 38.5 MB q/k/v block.  Costs three
            # launches of 1792 x 1792 x rows (98 us each at 16384 rows) against one of 5376 x 1792 x rows (276 us): qkv_release = 'block'
            # (MTS_DP_QKV_RELEASE=block) keeps the one GEMM and announces the whole block behind it
            gq = Gv('wqkv', 3 * D, D)
            for i in range(3):
                wgrad(dqkv[:, i * D:(i + 1) * D], S['hin'], gq[i * D:(i + 1) * D])
                self._grads_ready(o_qkv + i * D * D, o_qkv + (i + 1) * D * D if i < 2 else o_wo)   # the last one carries the biases
        else:
            wgrad(dqkv, S['hin'], Gv('wqkv', 3 * D, D))
            self._grads_ready(o_qkv, o_wo)                         # q/k/v weights + biases
        dhin = ws.get(f'dhin{slot}', N, D, dt, dev)
        ops.linear_dgrad(dqkv, self._lt(wf, names, 'wqkv', 3 * D, D), dhin, residual=ds1)
        return dhin

    # ---- autograd bridge ---------------------------------------------------------------------------
    def _autograd_loss(self, run_fwd_bwd):
        """run_fwd_bwd() must run forward+backward natively, fill grad_flat and return the loss tensor (0-d)."""
        params = list(self._flat_params.values())
        return _NativeLoss.apply(self, run_fwd_bwd, *params)


class _NativeLoss(torch.autograd.Function):
    """One autograd node for the whole tagger: forward runs the native forward AND backward (activations never
    leave the workspace), backward hands the flat gradient slices to autograd scaled by the incoming grad."""

    @staticmethod
    def forward(ctx, model, run, *params):
        loss = run()
        ctx.model = model
        return loss.clone()

    @staticmethod
    def backward(ctx, gout):
        views = ctx.model.grad_views()
        grads = [(v * gout).clone() for v in views.values()]
        return (None, None, *grads)


# =====================================================================================================
# Restricted-window transformer tagger
# =====================================================================================================
class Transformer_segmenter(_TaggerBase):
    """models/CRF.py:508-610 with restricted=True: HF-Longformer-style local attention encoder + linear head."""
    grad_hooks_cover_all = True

    def __init__(self, tagset_size, embedding_dim, hidden_dim, num_layers=6, nheads=8, dropout_in=0.0, dropout_out=0.0,
                 batch_first=True, loss_fn='CrossEntropy', positional_encoding=True, threshold=None, restricted=True,
                 window_size=127, alpha=0.9, gamma=2, compute_dtype=None, max_position_embedding=4096, seed=None):
        super().__init__()
        self._init_common(loss_fn, threshold, alpha, gamma, compute_dtype)
        if not restricted:
            raise NotImplementedError('restricted=False (HF BertModel full attention, models/CRF.py:544) is outside the hot path')
        # dropout_in -> HF hidden_dropout_prob (embeddings, attention-output and FFN-output dense layers; training mode only),
        # dropout_out -> attention_probs_dropout_prob (RestrictedTransformerLayer.py:88-89)
        if not 0.0 <= float(dropout_in) < 1.0:
            raise ValueError(f'dropout probability has to be between 0 and 1, but got {dropout_in}')
        if not 0.0 <= float(dropout_out) < 1.0:
            raise ValueError(f'dropout probability has to be between 0 and 1, but got {dropout_out}')
        self.dropout_in, self.dropout_out = float(dropout_in), float(dropout_out)
        self._drop_calls = 0
        self.embedding_dim, self.hidden_dim, self.tagset_size = embedding_dim, hidden_dim, tagset_size
        self.nheads, self.num_layers = nheads, num_layers
        # pyramidal windows, models/CRF.py:529; every entry must be even, RestrictedTransformerLayer.py:77-80
        windows = [k * window_size for k in range(num_layers, 0, -1)]
        assert all(w % 2 == 0 for w in windows), 'All window sizes must be divisible by 2!'
        self.radii = [w // 2 for w in windows]            # one-sided radius, modeling_longformer.py:478
        if embedding_dim % nheads != 0:
            raise ValueError(f'The hidden size ({embedding_dim}) is not a multiple of the number of attention heads ({nheads})')
        self.n_out = tagset_size if loss_fn == 'CrossEntropy' else 1
        self.ln_eps = 1e-12                                # HF default; the wrapper's layer_norm_eps is ignored
        self.max_pos = max_position_embedding
        D, F = embedding_dim, hidden_dim

        gen = torch.Generator().manual_seed(torch.initial_seed() if seed is None else seed)
        std = 0.02                                         # HF initializer_range
        groups, init, pads = [], {}, {}
        Fp = self._ffp = round_up(F, 8)                    # FFN width as stored (the reference's default is 25): flat.py "padded storage"

        def add(group, name, shape, value, pad=None, storage=None):
            group.append((name, storage or shape))
            init[name] = value
            if pad and storage != shape:
                pads[name] = pad

        e = 'model.model.embeddings.'
        g = []
        pos = torch.randn(self.max_pos, D, generator=gen) * std
        pos[1].zero_()                                     # padding_idx = 1
        add(g, e + 'position_embeddings.weight', (self.max_pos, D), pos)
        gpos = g
        g = []
        add(g, e + 'token_type_embeddings.weight', (2, D), torch.randn(2, D, generator=gen) * std)
        add(g, e + 'LayerNorm.weight', (D,), torch.ones(D))
        add(g, e + 'LayerNorm.bias', (D,), torch.zeros(D))
        # token-type rows and the embedding LayerNorm sit IN FRONT of the position table: everything the embedding block's backward touches --
        # they and position rows [0, L + 2) -- is then ONE span of the flat gradient, announced (and exchanged between ranks) as one message at the
        # very end of the backward, where every collective is exposed (two messages cost a one-rank RCCL group 40 us of stream hand-offs per step)
        groups.append(g)
        groups.append(gpos)
        for li in range(num_layers):
            lp = f'model.model.encoder.layer.{li}.'
            g = []
            for n in ('query', 'key', 'value'):            # contiguous -> one [3D, D] GEMM operand
                add(g, lp + f'attention.self.{n}.weight', (D, D), torch.randn(D, D, generator=gen) * std)
            groups.append(g)
            g = []
            for n in ('query', 'key', 'value'):
                add(g, lp + f'attention.self.{n}.bias', (D,), torch.zeros(D))
            groups.append(g)
            for name, shape, val in (
                    ('attention.output.dense.weight', (D, D), torch.randn(D, D, generator=gen) * std),
                    ('attention.output.dense.bias', (D,), torch.zeros(D)),
                    ('attention.output.LayerNorm.weight', (D,), torch.ones(D)),
                    ('attention.output.LayerNorm.bias', (D,), torch.zeros(D)),
                    ('intermediate.dense.weight', (F, D), torch.randn(F, D, generator=gen) * std),
                    ('intermediate.dense.bias', (F,), torch.zeros(F)),
                    ('output.dense.weight', (D, F), torch.randn(D, F, generator=gen) * std),
                    ('output.dense.bias', (D,), torch.zeros(D)),
                    ('output.LayerNorm.weight', (D,), torch.ones(D)),
                    ('output.LayerNorm.bias', (D,), torch.zeros(D))):
                g = []
                if name == 'intermediate.dense.weight':
                    add(g, lp + name, shape, val, pad=[(0, 1, F, Fp)], storage=(Fp, D))
                elif name == 'intermediate.dense.bias':
                    add(g, lp + name, shape, val, pad=[(0, 1, F, Fp)], storage=(Fp,))
                elif name == 'output.dense.weight':
                    add(g, lp + name, shape, val, pad=[(1, 1, F, Fp)], storage=(D, Fp))
                else:
                    add(g, lp + name, shape, val)
                groups.append(g)
        cw, cb = _linear_init(self.n_out, D, gen)
        g = []
        add(g, 'classification.weight', (self.n_out, D), cw)
        add(g, 'classification.bias', (self.n_out,), cb)
        groups.append(g)
        self._init_flat(FlatLayout(groups, pads), init)
        self._register_load_state_dict_pre_hook(self._drop_dead_keys)

    @staticmethod
    def _drop_dead_keys(state_dict, prefix, *args):
        """A reference checkpoint carries ~68 M parameters the local-attention path never touches (HF word
        embeddings, global-attention projections, pooler): ignore them on load."""
        for k in [k for k in state_dict if k.startswith(prefix) and any(d in k for d in DEAD_HF_KEYS)]:
            del state_dict[k]

    def _layer_names(self, li):
        lp = f'model.model.encoder.layer.{li}.'
        a_, o_ = lp + 'attention.self.', lp + 'attention.output.'
        one = lambda n: (n, n)
        return {'wqkv': (a_ + 'query.weight', a_ + 'value.weight'), 'bqkv': (a_ + 'query.bias', a_ + 'value.bias'),
                'wo': one(o_ + 'dense.weight'), 'bo': one(o_ + 'dense.bias'), 'ln1w': one(o_ + 'LayerNorm.weight'),
                'ln1b': one(o_ + 'LayerNorm.bias'), 'w1': one(lp + 'intermediate.dense.weight'), 'b1': one(lp + 'intermediate.dense.bias'),
                'w2': one(lp + 'output.dense.weight'), 'b2': one(lp + 'output.dense.bias'), 'ln2w': one(lp + 'output.LayerNorm.weight'),
                'ln2b': one(lp + 'output.LayerNorm.bias')}

    # ---- native forward / backward ------------------------------------------------------------------
    def _forward_native(self, xs, lengths_i32, want_grad_state=True, pack=None, need_hidden=True, tail=False):
        """pack = {'row_src', 'row0', 'n'} (see _pack_plan): activations hold only the valid sentences.
        need_hidden = False: the last layer's output goes to the fused head only and is not written to memory."""
        xs, xs2, B, Lq, D = self._split_input(xs)
        dt, dev = self.compute_dtype, xs.device
        if D != self.embedding_dim:
            raise ValueError(f'expected input dim {self.embedding_dim}, got {D}')
        if Lq + 2 > self.max_pos:
            raise ValueError(f'sequence length {Lq} exceeds max_position_embeddings-2 = {self.max_pos - 2}')
        N, F, H = (pack['n'] if pack else B * Lq), self._ffp, self.nheads
        row_src, row0 = (pack['row_src'], pack['row0']) if pack else (None, None)
        ws, lay = self._ws, self._layout
        wf = self._weights()                  # compute-dtype mirror (GEMM operands)
        pf = self._flat                       # fp32 masters (biases, LayerNorm, embeddings, head)
        # a batch that crossed PCIe in bf16 (prefetch.DevicePrefetcher / AudioPortionDataset(wire_dtype='bf16')) is read as it is: the fp32 copy the
        # kernel used to be handed cost a cast launch and 117 MB of writes + reads per step
        keep16 = xs.dtype == torch.bfloat16 and dt == torch.bfloat16 and xs2 is None and D % 4 == 0
        x = xs.contiguous() if keep16 else xs.contiguous().to(torch.float32)
        x2 = xs2.contiguous().to(torch.float32) if xs2 is not None else None
        e = 'model.model.embeddings.'
        st = {'B': B, 'L': Lq, 'N': N, 'lengths': lengths_i32, 'layers': [], 'pack': pack}
        h = ws.get('h0', N, D, dt, dev)
        pre0 = ws.get('pre0', N, D, dt, dev)
        mean0 = ws.get('mean0', N, 1, torch.float32, dev)
        rstd0 = ws.get('rstd0', N, 1, torch.float32, dev)
        ops.embed_layernorm_fwd(x, lay.view(pf, e + 'position_embeddings.weight'), 2,
                                lay.view(pf, e + 'token_type_embeddings.weight')[0], lay.view(pf, e + 'LayerNorm.weight'),
                                lay.view(pf, e + 'LayerNorm.bias'), self.ln_eps, h, pre0, mean0, rstd0, row_src=row_src, x2=x2)
        st.update(pre0=pre0, mean0=mean0, rstd0=rstd0)
        pdrop = self.dropout_in if self.training else 0.0          # nn.Dropout: training mode only
        st['pdrop'] = pdrop
        if pdrop:
            st['m0'] = ws.get('dropmask0', N, D, torch.uint8, dev)
            ops.dropout_fwd(h, h, pdrop, self._drop_seed(), mask=st['m0'])          # modeling_longformer.py:424
        scores = ws.get('scores', N, self.n_out, torch.float32, dev)
        for li, radius in enumerate(self.radii):
            last = li == len(self.radii) - 1
            head = (self._w(pf, 'classification.weight'), self._w(pf, 'classification.bias'), scores) if (last and not tail) else None
            S = self._band_layer_fwd(self._layer_names(li), str(li), h, lengths_i32, B, Lq, N, radius, row0, pdrop,
                                     self.dropout_out if self.training else 0.0, head,
                                     # (a head wider than two outputs takes its parameter gradients from the stored output: _backward_native)
                                     store_out=need_hidden or not last or self.n_out > 2, skip_ln2=last and tail)
            st['layers'].append(S)
            h = S['hout']
        st['scores'] = scores if pack else scores.view(B, Lq, self.n_out)
        st['hidden'] = h
        return st

    def _backward_native(self, st, dscores, tail_done=False):
        """Fill grad_flat from the saved forward state; dscores fp32 [N, n_out]."""
        dt = self.compute_dtype
        B, Lq, N = st['B'], st['L'], st['N']
        D, F, H = self.embedding_dim, self._ffp, self.nheads
        dev = self._flat.device
        ws, lay = self._ws, self._layout
        wf, pf = self._weights(), self._flat
        g = self.grad_flat()
        G = lambda name: lay.view(g, name)
        nl = len(self.radii)
        dh = None
        # Weight-gradient GEMMs (MFMA-bound, needed by nobody until the optimizer) go to a side stream so that they overlap the
        # HBM-bound kernels of the data-gradient chain (LayerNorm backward, band attention backward, GELU, reductions).  Off when
        # a data-parallel hook is installed: the hook hands spans to RCCL in main-stream order.
        side = self._side_stream(dev) if (self.overlap_wgrad and self._grad_hook is None and dev.type == 'cuda') else None
        main = torch.cuda.current_stream(dev) if side is not None else None

        def wgrad(dy, x, gview):
            if side is None:
                ops.linear_wgrad(dy, x, gview)
                return
            ev = torch.cuda.Event()
            ev.record(main)                    # dy and x are complete on the main stream
            side.wait_event(ev)
            with torch.cuda.stream(side):
                ops.linear_wgrad(dy, x, gview)

        for li in range(nl - 1, -1, -1):
            last = li == nl - 1
            if side is not None:
                main.wait_stream(side)         # the previous layer's weight gradients are done with ds2 / du / ds1 / dqkv
            # the head's parameter gradients: from the last layer's LayerNorm backward (no pass over the stored output, which the
            # training forward does not even write); a head wider than two outputs keeps its own kernel
            fuse_hg = last and self.n_out <= 2 and not tail_done
            if last and not fuse_hg and not tail_done:
                ops.head_bwd_params(st['layers'][li]['hout'], dscores, G('classification.weight'), G('classification.bias'))
            tail_end = lay.entries[f'model.model.encoder.layer.{li + 1}.attention.self.query.weight'][0] if li + 1 < nl else g.numel()
            dh = self._band_layer_bwd(self._layer_names(li), st['layers'][li], dh, st['lengths'], B, Lq, N, st['pdrop'],
                                      st['pack']['row0'] if st['pack'] else None, wgrad, tail_end,
                                      head=(dscores, self._w(pf, 'classification.weight')) if (last and not tail_done) else None, slot=li & 1,
                                      head_grads=(G('classification.weight'), G('classification.bias')) if fuse_hg else None,
                                      tail_done=tail_done and last, wgrad_direct=side is None)
        e = 'model.model.embeddings.'
        if st['pdrop']:
            ops.dropout_bwd(dh, dh, st['m0'], st['pdrop'])         # through the dropout behind the embedding LayerNorm
        pos_g, type_g = G(e + 'position_embeddings.weight'), G(e + 'token_type_embeddings.weight')
        if D <= 2048:
            # one pass: LayerNorm backward + the sums that are all anybody wants of its dx -- over the documents of each position into
            # rows [2, L + 2) of the position table, over all rows into token-type row 0 (both OVERWRITTEN; type row 1 is never touched
            # and keeps the zero it was allocated with).  A longer batch earlier may have left rows beyond L + 2: cleared once.
            prev = getattr(self, '_pos_touched', 0)
            if prev > Lq + 2:
                pos_g[Lq + 2:prev].zero_()
            self._pos_touched = Lq + 2
            ops.embed_layernorm_bwd(st['pre0'], dh, self._w(pf, e + 'LayerNorm.weight'), st['mean0'], st['rstd0'], B, Lq,
                                    G(e + 'LayerNorm.weight'), G(e + 'LayerNorm.bias'), type_g[0], pos_g, 2,
                                    row0=st['pack']['row0'] if st['pack'] else None, lengths=st['lengths'])
        else:
            dpre = ws.get('ds2', N, D, dt, dev)
            type_g.zero_()
            # d(type row 0) = sum over all rows of dpre = the LayerNorm backward's column sum of dx
            ops.layernorm_bwd(st['pre0'], dh, self._w(pf, e + 'LayerNorm.weight'), st['mean0'], st['rstd0'], dpre,
                              G(e + 'LayerNorm.weight'), G(e + 'LayerNorm.bias'), dxsum=type_g[0])
            # only rows [2, L+2) of the position table ever receive a gradient: keep the rest of its gradient at the zeros it was
            # allocated with and clear what the longest batch so far could have touched
            self._pos_touched = max(getattr(self, '_pos_touched', 0), Lq + 2)
            pos_g[:self._pos_touched].zero_()
            ops.embed_bwd(dpre, B, Lq, pos_g, 2, row0=st['pack']['row0'] if st['pack'] else None, lengths=st['lengths'])
        # embeddings: only the position rows a batch of this length can touch, then type row + LayerNorm
        D_ = self.embedding_dim
        if side is not None:
            main.wait_stream(side)             # every weight gradient has landed before the optimizer (or anyone else) looks
        p0 = lay.entries[e + 'position_embeddings.weight'][0]
        t0 = lay.entries[e + 'token_type_embeddings.weight'][0]
        assert t0 < p0, 'flat layout: token-type rows and embedding LayerNorm in front of the position table'
        self._grads_ready(t0, p0 + (Lq + 2) * D_)      # (position rows 0 and 1 ride along: never touched, gradient 0)

    def _drop_seed(self):
        self._drop_calls += 1
        return (torch.initial_seed() * 1000003 + self._drop_calls * 7919) & 0x7FFFFFFFFFFFFFFF

    overlap_wgrad = os.environ.get('MTS_OVERLAP_WGRAD', '0') == '1'    # True: +2 % step throughput at the BASELINE shape (2.70 -> 2.64 ms in round 1), but co-running kernels stretch each
                             # other, so per-kernel timings (bench.py's roofline) stop describing a kernel; off by default

    def _side_stream(self, dev):
        return ops.side_stream(dev, 0)

    # ---- packed batches ---------------------------------------------------------------------------------
    pack_rows = 'auto'     # training path: 'auto' packs when >= 10 % of the B*L rows are padding; True / False force it

    def _pack_plan(self, lengths, B, Lq, dev):
        """The reference pads every Transformer batch to 3600 sentences (train_fit.py:104-106; real documents: median 359)
        and pushes all of them through the encoder.  Padded rows never reach a valid row (masked as keys) nor the loss, so
        the training path keeps only the valid sentences, document after document: row_src[r] = b*L + i, row0[b] = first
        packed row of document b.  Same loss and gradients up to fp32 summation order."""
        if self.pack_rows is False or lengths is None:
            return None
        lens = [max(0, min(int(v), Lq)) for v in (lengths.tolist() if hasattr(lengths, 'tolist') else lengths)]
        n = sum(lens)
        if n == 0 or (self.pack_rows == 'auto' and n > 0.9 * B * Lq):
            return None
        key = (tuple(lens), Lq, str(dev))
        if getattr(self, '_pack_key', None) != key:
            import numpy as np
            row0 = np.zeros(B, dtype=np.int32)
            row0[1:] = np.cumsum(lens[:-1], dtype=np.int64)
            row_src = np.concatenate([b * Lq + np.arange(n_b, dtype=np.int32) for b, n_b in enumerate(lens)]).astype(np.int32)
            self._pack_val = {'row_src': torch.from_numpy(row_src).to(dev), 'row0': torch.from_numpy(row0).to(dev), 'n': n}
            self._pack_key = key
        return self._pack_val

    # ---- public API ------------------------------------------------------------------------------------
    def loss_and_grad(self, xs, lengths, tags, want_grad=True):
        """Native forward(+backward): returns (loss 0-d fp32 tensor, scores [B,L,n_out] -- or [n_valid, n_out] when the batch
        was packed, see _pack_plan); gradients land in grad_flat."""
        L.require_gpu()
        x1, _, B, Lq, _ = self._split_input(xs)
        dev = x1.device
        li32 = self._prep_lengths(lengths, B, Lq, dev)
        pack = self._pack_plan(lengths, B, Lq, dev)
        tg = tags.to(device=dev, dtype=torch.float32).contiguous()
        loss_out = torch.empty(2, dtype=torch.float32, device=dev)
        D = self.embedding_dim
        if (want_grad and self.fuse_tail and self.n_out == (2 if self.loss_kind == L.LOSS_CE else 1) and tg.shape[1] >= Lq
                and ops.loss_tail_supported(self.compute_dtype, D, self.n_out)):
            # the last layer's LayerNorm, the head, the loss and their backward in ONE pass over the layer's pre-LayerNorm sum (mts_layernorm_loss_tail)
            st = self._forward_native(xs, li32, pack=pack, need_hidden=False, tail=True)
            nm, pf, g, lay = self._layer_names(len(self.radii) - 1), self._flat, self.grad_flat(), self._layout
            N = st['N']
            scores = self._ws.get('scores', N, self.n_out, torch.float32, dev)
            ds2 = self._ws.get('ds2', N, D, self.compute_dtype, dev)
            ops.layernorm_loss_tail(self.loss_kind, st['layers'][-1]['s2'], self._lt(pf, nm, 'ln2w', 1, D).view(-1), self._lt(pf, nm, 'ln2b', 1, D).view(-1),
                                    self.ln_eps, self._w(pf, 'classification.weight'), self._w(pf, 'classification.bias'), tg, li32, self.alpha,
                                    self.gamma, self.loss_grad_scale, scores, loss_out, ds2, self._lt(g, nm, 'ln2w', 1, D).view(-1),
                                    self._lt(g, nm, 'ln2b', 1, D).view(-1), self._lt(g, nm, 'b2', 1, D).view(-1), lay.view(g, 'classification.weight'),
                                    lay.view(g, 'classification.bias'), (B, Lq), row_src=pack['row_src'] if pack else None)
            st['scores'] = scores if pack else scores.view(B, Lq, self.n_out)
            self._backward_native(st, None, tail_done=True)
            return loss_out[0], st['scores']
        st = self._forward_native(xs, li32, pack=pack, need_hidden=False)
        dsc = self._ws.get('dscores', st['N'], self.n_out, torch.float32, dev) if want_grad else None
        ops.tagger_loss(self.loss_kind, st['scores'], tg, li32, self.alpha, self.gamma, loss_out, dsc,
                        row_src=pack['row_src'] if pack else None, batch_shape=(B, Lq))
        if want_grad:
            ops.scale_(dsc, self.loss_grad_scale)
            self._backward_native(st, dsc)
        return loss_out[0], st['scores']

    def loss(self, xs, lengths, tags):
        """models/CRF.py:574-595."""
        if torch.is_grad_enabled() and any(p.requires_grad for p in self._flat_params.values()):
            return self._autograd_loss(lambda: self.loss_and_grad(xs, lengths, tags, True)[0])
        return self.loss_and_grad(xs, lengths, tags, False)[0].clone()

    def encode(self, xs, lengths):
        """Encoder output [B, L, D] (``Longformer_Local_Attention.forward``, RestrictedTransformerLayer.py:118-133)."""
        x1, _, B, Lq, _ = self._split_input(xs)
        st = self._forward_native(xs, self._prep_lengths(lengths, B, Lq, x1.device))
        return st['hidden'].view(B, Lq, -1).to(torch.float32)

    # Inference at one document per call is launch-bound (a dozen kernels of a few microseconds each): with inference_graphs = True
    # the forward + decode of a given (B, L) shape is captured once as a hipGraph (torch.cuda.CUDAGraph: the kernels are enqueued
    # on torch's current stream, which is the capturing stream inside the capture) and replayed on static input / output buffers.
    # Same kernels, same order, same buffers: bitwise the eager results.  Eval mode, single-tensor input only; anything else, and
    # every shape after the 16th, takes the eager path.
    inference_graphs = False

    def _graph_entry(self, xs, li32, threshold):
        th = float(self.th if self.th is not None else threshold)
        key = (tuple(xs.shape), xs.dtype, th, self._ws.generation, self._flat.data_ptr())
        ent = self._graphs.get(key) if hasattr(self, '_graphs') else None
        if ent is not None:
            return ent
        if not hasattr(self, '_graphs'):
            self._graphs = {}
        if len(self._graphs) >= 16:
            return None
        B, Lq = xs.shape[0], xs.shape[1]
        dev = xs.device
        xs_s, li_s = torch.empty_like(xs), torch.empty_like(li32)
        tags = torch.empty(B, Lq, dtype=torch.uint8, device=dev)
        xs_s.copy_(xs)
        li_s.copy_(li32)

        def run():
            st = self._forward_native(xs_s, li_s, need_hidden=False)
            ops.greedy_decode(st['scores'], li_s, th, tags)
            return st['scores']
        self._weights()                                   # the bf16 mirror is refreshed OUTSIDE the graph (a cast inside would replay forever)
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(dev)
        side.wait_stream(cur)
        with torch.cuda.stream(side):                     # warm-up: buffer allocation, first-launch attribute calls
            run()
        cur.wait_stream(side)
        key = (tuple(xs.shape), xs.dtype, th, self._ws.generation, self._flat.data_ptr())       # (the warm-up may have grown the workspace)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            scores = run()
        ent = self._graphs[key] = (g, xs_s, li_s, scores, tags)
        return ent

    def forward(self, xs, lenghts, threshold=0.4):
        """models/CRF.py:597-610 -> (scores [B,L,n_out], list of per-document bool lists)."""
        L.require_gpu()
        x1, x2, B, Lq, _ = self._split_input(xs)
        li32 = self._prep_lengths(lenghts, B, Lq, x1.device)
        with torch.no_grad():
            ent = None
            if self.inference_graphs and not self.training and x2 is None and torch.is_tensor(xs) and xs.dtype == torch.float32 and xs.is_contiguous():
                ent = self._graph_entry(xs, li32, threshold)
            if ent is not None:
                g, xs_s, li_s, scores_s, tags_s = ent
                self._weights()
                xs_s.copy_(xs)
                li_s.copy_(li32)
                g.replay()
                scores = scores_s.clone()
                tags_h = tags_s.cpu().numpy().astype(bool)
                L.check_async()
                lens = [int(v) for v in (lenghts.tolist() if lenghts is not None else [Lq] * B)]
                return scores, [tags_h[i, :lens[i]].tolist() for i in range(B)]
            st = self._forward_native(xs, li32, need_hidden=False)
            scores = st['scores'].clone()
            tags = self._decode(scores, li32, lenghts, threshold)
        return scores, tags


# =====================================================================================================
# Legacy restricted-window encoder layer
# =====================================================================================================
class _LegacyLayerFn(torch.autograd.Function):
    """One autograd node for the whole layer (same bridge as _NativeLoss): forward keeps the saved state in the model's
    workspace, backward runs the native layer backward and hands out dx and the flat gradient slices."""

    @staticmethod
    def forward(ctx, model, x, *params):
        y, st = model._run_forward(x)
        ctx.model, ctx.st, ctx.shape = model, st, tuple(x.shape)
        return y

    @staticmethod
    def backward(ctx, gy):
        m = ctx.model
        dx = m._run_backward(ctx.st, gy)
        grads = [v.clone() for v in m.grad_views().values()]
        return (None, dx.view(ctx.shape), *grads)


class RestrictedTransformerEncoderLayer(_TaggerBase):
    """models/RestrictedTransformerLayer.py:269-310 (layer) + :413-644 (RestrictedMultiheadAttention): the reference's own,
    self-contained statement of the band attention -- for every position i a full multi-head attention over the slice
    [i - window_size, i + window_size], keeping row i (:509-636; L Python iterations) -- inside a post-LN encoder layer with a
    ReLU feed-forward, LayerNorm eps 1e-5 and a packed in_proj [3d, d].  No padding mask exists on this path
    (key_padding_mask is forced to None, :467).  Here: the same HIP kernels as Transformer_segmenter (lengths = NULL), one
    launch per stage instead of L attention calls.  ``state_dict`` keys are the reference's
    (self_attn.in_proj_weight, self_attn.out_proj.weight, linear1/2, norm1/2).  Post-LN only: norm_first=True and
    batch_first=False raise NotImplementedError; src_mask / src_key_padding_mask must be None (the reference ignores the
    latter and adds the former to every window)."""
    ffn_act = 'relu'

    def __init__(self, d_model, nhead, dim_feedforward=2048, window_size=None, dropout=0.1, activation='relu', layer_norm_eps=1e-5,
                 batch_first=False, norm_first=False, device=None, dtype=None, compute_dtype=None, seed=None):
        super().__init__()
        self._init_common('BinaryCrossEntropy', None, 0.9, 2, compute_dtype)
        if norm_first:
            raise NotImplementedError('norm_first=True (pre-LN variant, RestrictedTransformerLayer.py:291-296) is not built')
        if not batch_first:
            raise NotImplementedError('batch_first=False: every caller in the reference passes batch_first=True')
        act = getattr(activation, '__name__', activation)
        if act not in ('relu', 'gelu'):
            raise RuntimeError(f'activation should be relu/gelu, not {activation}')      # _get_activation_fn
        if window_size is None or int(window_size) < 0:
            raise ValueError('window_size (one-sided radius of the attention band) is required')
        if d_model % nhead != 0:
            raise AssertionError('embed_dim must be divisible by num_heads')               # RestrictedMultiheadAttention.__init__
        if not 0.0 <= float(dropout) < 1.0:
            raise ValueError(f'dropout probability has to be between 0 and 1, but got {dropout}')
        self.ffn_act = act
        self.embedding_dim, self.nheads, self.hidden_dim = d_model, nhead, dim_feedforward
        self.radius = int(window_size)
        self.ln_eps = float(layer_norm_eps)
        self.dropout_p = float(dropout)
        self._drop_calls = 0
        D, F = d_model, dim_feedforward
        Fp = self._ffp = round_up(F, 8)
        gen = torch.Generator().manual_seed(torch.initial_seed() if seed is None else seed)
        w1, b1 = _linear_init(F, D, gen)
        w2, b2 = _linear_init(D, F, gen)
        wo, _ = _linear_init(D, D, gen)
        groups, init, pads = [], {}, {}
        for name, shape, val, pad, storage in (
                ('self_attn.in_proj_weight', (3 * D, D), _xavier_uniform((3 * D, D), gen), None, None),       # xavier_uniform_, :388
                ('self_attn.in_proj_bias', (3 * D,), torch.zeros(3 * D), None, None),
                ('self_attn.out_proj.weight', (D, D), wo, None, None),
                ('self_attn.out_proj.bias', (D,), torch.zeros(D), None, None),
                ('norm1.weight', (D,), torch.ones(D), None, None), ('norm1.bias', (D,), torch.zeros(D), None, None),
                ('linear1.weight', (F, D), w1, [(0, 1, F, Fp)], (Fp, D)), ('linear1.bias', (F,), b1, [(0, 1, F, Fp)], (Fp,)),
                ('linear2.weight', (D, F), w2, [(1, 1, F, Fp)], (D, Fp)), ('linear2.bias', (D,), b2, None, None),
                ('norm2.weight', (D,), torch.ones(D), None, None), ('norm2.bias', (D,), torch.zeros(D), None, None)):
            groups.append([(name, storage or shape)])
            init[name] = val
            if pad and storage != shape:
                pads[name] = pad
        self._init_flat(FlatLayout(groups, pads), init)
        one = lambda n: (n, n)
        self._names = {'wqkv': one('self_attn.in_proj_weight'), 'bqkv': one('self_attn.in_proj_bias'),
                       'wo': one('self_attn.out_proj.weight'), 'bo': one('self_attn.out_proj.bias'), 'ln1w': one('norm1.weight'),
                       'ln1b': one('norm1.bias'), 'w1': one('linear1.weight'), 'b1': one('linear1.bias'), 'w2': one('linear2.weight'),
                       'b2': one('linear2.bias'), 'ln2w': one('norm2.weight'), 'ln2b': one('norm2.bias')}

    def _drop_seed(self):
        self._drop_calls += 1
        return (torch.initial_seed() * 1000003 + self._drop_calls * 7919) & 0x7FFFFFFFFFFFFFFF

    def _run_forward(self, src):
        B, Lq, D = src.shape
        if D != self.embedding_dim:
            raise ValueError(f'expected input dim {self.embedding_dim}, got {D}')
        dt, dev = self.compute_dtype, src.device
        N = B * Lq
        x = self._ws.get('x', N, D, dt, dev)
        x.copy_(src.reshape(N, D))
        pdrop = self.dropout_p if self.training else 0.0          # nn.Dropout / F.dropout(training=self.training): training mode only
        S = self._band_layer_fwd(self._names, 'L', x, None, B, Lq, N, self.radius, None, pdrop, pdrop)
        return S['hout'].view(B, Lq, D).to(torch.float32), dict(S=S, B=B, L=Lq, N=N, pdrop=pdrop)

    def _run_backward(self, st, gy):
        dt = self.compute_dtype
        N, D = st['N'], self.embedding_dim
        dh = self._ws.get('dy', N, D, dt, gy.device)
        dh.copy_(gy.reshape(N, D))
        wgrad = lambda dy, x, gview: ops.linear_wgrad(dy, x, gview)
        dx = self._band_layer_bwd(self._names, st['S'], dh, None, st['B'], st['L'], N, st['pdrop'], None, wgrad, self.grad_flat().numel())
        return dx.to(torch.float32)

    def forward(self, src, src_mask=None, src_key_padding_mask=None):
        """-> [B, L, d_model] fp32.  RestrictedTransformerLayer.py:277-299."""
        L.require_gpu()
        if src_mask is not None or src_key_padding_mask is not None:
            raise NotImplementedError('src_mask / src_key_padding_mask: the reference drops the padding mask on this path (:467) and '
                                      'no caller passes an attention mask')
        if src.dim() != 3:
            raise ValueError('src must be [batch, sequence, d_model] (batch_first=True)')
        if torch.is_grad_enabled() and (src.requires_grad or any(p.requires_grad for p in self._flat_params.values())):
            return _LegacyLayerFn.apply(self, src, *self._flat_params.values())
        with torch.no_grad():
            return self._run_forward(src)[0]
