"""CPU restatement of the reference's sentence-tagger hot path.  TEST INFRASTRUCTURE ONLY.

This module is the *oracle*: a plain, explicit torch-CPU (fp32 or fp64) restatement of the
arithmetic of Ighina/MultimodalTopicSegmentation's tagger path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it; the product
package (``multimodaltopicsegmentation_amd``) never does and fails loudly without its HIP
library.

Parity pinning: the reference ships no tests/golden vectors, so the pins are the
fixtures in ``tests/golden/*.npz`` produced by ``tests/golden/make_golden.py`` by importing the
reference itself (CPU, stub modules for ``pytorch_lightning``/``segeval``/``models.longformer_noffn``)
and recording its outputs; ``tests/test_oracle_vs_golden.py`` checks every function here against them.

Every function cites the reference lines it restates (paths relative to the reference root; ``HF:`` =
``transformers/models/longformer/modeling_longformer.py``, the third-party module the reference's live
restricted-attention path calls; pinned ``transformers==4.24.0`` in requirements.txt:17, fixtures were
generated against 5.15.0, see SURVEY.md §8c for the version-skew evidence).

Parameters are passed as dicts keyed by the reference's own ``state_dict`` names so that a
checkpoint of the reference can be fed in unchanged.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch

Tensor = torch.Tensor
IMPOSSIBLE = -1e4  # models/CRF.py:95


# --------------------------------------------------------------------------------------------
# small pieces
# --------------------------------------------------------------------------------------------
def create_mask(max_len: int, lengths: Tensor) -> Tensor:
    """True = valid position.  models/NeuralArchitectures.py:11-21 (``(i+1) > len -> False``)."""
    pos = torch.arange(max_len).unsqueeze(0)
    return pos < lengths.view(-1, 1).to(pos.dtype)


def layer_norm(x: Tensor, gamma: Tensor, beta: Tensor, eps: float) -> Tensor:
    """Biased-variance LayerNorm over the last dim (torch.nn.LayerNorm semantics; HF:396,1065,1124)."""
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * gamma + beta


def gelu_erf(x: Tensor) -> Tensor:
    """Exact (erf) GELU = HF ACT2FN['gelu'] used by LongformerIntermediate (HF:1104-1117)."""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def sigmoid_focal_loss(logits: Tensor, targets: Tensor, alpha: float = 0.9, gamma: float = 2.0) -> Tensor:
    """models/focal_loss.py:38-57 with reduction='mean'."""
    p = 1.0 / (1.0 + torch.exp(-logits))
    # BCE-with-logits = (1-y)*x + softplus(-x) = (1-y)*x - logsigmoid(x)  (smooth at x == 0, unlike the
    # max/abs form, so autograd returns the analytic sigmoid(x)-y there too)
    ce = (1 - targets) * logits - torch.nn.functional.logsigmoid(logits)
    p_t = p * targets + (1 - p) * (1 - targets)
    loss = ce * (1 - p_t) ** gamma
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    return loss.mean()


def bce_on_probs(logits: Tensor, targets: Tensor) -> Tensor:
    """nn.BCELoss(sigmoid(x), y) as used at models/CRF.py:301-304,345-352 (log clamped at -100)."""
    p = 1.0 / (1.0 + torch.exp(-logits))
    logp = torch.clamp(torch.log(p), min=-100.0)
    log1mp = torch.clamp(torch.log(1 - p), min=-100.0)
    return (-(targets * logp + (1 - targets) * log1mp)).mean()


def cross_entropy_ignore(logits2: Tensor, targets: Tensor, ignore_index: int = -1) -> Tensor:
    """nn.CrossEntropyLoss(ignore_index=-1) over every row (models/CRF.py:298,354)."""
    t = targets.reshape(-1).long()
    x = logits2.reshape(-1, logits2.shape[-1])
    valid = t != ignore_index
    lse = torch.logsumexp(x, dim=-1)
    picked = x.gather(1, t.clamp(min=0).unsqueeze(1)).squeeze(1)
    return ((lse - picked) * valid).sum() / valid.sum()


def tagger_loss(scores: Tensor, lengths: Tensor, tags: Tensor, loss_fn: str,
                alpha: float = 0.9, gamma: float = 2.0) -> Tensor:
    """Loss tail shared by BiLSTM / BiLSTMLateFusion / Transformer_segmenter.

    models/CRF.py:340-356, :445-461, :579-595: BCE/Focal use the concatenation of the un-padded
    rows of every document; CrossEntropy uses all rows with ignore_index=-1.
    """
    if loss_fn == 'CrossEntropy':
        L = scores.shape[1]
        return cross_entropy_ignore(scores, tags[:, :L])
    xs, ys = [], []
    for b in range(scores.shape[0]):
        n = int(lengths[b])
        xs.append(scores[b, :n, 0])
        ys.append(tags[b, :n])
    x = torch.cat(xs)
    y = torch.cat(ys).to(x.dtype)
    if loss_fn == 'FocalLoss':
        return sigmoid_focal_loss(x, y, alpha, gamma)
    if loss_fn == 'BinaryCrossEntropy':
        return bce_on_probs(x, y)
    raise ValueError('Choose one of CrossEntropy or BinaryCrossEntropy as loss function')  # CRF.py:312


def greedy_decode(scores: Tensor, lengths: Tensor, th: Optional[float], bce: bool,
                  threshold: float = 0.4) -> List[List[bool]]:
    """models/CRF.py:358-369: strict ``>`` on sigmoid (1 logit) or softmax[...,1] (2 logits)."""
    if th is not None:
        threshold = th
    if bce:
        prob = (1.0 / (1.0 + torch.exp(-scores)))[:, :, 0]
    else:
        prob = torch.softmax(scores, dim=2)[:, :, 1]
    tag = prob > threshold
    return [tag[i].tolist()[: int(n)] for i, n in enumerate(lengths)]


# --------------------------------------------------------------------------------------------
# recurrent encoder
# --------------------------------------------------------------------------------------------
def lstm_direction(x: Tensor, lengths: Tensor, w_ih: Tensor, w_hh: Tensor, b_ih: Tensor, b_hh: Tensor,
                   reverse: bool) -> Tensor:
    """One direction of one nn.LSTM layer with packed-sequence semantics.

    models/NeuralArchitectures.py:98-115: pack(enforce_sorted=False) -> nn.LSTM(h0=c0=0) -> pad.
    Gate order i,f,g,o; c = sigmoid(f)*c + sigmoid(i)*tanh(g); h = sigmoid(o)*tanh(c); rows past a
    document's length are exactly 0; the reverse direction starts at the document's own last
    sentence (len-1) and walks to 0.
    """
    B, L, _ = x.shape
    H = w_hh.shape[1]
    out = x.new_zeros(B, L, H)
    xp = x @ w_ih.t() + b_ih + b_hh  # [B, L, 4H]
    for b in range(B):
        n = int(lengths[b])
        h = x.new_zeros(H)
        c = x.new_zeros(H)
        steps = range(n - 1, -1, -1) if reverse else range(n)
        hs = {}
        for t in steps:
            g = xp[b, t] + w_hh @ h
            i_, f_, g_, o_ = g[:H], g[H:2 * H], g[2 * H:3 * H], g[3 * H:]
            c = torch.sigmoid(f_) * c + torch.sigmoid(i_) * torch.tanh(g_)
            h = torch.sigmoid(o_) * torch.tanh(c)
            hs[t] = h
        if n > 0:
            out[b, :n] = torch.stack([hs[t] for t in range(n)])
    return out


def lstm_direction_batched(x: Tensor, lengths: Tensor, w_ih: Tensor, w_hh: Tensor, b_ih: Tensor,
                           b_hh: Tensor, reverse: bool) -> Tensor:
    """Same arithmetic as :func:`lstm_direction`, vectorised over documents (one matmul per step).

    Used for the larger fixtures and for the CPU baseline; the per-document loop above is the
    readable statement of the semantics and the two are cross-checked in the tests.
    """
    B, L, _ = x.shape
    H = w_hh.shape[1]
    xp = x @ w_ih.t() + (b_ih + b_hh)
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    outs: List[Optional[Tensor]] = [None] * L
    lengths = lengths.to(torch.long)
    idx = torch.arange(B)
    for s in range(L):
        if reverse:
            # step s processes, for every doc, its own position len-1-s (if still >= 0)
            t = lengths - 1 - s
            act = t >= 0
            tt = t.clamp(min=0)
        else:
            tt = torch.full((B,), s, dtype=torch.long)
            act = tt < lengths
        g = xp[idx, tt] + h @ w_hh.t()
        i_, f_, g_, o_ = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
        c_new = torch.sigmoid(f_) * c + torch.sigmoid(i_) * torch.tanh(g_)
        h_new = torch.sigmoid(o_) * torch.tanh(c_new)
        m = act.unsqueeze(1)
        c = torch.where(m, c_new, c)
        h = torch.where(m, h_new, h)
        outs[s] = h_new
    steps = torch.stack(outs)                                   # [step, B, H]
    t = torch.arange(L).view(1, L)
    step_of = (lengths.view(B, 1) - 1 - t) if reverse else t.expand(B, L)
    valid = (t < lengths.view(B, 1))
    out = steps[step_of.clamp(min=0), idx.view(B, 1)]            # [B, L, H] gather (autograd-friendly)
    return out * valid.unsqueeze(2).to(x.dtype)


def rnn_forward(x: Tensor, lengths: Tensor, p: Dict[str, Tensor], prefix: str, num_layers: int,
                bidirectional: bool = True, batched: bool = False) -> Tensor:
    """``RNN.forward`` with dropout 0 (models/NeuralArchitectures.py:83-145).

    Output length is max(lengths), not the padded input length (pad_packed_sequence, :115).
    ``p`` uses nn.LSTM names: ``{prefix}rnn.weight_ih_l{k}[_reverse]`` etc.
    """
    fn = lstm_direction_batched if batched else lstm_direction
    maxlen = int(lengths.max())
    h = x[:, :maxlen]
    for k in range(num_layers):
        outs = []
        for sfx, rev in (('', False), ('_reverse', True)):
            if rev and not bidirectional:
                continue
            outs.append(fn(h, lengths, p[f'{prefix}rnn.weight_ih_l{k}{sfx}'], p[f'{prefix}rnn.weight_hh_l{k}{sfx}'],
                           p[f'{prefix}rnn.bias_ih_l{k}{sfx}'], p[f'{prefix}rnn.bias_hh_l{k}{sfx}'], rev))
        h = torch.cat(outs, dim=2)
    return h


def bilstm_scores(x: Tensor, lengths: Tensor, p: Dict[str, Tensor], num_layers: int,
                  batched: bool = False) -> Tensor:
    """BiLSTM tagger scores: RNN -> Linear(2H -> 1|2).  models/CRF.py:319-321,340 / :360-361."""
    h = rnn_forward(x, lengths, p, 'model.', num_layers, True, batched)
    return h @ p['classification.weight'].t() + p['classification.bias']


def late_fusion_scores(x1: Tensor, x2: Tensor, lengths: Tensor, p: Dict[str, Tensor], num_layers: int,
                       batched: bool = False) -> Tensor:
    """BiLSTMLateFusion: two RNNs -> concat(dim=2) -> Linear(4H -> 1|2).  models/CRF.py:420-425,445."""
    h1 = rnn_forward(x1, lengths, p, 'model1.', num_layers, True, batched)
    h2 = rnn_forward(x2, lengths, p, 'model2.', num_layers, True, batched)
    h = torch.cat((h1, h2), dim=2)
    return h @ p['classification.weight'].t() + p['classification.bias']


# --------------------------------------------------------------------------------------------
# restricted-window self-attention encoder (live path = HF LongformerModel, local attention only)
# --------------------------------------------------------------------------------------------
def band_attention(q: Tensor, k: Tensor, v: Tensor, lengths: Tensor, radius: int) -> Tensor:
    """Per-query softmax over keys j in [i-radius, i+radius] ∩ [0, len_b).

    q,k,v: [B, L, heads, hd]; q already scaled by 1/sqrt(hd) (HF:514).
    HF:759-823 builds the same band scores by chunking; keys outside the sequence get -inf (HF:744-757),
    padded keys get finfo.min added (HF:524-536) => probability exactly 0; rows of masked queries
    are zeroed (HF:579).  The legacy implementation models/RestrictedTransformerLayer.py:509-636
    computes the same window softmax position by position (without the padding mask).
    """
    B, L, Hh, hd = q.shape
    W = 2 * radius + 1
    pos = torch.arange(L)

    def shifted(t: Tensor, off: int) -> Tensor:
        """t[:, i+off] for every i, zero where i+off is outside [0, L)."""
        if off == 0:
            return t
        z = t.new_zeros(B, min(abs(off), L), Hh, hd)
        if abs(off) >= L:
            return t.new_zeros(t.shape)
        return torch.cat([t[:, off:], z], dim=1) if off > 0 else torch.cat([z, t[:, :off]], dim=1)

    cols = []
    oks = []
    for c in range(W):                                                   # slot c <-> key j = i - radius + c
        off = c - radius
        cols.append((q * shifted(k, off)).sum(-1))                       # [B, L, heads]
        j = pos + off
        oks.append(((j >= 0) & (j < L)).view(1, L) & (j.view(1, L) < lengths.view(B, 1)))
    s = torch.stack(cols, dim=-1)                                        # [B, L, heads, W]
    key_ok = torch.stack(oks, dim=-1)                                    # [B, L, W]
    s = s.masked_fill(~key_ok.unsqueeze(2), float('-inf'))
    q_ok = pos.view(1, L) < lengths.view(B, 1)                           # [B, L]
    # a masked query has no valid key when it sits >radius past the end: avoid NaN, then zero the row
    s = torch.where(q_ok.view(B, L, 1, 1), s, torch.zeros_like(s))
    pr = torch.softmax(s, dim=-1)
    pr = pr * q_ok.view(B, L, 1, 1).to(pr.dtype)
    out = torch.zeros_like(q)
    for c in range(W):
        out = out + pr[..., c].unsqueeze(-1) * shifted(v, c - radius)
    return out


def band_attention_blocked(q: Tensor, k: Tensor, v: Tensor, lengths: Tensor, radius: int, block: int = 64) -> Tensor:
    """The same function as :func:`band_attention` computed block-wise with dense matmuls -- the way HF's
    ``_sliding_chunks_query_key_matmul`` (HF:759-823) gets its speed: queries in blocks of ``block`` rows, each block against the
    ``block + 2*radius`` keys it can see, out-of-band entries masked.  Used by bench.py's ``cpu_baseline`` (the shifted-product
    form above does 2w+1 elementwise passes over q/k/v and is ~2x slower than the reference on CPU); cross-checked against
    :func:`band_attention` in tests/test_oracle_vs_golden.py."""
    B, L, Hh, hd = q.shape
    w = radius
    nb = (L + block - 1) // block
    Lp = nb * block
    pad_q = Lp - L
    qp = torch.nn.functional.pad(q, (0, 0, 0, 0, 0, pad_q))                                 # [B, Lp, H, hd]
    kp = torch.nn.functional.pad(k, (0, 0, 0, 0, w, w + pad_q))                             # key j sits at j + w
    vp = torch.nn.functional.pad(v, (0, 0, 0, 0, w, w + pad_q))
    span = block + 2 * w
    kb = kp.unfold(1, span, block)                                                           # [B, nb, H, hd, span]
    vb = vp.unfold(1, span, block)
    qb = qp.view(B, nb, block, Hh, hd)
    s = torch.einsum('bnqhd,bnhdk->bnhqk', qb, kb)                                          # [B, nb, H, block, span]
    qi = torch.arange(block).view(block, 1)
    kj = torch.arange(span).view(1, span)
    in_band = (kj >= qi) & (kj <= qi + 2 * w)                                               # key offset kj <-> j = n*block + kj - w
    jabs = (torch.arange(nb).view(nb, 1) * block + torch.arange(span).view(1, span) - w)    # [nb, span]
    key_ok = (jabs.view(1, nb, span) >= 0) & (jabs.view(1, nb, span) < lengths.view(B, 1, 1).clamp(max=L))
    ok = in_band.view(1, 1, 1, block, span) & key_ok.view(B, nb, 1, 1, span)
    iabs = torch.arange(Lp).view(1, nb, 1, block, 1)
    q_ok = iabs < lengths.view(B, 1, 1, 1, 1).clamp(max=L)
    s = s.masked_fill(~ok, float('-inf'))
    s = torch.where(q_ok, s, torch.zeros_like(s))
    pr = torch.softmax(s, dim=-1) * q_ok.to(s.dtype)
    out = torch.einsum('bnhqk,bnhdk->bnqhd', pr, vb).reshape(B, Lp, Hh, hd)
    return out[:, :L]


def band_encoder(x: Tensor, lengths: Tensor, p: Dict[str, Tensor], heads: int, radii: Sequence[int],
                 prefix: str = 'model.model.', ln_eps: float = 1e-12, attention=None) -> Tensor:
    """``Longformer_Local_Attention.forward`` (models/RestrictedTransformerLayer.py:118-133) =
    HF ``LongformerModel`` with local attention only.

    HF:402-426 embeddings: x + pos_emb[2+i] + type_emb[0] -> LayerNorm(eps 1e-12; the wrapper's
    ``layer_norm_eps`` argument is ignored, RestrictedTransformerLayer.py:82-92) ; per layer
    (HF:482-640, 1061-1172): q=(Wq h+b)/sqrt(hd), k, v, band softmax, ctx; a=LN(Wo ctx+b+h);
    f=GELU(W1 a+b); h'=LN(W2 f+b+a).  HF pads L to a multiple of the largest window with masked
    rows and slices them off again (HF:1343-1390, 1229): no effect on rows < L, so not restated.
    """
    B, L, D = x.shape
    hd = D // heads
    e = prefix + 'embeddings.'
    h = x + p[e + 'position_embeddings.weight'][2:2 + L].unsqueeze(0) + p[e + 'token_type_embeddings.weight'][0]
    h = layer_norm(h, p[e + 'LayerNorm.weight'], p[e + 'LayerNorm.bias'], ln_eps)
    for li, radius in enumerate(radii):
        lp = f'{prefix}encoder.layer.{li}.'
        a = lp + 'attention.self.'
        q = (h @ p[a + 'query.weight'].t() + p[a + 'query.bias']) / math.sqrt(hd)
        k = h @ p[a + 'key.weight'].t() + p[a + 'key.bias']
        v = h @ p[a + 'value.weight'].t() + p[a + 'value.bias']
        ctx = (attention or band_attention)(q.view(B, L, heads, hd), k.view(B, L, heads, hd), v.view(B, L, heads, hd),
                                            lengths, radius).reshape(B, L, D)
        o = lp + 'attention.output.'
        a1 = layer_norm(ctx @ p[o + 'dense.weight'].t() + p[o + 'dense.bias'] + h,
                        p[o + 'LayerNorm.weight'], p[o + 'LayerNorm.bias'], ln_eps)
        f = gelu_erf(a1 @ p[lp + 'intermediate.dense.weight'].t() + p[lp + 'intermediate.dense.bias'])
        h = layer_norm(f @ p[lp + 'output.dense.weight'].t() + p[lp + 'output.dense.bias'] + a1,
                       p[lp + 'output.LayerNorm.weight'], p[lp + 'output.LayerNorm.bias'], ln_eps)
    return h


def pyramidal_radii(num_layers: int, window: int) -> List[int]:
    """models/CRF.py:529 ``[k*window for k in num_layers..1]`` -> one-sided radius = window//2 (HF:478)."""
    assert window % 2 == 0, 'Window size must be divisible by 2!'  # RestrictedTransformerLayer.py:77-80
    return [(k * window) // 2 for k in range(num_layers, 0, -1)]


def transformer_scores(x: Tensor, lengths: Tensor, p: Dict[str, Tensor], heads: int, radii: Sequence[int], attention=None) -> Tensor:
    """Transformer_segmenter: encoder -> Linear(D -> 1|2).  models/CRF.py:578-579, :601-602."""
    h = band_encoder(x, lengths, p, heads, radii, attention=attention)
    return h @ p['classification.weight'].t() + p['classification.bias']


def legacy_restricted_layer(x: Tensor, p: Dict[str, Tensor], heads: int, radius: int, eps: float = 1e-5) -> Tensor:
    """Legacy ``RestrictedTransformerEncoderLayer`` (post-LN, ReLU) with its per-position sliding MHA.

    models/RestrictedTransformerLayer.py:269-310 (layer), :413-644 (attention; key_padding_mask forced
    None :467, packed in_proj [3d,d]).  Same band softmax as the live path, no padding mask, q scaled by
    1/sqrt(hd) inside F.multi_head_attention_forward.
    """
    B, L, D = x.shape
    hd = D // heads
    w, b = p['self_attn.in_proj_weight'], p['self_attn.in_proj_bias']
    q = (x @ w[:D].t() + b[:D]) / math.sqrt(hd)
    k = x @ w[D:2 * D].t() + b[D:2 * D]
    v = x @ w[2 * D:].t() + b[2 * D:]
    full = torch.full((B,), L, dtype=torch.long)
    ctx = band_attention(q.view(B, L, heads, hd), k.view(B, L, heads, hd), v.view(B, L, heads, hd), full, radius)
    sa = ctx.reshape(B, L, D) @ p['self_attn.out_proj.weight'].t() + p['self_attn.out_proj.bias']
    y = layer_norm(x + sa, p['norm1.weight'], p['norm1.bias'], eps)
    ff = torch.relu(y @ p['linear1.weight'].t() + p['linear1.bias']) @ p['linear2.weight'].t() + p['linear2.bias']
    return layer_norm(y + ff, p['norm2.weight'], p['norm2.bias'], eps)


# --------------------------------------------------------------------------------------------
# CRF head
# --------------------------------------------------------------------------------------------
def crf_forward_score(feats: Tensor, mask: Tensor, trans: Tensor) -> Tensor:
    """log Z by the forward algorithm.  models/CRF.py:218-240 (T[i,j] = score of j -> i)."""
    B, L, C = feats.shape
    start, stop = C - 2, C - 1
    scores = feats.new_full((B, C), IMPOSSIBLE)
    scores[:, start] = 0.0
    for t in range(L):
        sc = scores.unsqueeze(1) + trans.unsqueeze(0) + feats[:, t].unsqueeze(2)   # [B, C(to), C(from)]
        sc = torch.logsumexp(sc, dim=-1)
        m = mask[:, t].unsqueeze(1)
        scores = sc * m + scores * (1 - m)
    return torch.logsumexp(scores + trans[stop], dim=-1)


def crf_gold_score(feats: Tensor, tags: Tensor, mask: Tensor, trans: Tensor) -> Tensor:
    """Score of the provided tag path.  models/CRF.py:148-170."""
    B, L, C = feats.shape
    start, stop = C - 2, C - 1
    emit = feats.gather(2, tags.unsqueeze(-1)).squeeze(-1)
    tg = torch.cat([torch.full((B, 1), start, dtype=torch.long), tags], dim=1)
    tr = trans[tg[:, 1:], tg[:, :-1]]
    last = tg.gather(1, mask.sum(1).long().unsqueeze(1)).squeeze(1)
    return ((tr + emit) * mask).sum(1) + trans[stop, last]


def crf_nll(features: Tensor, tags: Tensor, mask: Tensor, fc_w: Tensor, fc_b: Tensor, trans: Tensor) -> Tensor:
    """``CRF.loss``: fc -> mean(logZ - gold).  models/CRF.py:130-146."""
    feats = features @ fc_w.t() + fc_b
    L = feats.shape[1]
    m = mask[:, :L].to(feats.dtype)
    return (crf_forward_score(feats, m, trans) - crf_gold_score(feats, tags[:, :L].long(), m, trans)).mean()


def crf_viterbi(features: Tensor, mask: Tensor, fc_w: Tensor, fc_b: Tensor, trans: Tensor
                ) -> Tuple[Tensor, List[List[int]]]:
    """``CRF.forward`` -> (best_score [B], best_paths).  models/CRF.py:119-128, 172-216."""
    feats = features @ fc_w.t() + fc_b
    B, L, C = feats.shape
    start, stop = C - 2, C - 1
    m = mask[:, :L].to(feats.dtype)
    bps = torch.zeros(B, L, C, dtype=torch.long)
    best = feats.new_full((B, C), IMPOSSIBLE)
    best[:, start] = 0
    for t in range(L):
        acc = best.unsqueeze(1) + trans                      # [B, C(to), C(from)]
        acc, bps[:, t] = acc.max(dim=-1)
        acc = acc + feats[:, t]
        mt = m[:, t].unsqueeze(1)
        best = acc * mt + best * (1 - mt)
    best = best + trans[stop]
    score, tag = best.max(dim=-1)
    paths = []
    for b in range(B):
        cur = int(tag[b])
        n = int(m[b].sum())
        path = [cur]
        for t in range(n - 1, -1, -1):
            cur = int(bps[b, t, cur])
            path.append(cur)
        paths.append(path[-2::-1])
    return score, paths


# --------------------------------------------------------------------------------------------
# batch layout (EncoderDataset.AudioPortionDataset.collater)
# --------------------------------------------------------------------------------------------
def collate(samples: List[dict], crf: bool, truncate: bool, truncate_value: int, has_second: bool = False,
            domain_adapt: bool = False) -> dict:
    """EncoderDataset.py:91-152.  samples: dicts with 'id','target','embeddings'[,'embeddings2','domain']."""
    if len(samples) == 0:
        return {}
    minus = 0 if crf else 1                                       # :23

    def merge(values):
        max_len = truncate_value if truncate else max(v.shape[0] for v in values)
        out = torch.zeros((len(values), max_len, values[0].shape[1]))
        for i, v in enumerate(values):
            n = min(truncate_value, len(v)) if truncate else len(v)
            out[i, :n] = v[:n]
        return out

    def merge_tags(tags):
        max_len = truncate_value if truncate else max(len(v) for v in tags)
        out = torch.zeros((len(tags), max_len)) - minus
        for i, v in enumerate(tags):
            n = min(truncate_value, len(v)) if truncate else len(v)
            out[i, :n] = torch.as_tensor(v[:n], dtype=torch.float32)
        return out

    if truncate:
        lens = torch.LongTensor([min(truncate_value, len(s['embeddings'])) for s in samples])
    else:
        lens = torch.LongTensor([len(s['embeddings']) for s in samples])
    return {
        'id': torch.tensor([int(s['id']) for s in samples]),
        'src_tokens': merge([s['embeddings'] for s in samples]),
        'src_lengths': lens,
        'tgt_tokens': merge_tags([s['target'] for s in samples]),
        'src_tokens2': merge([s['embeddings2'] for s in samples]) if has_second else None,
        'domain': [s['domain'] for s in samples] if domain_adapt else None,
    }


# --------------------------------------------------------------------------------------------
# segmentation metrics (models/lightning_model.py:16-55; segeval 2.0.11 is a third-party dependency
# absent here -- its published Pk / WindowDiff definitions are restated; "parity unpinned" for the
# exact default-window convention)
# --------------------------------------------------------------------------------------------
def get_boundaries(boundaries: Sequence) -> List[int]:
    """bool list -> segment masses.  models/lightning_model.py:16-24."""
    tot, masses = 0, []
    for b in boundaries:
        tot += 1
        if b:
            masses.append(tot)
            tot = 0
    return masses


def _masses_to_positions(masses: Sequence[int]) -> List[int]:
    pos = []
    for seg, m in enumerate(masses):
        pos.extend([seg] * m)
    return pos


def _default_window(ref_masses: Sequence[int]) -> int:
    """segeval: k = round(mean reference segment mass / 2), at least 2 (Beeferman et al. 1999)."""
    avg = sum(ref_masses) / float(len(ref_masses))
    k = int(round(avg / 2.0))  # python3 round-half-even, as segeval's ``int(round(...))`` under py3
    return max(k, 2)


def pk(hyp_masses: Sequence[int], ref_masses: Sequence[int], window_size: Optional[int] = None) -> float:
    """P_k (Beeferman et al. 1999) as in segeval.pk(h, t): fraction of windows whose end points are
    'same segment' in one segmentation and 'different segment' in the other."""
    h, r = _masses_to_positions(hyp_masses), _masses_to_positions(ref_masses)
    assert len(h) == len(r)
    k = window_size or _default_window(ref_masses)
    n = len(r) - k
    if n <= 0:
        return 0.0
    err = sum(1 for i in range(n) if (h[i] == h[i + k]) != (r[i] == r[i + k]))
    return err / float(n)


def window_diff(hyp_masses: Sequence[int], ref_masses: Sequence[int], window_size: Optional[int] = None) -> float:
    """WindowDiff (Pevzner & Hearst 2002) as in segeval.window_diff(h, t): windows whose boundary
    counts differ."""
    h, r = _masses_to_positions(hyp_masses), _masses_to_positions(ref_masses)
    assert len(h) == len(r)
    k = window_size or _default_window(ref_masses)
    n = len(r) - k
    if n <= 0:
        return 0.0
    err = sum(1 for i in range(n) if (h[i + k] - h[i]) != (r[i + k] - r[i]))
    return err / float(n)


def compute_pk(boundaries: Sequence, ground_truth: Sequence, window_size: Optional[int] = None) -> float:
    """models/lightning_model.py:26-39: last position forced to a boundary on both sides."""
    b = list(boundaries); g = list(ground_truth)
    b[-1] = 1; g[-1] = 1
    return pk(get_boundaries(b), get_boundaries(g), window_size)


def compute_window_diff(boundaries: Sequence, ground_truth: Sequence, window_size: Optional[int] = None) -> float:
    """models/lightning_model.py:41-55."""
    b = list(boundaries); g = list(ground_truth)
    b[-1] = 1; g[-1] = 1
    return window_diff(get_boundaries(b), get_boundaries(g), window_size)
