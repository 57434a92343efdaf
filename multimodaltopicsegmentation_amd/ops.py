"""Tensor-level wrappers over the C ABI (include/mts.h).  torch is used for device memory and streams only:
every function enqueues hand-written HIP kernels on torch's current stream through raw device pointers.
"""
import ctypes
import math

import torch

from . import _lib as L
from ._lib import check, lib, ptr, stream_ptr, dtype_code

_ws_cache = {}

TIMER = None   # set to a KernelTimer() to time individual launches with HIP events on the launch stream (bench.py)


class KernelTimer:
    """Per-launch HIP-event timing (torch.cuda.Event on torch's current stream = the stream the kernels run on)."""

    def __init__(self, only=None):
        """only: optional predicate on the tag -- every event pair costs a few microseconds of stream time, so the timed region
        of bench.py brackets just the launches its roofline line needs."""
        self.records = {}          # tag -> list of (start_event, end_event)
        self.only = only

    def begin(self, tag):
        if self.only is not None and not self.only(tag):
            return None
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.records.setdefault(tag, []).append((s, e))
        s.record()
        return e

    def summary(self):
        """tag -> (launches, total_ms); call after torch.cuda.synchronize()."""
        return {t: (len(v), sum(s.elapsed_time(e) for s, e in v)) for t, v in self.records.items()}


_pending_end = [None]     # end event of the bracket in progress (one at a time: launches are issued from one host thread)


@ctypes.CFUNCTYPE(None)
def _mid_hook():
    """called by mts_gemm between its GEMM launch and its split-K reduce launch (mts_gemm_set_mid_hook): the bracket ends HERE, so that a
    timed weight-gradient GEMM reports the GEMM kernel alone -- what a rocprofv3 kernel trace lists under that symbol"""
    e = _pending_end[0]
    if e is not None:
        e.record()
        _pending_end[0] = None


_hook_installed = [False]


class _timed:
    def __init__(self, tag):
        self.tag = tag

    def __enter__(self):
        self.e = TIMER.begin(self.tag) if TIMER is not None else None
        if self.e is not None and self.tag[0] == 'gemm':
            if not _hook_installed[0]:
                lib.mts_gemm_set_mid_hook(ctypes.cast(_mid_hook, ctypes.c_void_p))
                _hook_installed[0] = True
            _pending_end[0] = self.e

    def __exit__(self, *a):
        if self.e is not None:
            if self.tag[0] == 'gemm':
                if _pending_end[0] is None:            # the hook recorded it
                    return False
                _pending_end[0] = None
            self.e.record()
        return False


def _scratch(nbytes, device, tag):
    """Grow-only scratch buffers keyed by (device, tag): never reallocated inside a steady-state step."""
    # one buffer per (device, purpose, HIP stream): kernels launched on different streams may run concurrently
    key = (str(device), tag, torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == 'cuda' else 0)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


_side_streams = {}


def side_stream(device, idx=0):
    """One of a SMALL fixed pool of side HIP streams per device, shared by every model of the process.  HIP maps streams onto a
    handful of hardware queues (GPU_MAX_HW_QUEUES, 8 here) in creation order: a process that builds several taggers, each with
    side streams of its own, ends up with two streams that should overlap on one queue (the late-fusion line of bench.py's
    other_configs ran 8.5 ms in such a process against 7.6 alone)."""
    key = (str(torch.device(device)), int(idx))
    s = _side_streams.get(key)
    if s is None:
        s = _side_streams[key] = torch.cuda.Stream(device=device)
    return s


_plan_cache = {}


def gemm(layout, A, B, out, *, M, N, K, lda=None, ldb=None, ldc=None, bias=None, residual=None, aux=None, gelu=False,
         colscale=None, ncols_scaled=0, accumulate=False, relu=False):
    """C[M,N] = op(A) op(B) (+ epilogue).  A/B share a dtype (fp32 or bf16); out is fp32 or that dtype."""
    a_dt = dtype_code(A.dtype)
    assert B.dtype == A.dtype
    c_dt = dtype_code(out.dtype)
    epi = 0
    if bias is not None:
        epi |= L.EPI_BIAS
    if residual is not None:
        epi |= L.EPI_RESIDUAL
    if gelu:
        epi |= L.EPI_GELU
    if relu:
        epi |= L.EPI_RELU
    if colscale is not None:
        epi |= L.EPI_COLSCALE
    if accumulate:
        epi |= L.EPI_ACCUM
    lda = lda if lda is not None else A.stride(0)
    ldb = ldb if ldb is not None else B.stride(0)
    ldc = ldc if ldc is not None else out.stride(0)
    ws, ws_bytes = None, 0
    if layout in (L.TN, L.TT) and c_dt == L.F32 and K >= 2048:
        if a_dt == L.BF16:
            ws_bytes = 8192 + min(16, max(1, K // 1024)) * M * N * 4   # arrival tickets of the in-launch combine + up to 16 split-K partial planes
        elif -(-M // 128) * -(-N // 128) <= 128:                       # the fp32 kernel only splits K when it has at most 128 output tiles
            ws_bytes = min(16, 256 // (-(-M // 128) * -(-N // 128)), max(1, K // 512)) * M * N * 4
        if ws_bytes:
            ws = _scratch(ws_bytes, A.device, 'splitk')
    key = (layout, a_dt, c_dt, M, N, K)
    with _timed(('gemm', *key, _plan_cache.get(key, 0))):
        check(lib.mts_gemm(stream_ptr(), a_dt, c_dt, layout, M, N, K, ptr(A), lda, ptr(B), ldb, ptr(out), ldc, ptr(bias),
                           ptr(residual), residual.stride(0) if residual is not None else 0, ptr(aux),
                           aux.stride(0) if aux is not None else 0, epi, float(colscale or 1.0), int(ncols_scaled),
                           ptr(ws), ws_bytes))
    if a_dt == L.BF16 and key not in _plan_cache:       # which kernel the cost model picked (128 | 224 | 256): timer label
        t = ctypes.c_int(0)
        lib.mts_gemm_last_plan(ctypes.byref(t), None)
        _plan_cache[key] = t.value
    return out


def linear_fwd(x, w, b, out, **kw):
    """out[M,N] = x[M,K] w[N,K]^T + b."""
    return gemm(L.NT, x, w, out, M=x.shape[0], N=w.shape[0], K=x.shape[1], bias=b, **kw)


def linear_dgrad(dy, w, out, **kw):
    """out[M,K] = dy[M,N] w[N,K]."""
    return gemm(L.NN, dy, w, out, M=dy.shape[0], N=w.shape[1], K=dy.shape[1], **kw)


def linear_wgrad(dy, x, out, **kw):
    """out[N,K] (fp32) = dy[M,N]^T x[M,K]."""
    return gemm(L.TN, dy, x, out, M=dy.shape[1], N=x.shape[1], K=dy.shape[0], **kw)


def wgrad_pair_supported(dy1, x1):
    """Can the two weight gradients dy1^T x1 and (their twin of the same shape) go out as one launch?  (include/mts.h mts_wgrad_pair)"""
    return dy1.dtype == torch.bfloat16 and dy1.is_cuda and lib.mts_wgrad_pair_workspace(dy1.shape[1], x1.shape[1], dy1.shape[0]) > 0


def wgrad_pair(dy1, x1, out1, dy2, x2, out2_t, accumulate=False):
    """out1[M, N] (fp32) = dy1[K, M]^T x1[K, N]  and  out2_t[N, M] = (dy2[K, M]^T x2[K, N])^T  in ONE launch + two fixed-order reduces."""
    K, M = dy1.shape
    N = x1.shape[1]
    assert dy2.shape == dy1.shape and x2.shape == x1.shape and dy2.stride(0) == dy1.stride(0) and x2.stride(0) == x1.stride(0)
    assert tuple(out1.shape) == (M, N) and out2_t.shape[0] == N and out2_t.shape[1] >= M
    nbytes = lib.mts_wgrad_pair_workspace(M, N, K)
    ws = _scratch(nbytes, dy1.device, 'wgrad_pair')
    with _timed(('wgrad_pair', M, N, K)):
        check(lib.mts_wgrad_pair(stream_ptr(), M, N, K, ptr(dy1), ptr(x1), ptr(out1), out1.stride(0), ptr(dy2), ptr(x2), ptr(out2_t), out2_t.stride(0),
                                 dy1.stride(0), x1.stride(0), 1 if accumulate else 0, ptr(ws), nbytes))


def colsum(x, out, accumulate=False):
    M, N = x.shape
    ws = _scratch(lib.mts_colsum_workspace(N), x.device, 'colsum')
    check(lib.mts_colsum(stream_ptr(), dtype_code(x.dtype), M, N, ptr(x), x.stride(0), ptr(out), int(accumulate), ptr(ws)))
    return out


def cast(src, dst):
    check(lib.mts_cast(stream_ptr(), dtype_code(dst.dtype), ptr(src), ptr(dst), src.numel()))
    return dst


def cast_concat(x1, x2, dst):
    """dst[r] = act-dtype(x1[r] | x2[r]) for two fp32 matrices of equal row count (K-split input of the recurrent taggers)."""
    check(lib.mts_cast_concat(stream_ptr(), dtype_code(dst.dtype), x1.shape[0], x1.shape[1], x2.shape[1], ptr(x1), ptr(x2), ptr(dst)))
    return dst


def embed_layernorm_fwd(x, pos, pos_offset, type0, gamma, beta, eps, y, pre, mean, rstd, row_src=None, x2=None):
    """row_src (int32 [n_rows], optional): packed batch -- output row r is sentence row_src[r] = b*L + i of x.
    x2 (fp32 [B, L, D2], optional): K-split input -- the row is x[b, i] | x2[b, i] and the concatenation is never materialised."""
    B, Lq, D = x.shape
    if x.dtype == torch.bfloat16:
        assert x2 is None and y.dtype == torch.bfloat16
        check(lib.mts_embed_layernorm_fwd_x16(stream_ptr(), B, Lq, D, ptr(x), ptr(pos), pos_offset, ptr(type0), ptr(gamma), ptr(beta), eps, ptr(y),
                                              ptr(pre), ptr(mean), ptr(rstd), ptr(row_src), row_src.numel() if row_src is not None else 0))
        return
    if x2 is not None:
        check(lib.mts_embed_layernorm_fwd2(stream_ptr(), dtype_code(y.dtype), B, Lq, D, x2.shape[2], ptr(x), ptr(x2), ptr(pos), pos_offset,
                                           ptr(type0), ptr(gamma), ptr(beta), eps, ptr(y), ptr(pre), ptr(mean), ptr(rstd), ptr(row_src),
                                           row_src.numel() if row_src is not None else 0))
        return
    check(lib.mts_embed_layernorm_fwd(stream_ptr(), dtype_code(y.dtype), B, Lq, D, ptr(x), ptr(pos), pos_offset, ptr(type0),
                                      ptr(gamma), ptr(beta), eps, ptr(y), ptr(pre), ptr(mean), ptr(rstd), ptr(row_src),
                                      row_src.numel() if row_src is not None else 0))


def layernorm_fwd(x, gamma, beta, eps, y, mean, rstd, head_w=None, head_b=None, scores=None):
    """y may be None with a fused head: only the scores and the statistics are produced."""
    rows, D = x.shape
    n_out = head_w.shape[0] if head_w is not None else 0
    check(lib.mts_layernorm_fwd(stream_ptr(), dtype_code(x.dtype), rows, D, ptr(x), ptr(gamma), ptr(beta), eps, ptr(y),
                                ptr(mean), ptr(rstd), ptr(head_w), ptr(head_b), n_out, ptr(scores)))


def layernorm_bwd(x, dy, gamma, mean, rstd, dx, dgamma, dbeta, dxsum=None, dlogit=None, head_w=None, beta=None, dhead_w=None, dhead_b=None):
    """dhead_w / dhead_b (with beta): the fused head's parameter gradients come out of the same pass (n_out <= 2)."""
    rows, D = x.shape
    ws = _scratch(lib.mts_layernorm_bwd_workspace(D), x.device, 'ln_bwd')
    n_out = head_w.shape[0] if head_w is not None else 0
    check(lib.mts_layernorm_bwd(stream_ptr(), dtype_code(x.dtype), rows, D, ptr(x), ptr(dy), ptr(dlogit), ptr(head_w), n_out,
                                ptr(gamma), ptr(mean), ptr(rstd), ptr(dx), ptr(dgamma), ptr(dbeta), ptr(dxsum), ptr(ws),
                                ptr(beta), ptr(dhead_w), ptr(dhead_b)))


def loss_tail_supported(dtype, D, n_out):
    return dtype in (torch.float32, torch.bfloat16) and bool(lib.mts_layernorm_loss_tail_supported(dtype_code(dtype), D, n_out))


def layernorm_loss_tail(kind, x, gamma, beta, eps, head_w, head_b, targets, lengths, alpha, gamma_f, grad_scale, scores, loss_out, dx, dgamma,
                        dbeta, dxsum, dhead_w, dhead_b, batch_shape, row_src=None):
    """The last layer's LayerNorm + head + loss + their backward in one pass over x = s2 [rows, D] (see include/mts.h)."""
    rows, D = x.shape
    B, Lq = batch_shape
    ws = _scratch(lib.mts_layernorm_bwd_workspace(D), x.device, 'ln_bwd')
    with _timed(('ln_tail', rows, D, head_w.shape[0])):
        check(lib.mts_layernorm_loss_tail(stream_ptr(), dtype_code(x.dtype), rows, D, ptr(x), ptr(gamma), ptr(beta), eps, ptr(head_w), ptr(head_b),
                                          head_w.shape[0], kind, B, Lq, targets.shape[1], ptr(targets), ptr(lengths), float(alpha), float(gamma_f),
                                          float(grad_scale), ptr(row_src), row_src.numel() if row_src is not None else 0, ptr(scores),
                                          ptr(loss_out), ptr(dx), ptr(dgamma), ptr(dbeta), ptr(dxsum), ptr(dhead_w), ptr(dhead_b), ptr(ws)))


def embed_layernorm_bwd(pre, dh, gamma, mean, rstd, B, Lq, dgamma, dbeta, dtype0, dpos, pos_offset, row0=None, lengths=None):
    """Embedding block backward in one pass: dgamma / dbeta / dtype0 and rows pos_offset .. pos_offset + Lq - 1 of dpos are
    overwritten; the pre-LayerNorm gradient is never stored."""
    rows, D = pre.shape
    nb = lib.mts_embed_layernorm_bwd_workspace(B, Lq, D)
    ws = _scratch(nb, pre.device, 'emb_bwd')
    check(lib.mts_embed_layernorm_bwd(stream_ptr(), dtype_code(pre.dtype), B, Lq, D, ptr(pre), ptr(dh), ptr(gamma), ptr(mean), ptr(rstd),
                                      ptr(dgamma), ptr(dbeta), ptr(dtype0), ptr(dpos), pos_offset, ptr(row0),
                                      ptr(lengths) if row0 is not None else None, rows if row0 is not None else 0, ptr(ws), nb))


def embed_bwd(dpre, B, Lq, dpos, pos_offset, row0=None, lengths=None):
    D = dpre.shape[-1]
    check(lib.mts_embed_bwd(stream_ptr(), dtype_code(dpre.dtype), B, Lq, D, ptr(dpre), ptr(dpos), pos_offset, ptr(row0),
                            ptr(lengths) if row0 is not None else None))


def dropout_fwd(x, y, p, seed, mask=None, residual=None):
    """y = dropout(x) (+ residual); mask (uint8, same numel) records the kept elements for the backward."""
    check(lib.mts_dropout_fwd(stream_ptr(), dtype_code(x.dtype), x.numel(), ptr(x), ptr(residual), ptr(y), ptr(mask), float(p), int(seed)))


def dropout_bwd(dy, dx, mask, p):
    check(lib.mts_dropout_bwd(stream_ptr(), dtype_code(dy.dtype), dy.numel(), ptr(dy), ptr(dx), ptr(mask), float(p)))


def gelu_bwd(u, dy):
    check(lib.mts_gelu_bwd(stream_ptr(), dtype_code(u.dtype), u.numel(), ptr(u), ptr(dy)))


def relu_bwd(u, dy):
    check(lib.mts_relu_bwd(stream_ptr(), dtype_code(u.dtype), u.numel(), ptr(u), ptr(dy)))


def ffn_supported(dtype, M, D, F):
    return bool(lib.mts_ffn_supported(dtype_code(dtype), M, D, F)) if dtype in (torch.float32, torch.bfloat16) else False


def ffn_fwd(a1, w1, b1, w2, b2, u, f, s2, relu=False):
    """Fused feed-forward block (F = 256): u, f [M, F] and s2 [M, D] are views of buffers with room for ceil(M / 64) * 64 rows."""
    M, D = a1.shape
    with _timed(('ffn_fwd', M, D, w1.shape[0])):
        check(lib.mts_ffn_fwd(stream_ptr(), M, D, w1.shape[0], ptr(a1), ptr(w1), ptr(b1), ptr(w2), ptr(b2), int(relu), ptr(u), ptr(f), ptr(s2)))


def ffn_bwd_data(ds2, w1, w2, u, du, da1, relu=False):
    M, D = ds2.shape
    with _timed(('ffn_bwd', M, D, w1.shape[0])):
        check(lib.mts_ffn_bwd_data(stream_ptr(), M, D, w1.shape[0], ptr(ds2), ptr(w1), ptr(w2), ptr(u), int(relu), ptr(du), ptr(da1)))


def band_slots(radius):
    return lib.mts_band_slots(radius)


def band_attn_fwd(qkv, lengths, B, Lq, D, heads, radius, ctx, probs, row0=None, drop_p=0.0, drop_seed=0):
    """row0 (int32 [B], optional): packed batch -- document b owns rows row0[b] .. row0[b] + lengths[b] - 1 of qkv / ctx / probs."""
    with _timed(('band_fwd', B, Lq, D, heads, radius)):
        check(lib.mts_band_attn_fwd(stream_ptr(), dtype_code(qkv.dtype), B, Lq, D, heads, radius, ptr(qkv), ptr(lengths), ptr(ctx), ptr(probs),
                                    ptr(row0), float(drop_p), int(drop_seed)))


def band_attn_bwd(qkv, lengths, probs, dctx, B, Lq, D, heads, radius, dqkv, dscores, dbias=None, row0=None, drop_p=0.0, drop_seed=0):
    """dbias (fp32 [3D], optional): column sums of dqkv = q/k/v bias gradients, fused into the kernels' output stage."""
    q_scale = 1.0 / math.sqrt(D // heads)
    ws = _scratch(lib.mts_band_attn_bwd_workspace(B, Lq, D), qkv.device, 'band_bwd') if dbias is not None else None
    with _timed(('band_bwd', B, Lq, D, heads, radius)):
        check(lib.mts_band_attn_bwd(stream_ptr(), dtype_code(qkv.dtype), B, Lq, D, heads, radius, q_scale, ptr(qkv), ptr(lengths),
                                    ptr(probs), ptr(dctx), ptr(dqkv), ptr(dscores), ptr(dbias), ptr(ws), ptr(row0),
                                    qkv.shape[0] if row0 is not None else 0, float(drop_p), int(drop_seed)))


def tagger_loss(kind, scores, targets, lengths, alpha, gamma, loss_out, dscores, row_src=None, batch_shape=None):
    """scores [B, L, n_out]; or, for a packed batch, [n_rows, n_out] with row_src (int32 [n_rows]) and batch_shape = (B, L)."""
    if row_src is None:
        B, Lq, n_out = scores.shape
    else:
        (B, Lq), n_out = batch_shape, scores.shape[-1]
    nb = lib.mts_tagger_loss_workspace(B, Lq)
    ws = _scratch(nb, scores.device, 'loss')
    check(lib.mts_tagger_loss(stream_ptr(), kind, B, Lq, targets.shape[1], n_out, ptr(scores), ptr(targets), ptr(lengths),
                              float(alpha), float(gamma), ptr(loss_out), ptr(dscores), ptr(ws), nb, ptr(row_src),
                              row_src.numel() if row_src is not None else 0))


def greedy_decode(scores, lengths, threshold, tags_out):
    B, Lq, n_out = scores.shape
    check(lib.mts_greedy_decode(stream_ptr(), B, Lq, n_out, ptr(scores), ptr(lengths), float(threshold), ptr(tags_out)))


def head_fwd(x, w, b, scores):
    rows, D = x.shape
    check(lib.mts_head_fwd(stream_ptr(), dtype_code(x.dtype), rows, D, w.shape[0], ptr(x), x.stride(0), ptr(w), ptr(b), ptr(scores)))


def head_bwd_params(x, dscores, dw, db):
    rows, D = x.shape
    ws = _scratch(lib.mts_layernorm_bwd_workspace(D), x.device, 'ln_bwd')
    check(lib.mts_head_bwd_params(stream_ptr(), dtype_code(x.dtype), rows, D, dw.shape[0], ptr(x), x.stride(0), ptr(dscores),
                                  ptr(dw), ptr(db), ptr(ws)))


def head_bwd_data(dscores, w, dx, accumulate=False):
    rows, D = dx.shape
    check(lib.mts_head_bwd_data(stream_ptr(), dtype_code(dx.dtype), rows, D, w.shape[0], ptr(dscores), ptr(w), ptr(dx),
                                dx.stride(0), int(accumulate)))


def lstm_workspace(dtype, B, Lq, H, ndir, device, tag='lstm'):
    return _scratch(lib.mts_lstm_workspace(dtype_code(dtype), B, Lq, H, ndir), device, tag)


def lstm_fwd(xproj, w_hh, b_hh, lengths, B, Lq, H, ndir, out, gates, cells):
    ws = lstm_workspace(xproj.dtype, B, Lq, H, ndir, xproj.device)
    check(lib.mts_lstm_fwd(stream_ptr(), dtype_code(xproj.dtype), B, Lq, H, ndir, ptr(xproj), ptr(w_hh), ptr(b_hh), ptr(lengths), ptr(out),
                           ptr(gates), ptr(cells), ptr(ws)))


def lstm_bwd(w_hh, lengths, out, gates, cells, dout, B, Lq, H, ndir, dxproj, dw_hh, ws=None):
    if ws is None:
        ws = lstm_workspace(out.dtype, B, Lq, H, ndir, out.device)
    check(lib.mts_lstm_bwd(stream_ptr(), dtype_code(out.dtype), B, Lq, H, ndir, ptr(w_hh), ptr(lengths), ptr(out), ptr(gates),
                           ptr(cells), ptr(dout), ptr(dxproj), ptr(dw_hh), ptr(ws)))


def crf_nll(feats, tags, lengths, trans, loss_out, dfeats=None, dtrans=None):
    B, Lq, C = feats.shape
    ws = _scratch(lib.mts_crf_workspace(B, Lq, C), feats.device, 'crf')
    check(lib.mts_crf_nll(stream_ptr(), B, Lq, C, ptr(feats), ptr(tags), tags.shape[1], ptr(lengths), ptr(trans), ptr(loss_out),
                          ptr(dfeats), ptr(dtrans), ptr(ws)))


def crf_viterbi(feats, lengths, trans, best_score, paths):
    B, Lq, C = feats.shape
    ws = _scratch(B * Lq * C * 4, feats.device, 'crf_bp')
    check(lib.mts_crf_viterbi(stream_ptr(), B, Lq, C, ptr(feats), ptr(lengths), ptr(trans), ptr(best_score), ptr(paths), ptr(ws)))


def adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step, grad_scale=1.0, bf16_copy=None):
    check(lib.mts_adam_step(stream_ptr(), param.numel(), ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), lr, beta1, beta2,
                            eps, step, grad_scale, ptr(bf16_copy)))


def sgd_step(param, grad, buf, lr, momentum, weight_decay, first_step, grad_scale=1.0, bf16_copy=None):
    check(lib.mts_sgd_step(stream_ptr(), param.numel(), ptr(param), ptr(grad), ptr(buf), lr, momentum, weight_decay,
                           int(first_step), grad_scale, ptr(bf16_copy)))


def scale_(x, scale):
    """x *= scale in place (fp32, contiguous)."""
    if scale != 1.0:
        assert x.dtype == torch.float32 and x.is_contiguous()
        check(lib.mts_scale(stream_ptr(), x.numel(), ptr(x), float(scale)))
    return x


def lstm_bwd_recurrence(w_hh, lengths, out, gates, cells, dout, B, Lq, H, ndir, dxproj, ws):
    """first half of lstm_bwd (the recurrence); ws: a workspace of lstm_workspace() size that lstm_bwd_whh gets again, untouched"""
    check(lib.mts_lstm_bwd_recurrence(stream_ptr(), dtype_code(out.dtype), B, Lq, H, ndir, ptr(w_hh), ptr(lengths), ptr(out), ptr(gates),
                                      ptr(cells), ptr(dout), ptr(dxproj), ptr(ws)))


def lstm_bwd_whh(lengths, out, dxproj, B, Lq, H, ndir, dw_hh, ws):
    """second half: h_{t-1} and dW_hh (any stream that is ordered behind the recurrence)"""
    check(lib.mts_lstm_bwd_whh(stream_ptr(), dtype_code(out.dtype), B, Lq, H, ndir, ptr(lengths), ptr(out), ptr(dxproj), ptr(dw_hh), ptr(ws)))
