/*
 * mts.h -- C ABI of the MI355X-native sentence-boundary tagger hot path (libmts_hip.so).
 *
 * Drop-in boundary: the reference (Ighina/MultimodalTopicSegmentation) is pure Python; its "operator
 * interface" for this path is the duck-type between models/lightning_model.py::TextSegmenter and the
 * tagger object it owns (models/CRF.py: .loss(xs, lengths, tags) -> scalar, .forward(xs, lengths) ->
 * (scores, tags), models/CRF.py:274-369, :371-479, :508-610) plus the batch dict of
 * EncoderDataset.py:91-152.  There is no FFI in the reference, so this header defines the C entry
 * points a ctypes binding on the reference side would call (INTEGRATION.md shows that binding).  Each
 * entry point names the reference code it replaces.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the name ends in _host; no ownership is transferred, the
 *    caller allocates outputs and workspaces (through torch, hipMalloc, ...);
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises;
 *  - row-major tensors, leading dimensions in ELEMENTS;
 *  - return value: 0 = MTS_OK, otherwise an mts_status; mts_last_error() gives a message (thread local);
 *  - Threads: every entry point may be called from any host thread, concurrently, on different streams.  The library keeps no
 *    mutable process-wide state that influences results or kernel choice: mts_last_error() and the tuning switches of
 *    mts_set_option() are PER HOST THREAD (a switch set by one thread applies to the calls that thread issues, and only
 *    those -- note that torch's autograd runs backward nodes on its own thread), mts_gemm_last_plan() reports the calling
 *    thread's most recent mts_gemm.  The only shared words are idempotent "function attribute already set" flags
 *    (atomics) and the sticky device-error word of mts_async_status();
 *  - "act dtype" = the arithmetic/storage type of activations: MTS_F32 (parity mode, fp32 everywhere) or
 *    MTS_BF16 (bf16 storage + MFMA, fp32 accumulate/statistics).  Parameters and gradients are always fp32.
 */
#ifndef MTS_H
#define MTS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  MTS_OK = 0,
  MTS_ERR_INVALID = 1,      /* bad argument (shape, alignment, enum) */
  MTS_ERR_UNSUPPORTED = 2,  /* valid but outside what the kernels cover */
  MTS_ERR_LAUNCH = 3,       /* HIP runtime error at launch */
  MTS_ERR_WORKSPACE = 4,    /* workspace too small */
  MTS_ERR_TIMEOUT = 5       /* an EARLIER launch reported a device-side timeout (see mts_async_status): its results are invalid */
} mts_status;

typedef enum { MTS_F32 = 0, MTS_BF16 = 1 } mts_dtype;

/* loss kinds: models/CRF.py:294-312 */
typedef enum { MTS_LOSS_CE = 0, MTS_LOSS_BCE = 1, MTS_LOSS_FOCAL = 2 } mts_loss_kind;

/* GEMM operand layouts.  C[M,N] = op(A) * op(B):
 *   MTS_NT : A is [M,K] (K contiguous), B is [N,K] (K contiguous)   -- y = x W^T        (forward)
 *   MTS_NN : A is [M,K] (K contiguous), B is [K,N] (N contiguous)   -- dx = dy W        (data grad)
 *   MTS_TN : A is [K,M] (M contiguous), B is [K,N] (N contiguous)   -- dW = dy^T x      (weight grad)
 *   MTS_TT : A is [K,M] (M contiguous), B is [N,K] (K contiguous)   -- dW = dy^T x for a caller that holds x^T: one
 *            operand less goes through the transposing LDS reads (measured +6 % on the wide weight gradients)            */
typedef enum { MTS_NT = 0, MTS_NN = 1, MTS_TN = 2, MTS_TT = 3 } mts_gemm_layout;

/* epilogue flags for mts_gemm */
#define MTS_EPI_BIAS      1u   /* += bias[n]  (fp32 [N])                                  */
#define MTS_EPI_RESIDUAL  2u   /* += residual[m,n]  (act dtype, ld = ldr)                 */
#define MTS_EPI_GELU      4u   /* C = gelu_erf(acc); pre-activation stored to `aux` if non-null */
#define MTS_EPI_COLSCALE  8u   /* columns n < ncols_scaled are multiplied by colscale (q / sqrt(hd)) */
#define MTS_EPI_ACCUM    16u   /* C += result (fp32 C only) */
#define MTS_EPI_RELU     32u   /* C = max(acc, 0); pre-activation stored to `aux` if non-null (legacy layer's FFN,
                                * models/RestrictedTransformerLayer.py:308); exclusive with MTS_EPI_GELU */

const char* mts_last_error(void);
/* version / build info: "mts-hip <n> gfx950" */
const char* mts_version(void);
/* tuning / A-B switches: "gemm_variant" = 0 | 1 (generic epilogue in the 256x224 kernel) | 5 (its K loop with the barrier at the end of the K-tile for every layout) ; "gemm_f32_mfma" = 1
 * (fp32 GEMM on v_mfma_f32_16x16x4_f32) | 0 (VALU kernel, bitwise the same results) ; "gemm_tile" = 0 (cost model) | 128 | 224 | 256 ; "gemm_glds" = 1 (LDS-DMA staging) | 0 (register
 * staging) ; "gemm_splits" = 0 (cost model) | n ; "gemm_order" = 1 (L2-blocked tile order) | 0 ; "gemm_chain" = 0 | 1 (split-K
 * of the 128x128 kernel accumulates in place) ; "gemm_deep" = 1 (four-buffer copy pipeline of the 128x128 kernel for grids of at
 * most one workgroup per CU) | 0 ; "band_mfma" = 1 | 0 ; "band_fused_bwd" = 1 (bf16 band attention backward in one pass where it applies:
 * radius <= 15, head dim <= 224) | 0 (two kernels) ; "lstm_parts" = 4 (CU-quad recurrences at H = 256, bf16 and fp32) | 2 (CU pair, bf16) ;
 * "lstm_pair_spin_limit" = re-polls before a CU-pair LSTM workgroup gives
 * up on its partner (-1 = default 2^22; tests use 0) ; "lstm_pair_max_pairs" = CU pairs per recurrence launch (1..64, default 64) */
int mts_set_option(const char* key, int value);
/* Device-side errors that cannot be known at launch time, polled WITHOUT synchronising (a pinned host word the kernels write
 * with a system-scope store): 0, or MTS_ERR_TIMEOUT when a CU-pair LSTM launch gave up waiting for its partner workgroup since the
 * last poll (mts_last_error() says which).  mts_lstm_fwd / mts_lstm_bwd poll first themselves, so a training loop sees the error
 * at the next step at the latest; a host that has synchronised (decode, checkpoint) calls this.  Reading clears. */
int mts_async_status(void);
/* tile width (128 | 224 | 256; 225 / 226 = the 224-wide tile on the four-wave data-gradient kernel / the one-tile-per-workgroup forward kernel, symbols
 * of their own in a kernel trace) and K split the
 * cost model chose for the calling thread's most recent bf16 mts_gemm (bench / profiling labels) */
int mts_gemm_last_plan(int* tile, int* splits);
/* Measurement aid (per host thread; NULL clears it): fn() is called between the GEMM launch and the split-K reduce launch of every bf16 mts_gemm,
 * so that an event bracket can time the GEMM kernel by itself. */
int mts_gemm_set_mid_hook(void (*fn)(void));
/* Two weight gradients of ONE shape in one launch (bf16 operands, fp32 results; the fused feed-forward block's dW1 and dW2, whose 8 output tiles each
 * fill half the chip at best when launched alone):  C1[M, N] (+)= A1[K, M]^T B1[K, N]  and  C2t[N, M] (+)= (A2[K, M]^T B2[K, N])^T -- the second one
 * is STORED TRANSPOSED (dW2 = ds2^T f is computed as f^T ds2, the shape of dW1).  A1 / A2 share lda, B1 / B2 share ldb.  Split-K slabs + fixed-order
 * reduces as mts_gemm (bitwise reproducible); workspace >= mts_wgrad_pair_workspace(M, N, K) bytes (0: shape not covered: M % 256, N % 224, K % 64 --
 * MTS_ERR_UNSUPPORTED, the caller issues two mts_gemm calls instead).  Replaces the backward of modeling_longformer.py:1113-1131 (parameter part). */
size_t mts_wgrad_pair_workspace(int M, int N, int K);
int mts_wgrad_pair(void* stream, int M, int N, int K, const void* A1, const void* B1, float* C1, int ldc1, const void* A2, const void* B2,
                   float* C2t, int ldc2t, int lda, int ldb, int accumulate, void* workspace, size_t workspace_bytes);
/* The planner by itself (pure host code, no device call): the tile width and K split mts_gemm WOULD use for this call under the
 * calling thread's options; workspace_bytes = 0 means "no split-K workspace".  tile / splits may be NULL. */
int mts_gemm_plan(int a_dtype, int c_dtype, int layout, int M, int N, int K, unsigned epilogue, size_t workspace_bytes,
                  int* tile, int* splits);

/* ---------------------------------------------------------------------------------------------
 * Dense projection GEMM (MFMA for bf16, exact-fp32 VALU kernel for parity mode).
 * Replaces: nn.Linear inside HF Longformer (modeling_longformer.py:504-506, :1069, :1114, :1128),
 * the LSTM input projection inside aten::lstm (models/NeuralArchitectures.py:113) and the tagger
 * heads (models/CRF.py:299-310, :554-566); plus their autograd backward.
 * a_dtype: dtype of A and B (and residual/aux); c_dtype: dtype of C (MTS_F32 or a_dtype).
 * workspace (optional, workspace_bytes, 16-byte aligned): scratch for split-K partial tiles of weight-gradient shapes
 * (fp32 C, no epilogue besides MTS_EPI_ACCUM); without it such shapes run unsplit.  bf16 operands: the first 8192 bytes
 * hold the arrival tickets of the in-launch combine (zeroed by the call itself), the partial planes follow -- a call that
 * may split K into S slices wants 8192 + S * M * N * 4 bytes.  Partials are summed in slice order whichever slice adds
 * them (the last one to arrive, inside the launch, or a reduce launch), so results are bitwise reproducible.
 * ------------------------------------------------------------------------------------------- */
int mts_gemm(void* stream, int a_dtype, int c_dtype, int layout, int M, int N, int K,
             const void* A, int lda, const void* B, int ldb, void* C, int ldc,
             const float* bias, const void* residual, int ldr, void* aux, int ldaux,
             unsigned epilogue, float colscale, int ncols_scaled, void* workspace, size_t workspace_bytes);

/* column sums of an [M,N] activation matrix into fp32 out[N] (bias gradients); deterministic two-stage
 * reduction through `partial` (mts_colsum_workspace(N) bytes). */
size_t mts_colsum_workspace(int N);
int mts_colsum(void* stream, int dtype, int M, int N, const void* X, int ldx, float* out, int accumulate,
               void* partial);

/* fp32 -> act dtype copy (weights to bf16), n elements */
int mts_cast(void* stream, int dst_dtype, const float* src, void* dst, size_t n);
/* K-SPLIT INPUT.  Early fusion in the reference is a host-side torch.cat of the text and audio embedding matrices of a document
 * (utils/load_datasets_precomputed.py:158-161).  The *2 / _concat entry points take the two fp32 matrices separately
 * (src1 [rows, D1] | src2 [rows, D2], D1 and D2 multiples of 4) so that the concatenated batch never exists: dst [rows, D1+D2] in the
 * act dtype for the recurrent taggers' first projection ... */
int mts_cast_concat(void* stream, int dst_dtype, size_t rows, int D1, int D2, const float* src1, const float* src2, void* dst);
/* HOST-SIDE COLLATION (no device call, no stream).  Replaces the `merge` closure of AudioPortionDataset.collater (EncoderDataset.py:20-27,
 * :103-109): B ragged documents docs[b] = [doc_rows[b], D] (src_dtype MTS_F32 | MTS_BF16, row-major, contiguous) -> dst [B, Lmax, D] in
 * dst_dtype, rows past min(doc_rows[b], Lmax) = pad_value (0 for the embeddings, -1 / 0 for the targets: EncoderDataset.py:23; the
 * reference's truncate = "pad / cut to exactly Lmax").  The one pass that pads also
 * narrows fp32 -> bf16 (round to nearest even) where the dtypes differ, and is split over `nthreads` host threads (a persistent pool inside the library).  dst may be pinned
 * memory: the batch is then ready for an asynchronous host-to-device copy as it stands (prefetch.DevicePrefetcher sends it unstaged). */
int mts_collate_pad(int src_dtype, int dst_dtype, int B, int Lmax, int D, const void* const* docs, const int64_t* doc_rows, void* dst,
                    float pad_value, int nthreads);

/* ---------------------------------------------------------------------------------------------
 * LayerNorm family (biased variance, eps inside sqrt).
 * Replaces: LongformerEmbeddings (modeling_longformer.py:402-426: x + pos_emb[2+i] + type_emb[0] -> LN),
 * LongformerSelfOutput / LongformerOutput LayerNorm (:1068-1072, :1127-1131) and their backward.
 * ------------------------------------------------------------------------------------------- */
/* y[b,i,:] = LN(x[b,i,:] + pos[pos_offset+i,:] + type0[:]); x fp32 [B*L, D] (the batch as the collater made it);
 * pre (act dtype, optional) receives the pre-LN sum for the backward; mean/rstd fp32 [B*L]. */
int mts_embed_layernorm_fwd(void* stream, int dtype, int B, int L, int D, const float* x, const float* pos,
                            int pos_offset, const float* type0, const float* gamma, const float* beta, float eps,
                            void* y, void* pre, float* mean, float* rstd, const int32_t* row_src, int n_rows);
/* ... on a batch that arrived in bf16 (x_bf16: bf16 [B, L, D]; bf16 activations; one source): the same bits as mts_embed_layernorm_fwd on the fp32
 * values of the same numbers, without an fp32 copy of the batch (prefetch.DevicePrefetcher / AudioPortionDataset(wire_dtype='bf16')). */
int mts_embed_layernorm_fwd_x16(void* stream, int B, int L, int D, const void* x_bf16, const float* pos, int pos_offset, const float* type0,
                                const float* gamma, const float* beta, float eps, void* y, void* pre, float* mean, float* rstd,
                                const int32_t* row_src, int n_rows);
/* ... and the same embedding + LayerNorm as mts_embed_layernorm_fwd on x = x1[b,i,0:D1] | x2[b,i,0:D2] (D = D1 + D2). */
int mts_embed_layernorm_fwd2(void* stream, int dtype, int B, int L, int D1, int D2, const float* x1, const float* x2, const float* pos,
                             int pos_offset, const float* type0, const float* gamma, const float* beta, float eps,
                             void* y, void* pre, float* mean, float* rstd, const int32_t* row_src, int n_rows);
/* If head_w != NULL the tagger head is fused in: scores[r,c] = y[r,:].head_w[c,:] + head_b[c] (fp32 [rows,n_out],
 * n_out <= 4), computed on the stored (act dtype) y.  models/CRF.py:579 on top of modeling_longformer.py:1127-1131.
 * With a fused head and mean/rstd given, y may be NULL: the row is normalised, rounded to the act dtype and fed to the head, but
 * not written (training step of the last layer: mts_layernorm_bwd recomputes it, see dhead_w there). */
int mts_layernorm_fwd(void* stream, int dtype, int rows, int D, const void* x, const float* gamma,
                      const float* beta, float eps, void* y, float* mean, float* rstd,
                      const float* head_w, const float* head_b, int n_out, float* scores);
/* dx = LN'(x) dy ; dgamma/dbeta (fp32 [D]) are OVERWRITTEN with the column reductions; dxsum (optional,
 * fp32 [D]) receives colsum(dx) = the bias gradient of the linear layer that produced x.
 * If head_w != NULL the incoming gradient is dy[r,:] (if non-null) + sum_c dlogit[r,c]*head_w[c,:]
 * (the tagger head's data gradient fused in; n_out columns).
 * If additionally dhead_w != NULL (needs beta, dhead_b, n_out <= 2) the head's PARAMETER gradients are produced in the same pass:
 * dhead_w[c,:] = sum_r dlogit[r,c] * y[r,:], dhead_b[c] = sum_r dlogit[r,c] (OVERWRITE; models/CRF.py:579's nn.Linear), with
 * y[r,:] = LN(x[r,:]) recomputed from the saved statistics exactly as the forward rounded it -- the forward of that layer did not
 * have to store y and no separate mts_head_bwd_params pass reads it.  beta / dhead_w / dhead_b may be NULL together.
 * partial: fp32 workspace of mts_layernorm_bwd_workspace(D) bytes. */
size_t mts_layernorm_bwd_workspace(int D);
int mts_layernorm_bwd(void* stream, int dtype, int rows, int D, const void* x, const void* dy,
                      const float* dlogit, const float* head_w, int n_out,
                      const float* gamma, const float* mean, const float* rstd,
                      void* dx, float* dgamma, float* dbeta, float* dxsum, void* partial,
                      const float* beta, float* dhead_w, float* dhead_b);
/* THE LAST LAYER'S TAIL IN ONE PASS (training): LayerNorm forward + tagger head + masked loss + loss gradient + head data gradient + LayerNorm
 * backward of x = the layer's pre-LayerNorm sum s2 [rows, D], row by row -- what mts_layernorm_fwd(head) + mts_tagger_loss (+ mts_scale) +
 * mts_layernorm_bwd(head, dhead_w) do in four launches and two reads of x.  (modeling_longformer.py:1127-1131, models/CRF.py:579-595,
 * focal_loss.py:38-57 and their backward.)  scores fp32 [rows, n_out]; loss_out fp32 [2] = {loss, rows averaged}; dx act dtype [rows, D];
 * dgamma / dbeta / dxsum (may be NULL) fp32 [D]; dhead_w fp32 [n_out, D], dhead_b [n_out] (all OVERWRITTEN).  Batch description as
 * mts_tagger_loss (targets fp32 [B, Lt], lengths, packed rows through row_src / n_rows); grad_scale multiplies d loss / d scores.
 * n_out = 1 (BCE / focal) or 2 (CrossEntropy); D in {256, 512, 1024, 1792, 2048}: mts_layernorm_loss_tail_supported.  Scores and gradients are
 * bitwise those of the four launches; the loss differs by the grouping of its partial sums.  workspace: mts_layernorm_bwd_workspace(D) bytes. */
int mts_layernorm_loss_tail_supported(int dtype, int D, int n_out);
int mts_layernorm_loss_tail(void* stream, int dtype, int rows, int D, const void* x, const float* gamma, const float* beta, float eps,
                            const float* head_w, const float* head_b, int n_out, int loss_kind, int B, int L, int Lt, const float* targets,
                            const int32_t* lengths, float alpha, float gamma_focal, float grad_scale, const int32_t* row_src, int n_rows,
                            float* scores, float* loss_out, void* dx, float* dgamma, float* dbeta, float* dxsum, float* dhead_w,
                            float* dhead_b, void* workspace);
/* Backward of the embedding block in ONE pass (modeling_longformer.py:402-426: LN(x + pos[2+i] + type0)): dh (act dtype [rows, D]) is
 * the gradient at the LayerNorm's output, pre / mean / rstd what mts_embed_layernorm_fwd saved.  OVERWRITES dgamma, dbeta, dtype0
 * (= sum over all rows of the pre-LN gradient: the token-type row 0) and rows pos_offset .. pos_offset+L-1 of dpos (= the sum over the
 * documents of each position; positions no document reaches get 0).  The pre-LN gradient itself is never written to memory.
 * row0 / lengths / n_rows: packed batches (below), else NULL / NULL / 0.  D <= 2048. */
size_t mts_embed_layernorm_bwd_workspace(int B, int L, int D);
int mts_embed_layernorm_bwd(void* stream, int dtype, int B, int L, int D, const void* pre, const void* dh, const float* gamma,
                            const float* mean, const float* rstd, float* dgamma, float* dbeta, float* dtype0, float* dpos,
                            int pos_offset, const int32_t* row0, const int32_t* lengths, int n_rows, void* workspace, size_t workspace_bytes);
/* gradient of the position table from a STORED pre-LN gradient: dpos[pos_offset+i,:] += sum_b dpre[b,i,:].  (The token-type row's
 * gradient is sum_{b,i} dpre = the `dxsum` output of the embedding LayerNorm's mts_layernorm_bwd.)  Kept for D > 2048. */
int mts_embed_bwd(void* stream, int dtype, int B, int L, int D, const void* dpre, float* dpos, int pos_offset,
                  const int32_t* row0, const int32_t* lengths);

/* Inverted dropout: y = keep ? x / (1-p) : 0 (+ residual, if given), keep decided by a counter-based hash of (seed, element index)
 * (F.dropout in RNN.forward, NeuralArchitectures.py:94,119 -- active even in eval mode there; nn.Dropout of the HF layers,
 * modeling_longformer.py:424,1070,1129).  mask (optional, uint8 [n]) records keep = 1; x == y is allowed; n % 4 == 0.
 * mts_dropout_bwd: dx = mask ? dy / (1-p) : 0 (dx == dy allowed). */
int mts_dropout_fwd(void* stream, int dtype, size_t n, const void* x, const void* residual, void* y, uint8_t* mask,
                    float p, uint64_t seed);
int mts_dropout_bwd(void* stream, int dtype, size_t n, const void* dy, void* dx, const uint8_t* mask, float p);

/* dy *= gelu_erf'(u) in place (FFN backward; modeling_longformer.py:1113-1116); n elements, n % 4 == 0 */
int mts_gelu_bwd(void* stream, int dtype, size_t n, const void* u, void* dy);
/* dy *= (u > 0) in place (backward of the legacy layer's ReLU, models/RestrictedTransformerLayer.py:308); n % 4 == 0 */
int mts_relu_bwd(void* stream, int dtype, size_t n, const void* u, void* dy);

/* ---------------------------------------------------------------------------------------------
 * Fused feed-forward block (hidden width F = 256, D a multiple of 256, bf16): LongformerIntermediate + LongformerOutput.dense +
 * residual (modeling_longformer.py:1113-1131) in one launch, and the data gradient of the same block in one launch.
 *   forward : u = a1 W1^T + b1 ; f = gelu_erf(u) (relu != 0: max(u, 0)) ; s2 = f W2^T + b2 + a1
 *   backward: du = (ds2 W2) * act'(u) ; da1 = du W1 + ds2        (weight / bias gradients: mts_gemm TN, mts_colsum on du)
 * a1, ds2 [M, D]; w1 [F, D]; w2 [D, F] (bf16 mirrors); b1 [F], b2 [D] fp32; u, f, du [., F]; s2, da1 [., D].  OUTPUT buffers must
 * have room for ceil(M / 64) * 64 rows (rows past M are written, not masked).  Bitwise the results of the two-GEMM path.
 * mts_ffn_supported: 1 when the fused block covers (dtype, M, D, F), else the caller uses mts_gemm.
 * ------------------------------------------------------------------------------------------- */
int mts_ffn_supported(int dtype, int M, int D, int F);
int mts_ffn_fwd(void* stream, int M, int D, int F, const void* a1, const void* w1, const float* b1, const void* w2, const float* b2,
                int relu, void* u, void* f, void* s2);
int mts_ffn_bwd_data(void* stream, int M, int D, int F, const void* ds2, const void* w1, const void* w2, const void* u, int relu,
                     void* du, void* da1);

/* ---------------------------------------------------------------------------------------------
 * PACKED BATCHES (training path; the reference pads every Transformer batch to 3600 sentences, train_fit.py:104-106,
 * and runs all of them through the encoder).  Activations may hold only the valid sentences, document after document:
 * n_rows = sum of lengths; `row_src[r]` = b*L + i names the sentence of the padded batch that packed row r holds
 * (mts_embed_layernorm_fwd gathers x and the position, mts_tagger_loss finds the target); `row0[b]` = first packed row of
 * document b (band attention, mts_embed_bwd).  All four are NULL for the padded [B, L] layout.  Padded rows never
 * influence valid rows or any gradient (masked as keys, excluded from the loss), so results on valid rows are the same.
 * ------------------------------------------------------------------------------------------- */

/* ---------------------------------------------------------------------------------------------
 * Restricted-window (band) self-attention.
 * Replaces: LongformerSelfAttention.forward local path (modeling_longformer.py:482-640:
 * _sliding_chunks_query_key_matmul :759-823, padding mask :524-536, softmax fp32 :574-576, masked
 * query rows zeroed :579, _sliding_chunks_matmul_attn_probs_value :825-867) and, equivalently, the
 * legacy per-position loop models/RestrictedTransformerLayer.py:509-636; plus the backward.
 * qkv: [B*L, 3*D] act dtype, row = [q(D) | k(D) | v(D)], q already scaled; head h owns columns
 * h*hd..(h+1)*hd of each third.  lengths: int32 [B] (NULL = all L).  ctx: [B*L, D].
 * probs: fp32 [B*L, heads, slots] with slots = mts_band_slots(radius), saved for the backward.
 * drop_p > 0: dropout on the attention probabilities (modeling_longformer.py:590; HF attention_probs_dropout_prob = the reference's
 * dropout_out): what multiplies V is keep ? p / (1 - drop_p) : 0 with keep = hash(drop_seed, (row, head, slot)); `probs` stays whole
 * and the backward, given the same drop_p / drop_seed, regenerates the mask.
 * ------------------------------------------------------------------------------------------- */
int mts_band_slots(int radius);
int mts_band_attn_fwd(void* stream, int dtype, int B, int L, int D, int heads, int radius,
                      const void* qkv, const int32_t* lengths, void* ctx, float* probs, const int32_t* row0,
                      float drop_p, uint64_t drop_seed);
/* dqkv [B*L, 3D] (dq already multiplied by q_scale so it is the gradient wrt the unscaled projection);
 * dscores: fp32 scratch of the same size as probs (the one-pass bf16 kernel leaves it untouched: dS stays on the chip).  dbias (optional): fp32 [3D] column sums of dqkv as stored =
 * the gradient of the q/k/v biases (modeling_longformer.py:504-506), produced from the kernels' output tiles instead of
 * re-reading dqkv; needs `workspace` of mts_band_attn_bwd_workspace(B, L, D) bytes.  Bitwise reproducible. */
size_t mts_band_attn_bwd_workspace(int B, int L, int D);
int mts_band_attn_bwd(void* stream, int dtype, int B, int L, int D, int heads, int radius, float q_scale,
                      const void* qkv, const int32_t* lengths, const float* probs, const void* dctx,
                      void* dqkv, float* dscores, float* dbias, void* workspace, const int32_t* row0, int n_rows,
                      float drop_p, uint64_t drop_seed);

/* ---------------------------------------------------------------------------------------------
 * Tagger head tail: loss + its gradient, and greedy decode.
 * Replaces: the un-pad loop + BCE/Focal/CE of models/CRF.py:342-356 (=:447-461, :581-595),
 * models/focal_loss.py:38-57, and decode models/CRF.py:362-369.
 * scores: fp32 [B, L, n_out] (n_out = 1 for BCE / focal, 2..4 = tagset_size for CrossEntropy); targets: fp32 [B, Lt] (pad -1 / 0 as
 * the collater wrote them), Lt >= L; lengths int32 [B].  loss_out: fp32 [2] = {loss, number of rows averaged}.  dscores may be NULL.
 * workspace: mts_tagger_loss_workspace(B, L) bytes of device scratch for the per-workgroup partial sums
 * (NULL = single-workgroup path; same result up to fp32 summation order).
 * ------------------------------------------------------------------------------------------- */
size_t mts_tagger_loss_workspace(int B, int L);
int mts_tagger_loss(void* stream, int loss_kind, int B, int L, int Lt, int n_out, const float* scores,
                    const float* targets, const int32_t* lengths, float alpha, float gamma,
                    float* loss_out, float* dscores, void* workspace, size_t workspace_bytes,
                    const int32_t* row_src, int n_rows);
/* tags_out: uint8 [B, L]; positions >= length are 0.  prob > threshold, strict; prob = sigmoid (n_out 1) or softmax[..., 1] (n_out 2..4). */
int mts_greedy_decode(void* stream, int B, int L, int n_out, const float* scores, const int32_t* lengths,
                      float threshold, uint8_t* tags_out);
/* scores[r, c] = x[r,:] . w[c,:] + b[c]  (x act dtype [rows, D]; w fp32 [n_out, D]); n_out in 1..4 */
int mts_head_fwd(void* stream, int dtype, int rows, int D, int n_out, const void* x, int ldx, const float* w,
                 const float* b, float* scores);
/* dw[c,:] = sum_r dscores[r,c] x[r,:], db[c] = sum_r dscores[r,c] (OVERWRITE); workspace as layernorm_bwd */
int mts_head_bwd_params(void* stream, int dtype, int rows, int D, int n_out, const void* x, int ldx,
                        const float* dscores, float* dw, float* db, void* partial);
/* dx[r,:] (+)= sum_c dscores[r,c] w[c,:] */
int mts_head_bwd_data(void* stream, int dtype, int rows, int D, int n_out, const float* dscores, const float* w,
                      void* dx, int lddx, int accumulate);

/* ---------------------------------------------------------------------------------------------
 * LSTM recurrence with packed-sequence semantics.
 * Replaces: aten::lstm under pack_padded_sequence / pad_packed_sequence
 * (models/NeuralArchitectures.py:98-115): gate order i,f,g,o, zero initial state, rows >= len are 0,
 * the reverse direction starts at each document's own last sentence.
 * xproj: [B*L, ndir*4H] act dtype = x W_ih^T + b_ih for both directions (direction d owns columns
 *   d*4H..(d+1)*4H), produced by mts_gemm.  w_hh: fp32 [ndir, 4H, H]; b_hh: fp32 [ndir, 4H] (may be NULL),
 *   added to the gate pre-activations in fp32 inside the recurrence.
 * out: [B*L, ndir*H] act dtype.  gates (saved for backward): act dtype, B*L*ndir*4H elements, post-activation
 *   i,f,g,o; cells: fp32, B*L*ndir*H elements.  Both are OPAQUE saved state handed unchanged to mts_lstm_bwd under the same options
 *   (the generic kernels keep them as [B*L, ndir*4H] / [B*L, ndir*H]; the bf16 CU-quad recurrences keep step-major blocks).
 * ------------------------------------------------------------------------------------------- */
/* workspace (both calls): mts_lstm_workspace(dtype, B, L, H, ndir) bytes */
size_t mts_lstm_workspace(int dtype, int B, int L, int H, int ndir);
int mts_lstm_fwd(void* stream, int dtype, int B, int L, int H, int ndir, const void* xproj, const float* w_hh,
                 const float* b_hh, const int32_t* lengths, void* out, void* gates, float* cells, void* workspace);
/* dxproj: [B*L, ndir*4H] act dtype (gradient wrt pre-activation gates); dw_hh fp32 [ndir,4H,H] OVERWRITTEN. */
int mts_lstm_bwd(void* stream, int dtype, int B, int L, int H, int ndir, const float* w_hh, const int32_t* lengths,
                 const void* out, const void* gates, const float* cells, const void* dout, void* dxproj,
                 float* dw_hh, void* workspace);
/* mts_lstm_bwd in two calls (recurrence | h_{t-1} + dW_hh), so that the weight-gradient half can run on another stream while the caller's next
 * dependent launch follows the recurrence directly.  `workspace`: the same buffer for both calls of a layer, untouched in between. */
int mts_lstm_bwd_recurrence(void* stream, int dtype, int B, int L, int H, int ndir, const float* w_hh, const int32_t* lengths,
                            const void* out, const void* gates, const float* cells, const void* dout, void* dxproj, void* workspace);
int mts_lstm_bwd_whh(void* stream, int dtype, int B, int L, int H, int ndir, const int32_t* lengths, const void* out, const void* dxproj,
                     float* dw_hh, void* workspace);

/* ---------------------------------------------------------------------------------------------
 * CRF head.  Replaces models/CRF.py:98-240 (fc output = `feats` is produced by mts_gemm/mts_head_fwd).
 * feats: fp32 [B, L, C]; C = num_tags + 2 (START = C-2, STOP = C-1); trans fp32 [C, C], T[i,j] = j -> i.
 * ------------------------------------------------------------------------------------------- */
/* loss_out[0] = mean_b(logZ_b - gold_b); dfeats [B,L,C] and dtrans [C,C] are OVERWRITTEN (may be NULL). */
size_t mts_crf_workspace(int B, int L, int C);
int mts_crf_nll(void* stream, int B, int L, int C, const float* feats, const float* tags, int Lt,
                const int32_t* lengths, const float* trans, float* loss_out, float* dfeats, float* dtrans,
                float* workspace /* mts_crf_workspace(B, L, C) bytes */);
/* best_score fp32 [B]; paths int32 [B, L] (entries >= length undefined); bp_ws int32 [B, L, C] */
int mts_crf_viterbi(void* stream, int B, int L, int C, const float* feats, const int32_t* lengths,
                    const float* trans, float* best_score, int32_t* paths, int32_t* bp_ws);

/* ---------------------------------------------------------------------------------------------
 * Optimizer step over a flat fp32 parameter buffer.
 * Replaces torch.optim.Adam(eps=1e-7) / SGD(momentum .9, wd 1e-4) of models/lightning_model.py:759-765.
 * grad_scale multiplies the gradient first (1/world_size after an RCCL sum all-reduce).
 * If bf16_copy != NULL the updated parameters are also written as bf16 (next step's GEMM weights).
 * ------------------------------------------------------------------------------------------- */
int mts_adam_step(void* stream, size_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq,
                  float lr, float beta1, float beta2, float eps, int step, float grad_scale, void* bf16_copy);
int mts_sgd_step(void* stream, size_t n, float* param, const float* grad, float* momentum_buf, float lr,
                 float momentum, float weight_decay, int first_step, float grad_scale, void* bf16_copy);
/* x[0..n) *= scale (fp32).  Token-weighted data parallelism: the reference's loss is a mean over the LOCAL batch's valid
 * sentences (models/CRF.py:352); multiplying d loss / d scores by world * n_local / n_global before the SUM all-reduce (and
 * grad_scale = 1/world in the optimizer) gives the single-process gradient of the global batch. */
int mts_scale(void* stream, size_t n, float* x, float scale);

#ifdef __cplusplus
}
#endif
#endif /* MTS_H */
