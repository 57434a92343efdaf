// Restricted-window ("band") self-attention on the gfx950 matrix cores: bf16 operands, fp32 accumulation.
// Same semantics and the same ABI buffers as band_attn.hip (probs / dscores fp32 [B*L, heads, slots], slot c of
// query i <-> key j = i - radius + c); covers head dims that are multiples of 32 and radius <= 63.
//
// Work decomposition: one 256-thread workgroup per (document, head, 128 rows); each of the 4 waves owns 32 query
// rows (32 key rows in the dK/dV pass) and the 16*NKB opposite rows that its band can touch.
//
//   qk_phase   T^T[key, query] = sum_d X[key, d] Y[query, d]   (scores: X=K, Y=Q; dP: X=V, Y=dCtx)
//              both operands are read straight from HBM/L2 in MFMA fragment order (lane = row l&15, 8 consecutive d
//              at 8*(l>>4)): 16-byte loads, nothing staged.  Computing the TRANSPOSE puts the 16*NKB coefficients of one
//              query in 4 lanes x (4*NKB) registers, so the softmax needs two cross-lane steps and -- the point of the
//              transposition -- the accumulator registers are already a valid B operand for the next product.
//   cv_phase   O^T[d, n] = sum_k R[k, d] C[k, n]                 (ctx: R=V, C=P; dQ: R=K, C=dS; dV: R=dCtx, C=P; dK: R=Q, C=dS)
//              R rows are staged ONCE per workgroup into LDS by LDS-DMA (global_load_lds, 16 B per lane, no VGPRs) and
//              read with ds_read_b64_tr_b16 (the transposing LDS read: a lane gets 4 consecutive k for its d column).
//              The contraction index inside a lane group is permuted (registers of two 16-row accumulator blocks form one
//              32-deep k-step); the R fragment is fetched with the same permutation, so no data is shuffled.
//
// Algorithmic HBM bytes per (row, head), bf16: forward reads q,k,v (6*hd) + writes ctx (2*hd) + probs (4*slots);
// backward reads q,k,v,dctx (8*hd), probs, writes dqkv (6*hd) + dscores.  The matrix-core work is ~3 us per pass at
// the BASELINE shape, i.e. the kernels are HBM/L2-latency bound by construction.
#include "band_common.h"

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

#define WROWS 32                 // rows per wave
#define TROWS (4 * WROWS)        // rows per workgroup

// ---- LDS-DMA staging: image row r <- matrix row clamp(first + r, 0, L-1) of one head's slice; rows are packed (hd*2 bytes).
// Out-of-document rows are clamped (finite data); their coefficients are exactly 0.
template <int KK>
__device__ __forceinline__ void stage_dma(char* img, const bf16_t* __restrict__ doc_base, int ld, int first, int nrows, int L) {
  constexpr int CPR = 4 * KK;    // 16-byte chunks per row
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int total = nrows * CPR;
  for (int base = wave * 64; base < total; base += 256) {
    const int idx = min(base + lane, total - 1);
    const int r = idx / CPR, ch = idx - r * CPR;
    const int j = min(max(first + r, 0), L - 1);
    const bf16_t* src = doc_base + (size_t)j * ld + ch * 8;
    __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(img + base * 16), 16, 0, 0);
  }
}

template <int KK, int NKB>
__device__ __forceinline__ void qk_phase(const bf16_t* __restrict__ xrows, int ldx, int k0, const bf16_t* __restrict__ yrows, int ldy, int q0,
                                         int L, int lane, f32x4 (&acc)[NKB][2]) {
  const int l15 = lane & 15, g = lane >> 4;
  bf16x8 yq[2][KK];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int row = min(max(q0 + 16 * qb + l15, 0), L - 1);
    const bf16_t* p = yrows + (size_t)row * ldy + 8 * g;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) yq[qb][kk] = *reinterpret_cast<const bf16x8*>(p + 32 * kk);
  }
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    const int row = min(max(k0 + 16 * kb + l15, 0), L - 1);
    const bf16_t* p = xrows + (size_t)row * ldx + 8 * g;
    bf16x8 xk[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) xk[kk] = *reinterpret_cast<const bf16x8*>(p + 32 * kk);
#pragma unroll
    for (int kk = 0; kk < KK; ++kk)
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) acc[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xk[kk], yq[qb][kk], acc[kb][qb], 0, 0, 0);
  }
}

// o[db][nb] += sum over the NS k-steps; lo/hi = image rows (incl. the lane group's 4g / 8g part) of elements 0..3 / 4..7
template <int KK, int NS>
__device__ __forceinline__ void cv_phase(const char* img, const int (&lo)[NS], const int (&hi)[NS], const bf16x8 (&coef)[NS][2], int lane,
                                         f32x4 (&o)[2 * KK][2]) {
  constexpr int RS = 64 * KK;    // row stride in bytes
  const int r = lane & 15, q = r >> 2, p = r & 3;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const char* plo = img + (lo[s] + q) * RS + 8 * p;
    const char* phi = img + (hi[s] + q) * RS + 8 * p;
#pragma unroll
    for (int db = 0; db < 2 * KK; ++db) {
      const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(plo + 32 * db));
      const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(phi + 32 * db));
      const bf16x8 av = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) o[db][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, coef[s][nb], o[db][nb], 0, 0, 0);
    }
  }
}

// O^T accumulators (lane: row n = n0 + 16 nb + l15, columns 16 db + 4g .. +3) -> out[row][0..hd).  The staged-row image is
// dead once every wave has finished its cv_phase, so each wave parks its 32 x hd tile there (row stride hd*2+16 bytes)
// and writes it out as whole 16-byte chunks of contiguous rows; with `slab` the column sums of the tile as stored (bf16)
// are combined over the 4 waves and left in slab[0..hd) -- the q/k/v bias gradient, reduced over tiles by the caller.
template <int KK>
__device__ __forceinline__ void emit_rows(char* img, float* red, bf16_t* __restrict__ out, int ldo, int n0, int L, float scale,
                                          const f32x4 (&o)[2 * KK][2], float* __restrict__ slab) {
  constexpr int HD = 32 * KK, RSO = HD * 2 + 16, CPR = 4 * KK;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, g = lane >> 4;
  char* mine = img + wave * WROWS * RSO;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    char* p = mine + (16 * nb + l15) * RSO + 8 * g;
#pragma unroll
    for (int db = 0; db < 2 * KK; ++db) {
      bf16x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (bf16_t)(o[db][nb][e] * scale);
      *reinterpret_cast<bf16x4*>(p + 32 * db) = v;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // same wave, in-order LDS: the tile is complete
#pragma unroll
  for (int it = 0; it < CPR / 2; ++it) {
    const int idx = it * 64 + lane, r = idx / CPR, ch = idx - r * CPR;
    const uint4 v = *reinterpret_cast<const uint4*>(mine + r * RSO + ch * 16);
    if (n0 + r < L) *reinterpret_cast<uint4*>(out + (size_t)(n0 + r) * ldo + ch * 8) = v;
  }
  if (slab) {
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    if (lane < HD / 4) {
#pragma unroll 8
      for (int r = 0; r < WROWS; ++r) {
        const uint2 u = *reinterpret_cast<const uint2*>(mine + r * RSO + lane * 8);
        cs[0] += bf16_lo(u.x); cs[1] += bf16_hi(u.x); cs[2] += bf16_lo(u.y); cs[3] += bf16_hi(u.y);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) red[wave * HD + 4 * lane + j] = cs[j];
    }
    __syncthreads();
    if (threadIdx.x < HD) {
      const int t = threadIdx.x;
      slab[t] = (red[t] + red[HD + t]) + (red[2 * HD + t] + red[3 * HD + t]);
    }
  }
}

__device__ __forceinline__ float quad_max(float v) {   // over the 4 lane groups that share a query column
  v = fmaxf(v, __shfl_xor(v, 16, 64)); v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}
__device__ __forceinline__ float quad_sum(float v) {
  v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
  return v;
}

template <int N>
__device__ __forceinline__ void zero_acc(f32x4 (&x)[N][2]) {
#pragma unroll
  for (int i = 0; i < N; ++i) { x[i][0] = f32x4{0.f, 0.f, 0.f, 0.f}; x[i][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <int KK, int NKB, int OCC>
__global__ __launch_bounds__(256, OCC) void band_mfma_fwd_kernel(const BandArgs a) {
  extern __shared__ __attribute__((aligned(16))) char img[];
  const int ntiles = (a.L + TROWS - 1) / TROWS;
  int tile, h, b;
  decode_block(ntiles, a.heads, ntiles * a.heads * a.B, tile, h, b);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, g = lane >> 4;
  const int t0 = tile * TROWS, q0 = t0 + WROWS * wave;
  const int w = a.radius, W = 2 * w + 1, hd = a.hd, ld = 3 * a.D;
  const DocView doc = doc_view(a, b);
  const int L = doc.Lb;                     // rows of this document (packed batches: its length)
  if (t0 >= L) return;                     // packed batches: this document has no rows in the tile (uniform: before any barrier)
  const int len = a.lengths ? min(a.lengths[b], L) : L;
  const bf16_t* qbase = reinterpret_cast<const bf16_t*>(a.qkv) + (size_t)doc.base * ld + h * hd;

  stage_dma<KK>(img, qbase + 2 * a.D, ld, t0 - w, TROWS - WROWS + 16 * NKB, L);      // V rows

  f32x4 acc[NKB][2];
  zero_acc<NKB>(acc);
  qk_phase<KK, NKB>(qbase + a.D, ld, q0 - w, qbase, ld, q0, L, lane, acc);

  bf16x8 coef[NKB / 2][2];
  float pr[NKB][2][4];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int i = q0 + 16 * qb + l15;
    const bool qok = i < len;
    float m = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int jl = 16 * kb + 4 * g + r, c = jl - 16 * qb - l15, j = q0 - w + jl;
        const bool ok = qok && c >= 0 && c < W && j >= 0 && j < len;
        const float s = ok ? acc[kb][qb][r] : -INFINITY;
        pr[kb][qb][r] = s;
        m = fmaxf(m, s);
      }
    m = quad_max(m);
    float sum = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s = pr[kb][qb][r];
        const float e = (s > -INFINITY) ? __expf(s - m) : 0.f;
        pr[kb][qb][r] = e;
        sum += e;
      }
    sum = quad_sum(sum);
    const float inv = qok ? 1.0f / sum : 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) pr[kb][qb][r] *= inv;
#pragma unroll
    for (int s = 0; s < NKB / 2; ++s) {
      bf16x8 c8;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p0 = pr[2 * s][qb][r], p1 = pr[2 * s + 1][qb][r];
        if (a.drop_thr) {                                  // dropout acts on what multiplies V; the saved probabilities stay whole
          const int c0 = 16 * (2 * s) + 4 * g + r - 16 * qb - l15;
          p0 = band_keep(a, doc.base + i, h, c0) ? p0 * a.drop_scale : 0.f;
          p1 = band_keep(a, doc.base + i, h, c0 + 16) ? p1 * a.drop_scale : 0.f;
        }
        c8[r] = (bf16_t)p0; c8[4 + r] = (bf16_t)p1;
      }
      coef[s][qb] = c8;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the V image (older than every fragment load) has landed
  __syncthreads();
  // probabilities for the backward (after the wait so the stores do not sit in front of it)
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int i = q0 + 16 * qb + l15;
    if (i < L) {
      float* prow = a.probs + ((size_t)(doc.base + i) * a.heads + h) * a.slots;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * kb + 4 * g + r - 16 * qb - l15;
          if (c >= 0 && c < a.slots) prow[c] = pr[kb][qb][r];
        }
    }
  }
  f32x4 o[2 * KK][2];
  zero_acc<2 * KK>(o);
  int lo[NKB / 2], hi[NKB / 2];
#pragma unroll
  for (int s = 0; s < NKB / 2; ++s) { lo[s] = WROWS * wave + 32 * s + 4 * g; hi[s] = lo[s] + 16; }
  cv_phase<KK, NKB / 2>(img, lo, hi, coef, lane, o);
  emit_rows<KK>(img, nullptr, reinterpret_cast<bf16_t*>(a.ctx) + (size_t)doc.base * a.D + h * hd, a.D, q0, L, 1.f, o, nullptr);
}

// ------------------------------------------------------------------------------------------------
// backward pass A (per query rows): dP = dCtx.V^T, dS = P*(dP - rowsum(P*dP)) (stored), dQ = q_scale * dS.K
// ------------------------------------------------------------------------------------------------
template <int KK, int NKB, int OCC>
__global__ __launch_bounds__(256, OCC) void band_mfma_bwd_q_kernel(const BandArgs a) {
  extern __shared__ __attribute__((aligned(16))) char img[];
  const int ntiles = (a.L + TROWS - 1) / TROWS;
  int tile, h, b;
  decode_block(ntiles, a.heads, ntiles * a.heads * a.B, tile, h, b);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l15 = lane & 15, g = lane >> 4;
  const int t0 = tile * TROWS, q0 = t0 + WROWS * wave;
  const int w = a.radius, W = 2 * w + 1, hd = a.hd, ld = 3 * a.D;
  const DocView doc = doc_view(a, b);
  const int L = doc.Lb;                     // rows of this document (packed batches: its length)
  if (t0 >= L) {                           // nothing to do, but the bias-gradient slab row of this (document, tile) is summed later
    if (a.bias_slab && threadIdx.x < hd) a.bias_slab[(size_t)(b * ntiles + tile) * ld + h * hd + threadIdx.x] = 0.f;
    return;
  }
  const bf16_t* qbase = reinterpret_cast<const bf16_t*>(a.qkv) + (size_t)doc.base * ld + h * hd;
  const bf16_t* dcbase = reinterpret_cast<const bf16_t*>(a.dctx) + (size_t)doc.base * a.D + h * hd;

  stage_dma<KK>(img, qbase + a.D, ld, t0 - w, TROWS - WROWS + 16 * NKB, L);          // K rows

  float pr[NKB][2][4];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int i = q0 + 16 * qb + l15;
    const float* prow = a.probs + ((size_t)(doc.base + min(i, L - 1)) * a.heads + h) * a.slots;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = 16 * kb + 4 * g + r - 16 * qb - l15;
        pr[kb][qb][r] = (i < L && c >= 0 && c < W) ? prow[c] : 0.f;   // (the unconditional-load form of load_coef_T measured slower HERE)
      }
  }
  f32x4 acc[NKB][2];
  zero_acc<NKB>(acc);
  qk_phase<KK, NKB>(qbase + 2 * a.D, ld, q0 - w, dcbase, a.D, q0, L, lane, acc);      // dP^T

  bf16x8 coef[NKB / 2][2];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    float delta = 0.f;
    if (a.drop_thr) {                                      // gradient through the dropout: dP = keep ? dP_dropped / (1-p) : 0
      const int i = q0 + 16 * qb + l15;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * kb + 4 * g + r - 16 * qb - l15;
          acc[kb][qb][r] = band_keep(a, doc.base + i, h, c) ? acc[kb][qb][r] * a.drop_scale : 0.f;
        }
    }
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) delta = fmaf(pr[kb][qb][r], acc[kb][qb][r], delta);
    delta = quad_sum(delta);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) pr[kb][qb][r] *= (acc[kb][qb][r] - delta);
#pragma unroll
    for (int s = 0; s < NKB / 2; ++s) {
      bf16x8 c8;
#pragma unroll
      for (int r = 0; r < 4; ++r) { c8[r] = (bf16_t)pr[2 * s][qb][r]; c8[4 + r] = (bf16_t)pr[2 * s + 1][qb][r]; }
      coef[s][qb] = c8;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    const int i = q0 + 16 * qb + l15;
    if (i < L) {
      float* dsrow = a.dscores + ((size_t)(doc.base + i) * a.heads + h) * a.slots;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * kb + 4 * g + r - 16 * qb - l15;
          if (c >= 0 && c < a.slots) dsrow[c] = pr[kb][qb][r];
        }
    }
  }
  f32x4 o[2 * KK][2];
  zero_acc<2 * KK>(o);
  int lo[NKB / 2], hi[NKB / 2];
#pragma unroll
  for (int s = 0; s < NKB / 2; ++s) { lo[s] = WROWS * wave + 32 * s + 4 * g; hi[s] = lo[s] + 16; }
  cv_phase<KK, NKB / 2>(img, lo, hi, coef, lane, o);
  float* red = reinterpret_cast<float*>(img + a.img_bytes);
  float* slab = a.bias_slab ? a.bias_slab + (size_t)(b * ntiles + tile) * ld + h * hd : nullptr;
  emit_rows<KK>(img, red, reinterpret_cast<bf16_t*>(a.dqkv) + (size_t)doc.base * ld + h * hd, ld, q0, L, a.q_scale, o, slab);
}

// ------------------------------------------------------------------------------------------------
// backward pass B (per KEY rows): dV = P^T.dCtx, dK = dS^T.Q.  Key j sees queries i = j - w .. j + w; the coefficient
// of (i, j) sits in slot c = j - i + w of row i.  No atomics: bitwise reproducible.
// ------------------------------------------------------------------------------------------------
template <int NS>
__device__ __forceinline__ void load_coef_T(const float* __restrict__ X, size_t row_stride, int ibase, int j0, int w, int W, int L, int lane,
                                            bf16x8 (&coef)[NS][2], const BandArgs& a, int base_row, int h, bool dropped) {
  const int l15 = lane & 15, g = lane >> 4;
  float v[NS][2][8];
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      const int j = j0 + 16 * nb + l15;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int i = ibase + 32 * s + 8 * g + e, c = j - i + w;
        const bool ok = i >= 0 && i < L && j < L && c >= 0 && c < W;
        // unconditional load from a clamped (always valid) address, then a select: a load inside a divergent branch makes the
        // compiler wait for every outstanding memory operation before the next one (32 serial round trips here)
        float x = X[(size_t)min(max(i, 0), L - 1) * row_stride + min(max(c, 0), W - 1)];
        x = ok ? x : 0.f;
        if (dropped && ok) x = band_keep(a, base_row + i, h, c) ? x * a.drop_scale : 0.f;        // dV sees the dropped probabilities
        v[s][nb][e] = x;
      }
    }
#pragma unroll
  for (int s = 0; s < NS; ++s)
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      bf16x8 c8;
#pragma unroll
      for (int e = 0; e < 8; ++e) c8[e] = (bf16_t)v[s][nb][e];
      coef[s][nb] = c8;
    }
}

template <int KK, int NKB, int OCC>
__global__ __launch_bounds__(256, OCC) void band_mfma_bwd_kv_kernel(const BandArgs a) {
  extern __shared__ __attribute__((aligned(16))) char img[];
  constexpr int NS = NKB / 2;
  const int ntiles = (a.L + TROWS - 1) / TROWS;
  int tile, h, b;
  decode_block(ntiles, a.heads, ntiles * a.heads * a.B, tile, h, b);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane >> 4;
  const int t0 = tile * TROWS, j0 = t0 + WROWS * wave;
  const int w = a.radius, W = 2 * w + 1, hd = a.hd, ld = 3 * a.D;
  const DocView doc = doc_view(a, b);
  const int L = doc.Lb;                     // rows of this document (packed batches: its length)
  if (t0 >= L) {
    if (a.bias_slab && threadIdx.x < hd) {
      float* z = a.bias_slab + (size_t)(b * ntiles + tile) * ld + h * hd + threadIdx.x;
      z[a.D] = 0.f; z[2 * a.D] = 0.f;
    }
    return;
  }
  const bf16_t* qbase = reinterpret_cast<const bf16_t*>(a.qkv) + (size_t)doc.base * ld + h * hd;
  const bf16_t* dcbase = reinterpret_cast<const bf16_t*>(a.dctx) + (size_t)doc.base * a.D + h * hd;
  const size_t xrow = (size_t)a.heads * a.slots;
  const float* pb = a.probs + ((size_t)doc.base * a.heads + h) * a.slots;
  const float* dsb = a.dscores + ((size_t)doc.base * a.heads + h) * a.slots;
  bf16_t* out = reinterpret_cast<bf16_t*>(a.dqkv) + (size_t)doc.base * ld + h * hd;
  const int nrows = TROWS - WROWS + 16 * NKB;

  int lo[NS], hi[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) { lo[s] = WROWS * wave + 32 * s + 8 * g; hi[s] = lo[s] + 4; }
  bf16x8 coef[NS][2];
  f32x4 o[2 * KK][2];

  stage_dma<KK>(img, dcbase, a.D, t0 - w, nrows, L);                                  // dCtx rows
  load_coef_T<NS>(pb, xrow, j0 - w, j0, w, W, L, lane, coef, a, doc.base, h, a.drop_thr != 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  zero_acc<2 * KK>(o);
  cv_phase<KK, NS>(img, lo, hi, coef, lane, o);
  float* red = reinterpret_cast<float*>(img + a.img_bytes);
  float* slab = a.bias_slab ? a.bias_slab + (size_t)(b * ntiles + tile) * ld + h * hd : nullptr;
  emit_rows<KK>(img, red, out + 2 * a.D, ld, j0, L, 1.f, o, slab ? slab + 2 * a.D : nullptr);      // dV

  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();                                                                    // every wave is done with the dCtx image
  stage_dma<KK>(img, qbase, ld, t0 - w, nrows, L);                                    // (scaled) q rows
  load_coef_T<NS>(dsb, xrow, j0 - w, j0, w, W, L, lane, coef, a, doc.base, h, false);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  zero_acc<2 * KK>(o);
  cv_phase<KK, NS>(img, lo, hi, coef, lane, o);
  emit_rows<KK>(img, red, out + a.D, ld, j0, L, 1.f, o, slab ? slab + a.D : nullptr);              // dK
}

// ------------------------------------------------------------------------------------------------
// backward in ONE pass (radius <= 15, head dim <= 224): q, k, v, dCtx and the probabilities are read once, dq, dk, dv written once,
// dS never leaves the chip.  A 512-thread workgroup owns FROWS = 256 rows of one (document, head): every document of at most 256
// sentences is ONE tile (no halo at all: the BASELINE shape); longer documents are cut into tiles of 192 key rows whose 224 query
// rows include a 16-row halo on either side (key j needs the coefficients of queries j - w .. j + w, and a query's dS needs its whole
// window, so halo queries are recomputed by both neighbours: 224 / 192 of the per-query work).
//   part 1 (wave = 32 query rows, as band_mfma_bwd_q_kernel): dP^T = V.dCtx^T (dCtx fragments out of the dCtx image, which part 2 needs
//           anyway: dCtx is fetched once), dS = P (dP - rowsum(P dP)), and -- after dV -- dQ = q_scale dS.K from the K
//           image; P (dropped form) and dS are left in LDS as bf16, transposed: XT[key - t0][query & 31] (a key's 2w + 1 queries are
//           distinct modulo 32), 16 KB each;
//   part 2 (wave = 32 key rows, as band_mfma_bwd_kv_kernel): the B operand of key j is 8 consecutive queries = ONE 16-byte LDS read of
//           XT (the window is anchored at j0 - 16, a multiple of 8), masked to |i - j| <= w; dV = P^T.dCtx and dK = dS^T.Q from the
//           dCtx and Q images.  The three 256-row images share one LDS region (3 x 112 KB do not fit), staged one after the other.
// ------------------------------------------------------------------------------------------------
#define FROWS 256

template <int KK>
__device__ __forceinline__ void stage_dma8(char* img, const bf16_t* __restrict__ doc_base, int ld, int first, int nrows, int L) {
  constexpr int CPR = 4 * KK;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int total = nrows * CPR;                         // nrows is a multiple of 16: whole 64-chunk pieces
  for (int base = wave * 64; base < total; base += 512) {
    const int idx = base + lane;
    const int r = idx / CPR, ch = idx - r * CPR;
    const int j = min(max(first + r, 0), L - 1);
    const bf16_t* src = doc_base + (size_t)j * ld + ch * 8;
    __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(img + base * 16), 16, 0, 0);
  }
}

// cv_phase with the image rows given per lane (already including the lane's row-in-quad and clamped into the staged rows)
template <int KK, int NS>
__device__ __forceinline__ void cv_phase_rows(const char* img, const int (&rlo)[NS], const int (&rhi)[NS], const bf16x8 (&coef)[NS][2], int lane,
                                              f32x4 (&o)[2 * KK][2]) {
  constexpr int RS = 64 * KK;
  const int p = lane & 3;
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const char* plo = img + rlo[s] * RS + 8 * p;
    const char* phi = img + rhi[s] * RS + 8 * p;
#pragma unroll
    for (int db = 0; db < 2 * KK; ++db) {
      const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(plo + 32 * db));
      const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(phi + 32 * db));
      const bf16x8 av = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7));
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) o[db][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, coef[s][nb], o[db][nb], 0, 0, 0);
    }
  }
}

// emit_rows for 8 waves; only rows [rlo, rhi) of the wave's 32 (wave-uniform bounds) are stored and enter the column sums
template <int KK>
__device__ __forceinline__ void emit_rows8(char* img, float* red, bf16_t* __restrict__ out, int ldo, int n0, int rlo, int rhi, float scale,
                                           const f32x4 (&o)[2 * KK][2], float* __restrict__ slab) {
  constexpr int HD = 32 * KK, RSO = HD * 2 + 16, CPR = 4 * KK;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l15 = lane & 15, g = lane >> 4;
  char* mine = img + wave * WROWS * RSO;
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();                                         // every wave has finished reading the image
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    char* p = mine + (16 * nb + l15) * RSO + 8 * g;
#pragma unroll
    for (int db = 0; db < 2 * KK; ++db) {
      bf16x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = (bf16_t)(o[db][nb][e] * scale);
      *reinterpret_cast<bf16x4*>(p + 32 * db) = v;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int it = 0; it < CPR / 2; ++it) {
    const int idx = it * 64 + lane, r = idx / CPR, ch = idx - r * CPR;
    const uint4 v = *reinterpret_cast<const uint4*>(mine + r * RSO + ch * 16);
    if (r >= rlo && r < rhi) *reinterpret_cast<uint4*>(out + (size_t)(n0 + r) * ldo + ch * 8) = v;
  }
  if (slab) {
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
    if (lane < HD / 4) {
      for (int r = rlo; r < rhi; ++r) {
        const uint2 u = *reinterpret_cast<const uint2*>(mine + r * RSO + lane * 8);
        cs[0] += bf16_lo(u.x); cs[1] += bf16_hi(u.x); cs[2] += bf16_lo(u.y); cs[3] += bf16_hi(u.y);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) red[wave * HD + 4 * lane + j] = cs[j];
    }
    __syncthreads();
    if (threadIdx.x < HD) {
      const int t = threadIdx.x;
      slab[t] = ((red[t] + red[HD + t]) + (red[2 * HD + t] + red[3 * HD + t])) + ((red[4 * HD + t] + red[5 * HD + t]) + (red[6 * HD + t] + red[7 * HD + t]));
    }
  }
}

template <int KK>
struct FusedLds {
  static constexpr int HD = 32 * KK;
  static constexpr int PARK = 8 * WROWS * (HD * 2 + 16), ROWS = FROWS * HD * 2;
  static constexpr int IMG = (((PARK > ROWS ? PARK : ROWS) + 1023) / 1024) * 1024;
  static constexpr int XT = FROWS * 64;                  // one transposed coefficient array: 256 keys x 32 queries x bf16
  static constexpr int TOTAL = IMG + 2 * XT + 8 * HD * 4;
};

template <int KK>
__global__ __launch_bounds__(512, 1) void band_mfma_bwd_fused_kernel(const BandArgs a, int tk, int halo, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char img[];
  constexpr int NKB = 4, NS = 2;
  char* xp = img + FusedLds<KK>::IMG;
  char* xs = xp + FusedLds<KK>::XT;
  float* red = reinterpret_cast<float*>(xs + FusedLds<KK>::XT);
  int tile, h, b;
  decode_block(ntiles, a.heads, ntiles * a.heads * a.B, tile, h, b);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l15 = lane & 15, g = lane >> 4;
  const int t0 = tile * tk, qs = t0 - halo, fq = tk + 2 * halo;
  const int w = a.radius, W = 2 * w + 1, hd = a.hd, ld = 3 * a.D;
  const DocView doc = doc_view(a, b);
  const int L = doc.Lb;
  float* slab = a.bias_slab ? a.bias_slab + (size_t)(b * ntiles + tile) * ld + h * hd : nullptr;
  if (t0 >= L) {
    if (slab && threadIdx.x < hd) { slab[threadIdx.x] = 0.f; slab[a.D + threadIdx.x] = 0.f; slab[2 * a.D + threadIdx.x] = 0.f; }
    return;
  }
  const int khi = min(t0 + tk, L);                         // this workgroup's key rows (and the queries whose dQ it writes): [t0, khi)
  const int io = halo ? qs - w : 0;                        // matrix row of image row 0
  const int nst = min(FROWS, ((L - io + 15) / 16) * 16);   // staged rows (the rest of the image is never addressed)
  const bf16_t* qbase = reinterpret_cast<const bf16_t*>(a.qkv) + (size_t)doc.base * ld + h * hd;
  const bf16_t* dcbase = reinterpret_cast<const bf16_t*>(a.dctx) + (size_t)doc.base * a.D + h * hd;
  bf16_t* out = reinterpret_cast<bf16_t*>(a.dqkv) + (size_t)doc.base * ld + h * hd;
  const int rq = (lane & 15) >> 2;                         // the lane's row inside a 4-row group of the transposing read

  stage_dma8<KK>(img, dcbase, a.D, io, nst, L);            // dCtx rows: B operand of dP in part 1, R operand of dV in part 2

  // ---- part 1: this wave's 32 query rows ------------------------------------------------------------------------------------------
  const int q0 = qs + WROWS * wave;
  const bool act1 = WROWS * wave < fq && q0 < L && q0 + WROWS > 0;
  bf16x8 coefq[NS][2];                                     // dS^T of this wave's queries: B operand of dQ (used after the dV phase)
#pragma unroll
  for (int s = 0; s < NS; ++s) { coefq[s][0] = bf16x8{}; coefq[s][1] = bf16x8{}; }
  float pr[NKB][2][4];
  if (act1) {
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int i = q0 + 16 * qb + l15;
      const bool iok = i >= 0 && i < L;
      const float* prow = a.probs + ((size_t)(doc.base + min(max(i, 0), L - 1)) * a.heads + h) * a.slots;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * kb + 4 * g + r - 16 * qb - l15;
          pr[kb][qb][r] = (iok && c >= 0 && c < W) ? prow[c] : 0.f;
        }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the dCtx image has landed
  __syncthreads();
  if (act1) {
    f32x4 acc[NKB][2];
    zero_acc<NKB>(acc);
    {                                                      // dP^T = V . dCtx^T: V fragments straight from L2, dCtx fragments out of the image
      constexpr int RS = 64 * KK;
      bf16x8 yq[2][KK];
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        const int row = min(max(q0 - io + 16 * qb + l15, 0), nst - 1);
        const char* py = img + row * RS + 16 * g;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) yq[qb][kk] = *reinterpret_cast<const bf16x8*>(py + 64 * kk);
      }
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb) {
        const int row = min(max(q0 - w + 16 * kb + l15, 0), L - 1);
        const bf16_t* px = qbase + 2 * a.D + (size_t)row * ld + 8 * g;
        bf16x8 xk[KK];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) xk[kk] = *reinterpret_cast<const bf16x8*>(px + 32 * kk);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
          for (int qb = 0; qb < 2; ++qb) acc[kb][qb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xk[kk], yq[qb][kk], acc[kb][qb], 0, 0, 0);
      }
    }
#pragma unroll
    for (int qb = 0; qb < 2; ++qb) {
      const int i = q0 + 16 * qb + l15;
      const bool iok = i >= 0 && i < L;
      float pd[NKB][4];                                    // the probabilities as they multiplied V (dropout applied)
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float x = pr[kb][qb][r];
          if (a.drop_thr) {
            const int c = 16 * kb + 4 * g + r - 16 * qb - l15;
            const bool keep = band_keep(a, doc.base + i, h, c);
            acc[kb][qb][r] = keep ? acc[kb][qb][r] * a.drop_scale : 0.f;
            x = keep ? x * a.drop_scale : 0.f;
          }
          pd[kb][r] = x;
        }
      float delta = 0.f;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) delta = fmaf(pr[kb][qb][r], acc[kb][qb][r], delta);
      delta = quad_sum(delta);
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) pr[kb][qb][r] *= (acc[kb][qb][r] - delta);
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        bf16x8 c8;
#pragma unroll
        for (int r = 0; r < 4; ++r) { c8[r] = (bf16_t)pr[2 * s][qb][r]; c8[4 + r] = (bf16_t)pr[2 * s + 1][qb][r]; }
        coefq[s][qb] = c8;
      }
      // transposed hand-over to part 2: XT[key - t0][query & 31]
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int jl = q0 - w + 16 * kb + 4 * g + r - t0, c = 16 * kb + 4 * g + r - 16 * qb - l15;
          if (iok && jl >= 0 && jl < tk && c >= 0 && c < W) {
            const int off = jl * 64 + (i & 31) * 2;
            *reinterpret_cast<bf16_t*>(xp + off) = (bf16_t)pd[kb][r];
            *reinterpret_cast<bf16_t*>(xs + off) = (bf16_t)pr[kb][qb][r];
          }
        }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // this wave's XT entries are written
  __syncthreads();

  // ---- part 2a: dV of this wave's 32 key rows, out of the dCtx image that is already there ---------------------------------------------
  const int j0 = t0 + WROWS * wave;
  const bool act2 = WROWS * wave < tk && j0 < L;
  const int jhi = min(max(khi - j0, 0), WROWS);
  bf16x8 coef[NS][2];
#pragma unroll
  for (int s = 0; s < NS; ++s) { coef[s][0] = bf16x8{}; coef[s][1] = bf16x8{}; }
  auto load_xt = [&](const char* xt) {
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const int j = j0 + 16 * nb + l15, i8 = j0 - 16 + 32 * s + 8 * g;
        const s16x8 raw = *reinterpret_cast<const s16x8*>(xt + (j - t0) * 64 + (i8 & 31) * 2);
        s16x8 v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int i = i8 + e, c = j - i + w;
          v[e] = (i >= 0 && i < L && j < L && c >= 0 && c < W) ? raw[e] : (short)0;
        }
        coef[s][nb] = __builtin_bit_cast(bf16x8, v);
      }
  };
  int rlo[NS], rhi[NS];
  auto kv_rows = [&]() {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int r0 = j0 - 16 - io + 32 * s + 8 * g + rq;
      rlo[s] = min(max(r0, 0), nst - 1); rhi[s] = min(max(r0 + 4, 0), nst - 1);
    }
  };
  f32x4 o[2 * KK][2];
  zero_acc<2 * KK>(o);
  if (act2) { load_xt(xp); kv_rows(); cv_phase_rows<KK, NS>(img, rlo, rhi, coef, lane, o); }
  emit_rows8<KK>(img, red, out + 2 * a.D, ld, j0, 0, act2 ? jhi : 0, 1.f, o, slab ? slab + 2 * a.D : nullptr);                         // dV

  // ---- part 1, continued: dQ = q_scale dS.K ---------------------------------------------------------------------------------------------
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();                                         // every wave is done with the dCtx image and with its parked tile
  stage_dma8<KK>(img, qbase + a.D, ld, io, nst, L);        // K rows
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  zero_acc<2 * KK>(o);
  if (act1) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
      const int r0 = q0 - w - io + 32 * s + 4 * g + rq;
      rlo[s] = min(max(r0, 0), nst - 1); rhi[s] = min(max(r0 + 16, 0), nst - 1);
    }
    cv_phase_rows<KK, NS>(img, rlo, rhi, coefq, lane, o);
  }
  emit_rows8<KK>(img, red, out, ld, q0, min(max(t0 - q0, 0), WROWS), min(max(khi - q0, 0), WROWS), a.q_scale, o, slab);                // dQ

  // ---- part 2b: dK ------------------------------------------------------------------------------------------------------------------------
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();
  stage_dma8<KK>(img, qbase, ld, io, nst, L);              // (scaled) q rows
  if (act2) load_xt(xs);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  zero_acc<2 * KK>(o);
  if (act2) { kv_rows(); cv_phase_rows<KK, NS>(img, rlo, rhi, coef, lane, o); }
  emit_rows8<KK>(img, red, out + a.D, ld, j0, 0, act2 ? jhi : 0, 1.f, o, slab ? slab + a.D : nullptr);                                 // dK
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
template <typename K>
static int launch(K kernel, BandArgs a, int nkb, hipStream_t st, const char* who) {
  const int nrows = TROWS - WROWS + 16 * nkb;
  a.img_bytes = (int)align_up((size_t)nrows * a.hd * 2, (size_t)1024);      // whole 1-KiB DMA pieces
  const size_t lds = (size_t)a.img_bytes + 4 * (size_t)a.hd * sizeof(float);  // + the 4 waves' column sums
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { mts_set_error("%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e)); return MTS_ERR_LAUNCH; }
  }
  const int nblocks = ceil_div(a.L, TROWS) * a.heads * a.B;
  hipLaunchKernelGGL(kernel, dim3(nblocks), dim3(256), lds, st, a);
  MTS_LAUNCH_CHECK(who);
  return MTS_OK;
}

static int pick_nkb(const BandArgs& a) {
  if (a.hd % 32 != 0 || a.hd > 256) return 0;
  const int need = 2 + a.slots / 16;
  return need <= 4 ? 4 : need <= 6 ? 6 : need <= 10 ? 10 : 0;
}

#define BAND_DISPATCH(KERNEL, NKBV, OCC, ...)                                                         \
  switch (a.hd / 32) {                                                                                \
    case 1: return launch(KERNEL<1, NKBV, OCC>, __VA_ARGS__);                                         \
    case 2: return launch(KERNEL<2, NKBV, OCC>, __VA_ARGS__);                                         \
    case 3: return launch(KERNEL<3, NKBV, OCC>, __VA_ARGS__);                                         \
    case 4: return launch(KERNEL<4, NKBV, OCC>, __VA_ARGS__);                                         \
    case 5: return launch(KERNEL<5, NKBV, OCC>, __VA_ARGS__);                                         \
    case 6: return launch(KERNEL<6, NKBV, OCC>, __VA_ARGS__);                                         \
    case 7: return launch(KERNEL<7, NKBV, OCC>, __VA_ARGS__);                                         \
    case 8: return launch(KERNEL<8, NKBV, OCC>, __VA_ARGS__);                                         \
    default: return -1;                                                                               \
  }

template <int WHICH>
static int dispatch(const BandArgs& a, hipStream_t st, const char* who) {
  const int nkb = pick_nkb(a);
  if (nkb == 4) {
    if (WHICH == 0) { BAND_DISPATCH(band_mfma_fwd_kernel, 4, 2, a, 4, st, who) }
    if (WHICH == 1) { BAND_DISPATCH(band_mfma_bwd_q_kernel, 4, 2, a, 4, st, who) }
    if (WHICH == 2) { BAND_DISPATCH(band_mfma_bwd_kv_kernel, 4, 2, a, 4, st, who) }
  } else if (nkb == 6) {
    if (WHICH == 0) { BAND_DISPATCH(band_mfma_fwd_kernel, 6, 1, a, 6, st, who) }
    if (WHICH == 1) { BAND_DISPATCH(band_mfma_bwd_q_kernel, 6, 1, a, 6, st, who) }
    if (WHICH == 2) { BAND_DISPATCH(band_mfma_bwd_kv_kernel, 6, 1, a, 6, st, who) }
  } else if (nkb == 10) {
    if (WHICH == 0) { BAND_DISPATCH(band_mfma_fwd_kernel, 10, 1, a, 10, st, who) }
    if (WHICH == 1) { BAND_DISPATCH(band_mfma_bwd_q_kernel, 10, 1, a, 10, st, who) }
    if (WHICH == 2) { BAND_DISPATCH(band_mfma_bwd_kv_kernel, 10, 1, a, 10, st, who) }
  }
  return -1;
}

static bool covered(const BandArgs& a) {
  const int k = a.hd / 32;
  return pick_nkb(a) != 0 && k >= 1 && k <= 8;        // head dims 32, 64, .., 256 (768/8 = 96, 1024/8 = 128, 1536/8 = 192, 1792/8 = 224)
}

int mts_band_mfma_fwd(const BandArgs& a, hipStream_t st) {
  if (!covered(a)) return -1;
  return dispatch<0>(a, st, "mts_band_attn_fwd(mfma)");
}

static thread_local int g_band_fused = 1;       // mts_set_option("band_fused_bwd", 0): the two-kernel backward (A/B testing)
void mts_band_set_fused(int on) { g_band_fused = on; }

template <int KK>
static int launch_fused(const BandArgs& a, hipStream_t st, int* slab_rows) {
  auto k = band_mfma_bwd_fused_kernel<KK>;
  const size_t lds = FusedLds<KK>::TOTAL;
  static std::atomic<bool> attr{false};
  if (!attr) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { mts_set_error("mts_band_attn_bwd(fused): hipFuncSetAttribute: %s", hipGetErrorString(e)); return MTS_ERR_LAUNCH; }
    attr = true;
  }
  const int halo = a.L <= FROWS ? 0 : 16, tk = FROWS - (halo ? 64 : 0);      // 256 rows, or 192 key rows inside 224 query rows
  const int ntiles = ceil_div(a.L, tk);
  hipLaunchKernelGGL(k, dim3(ntiles * a.heads * a.B), dim3(512), lds, st, a, tk, halo, ntiles);
  MTS_LAUNCH_CHECK("mts_band_attn_bwd(fused)");
  *slab_rows = a.B * ntiles;
  return MTS_OK;
}

int mts_band_mfma_bwd(const BandArgs& a, hipStream_t st, int* slab_rows) {
  if (!covered(a)) return -1;
  *slab_rows = a.B * ceil_div(a.L, TROWS);
  if (g_band_fused && pick_nkb(a) == 4 && a.hd <= 224) {
    switch (a.hd / 32) {
      case 1: return launch_fused<1>(a, st, slab_rows);
      case 2: return launch_fused<2>(a, st, slab_rows);
      case 3: return launch_fused<3>(a, st, slab_rows);
      case 4: return launch_fused<4>(a, st, slab_rows);
      case 5: return launch_fused<5>(a, st, slab_rows);
      case 6: return launch_fused<6>(a, st, slab_rows);
      case 7: return launch_fused<7>(a, st, slab_rows);
    }
  }
  int rc = dispatch<1>(a, st, "mts_band_attn_bwd(mfma q)");
  if (rc) return rc;
  return dispatch<2>(a, st, "mts_band_attn_bwd(mfma kv)");
}
