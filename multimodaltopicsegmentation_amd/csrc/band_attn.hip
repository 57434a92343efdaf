// Restricted-window ("band") self-attention for gfx950: forward, and backward in two passes.
//
// Work decomposition: one 256-thread workgroup per (document, head, tile of 32 queries).  The tile's
// query rows and, per block of 32 key slots, the 63 key (or value) rows that the tile can see are staged
// once from HBM into LDS with 16-byte coalesced loads (each K/V row is then reused by up to 31 queries out
// of LDS instead of HBM/L2); scores live in registers (4 per lane), the softmax is an 8-lane DPP/shuffle
// reduction, probabilities go through a small LDS tile to the value phase.
//
//   score phase : lane = (query tq = tid/8, key group g = tid%8), 4 slots per lane (c = g + 8t): the query
//                 chunk is read once per 4 key chunks; bf16 uses v_dot2_f32_bf16 on packed pairs.
//   value phase : lane = (query tq, dim group dg = tid%8), 4-element chunks u = dg + 8*uu of the head dim.
//
// Backward: pass A (per query tile) recomputes nothing but dP = dCtx.V^T, forms dS = P*(dP - rowsum(P*dP)),
// stores it, and accumulates dQ = dS.K; pass B (per KEY tile) gathers the transposed coefficient bands
// dS^T / P^T from the stored [row, head, slot] arrays and accumulates dK = dS^T.Q, dV = P^T.dCtx -- no
// atomics, bitwise reproducible.
//
// Semantics (oracle/restatement.py::band_attention; modeling_longformer.py:482-640): key j = i - radius + c
// takes part iff 0 <= j < len_b; a query i >= len_b yields a zero row.
#include <algorithm>
#include "band_common.h"

#define TQ 32
#define KV_ROWS (TQ + 31)

// ---- staging: rows [first, first+nrows) of one head's slice of a [B*L, ld] matrix into LDS; rows outside [0,L) -> 0
template <typename T>
__device__ __forceinline__ void stage_rows(char* dst, int rs, const T* __restrict__ doc_base, int ld, int first, int nrows, int L, int hd) {
  constexpr int VEC = 16 / sizeof(T);
  const int cpr = hd / VEC;                      // 16-byte chunks per row
  for (int idx = threadIdx.x; idx < nrows * cpr; idx += 256) {
    const int r = idx / cpr, ch = idx % cpr;
    const int j = first + r;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (j >= 0 && j < L) v = *reinterpret_cast<const uint4*>(doc_base + (size_t)j * ld + ch * VEC);
    *reinterpret_cast<uint4*>(dst + r * rs + ch * 16) = v;
  }
}

__device__ __forceinline__ float dot16(const uint4& a, const uint4& b, float acc, float) {   // 4 fp32 pairs
  acc = fmaf(__uint_as_float(a.x), __uint_as_float(b.x), acc);
  acc = fmaf(__uint_as_float(a.y), __uint_as_float(b.y), acc);
  acc = fmaf(__uint_as_float(a.z), __uint_as_float(b.z), acc);
  acc = fmaf(__uint_as_float(a.w), __uint_as_float(b.w), acc);
  return acc;
}
__device__ __forceinline__ float dot16(const uint4& a, const uint4& b, float acc, bf16_t) {  // 8 bf16 pairs
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a.x), __builtin_bit_cast(bf16x2, b.x), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a.y), __builtin_bit_cast(bf16x2, b.y), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a.z), __builtin_bit_cast(bf16x2, b.z), acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a.w), __builtin_bit_cast(bf16x2, b.w), acc, false);
  return acc;
}

// s[t] = <A[tq,:], Brows[tq + g + 8t,:]>, t = 0..3
template <typename T>
__device__ __forceinline__ void score_phase(const char* As, const char* Bs, int rs, int hd, int tq, int g, float (&s)[4]) {
  constexpr int VEC = 16 / sizeof(T);
  const int cpr = hd / VEC;
  const char* a = As + tq * rs;
  const char* b0 = Bs + (tq + g) * rs;
  s[0] = s[1] = s[2] = s[3] = 0.f;
  for (int ch = 0; ch < cpr; ++ch) {
    const uint4 av = *reinterpret_cast<const uint4*>(a + ch * 16);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const uint4 bv = *reinterpret_cast<const uint4*>(b0 + (8 * t) * rs + ch * 16);
      s[t] = dot16(av, bv, s[t], T());
    }
  }
}

// acc[uu][0..3] += sum_{cc<ncc} coef[cc] * Rows[row0 + cc][4*(dg + 8*uu) ..]
template <typename T, int MAXU>
__device__ __forceinline__ void accum_phase(const float* coef, int coef_stride, const char* Rows, int rs, int hd, int row0, int dg, int ncc,
                                            float (&acc)[MAXU][4]) {
  const int nch = hd / 4;
  for (int cc = 0; cc < ncc; ++cc) {
    const float p = coef[cc * coef_stride];
    const T* row = reinterpret_cast<const T*>(Rows + (row0 + cc) * rs);
#pragma unroll
    for (int uu = 0; uu < MAXU; ++uu) {
      const int u = dg + 8 * uu;
      if (u < nch) {
        float v[4];
        load4<T>(row + 4 * u, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[uu][j] = fmaf(p, v[j], acc[uu][j]);
      }
    }
  }
}

__device__ __forceinline__ float oct_max(float v) {   // reduce over the 8 lanes that share a query
  v = fmaxf(v, __shfl_xor(v, 1, 64)); v = fmaxf(v, __shfl_xor(v, 2, 64)); v = fmaxf(v, __shfl_xor(v, 4, 64));
  return v;
}
__device__ __forceinline__ float oct_sum(float v) {
  v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
  return v;
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <typename T, int MAXU>
__global__ __launch_bounds__(256) void band_fwd_kernel(const BandArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qs = smem;
  char* KVs = Qs + TQ * a.rs;
  float* Ps = reinterpret_cast<float*>(KVs + KV_ROWS * a.rs);

  const int ntiles = (a.L + TQ - 1) / TQ;
  int tile, h, b;
  decode_block(ntiles, a.heads, ntiles * a.heads * a.B, tile, h, b);
  const DocView doc = doc_view(a, b);
  const int i0 = tile * TQ;
  const int len = a.lengths ? min(a.lengths[b], a.L) : a.L;
  const int tq = threadIdx.x >> 3, g = threadIdx.x & 7;
  const int i = i0 + tq;
  const int w = a.radius, W = 2 * w + 1, hd = a.hd, ld = 3 * a.D;
  const T* qbase = reinterpret_cast<const T*>(a.qkv) + (size_t)doc.base * ld + h * hd;
  const int nsb = a.slots / 32;

  stage_rows<T>(Qs, a.rs, qbase, ld, i0, TQ, doc.Lb, hd);
  for (int sb = 0; sb < nsb; ++sb) {
    __syncthreads();   // previous block's readers are done with KVs (and Qs is complete after the first pass)
    stage_rows<T>(KVs, a.rs, qbase + a.D, ld, i0 - w + 32 * sb, KV_ROWS, doc.Lb, hd);
    __syncthreads();
    float s[4];
    score_phase<T>(Qs, KVs, a.rs, hd, tq, g, s);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int c = 32 * sb + g + 8 * t;
      const int j = i - w + c;
      const bool ok = (c < W) && (j >= 0) && (j < len);
      Ps[tq * a.ps + c] = ok ? s[t] : -INFINITY;
    }
  }
  __syncthreads();
  // softmax over the row's slots: 8 lanes per query
  {
    const bool qok = (i < len);
    float m = -INFINITY;
    for (int c = g; c < a.slots; c += 8) m = fmaxf(m, Ps[tq * a.ps + c]);
    m = oct_max(m);
    float sum = 0.f;
    for (int c = g; c < a.slots; c += 8) {
      const float e = qok ? __expf(Ps[tq * a.ps + c] - m) : 0.f;   // exp(-inf) = 0 for masked keys
      Ps[tq * a.ps + c] = e;
      sum += e;
    }
    sum = oct_sum(sum);
    const float inv = qok ? 1.0f / sum : 0.f;
    float* prow = a.probs + ((size_t)(doc.base + i) * a.heads + h) * a.slots;
    for (int c = g; c < a.slots; c += 8) {
      const float p = Ps[tq * a.ps + c] * inv;
      // dropout (modeling_longformer.py:590) acts on what multiplies V; the saved probabilities stay whole
      Ps[tq * a.ps + c] = (a.drop_thr && !band_keep(a, doc.base + i, h, c)) ? 0.f : (a.drop_thr ? p * a.drop_scale : p);
      if (i < doc.Lb) prow[c] = p;
    }
  }
  float acc[MAXU][4];
#pragma unroll
  for (int uu = 0; uu < MAXU; ++uu) acc[uu][0] = acc[uu][1] = acc[uu][2] = acc[uu][3] = 0.f;
  for (int sb = 0; sb < nsb; ++sb) {
    __syncthreads();
    stage_rows<T>(KVs, a.rs, qbase + 2 * a.D, ld, i0 - w + 32 * sb, KV_ROWS, doc.Lb, hd);
    __syncthreads();
    const int ncc = min(32, W - 32 * sb);
    accum_phase<T, MAXU>(Ps + tq * a.ps + 32 * sb, 1, KVs, a.rs, hd, tq, g, ncc, acc);
  }
  if (i < doc.Lb) {
    T* o = reinterpret_cast<T*>(a.ctx) + (size_t)(doc.base + i) * a.D + h * hd;
    const int nch = hd / 4;
#pragma unroll
    for (int uu = 0; uu < MAXU; ++uu) {
      const int u = g + 8 * uu;
      if (u < nch) store4<T>(o + 4 * u, acc[uu]);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward pass A: dS (stored) and dQ, per query tile
// ------------------------------------------------------------------------------------------------
template <typename T, int MAXU>
__global__ __launch_bounds__(256) void band_bwd_q_kernel(const BandArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qs = smem;                        // dCtx tile
  char* KVs = Qs + TQ * a.rs;
  float* Ps = reinterpret_cast<float*>(KVs + KV_ROWS * a.rs);

  const int ntiles = (a.L + TQ - 1) / TQ;
  int tile, h, b;
  decode_block(ntiles, a.heads, ntiles * a.heads * a.B, tile, h, b);
  const DocView doc = doc_view(a, b);
  const int i0 = tile * TQ;
  const int tq = threadIdx.x >> 3, g = threadIdx.x & 7;
  const int i = i0 + tq;
  const int w = a.radius, W = 2 * w + 1, hd = a.hd, ld = 3 * a.D;
  const T* qbase = reinterpret_cast<const T*>(a.qkv) + (size_t)doc.base * ld + h * hd;
  const T* dcbase = reinterpret_cast<const T*>(a.dctx) + (size_t)doc.base * a.D + h * hd;
  const int nsb = a.slots / 32;
  const size_t prow_off = ((size_t)(doc.base + min(i, doc.Lb - 1)) * a.heads + h) * a.slots;
  const float* prow = a.probs + prow_off;

  stage_rows<T>(Qs, a.rs, dcbase, a.D, i0, TQ, doc.Lb, hd);
  float delta = 0.f;
  for (int sb = 0; sb < nsb; ++sb) {
    __syncthreads();
    stage_rows<T>(KVs, a.rs, qbase + 2 * a.D, ld, i0 - w + 32 * sb, KV_ROWS, doc.Lb, hd);   // V rows
    __syncthreads();
    float s[4];
    score_phase<T>(Qs, KVs, a.rs, hd, tq, g, s);     // dP
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int c = 32 * sb + g + 8 * t;
      const float p = (i < doc.Lb) ? prow[c] : 0.f;      // zero for masked keys / queries and for c >= W
      if (a.drop_thr) s[t] = band_keep(a, doc.base + i, h, c) ? s[t] * a.drop_scale : 0.f;      // gradient through the dropout
      Ps[tq * a.ps + c] = s[t];
      delta += p * s[t];
    }
  }
  delta = oct_sum(delta);
  __syncthreads();
  {
    float* dsrow = a.dscores + prow_off;
    for (int c = g; c < a.slots; c += 8) {
      const float p = (i < doc.Lb) ? prow[c] : 0.f;
      const float ds = p * (Ps[tq * a.ps + c] - delta);
      Ps[tq * a.ps + c] = ds;
      if (i < doc.Lb) dsrow[c] = ds;
    }
  }
  float acc[MAXU][4];
#pragma unroll
  for (int uu = 0; uu < MAXU; ++uu) acc[uu][0] = acc[uu][1] = acc[uu][2] = acc[uu][3] = 0.f;
  for (int sb = 0; sb < nsb; ++sb) {
    __syncthreads();
    stage_rows<T>(KVs, a.rs, qbase + a.D, ld, i0 - w + 32 * sb, KV_ROWS, doc.Lb, hd);       // K rows
    __syncthreads();
    const int ncc = min(32, W - 32 * sb);
    accum_phase<T, MAXU>(Ps + tq * a.ps + 32 * sb, 1, KVs, a.rs, hd, tq, g, ncc, acc);
  }
  if (i < doc.Lb) {
    T* o = reinterpret_cast<T*>(a.dqkv) + (size_t)(doc.base + i) * ld + h * hd;
    const int nch = hd / 4;
#pragma unroll
    for (int uu = 0; uu < MAXU; ++uu) {
      const int u = g + 8 * uu;
      if (u < nch) {
        float v[4] = {acc[uu][0] * a.q_scale, acc[uu][1] * a.q_scale, acc[uu][2] * a.q_scale, acc[uu][3] * a.q_scale};
        store4<T>(o + 4 * u, v);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// backward pass B: dK, dV per KEY tile.  For key j and "reverse slot" c' (query i = j - w + c'), the
// coefficient is X[i][2w - c'] with X = dS (for dK) or P (for dV).
// ------------------------------------------------------------------------------------------------
// raw[r][cidx] = X[row i = first_i + r][slot cbase + cidx], 32 slots wide, zero outside the valid ranges
__device__ __forceinline__ void stage_coef(float* raw, const float* __restrict__ X, size_t row_stride, int first_i, int L, int cbase, int W,
                                           const BandArgs& a, int base_row, int h, bool dropped) {
  for (int idx = threadIdx.x; idx < KV_ROWS * 32; idx += 256) {
    const int r = idx >> 5, ci = idx & 31;
    const int i = first_i + r, c = cbase + ci;
    const bool ok = (i >= 0 && i < L && c >= 0 && c < W);
    float x = ok ? X[(size_t)i * row_stride + c] : 0.f;
    if (dropped && ok) x = band_keep(a, base_row + i, h, c) ? x * a.drop_scale : 0.f;     // dV sees the dropped probabilities
    raw[idx] = x;
  }
}

template <typename T, int MAXU>
__global__ __launch_bounds__(256) void band_bwd_kv_kernel(const BandArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Rs = smem;                                               // KV_ROWS rows of q (or dCtx)
  float* raw = reinterpret_cast<float*>(Rs + KV_ROWS * a.rs);    // [KV_ROWS][32]
  float* PT = raw + KV_ROWS * 32;                                // [TQ][33]

  const int ntiles = (a.L + TQ - 1) / TQ;
  int tile, h, b;
  decode_block(ntiles, a.heads, ntiles * a.heads * a.B, tile, h, b);
  const DocView doc = doc_view(a, b);
  const int j0 = tile * TQ;
  const int tk = threadIdx.x >> 3, g = threadIdx.x & 7;
  const int j = j0 + tk;
  const int w = a.radius, W = 2 * w + 1, hd = a.hd, ld = 3 * a.D;
  const T* qbase = reinterpret_cast<const T*>(a.qkv) + (size_t)doc.base * ld + h * hd;
  const T* dcbase = reinterpret_cast<const T*>(a.dctx) + (size_t)doc.base * a.D + h * hd;
  const size_t xrow = (size_t)a.heads * a.slots;
  const float* dsb = a.dscores + ((size_t)doc.base * a.heads + h) * a.slots;
  const float* pb = a.probs + ((size_t)doc.base * a.heads + h) * a.slots;
  const int nsb = a.slots / 32;

  float dk[MAXU][4], dv[MAXU][4];
#pragma unroll
  for (int uu = 0; uu < MAXU; ++uu)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) dk[uu][jj] = dv[uu][jj] = 0.f;

  for (int sb = 0; sb < nsb; ++sb) {
    const int first_i = j0 - w + 32 * sb;          // query row of (tk = 0, cc = 0)
    const int cbase = 2 * w - 32 * sb - 31;        // slot of raw column 0; coefficient (tk, cc) = raw[tk + cc][31 - cc]
    const int ncc = min(32, W - 32 * sb);
    for (int pass = 0; pass < 2; ++pass) {
      __syncthreads();
      stage_coef(raw, pass == 0 ? dsb : pb, xrow, first_i, doc.Lb, cbase, W, a, doc.base, h, pass == 1 && a.drop_thr != 0);
      if (pass == 0) stage_rows<T>(Rs, a.rs, qbase, ld, first_i, KV_ROWS, doc.Lb, hd);          // scaled q rows
      else stage_rows<T>(Rs, a.rs, dcbase, a.D, first_i, KV_ROWS, doc.Lb, hd);                  // dCtx rows
      __syncthreads();
      for (int cc = g; cc < 32; cc += 8) PT[tk * 33 + cc] = raw[(tk + cc) * 32 + (31 - cc)];
      __syncthreads();
      if (pass == 0) accum_phase<T, MAXU>(PT + tk * 33, 1, Rs, a.rs, hd, tk, g, ncc, dk);
      else accum_phase<T, MAXU>(PT + tk * 33, 1, Rs, a.rs, hd, tk, g, ncc, dv);
    }
  }
  if (j < doc.Lb) {
    T* o = reinterpret_cast<T*>(a.dqkv) + (size_t)(doc.base + j) * ld + h * hd;
    const int nch = hd / 4;
#pragma unroll
    for (int uu = 0; uu < MAXU; ++uu) {
      const int u = g + 8 * uu;
      if (u < nch) { store4<T>(o + a.D + 4 * u, dk[uu]); store4<T>(o + 2 * a.D + 4 * u, dv[uu]); }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static thread_local int g_band_mfma = 1;           // mts_set_option("band_mfma", 0) forces the generic kernels (A/B testing)
void mts_band_set_mfma(int on) { g_band_mfma = on; }

extern "C" int mts_band_slots(int radius) { return band_slots(radius); }

static int band_row_stride(int hd, int esize) {
  int bytes = ((hd * esize + 15) / 16) * 16;
  if (((bytes / 16) & 1) == 0) bytes += 16;   // stride = 16 B x odd: 16 rows at one column hit 16 distinct 16-byte bank slots
  return bytes;
}

static int band_fill(BandArgs& a, int dtype, int B, int L, int D, int heads, int radius, const char* who) {
  MTS_CHECK_ARG(B > 0 && L > 0 && D > 0 && heads > 0 && radius > 0, "%s: bad shape", who);
  MTS_CHECK_ARG(D % heads == 0, "%s: D=%d not divisible by heads=%d", who, D, heads);
  MTS_CHECK_ARG(dtype == MTS_F32 || dtype == MTS_BF16, "%s: bad dtype %d", who, dtype);
  const int hd = D / heads;
  const int vec = dtype == MTS_F32 ? 4 : 8;
  MTS_UNSUPPORTED(hd % vec == 0 && hd <= 512, "%s: head dim %d must be a multiple of %d and <= 512", who, hd, vec);
  MTS_UNSUPPORTED((long)B * L * heads * (long)band_slots(radius) < (1L << 31), "%s: problem too large for 32-bit slot indexing", who);
  a.B = B; a.L = L; a.D = D; a.heads = heads; a.hd = hd; a.radius = radius; a.slots = band_slots(radius);
  a.rs = band_row_stride(hd, dtype == MTS_F32 ? 4 : 2);
  a.ps = a.slots + 1;
  a.q_scale = 1.f;
  a.bias_slab = nullptr; a.img_bytes = 0; a.row0 = nullptr;
  a.drop_scale = 1.f; a.drop_thr = 0; a.drop_seed = 0;
  a.lengths = nullptr; a.qkv = nullptr; a.ctx = nullptr; a.probs = nullptr; a.dctx = nullptr; a.dqkv = nullptr; a.dscores = nullptr;
  return MTS_OK;
}

static int band_set_dropout(BandArgs& a, float p, uint64_t seed, const char* who) {
  MTS_CHECK_ARG(p >= 0.f && p < 1.f, "%s: dropout probability has to be between 0 and 1, but got %f", who, (double)p);
  if (p > 0.f) {
    a.drop_thr = (uint32_t)std::max<double>(1.0, std::min<double>(4294967295.0, (double)p * 4294967296.0));
    a.drop_scale = 1.0f / (1.0f - p);
    a.drop_seed = seed;
  }
  return MTS_OK;
}

template <typename K> static int set_lds(K kernel, size_t bytes, const char* who) {
  MTS_UNSUPPORTED(bytes <= 160 * 1024, "%s: needs %zu bytes of LDS (> 160 KiB): window too wide for this head dim", who, bytes);
  if (bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { mts_set_error("%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e)); return MTS_ERR_LAUNCH; }
  }
  return MTS_OK;
}

template <typename T>
static int band_fwd_launch(const BandArgs& a, hipStream_t st) {
  const size_t lds = (size_t)(TQ + KV_ROWS) * a.rs + (size_t)TQ * a.ps * sizeof(float);
  const int nblocks = ceil_div(a.L, TQ) * a.heads * a.B;
  auto k = a.hd > 256 ? band_fwd_kernel<T, 16> : band_fwd_kernel<T, 8>;       // 8 lanes x MAXU chunks of 4 cover the head dim
  int rc = set_lds(k, lds, "mts_band_attn_fwd");
  if (rc) return rc;
  hipLaunchKernelGGL(k, dim3(nblocks), dim3(256), lds, st, a);
  MTS_LAUNCH_CHECK("mts_band_attn_fwd");
  return MTS_OK;
}

extern "C" int mts_band_attn_fwd(void* stream, int dtype, int B, int L, int D, int heads, int radius, const void* qkv, const int32_t* lengths,
                                 void* ctx, float* probs, const int32_t* row0, float drop_p, uint64_t drop_seed) {
  BandArgs a;
  int rc = band_fill(a, dtype, B, L, D, heads, radius, "mts_band_attn_fwd");
  if (rc) return rc;
  MTS_CHECK_ARG(qkv && ctx && probs, "mts_band_attn_fwd: null pointer");
  MTS_CHECK_ARG(!row0 || lengths, "mts_band_attn_fwd: packed rows (row0) need lengths");
  a.row0 = row0;
  rc = band_set_dropout(a, drop_p, drop_seed, "mts_band_attn_fwd");
  if (rc) return rc;
  a.qkv = qkv; a.lengths = lengths; a.ctx = ctx; a.probs = probs;
  if (dtype == MTS_BF16 && g_band_mfma) {
    rc = mts_band_mfma_fwd(a, (hipStream_t)stream);
    if (rc >= 0) return rc;             // -1: shape not covered by the matrix-core kernels
  }
  return dtype == MTS_F32 ? band_fwd_launch<float>(a, (hipStream_t)stream) : band_fwd_launch<bf16_t>(a, (hipStream_t)stream);
}

template <typename T>
static int band_bwd_launch(const BandArgs& a, hipStream_t st) {
  const int nblocks = ceil_div(a.L, TQ) * a.heads * a.B;
  {
    const size_t lds = (size_t)(TQ + KV_ROWS) * a.rs + (size_t)TQ * a.ps * sizeof(float);
    auto k = a.hd > 256 ? band_bwd_q_kernel<T, 16> : band_bwd_q_kernel<T, 8>;
    int rc = set_lds(k, lds, "mts_band_attn_bwd(q)");
    if (rc) return rc;
    hipLaunchKernelGGL(k, dim3(nblocks), dim3(256), lds, st, a);
  }
  {
    const size_t lds = (size_t)KV_ROWS * a.rs + (size_t)(KV_ROWS * 32 + TQ * 33) * sizeof(float);
    auto k = a.hd > 256 ? band_bwd_kv_kernel<T, 16> : band_bwd_kv_kernel<T, 8>;
    int rc = set_lds(k, lds, "mts_band_attn_bwd(kv)");
    if (rc) return rc;
    hipLaunchKernelGGL(k, dim3(nblocks), dim3(256), lds, st, a);
  }
  MTS_LAUNCH_CHECK("mts_band_attn_bwd");
  return MTS_OK;
}

// norm.hip: out[e] = sum_b partial[b][e] (fixed order)
int mts_slab_reduce_rows(hipStream_t st, const float* partial, int nblocks, int D, float* out);

extern "C" size_t mts_band_attn_bwd_workspace(int B, int L, int D) {
  const size_t slab = (size_t)B * ceil_div(L, 128) * 3 * (size_t)D * sizeof(float);
  return std::max(slab, mts_colsum_workspace(3 * D));
}

extern "C" int mts_band_attn_bwd(void* stream, int dtype, int B, int L, int D, int heads, int radius, float q_scale, const void* qkv,
                                 const int32_t* lengths, const float* probs, const void* dctx, void* dqkv, float* dscores, float* dbias,
                                 void* workspace, const int32_t* row0, int n_rows, float drop_p, uint64_t drop_seed) {
  BandArgs a;
  int rc = band_fill(a, dtype, B, L, D, heads, radius, "mts_band_attn_bwd");
  if (rc) return rc;
  MTS_CHECK_ARG(qkv && probs && dctx && dqkv && dscores, "mts_band_attn_bwd: null pointer");
  MTS_CHECK_ARG(!dbias || workspace, "mts_band_attn_bwd: dbias needs mts_band_attn_bwd_workspace() bytes of workspace");
  MTS_CHECK_ARG(!row0 || (lengths && n_rows > 0 && n_rows <= B * L), "mts_band_attn_bwd: packed rows (row0) need lengths and 0 < n_rows <= B*L");
  a.row0 = row0;
  rc = band_set_dropout(a, drop_p, drop_seed, "mts_band_attn_bwd");
  if (rc) return rc;
  a.qkv = qkv; a.lengths = lengths; a.probs = const_cast<float*>(probs); a.dctx = dctx; a.dqkv = dqkv; a.dscores = dscores;
  a.q_scale = q_scale;
  if (dtype == MTS_BF16 && g_band_mfma) {
    a.bias_slab = dbias ? (float*)workspace : nullptr;      // column sums fused into the kernels' output stage
    int slab_rows = 0;
    rc = mts_band_mfma_bwd(a, (hipStream_t)stream, &slab_rows);
    if (rc == MTS_OK && dbias) return mts_slab_reduce_rows((hipStream_t)stream, (const float*)workspace, slab_rows, 3 * D, dbias);
    if (rc >= 0) return rc;
    a.bias_slab = nullptr;
  }
  rc = dtype == MTS_F32 ? band_bwd_launch<float>(a, (hipStream_t)stream) : band_bwd_launch<bf16_t>(a, (hipStream_t)stream);
  if (rc == MTS_OK && dbias) rc = mts_colsum(stream, dtype, row0 ? n_rows : B * L, 3 * D, dqkv, 3 * D, dbias, 0, workspace);
  return rc;
}
