// Pieces shared by the two bf16 MFMA GEMM kernels (gemm.hip: 128x128 tile; gemm256.hip: 256x256 tile).
#pragma once
#include "common.h"

typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

struct GemmArgs {
  const void* A; const void* B; void* C;
  const float* bias; const void* residual; void* aux;
  int M, N, K;
  int lda, ldb, ldc, ldr, ldaux;
  unsigned epi;
  float colscale; int ncols_scaled;
  int ksplit;      // K elements per blockIdx.z slice (multiple of 64)
  int order;       // tile order of the persistent kernels: 1 = 4-row bands (L2-blocked), 0 = row-major
  int nbuf;        // 128x128 kernel, LDS-DMA form: K-tile buffers in LDS (2, or 4 = copies run three K-tiles ahead; launched
                   // with 128 KiB of LDS when the grid has at most one workgroup per CU and nothing else hides DMA latency)
  unsigned* chain; // split-K without slabs (128x128 kernel): per-tile turn counter, zeroed before the launch; slice z adds its
                   // partial tile into C when the counter reads z (fixed order -> bitwise reproducible), then bumps it
  float* slab;     // split-K: slice z stores its fp32 partial tile to slab[z][M][N] (plain stores); reduced afterwards
  int variant;     // A/B switch of the big-tile kernels (mts_set_option("gemm_variant", v)); 0 = production behaviour
#ifdef MTS_GEMM_STAMPS
  unsigned long long* stamps;   // diagnostic build only (tools/gemm_stamps.py): [workgroup][tile round][8] s_memtime / s_memrealtime
#endif
};

// activation of the epilogue: erf-GELU (HF intermediate layer) or ReLU (legacy RestrictedTransformerEncoderLayer)
#define MTS_EPI_ACT (MTS_EPI_GELU | MTS_EPI_RELU)
__device__ __forceinline__ float epi_act(unsigned epi, float x) { return (epi & MTS_EPI_RELU) ? fmaxf(x, 0.0f) : gelu_erf_f(x); }

// ------------------------------------------------------------------------------------------------
// shared epilogue: lane owns C[m][n..n+3]
// ------------------------------------------------------------------------------------------------
// epilogue arithmetic only (bias, q-scale, residual, GELU + pre-activation) for 4 consecutive in-range columns
template <typename TA>
__device__ __forceinline__ void epi_math4(const GemmArgs& a, int m, int n, float (&v)[4], bool first_slice) {
  const unsigned epi = a.epi;
  if ((epi & MTS_EPI_BIAS) && first_slice) {
    const float4 b = *reinterpret_cast<const float4*>(a.bias + n);
    v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
  }
  if (epi & MTS_EPI_COLSCALE) {
#pragma unroll
    for (int i = 0; i < 4; ++i) if (n + i < a.ncols_scaled) v[i] *= a.colscale;
  }
  if ((epi & MTS_EPI_RESIDUAL) && first_slice) {
    float r[4];
    load4<TA>(reinterpret_cast<const TA*>(a.residual) + (size_t)m * a.ldr + n, r);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] += r[i];
  }
  if (epi & MTS_EPI_ACT) {
    if (a.aux) store4<TA>(reinterpret_cast<TA*>(a.aux) + (size_t)m * a.ldaux + n, v);
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = epi_act(epi, v[i]);
  }
}

// the same arithmetic on 8 consecutive in-range columns of row m (bf16 activations): 16-byte aux accesses; r = the residual
// chunk of (m, n) and b0/b1 = the bias of columns n..n+7, ALL fetched by the caller ahead of the K loop: a global load between
// the stores of an epilogue is waited for with "s_waitcnt vmcnt(0)", i.e. together with every store issued before it (the
// vector-memory counter retires in order) -- eight serial store round trips per tile in the 128x128 kernel before this.
__device__ __forceinline__ void epi_math8(const GemmArgs& a, int m, int n, float (&v)[8], bool first_slice, const uint4& r,
                                          const float4& b0, const float4& b1) {
  const unsigned epi = a.epi;
  if ((epi & MTS_EPI_BIAS) && first_slice) {
    v[0] += b0.x; v[1] += b0.y; v[2] += b0.z; v[3] += b0.w; v[4] += b1.x; v[5] += b1.y; v[6] += b1.z; v[7] += b1.w;
  }
  if (epi & MTS_EPI_COLSCALE) {
#pragma unroll
    for (int i = 0; i < 8; ++i) if (n + i < a.ncols_scaled) v[i] *= a.colscale;
  }
  if ((epi & MTS_EPI_RESIDUAL) && first_slice) {
    v[0] += bf16_lo(r.x); v[1] += bf16_hi(r.x); v[2] += bf16_lo(r.y); v[3] += bf16_hi(r.y);
    v[4] += bf16_lo(r.z); v[5] += bf16_hi(r.z); v[6] += bf16_lo(r.w); v[7] += bf16_hi(r.w);
  }
  if (epi & MTS_EPI_ACT) {
    if (a.aux) {
      uint4 pk;
      pk.x = pack_bf16x2(v[0], v[1]); pk.y = pack_bf16x2(v[2], v[3]); pk.z = pack_bf16x2(v[4], v[5]); pk.w = pack_bf16x2(v[6], v[7]);
      *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.aux) + (size_t)m * a.ldaux + n) = pk;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = epi_act(epi, v[i]);
  }
}

template <typename TA, typename TC>
__device__ __forceinline__ void epilogue4(const GemmArgs& a, int m, int n, float (&v)[4], bool first_slice) {
  if (m >= a.M || n >= a.N) return;
  const unsigned epi = a.epi;
  const bool full = (n + 3 < a.N);
  if (full) {
    if ((epi & MTS_EPI_BIAS) && first_slice) {
      const float4 b = *reinterpret_cast<const float4*>(a.bias + n);
      v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
    }
    if (epi & MTS_EPI_COLSCALE) {
#pragma unroll
      for (int i = 0; i < 4; ++i) if (n + i < a.ncols_scaled) v[i] *= a.colscale;
    }
    if ((epi & MTS_EPI_RESIDUAL) && first_slice) {
      float r[4];
      load4<TA>(reinterpret_cast<const TA*>(a.residual) + (size_t)m * a.ldr + n, r);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] += r[i];
    }
    if (epi & MTS_EPI_ACT) {
      if (a.aux) store4<TA>(reinterpret_cast<TA*>(a.aux) + (size_t)m * a.ldaux + n, v);
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = epi_act(epi, v[i]);
    }
    TC* c = reinterpret_cast<TC*>(a.C) + (size_t)m * a.ldc + n;
    if constexpr (sizeof(TC) == 4) {
      float* cf = reinterpret_cast<float*>(c);
      if (a.slab) {
        store4<float>(a.slab + ((size_t)blockIdx.z * a.M + m) * a.N + n, v);
      } else if (epi & MTS_EPI_ACCUM) {
        float o[4];
        load4<float>(cf, o);
#pragma unroll
        for (int i = 0; i < 4; ++i) o[i] += v[i];
        store4<float>(cf, o);
      } else {
        store4<float>(cf, v);
      }
    } else {
      store4<TC>(c, v);
    }
  } else {
    for (int i = 0; i < 4 && n + i < a.N; ++i) {
      float x = v[i];
      if ((epi & MTS_EPI_BIAS) && first_slice) x += a.bias[n + i];
      if ((epi & MTS_EPI_COLSCALE) && n + i < a.ncols_scaled) x *= a.colscale;
      if ((epi & MTS_EPI_RESIDUAL) && first_slice) x += to_f32(reinterpret_cast<const TA*>(a.residual)[(size_t)m * a.ldr + n + i]);
      if (epi & MTS_EPI_ACT) {
        if (a.aux) reinterpret_cast<TA*>(a.aux)[(size_t)m * a.ldaux + n + i] = from_f32<TA>(x);
        x = epi_act(epi, x);
      }
      TC* c = reinterpret_cast<TC*>(a.C) + (size_t)m * a.ldc + n + i;
      if constexpr (sizeof(TC) == 4) {
        float* cf = reinterpret_cast<float*>(c);
        if (a.slab) a.slab[((size_t)blockIdx.z * a.M + m) * a.N + n + i] = x;
        else if (epi & MTS_EPI_ACCUM) *cf += x;
        else *cf = x;
      } else {
        *c = from_f32<TC>(x);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// bf16 MFMA kernel
// ------------------------------------------------------------------------------------------------
#define BM 128
#define BN 128
#define BK 64
#define TILE_BYTES (128 * 64 * 2)  // one operand stage = 16 KiB in either image

// K-major image: [128 rows][64 k] bf16, 128-B rows, 16-B chunk index XORed with (row>>1)&7.
__device__ __forceinline__ int kmajor_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }
// strided image: [64 k-rows][128 cols] bf16, 256-B rows, 32-B chunk index XORed with key(krow).
__device__ __forceinline__ int strided_key(int krow) { return (krow & 3) | (((krow >> 3) & 1) << 2); }
__device__ __forceinline__ int strided_off(int krow, int col) {
  return krow * 256 + ((((col >> 4) ^ strided_key(krow))) << 5) + ((col & 15) << 1);
}

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

__device__ __forceinline__ bf16x8 frag_kmajor(const char* tile, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(tile + kmajor_off(row, chunk));
}
// fragment X[k = kbase + 0..7][c = c0 + (lane&15)] of a strided image through two transposed reads.
//
// The reads are INLINE ASM on purpose.  Written with __builtin_amdgcn_ds_read_tr16_b64 the compiler's wait-count pass cannot
// tell them apart from the LDS range an outstanding LDS-DMA (global_load_lds) is writing, and puts "s_waitcnt vmcnt(0)" in
// front of the first transposed read of every phase: the DMA of the NEXT K-tile, issued a few instructions earlier, had to
// land before the current tile could be read -- no overlap of copy and MFMA inside a workgroup for every layout with a
// strided operand (NN, TN, TT), while NT (plain ds_read_b128) was unaffected.  The asm form is invisible to that pass;
// the price is that the consumer must wait for the data itself: lds_frags_wait() once after the last read, then
// frag_finish(raw) on every fragment read this way (an empty asm ties the halves to the wait's position).
// The two halves stay SEPARATE values until frag_finish(): assembling the four-register MFMA operand right after the asm
// would let the compiler place register copies between the read and the wait (i.e. copy registers the LDS has not written
// yet) whenever it cannot coalesce the halves into the tuple.
struct frag_raw { s16x4 lo, hi; };
__device__ __forceinline__ frag_raw frag_strided(const char* tile, int kbase, int c0, int lane) {
  const int r = lane & 15;
  const int q = r >> 2, p = r & 3;
  const int col = c0 + 4 * p;
  // rows kbase + q and kbase + 4 + q share the swizzle key (bits 0-1 and 3 of the row), so the second read is +4 rows = +1024 B
  const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(tile + strided_off(kbase + q, col));
  frag_raw f;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.lo) : "v"(addr));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(f.hi) : "v"(addr));
  return f;
}
// after lds_frags_wait(): ties both halves to the wait's position (empty asm), then forms the MFMA operand
__device__ __forceinline__ bf16x8 frag_finish(frag_raw f) {
  asm volatile("" : "+v"(f.lo), "+v"(f.hi));
  s16x8 v = __builtin_shufflevector(f.lo, f.hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ void lds_frags_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// ------------------------------------------------------------------------------------------------
// LDS fragment reads the CALLER waits for (big-tile kernels)
// ------------------------------------------------------------------------------------------------
// Both read forms as inline asm (K-major: one ds_read_b128; strided: two ds_read_b64_tr_b16) so that a phase can issue ALL of
// its reads -- and the next phase's -- up front and start its first MFMAs behind a COUNTED s_waitcnt lgkmcnt(N) as soon as the
// fragments those MFMAs need have arrived (LDS returns in order).  Written as plain loads the compiler waits for every
// outstanding read before the first MFMA of a burst: after a workgroup barrier all 8 waves issue 12 reads each at once and the
// matrix cores idle until the LAST of the 96 has been served.
template <bool KMAJOR> struct LFrag { s16x8 k; frag_raw s; };      // one member is live, the other is dead code
template <bool KMAJOR> struct lfrag_ops { static constexpr int value = KMAJOR ? 1 : 2; };   // LDS operations per fragment

// K-major: fragment of `row`, 16-byte chunk `chunk`; strided: X[k = kbase .. kbase+7][c0 + (lane & 15)]
template <bool KMAJOR>
__device__ __forceinline__ void lfrag_read(LFrag<KMAJOR>& f, const char* tile, int row_or_kbase, int chunk_or_c0, int lane) {
  if constexpr (KMAJOR) {
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(tile + kmajor_off(row_or_kbase, chunk_or_c0));
    asm volatile("ds_read_b128 %0, %1" : "=v"(f.k) : "v"(addr));
  } else {
    f.s = frag_strided(tile, row_or_kbase, chunk_or_c0, lane);
  }
}
// after the covering wait: ties the registers to the wait's position and forms the MFMA operand
template <bool KMAJOR>
__device__ __forceinline__ bf16x8 lfrag_get(LFrag<KMAJOR>& f) {
  if constexpr (KMAJOR) {
    asm volatile("" : "+v"(f.k));
    return __builtin_bit_cast(bf16x8, f.k);
  } else {
    return frag_finish(f.s);
  }
}
template <int N> __device__ __forceinline__ void lgkm_wait() {
  static_assert(N >= 0, "negative wait count");
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N < 15 ? N : 15) : "memory");     // the counter has 4 bits
}
