// 256x224x64 variant of the big-tile bf16 MFMA GEMM (gemm256.hip) for N that is a multiple of 224.
//
// Why: the reference's width is d = 1792 = 8 x 224 (768-d sentence + 1024-d audio embeddings), so every projection of
// the tagger has N in {1792, 5376}.  With 256-wide tiles those give 448 / 1344 output tiles = 1.75 / 5.25 rounds over
// the 256 CUs -- the last round is 3/4 or 1/4 full.  224-wide tiles give 512 / 1536 tiles = exactly 2 / 6 rounds, each
// 12.5 % shorter.
//
// Pipeline: three A images and two B images of two 16-KiB halves each, both operands copied by LDS-DMA TWO K-tiles ahead behind
// a counted vmcnt (A one tile ahead, as in gemm256.hip, left its latency exposed at the end of every K-step), raw s_barrier;
// the 8 waves are laid out 4 (M) x 2 (N): wave (wm, wn) owns C rows
// wm*64..+63 (half of A[wm>>1]) and cols wn*112..+111 (all of B[wn]; a B half image holds 112 used rows, the DMA still
// moves 128 so that every wave issues the same number of loads and one counted wait serves all):
//     phase 1: DMA A0(t+2) ; read A[rows 0..31], B[cols 0..63]   ; 16 MFMA  (0,0)
//     phase 2: DMA A1(t+2) ; read B[cols 64..111]                ; 12 MFMA  (0,1) ; barrier
//     phase 3: DMA B0(t+2) ; read A[rows 32..63]                 ; 12 MFMA  (1,1)
//     phase 4: DMA B1(t+2) ;                                       16 MFMA  (1,0) ; vmcnt(8) ; barrier
#include <algorithm>
#include <type_traits>
#include <math.h>
#include "gemm_common.h"

#if defined(MTS_GEMM_STAMPS) && !defined(G224_PHASES)
#define STAMP(slot) do { if (a.stamps && tid == 0) { a.stamps[((size_t)blockIdx.x * 8 + round) * 8 + (slot)] = __builtin_amdgcn_s_memtime(); \
                                                      if ((slot) == 0) a.stamps[((size_t)blockIdx.x * 8 + round) * 8 + 6] = __builtin_amdgcn_s_memrealtime(); \
                                                      if ((slot) == 3) a.stamps[((size_t)blockIdx.x * 8 + round) * 8 + 5] = __builtin_amdgcn_s_memrealtime(); \
                                                      if ((slot) == 4) a.stamps[((size_t)blockIdx.x * 8 + round) * 8 + 7] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define STAMP(slot) do { } while (0)
#endif
#define HT_BYTES 16384
// LDS: three A images (2 halves each) | two B images (2 halves each); the bf16 store staging (8 waves x 4 KiB) aliases the third
// A image, which is idle between a tile's last K-step and the next tile's first
#define A_BYTES (2 * HT_BYTES)
#define B_BASE (3 * A_BYTES)
#define LDS_TOTAL (3 * A_BYTES + 4 * HT_BYTES)
#define BN224 224
#define HN224 112

#ifdef G224_PHASES                      // per-wave cycle sums per phase of the K loop (tools/micro/gemm224_phases.py); needs MTS_GEMM_STAMPS
#define PST(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); pacc[i] += t_ - pprev; pprev = t_; } while (0)
#else
#define PST(i) do { } while (0)
#endif
#ifdef MTS_GEMM_STAMPS
static unsigned long long* g_gemm_stamps = nullptr;
extern "C" void mts_gemm_set_stamps(void* p) { g_gemm_stamps = (unsigned long long*)p; }
#endif

// ONLY >= 0: just that one of the wave's two pieces (the mid-tile-barrier schedule issues its copies one at a time)
template <bool KMAJOR, int ONLY = -1>
__device__ __forceinline__ void dma_half(const bf16_t* __restrict__ G, int ld, int row0, int dim, int k0, char* dst, int wave_u, int lane) {
#ifdef G224_DBG_NO_DMA                  // timing experiment only (tools/micro/gemm224_variants.sh): the K loop without its copies
  return;
#endif
#pragma unroll
  for (int i = (ONLY < 0 ? 0 : ONLY); i < (ONLY < 0 ? 2 : ONLY + 1); ++i) {
    const int piece = wave_u * 2 + i;   // 16 pieces of 1 KiB, 2 per wave
    const bf16_t* src;
    if constexpr (KMAJOR) {
      const int row = piece * 8 + (lane >> 3), pos = lane & 7;
      const int ch = pos ^ ((row >> 1) & 7);
      src = G + (size_t)min(row0 + row, dim - 1) * ld + k0 + ch * 8;
    } else {
      const int kr = piece * 4 + (lane >> 4), c16 = lane & 15;
      const int ch = ((((c16 >> 1) ^ strided_key(kr))) << 1) | (c16 & 1);
      src = G + (size_t)(k0 + kr) * ld + min(row0 + ch * 8, dim - 8);
    }
    __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(dst + piece * 1024), 16, 0, 0);
  }
}

// C tile of one wave (64 x 112) -> global; bf16 output through a wave-private LDS staging area, 16 rows at a time, so that
// HBM sees 16-byte pieces of contiguous 224-byte row segments instead of the 8-byte pieces of the accumulator layout.
template <typename TC>
__device__ __forceinline__ void store_tile_224(const GemmArgs& a, f32x4 (&acc)[4][7], int m0, int n0, bool first_slice, char* stage,
                                               int lane) {
  const int r16 = lane & 15, g = lane >> 4;
  const bool vec_ok = (a.N % 8 == 0) && (a.ldc % 8 == 0);
  if constexpr (sizeof(TC) == 2) {
    if (!a.slab && vec_ok) {
      // all of the tile's residual pieces are requested up front (the MFMA fragments are dead, their registers are free):
      // one exposed memory latency per tile instead of one per 16-row pass
      const bool has_res = (a.epi & MTS_EPI_RESIDUAL) && first_slice;
      uint2 rr[4][7];
      if (has_res) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int m = min(m0 + i * 16 + r16, a.M - 1);
#pragma unroll
          for (int j = 0; j < 7; ++j)
            rr[i][j] = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(a.residual) + (size_t)m * a.ldr + min(n0 + j * 16 + 4 * g, a.N - 4));
        }
      }
      GemmArgs a2 = a;
      a2.epi = a.epi & ~MTS_EPI_RESIDUAL;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + i * 16 + r16;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
          const int n = n0 + j * 16 + 4 * g;
          float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
          if (m < a.M && n < a.N) {
            // same order as epi_math4: bias, column scale, residual, GELU (GELU never comes with a residual in this model)
            if (has_res && !(a.epi & MTS_EPI_ACT)) {
              epi_math4<bf16_t>(a2, m, n, v, first_slice);
              v[0] += bf16_lo(rr[i][j].x); v[1] += bf16_hi(rr[i][j].x); v[2] += bf16_lo(rr[i][j].y); v[3] += bf16_hi(rr[i][j].y);
            } else {
              epi_math4<bf16_t>(a, m, n, v, first_slice);
            }
          }
          uint2 pk;
          pk.x = pack_bf16x2(v[0], v[1]);
          pk.y = pack_bf16x2(v[2], v[3]);
          *reinterpret_cast<uint2*>(stage + r16 * 240 + (j * 16 + 4 * g) * 2) = pk;      // 240-byte rows: 16 B of padding
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int idx = it * 64 + lane;
          if (idx < 16 * 14) {
            const int row = idx / 14, ch = idx - row * 14;
            const uint4 val = *reinterpret_cast<const uint4*>(stage + row * 240 + ch * 16);
            const int mm = m0 + i * 16 + row, n = n0 + ch * 8;
            if (mm < a.M && n < a.N) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.C) + (size_t)mm * a.ldc + n) = val;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + i * 16 + r16;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int n = n0 + j * 16 + 4 * g;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      epilogue4<bf16_t, TC>(a, m, n, v, first_slice);
    }
  }
}

// ---- fast epilogue (full tiles, bf16 C, bias / column scale / residual only) -----------------------------------------------
// The generic store_tile_224 above evaluates the epilogue flags per 16x16 fragment: 28 bias loads and up to 28 residual loads
// per wave sit BETWEEN the stores, each behind a compiler-inserted "s_waitcnt vmcnt(0)" (346 of them in the ISA) -- every load
// waited for the stores issued before it AND for the next tile's first DMAs, so a tile's epilogue ran as ~30 serial memory round
// trips: 24,400 of a tile's 111,000 cycles for the Q|K|V projection, 50,000 of 139,000 with a residual (in-kernel stamps,
// tools/gemm_stamps.py).  Here every global load of the epilogue is issued up front, BEFORE the next tile's DMAs (the vector
// memory counter retires in order: a load behind the DMAs could only be waited for together with them), the stores follow
// without a single load in between, and nobody waits for the stores until two K-tiles into the next tile (EPI_STORES below).

__device__ __forceinline__ bool epi224_fast_ok(const GemmArgs& a, int bm0) {
  const unsigned simple = MTS_EPI_BIAS | MTS_EPI_COLSCALE | MTS_EPI_RESIDUAL;
  return !a.slab && (a.epi & ~simple) == 0 && bm0 + 256 <= a.M && (a.ldc % 8 == 0) && (((uintptr_t)a.C & 15) == 0) &&
         (!(a.epi & MTS_EPI_COLSCALE) || a.ncols_scaled % 4 == 0) &&
         (!(a.epi & MTS_EPI_RESIDUAL) || (a.ldr % 4 == 0 && ((uintptr_t)a.residual & 7) == 0)) &&
         (!(a.epi & MTS_EPI_BIAS) || ((uintptr_t)a.bias & 15) == 0);
}

// The fast epilogue is BRANCH-FREE so that every vector-memory wait in it can be counted: without a bias / residual the loads
// still execute, from a few cache lines at the start of A (always mapped, >= 4 KiB), and their values are discarded by a select.
struct Epi224 {
  const float* bias_p; const bf16_t* res_p; size_t res_ld; bool has_bias, has_res; float colscale; int ncols_scaled;
  __device__ __forceinline__ void init(const GemmArgs& a, int m0, int n0, bool first_slice, int lane) {
    const int r16 = lane & 15, g = lane >> 4;
    has_bias = (a.epi & MTS_EPI_BIAS) && first_slice;
    has_res = (a.epi & MTS_EPI_RESIDUAL) && first_slice;
    bias_p = has_bias ? a.bias + n0 + 4 * g : reinterpret_cast<const float*>(a.A) + 4 * g;
    res_ld = has_res ? (size_t)a.ldr : 0;
    res_p = has_res ? reinterpret_cast<const bf16_t*>(a.residual) + (size_t)(m0 + r16) * a.ldr + n0 + 4 * g
                    : reinterpret_cast<const bf16_t*>(a.A) + 4 * g;
    colscale = (a.epi & MTS_EPI_COLSCALE) ? a.colscale : 1.0f;
    ncols_scaled = (a.epi & MTS_EPI_COLSCALE) ? a.ncols_scaled - n0 - 4 * g : 0;
  }
  __device__ __forceinline__ void load_bias(float4 (&b)[7]) const {
#pragma unroll
    for (int j = 0; j < 7; ++j) b[j] = *reinterpret_cast<const float4*>(bias_p + j * 16);
  }
  __device__ __forceinline__ void load_res(int i, uint2 (&r)[7]) const {       // rows m0 + i*16 + r16 of the wave's tile
#pragma unroll
    for (int j = 0; j < 7; ++j) r[j] = *reinterpret_cast<const uint2*>(res_p + (size_t)(i * 16) * res_ld + j * 16);
  }
  // acc = (acc + bias) * scale for the whole 64 x 112 wave tile
  __device__ __forceinline__ void fold_bias(f32x4 (&acc)[4][7], const float4 (&b)[7]) const {
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const float sc = (j * 16 < ncols_scaled) ? colscale : 1.0f;
      const float4 bb = has_bias ? b[j] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[i][j][0] = (acc[i][j][0] + bb.x) * sc;
        acc[i][j][1] = (acc[i][j][1] + bb.y) * sc;
        acc[i][j][2] = (acc[i][j][2] + bb.z) * sc;
        acc[i][j][3] = (acc[i][j][3] + bb.w) * sc;
      }
    }
  }
  // rows i*16 .. i*16+15 of the wave tile: (+ residual) -> bf16 -> LDS staging -> four 16-byte stores per lane group
  __device__ __forceinline__ void pass(const GemmArgs& a, const f32x4 (&acc)[7], const uint2 (&r)[7], int i, int m0, int n0, char* stage, int lane) const {
    const int r16 = lane & 15, g = lane >> 4;
    bf16_t* __restrict__ C = reinterpret_cast<bf16_t*>(a.C);
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const uint2 rr = has_res ? r[j] : make_uint2(0u, 0u);
      uint2 pk;
      pk.x = pack_bf16x2(acc[j][0] + bf16_lo(rr.x), acc[j][1] + bf16_hi(rr.x));
      pk.y = pack_bf16x2(acc[j][2] + bf16_lo(rr.y), acc[j][3] + bf16_hi(rr.y));
      *reinterpret_cast<uint2*>(stage + r16 * 240 + (j * 16 + 4 * g) * 2) = pk;      // 240-byte rows: 16 B of padding
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = it * 64 + lane;
      const int row = idx / 14, ch = idx - row * 14;
      if (idx < 16 * 14) {
        const uint4 val = *reinterpret_cast<const uint4*>(stage + row * 240 + ch * 16);
        *reinterpret_cast<uint4*>(C + (size_t)(m0 + i * 16 + row) * a.ldc + n0 + ch * 8) = val;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
};

// PERSIST = false: one tile per workgroup.  The workgroup ends right behind its last store instruction, so the stores drain
// while the CU's next workgroup starts its K loop, and no state of a next tile lives across the epilogue (the persistent form
// keeps ~60 registers of it and the fast epilogue spills around every store there).
// One workgroup barrier per K-tile in both schedules.
// MIDBAR = false (gemm_variant 5): the barrier sits at the END of the K-tile.  A is copied two K-tiles ahead into three images, B one
// K-tile ahead into two, issued right behind the barrier that freed its image.  Every wave starts a K-tile by requesting its 18
// first fragments: the LDS needs ~400 cycles to serve the 8 waves and the matrix cores trickle until it has (in-kernel stamps,
// tools/micro/gemm224_phases.py: the first 16 MFMAs of a K-tile take 720-1600 cycles, the other 40 run at the pipe's rate).
// MIDBAR = true (default): the products of a K-tile are ordered  b1 x aA, b1 x aB | b0 x aA, b0 x aB  (b0 / b1: columns 0-63 /
// 64-111 of the wave's B half, aA / aB: rows 0-31 / 32-63 of its A rows).  All LDS reads of the tile are done after the second
// product, so the barrier sits THERE; behind it the wave already fetches the next K-tile's b1 (its registers are free) and, one
// product later, its aA -- while the 32 MFMAs of the last two products run.  The next K-tile starts with b1 x aA on registers
// that landed long ago: no MFMA waits for the LDS at a K-tile boundary, and the burst of reads (aB, b0) has 24 MFMAs to hide
// behind.  A lives in two images (K-tile parity), B in three: B(kt+2) is copied under the first two products of K-tile kt, A(kt+2)
// under the last two (behind the barrier that freed its image) -- a full K-tile for either to land, and the CU's copy queue
// (~24 cycles per 1-KiB copy, 64 copies per K-tile) is loaded evenly.  The bf16 store staging aliases the third B image.
// Accumulation order per output element is unchanged (bitwise equal).  Measured (in-process A/B, tools/gemm_ab_libs.py @0 vs @5).
template <int LAYOUT, typename TC, bool PERSIST, bool MIDBAR>
__global__ __launch_bounds__(512, 2) void gemm_bf16_224_kernel(const GemmArgs a) {
  constexpr bool A_KMAJOR = (LAYOUT == MTS_NT || LAYOUT == MTS_NN);
  constexpr bool B_KMAJOR = (LAYOUT == MTS_NT || LAYOUT == MTS_TT);
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [3][A0 | A1]  [2][B0 | B1]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave_u >> 1, wn = wave_u & 1;
  const int r16 = lane & 15, g = lane >> 4;
  constexpr int B_OFF = MIDBAR ? 2 * A_BYTES : B_BASE;           // MIDBAR: two A images, THREE B images (the store staging aliases the third)
  char* stage = smem + (MIDBAR ? 2 * A_BYTES + 4 * HT_BYTES : 2 * A_BYTES) + wave_u * 4096;

  const int ntn = a.N / BN224;
  const int ntm = (a.M + 255) / 256;
  const int nt = ntn * ntm;
  const int kbeg = blockIdx.z * a.ksplit;
  const int kend = min(a.K, kbeg + a.ksplit);
  const int nk = (kend - kbeg) / BK;
  const bool first_slice = (blockIdx.z == 0);

  const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(a.A);
  const bf16_t* __restrict__ B = reinterpret_cast<const bf16_t*>(a.B);

  // Tile order.  Workgroups b, b+8, b+16.. run on one XCD (round-robin dispatch) and each XCD has its own 4-MiB L2, so an
  // XCD gets a contiguous run of tile ids, and ids walk the tile grid in bands of 4 tile-rows, column-major inside a band:
  // the 32 tiles an XCD works on at a time form a 4 x 8 block (4 A panels + 8 B panels per K-step) instead of a
  // 1.3 x 24 strip that streams ALL of the weight matrix through every L2 in every round.
  auto tile_origin = [&](int t, int& bm0, int& bn0) {
    const int q = nt >> 3, rr = nt & 7, xcd = t & 7, idx = t >> 3;
    const int id = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    if (a.order == 0) { bm0 = (id / ntn) * 256; bn0 = (id % ntn) * BN224; return; }
    const int band = id / (4 * ntn), within = id - band * 4 * ntn;
    const int rows = min(4, ntm - band * 4);
    bm0 = (band * 4 + within % rows) * 256;
    bn0 = (within / rows) * BN224;
  };
  int bm0, bn0;
  // MIDBAR = false: A(kt) lives in A image kt % 3 (`ab`), B(kt) in B image kt & 1;  MIDBAR = true: both in images kt & 1
  auto dmaA = [&](int h, int kt, int ab) {
    dma_half<A_KMAJOR>(A, a.lda, bm0 + h * 128, a.M, kbeg + kt * BK, smem + ab * A_BYTES + h * HT_BYTES, wave_u, lane);
  };
  auto dmaB = [&](int h, int kt, int img) {
    dma_half<B_KMAJOR>(B, a.ldb, bn0 + h * HN224, a.N, kbeg + kt * BK, smem + B_OFF + (img * 2 + h) * HT_BYTES, wave_u, lane);
  };
  auto dmaA1 = [&](int h, int kt, int ab, auto I) {
    dma_half<A_KMAJOR, decltype(I)::value>(A, a.lda, bm0 + h * 128, a.M, kbeg + kt * BK, smem + ab * A_BYTES + h * HT_BYTES, wave_u, lane);
  };
  auto dmaB1 = [&](int h, int kt, int img, auto I) {
    dma_half<B_KMAJOR, decltype(I)::value>(B, a.ldb, bn0 + h * HN224, a.N, kbeg + kt * BK, smem + B_OFF + (img * 2 + h) * HT_BYTES, wave_u, lane);
  };
  // copy instructions per wave and K-tile: 4 for A, 4 for B.  Issue order matters for the counted waits (in-order counter).
  auto prologue = [&]() {
    if (nk > 0) {
      dmaB(0, 0, 0); dmaB(1, 0, 0); dmaA(0, 0, 0); dmaA(1, 0, 0);
      if (nk > 1) {
        if constexpr (MIDBAR) { dmaB(0, 1, 1); dmaB(1, 1, 1); }
        dmaA(0, 1, 1); dmaA(1, 1, 1);
      }
    }
  };

  f32x4 acc[4][7];
  LFrag<B_KMAJOR> fb0[4][2], fb1[3][2];          // fragment reads in flight / landed (gemm_common.h)
  LFrag<A_KMAJOR> faA[2][2], faB[2][2];
  const int arow = (wm & 1) * 64;

  int t = blockIdx.x;
  if (t >= nt) return;
  tile_origin(t, bm0, bn0);
  auto wait_first_tile = [&]() {                 // K-tile 0 has landed: everything but the copies of K-tile 1
    if (nk > 1) { if constexpr (MIDBAR) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  prologue();
  wait_first_tile();
  __builtin_amdgcn_s_barrier();

  int round = 0;
#ifdef G224_PHASES
  unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long pprev = __builtin_amdgcn_s_memtime();
#endif
  for (;;) {
    STAMP(0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 7; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if constexpr (MIDBAR) {
      constexpr int cb = lfrag_ops<B_KMAJOR>::value, ca = lfrag_ops<A_KMAJOR>::value;
      auto a_img = [&](int kt) { return (const char*)(smem + (kt & 1) * A_BYTES + (wm >> 1) * HT_BYTES); };
      auto b_img = [&](int img) { return (const char*)(smem + B_OFF + (img * 2 + wn) * HT_BYTES); };
      // Fragment addresses = lane part (loop-invariant registers) + image base (one add per K-tile) + instruction immediate.
      //   K-major:  kmajor_off(16 t + r16, ks*4 + g) = t * 2048 + lk[ks]              -- 2 lane registers per kernel
      //   strided:  strided_off(ks*32 + 8g + q, 16 c + 4p) = ks * 8192 + (sx ^ (c << 5)) -- one lane register per column block c
      // One address register per fragment (what the generic lfrag_read costs once the compiler has hoisted it out of the loop,
      // twice for the two image parities) does not fit next to 112 accumulators and 88 fragment registers: it spilled fragment
      // registers whose LDS data had not arrived yet.
      const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
      unsigned lk[2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) lk[ks] = r16 * 128 + (((ks * 4 + g) ^ ((r16 >> 1) & 7)) << 4);
      const int q_ = r16 >> 2, p_ = r16 & 3;
      const unsigned sx = (8 * g + q_) * 256 + 8 * p_ + ((((8 * g + q_) & 3) | ((g & 1) << 2)) << 5);   // strided_off(8g + q, 4p), column block 0
      const unsigned sxa = sx ^ ((arow >> 4) << 5);          // the wave's 64 A rows start at column block arow / 16 of the image half
      // (The lane parts go through an empty asm at every use: otherwise the compiler hoists one address per fragment out of the
      // loops again -- and spills them, reloading through the vector-memory counter behind the copies in flight.)
      // t2 = first 16-row (column) block of the group inside the wave's 64 A rows: 0 (aA) or 2 (aB)
      auto rd_a = [&](LFrag<A_KMAJOR> (&f)[2][2], const char* At, auto T2) {
        constexpr int t2 = decltype(T2)::value;
        const unsigned ib = lds0 + (unsigned)(At - smem);
        if constexpr (A_KMAJOR) {
          unsigned l0 = lk[0], l1 = lk[1];
          asm volatile("" : "+v"(l0), "+v"(l1));
          const unsigned b0_ = ib + arow * 128 + l0, b1_ = ib + arow * 128 + l1;
#pragma unroll
          for (int i = 0; i < 2; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[i][0].k) : "v"(b0_), "n"((t2 + i) * 2048));
#pragma unroll
          for (int i = 0; i < 2; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f[i][1].k) : "v"(b1_), "n"((t2 + i) * 2048));
        } else {
          unsigned sl = sxa;
          asm volatile("" : "+v"(sl));
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              const unsigned ad = ib + (sl ^ ((t2 + i) << 5));
              if (ks == 0) {
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f[i][0].s.lo) : "v"(ad));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(f[i][0].s.hi) : "v"(ad));
              } else {
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8192" : "=v"(f[i][1].s.lo) : "v"(ad));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:9216" : "=v"(f[i][1].s.hi) : "v"(ad));
              }
            }
        }
      };
      // c0 = first 16-column block of the group inside the wave's B half: 0 (b0: 4 blocks) or 4 (b1: 3 blocks)
      auto rd_b = [&](auto& fb, const char* Bt, auto C0, auto NJ) {
        constexpr int c0 = decltype(C0)::value, nj = decltype(NJ)::value;
        const unsigned ib = lds0 + (unsigned)(Bt - smem);
        if constexpr (B_KMAJOR) {
          unsigned l0 = lk[0], l1 = lk[1];
          asm volatile("" : "+v"(l0), "+v"(l1));
          const unsigned b0_ = ib + l0, b1_ = ib + l1;
#pragma unroll
          for (int j = 0; j < nj; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[j][0].k) : "v"(b0_), "n"((c0 + j) * 2048));
#pragma unroll
          for (int j = 0; j < nj; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[j][1].k) : "v"(b1_), "n"((c0 + j) * 2048));
        } else {
          unsigned sl = sx;
          asm volatile("" : "+v"(sl));
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < nj; ++j) {
              const unsigned ad = ib + (sl ^ ((c0 + j) << 5));
              if (ks == 0) {
                asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fb[j][0].s.lo) : "v"(ad));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(fb[j][0].s.hi) : "v"(ad));
              } else {
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8192" : "=v"(fb[j][1].s.lo) : "v"(ad));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:9216" : "=v"(fb[j][1].s.hi) : "v"(ad));
              }
            }
        }
      };
      using I0 = std::integral_constant<int, 0>; using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
      using I4 = std::integral_constant<int, 4>;
      auto rd_b0 = [&](const char* Bt) { rd_b(fb0, Bt, I0{}, I4{}); };
      auto rd_b1 = [&](const char* Bt) { rd_b(fb1, Bt, I4{}, I3{}); };
      // K-tile 0 has landed (barrier above / at the end of the previous round): its first product's operands
      if (nk > 0) { rd_b1(b_img(0)); rd_a(faA, a_img(0), I0{}); }
      int bb = 0;                                  // kt % 3: B image of this K-tile
#pragma clang loop unroll(disable)
      for (int kt = 0; kt < nk; ++kt) {
        // outstanding LDS operations, oldest first: [b1 6cb, aA 4ca] (requested during the previous K-tile), aB 4ca, b0 8cb
        rd_a(faB, a_img(kt), I2{});
        rd_b0(b_img(bb));
        const int bb1 = bb == 2 ? 0 : bb + 1, bb2 = bb == 0 ? 2 : bb - 1;      // (kt + 1) % 3, (kt + 2) % 3
        const bool nxt = kt + 1 < nk, cpy = kt + 2 < nk;
        // one product: acc[i0 + i][j0 + j] += B fragment j x A fragment i, ks = 0 then 1 (the order every schedule uses).  The MFMA
        // operands are formed right before their use: for strided operands that is a register copy, kept short-lived.
        // MFMA operands: formed ONCE per fragment, at its first product (for a strided operand that joins the two halves of the
        // transposing reads; formed again at the second product the compiler copied registers: 68 moves per K-tile in the TN loop),
        // and kept for the second.  The raw fragment registers are dead from then on and take the next K-tile's prefetch.
        bf16x8 ob1[3][2], ob0[4][2], oaA[2][2], oaB[2][2];
        auto product = [&](auto& fbx, auto& obx, auto& fax, auto& oax, auto J0, auto NJ, auto I0_, auto FORM_B, auto FORM_A, auto&& between) {
          constexpr int j0 = decltype(J0)::value, nj = decltype(NJ)::value, i0 = decltype(I0_)::value;
          constexpr bool form_b = decltype(FORM_B)::value != 0, form_a = decltype(FORM_A)::value != 0;
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            if constexpr (form_b) {
#pragma unroll
              for (int j = 0; j < nj; ++j) obx[j][ks] = lfrag_get<B_KMAJOR>(fbx[j][ks]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              if constexpr (form_a) oax[i][ks] = lfrag_get<A_KMAJOR>(fax[i][ks]);
#pragma unroll
              for (int j = 0; j < nj; ++j) acc[i0 + i][j0 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(obx[j][ks], oax[i][ks], acc[i0 + i][j0 + j], 0, 0, 0);
              if (i == 0) between(ks);                 // in the middle of each ks group: 3-4 MFMAs issued, 3-4 to come
            }
          }
        };
        using I1 = std::integral_constant<int, 1>;
        __builtin_amdgcn_s_setprio(1);
        lgkm_wait<4 * ca + 8 * cb>();                // b1, aA
        // B(kt+2) goes out under the first two products (its image was last read before the barrier of K-tile kt-1), A(kt+2) under
        // the last two, ONE copy at a time between MFMA groups (8 per wave and K-tile, one per 7 MFMAs): the CU's copy queue takes
        // ~24 cycles per 1-KiB copy, and a wave whose copy it cannot take yet stalls, MFMAs included -- bursts of 16 queued copies
        // cost each wave ~90 cycles per copy.
        product(fb1, ob1, faA, oaA, I4{}, I3{}, I0{}, I1{}, I1{}, [&](int ks) { if (cpy) { if (ks == 0) dmaB1(0, kt + 2, bb2, I0{}); else dmaB1(0, kt + 2, bb2, I1{}); } });
        lgkm_wait<8 * cb>();                         // aB
        product(fb1, ob1, faB, oaB, I4{}, I3{}, I2{}, I0{}, I1{}, [&](int ks) { if (cpy) { if (ks == 0) dmaB1(1, kt + 2, bb2, I0{}); else dmaB1(1, kt + 2, bb2, I1{}); } });
        __builtin_amdgcn_s_setprio(0);
        // every LDS read of this K-tile has landed; so have this wave's copies of K-tile kt+1 (all but the 4 of B(kt+2) just issued)
        lgkm_wait<0>();
        PST(0);                                      // first two products (24 MFMAs) issued, all reads landed
        if (cpy) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PST(1);                                      // copies landed
        __builtin_amdgcn_s_barrier();                // images kt & 1 are free, K-tile kt+1 is complete in the other pair
        PST(2);                                      // barrier
        if (nxt) rd_b1(b_img(bb1));
        __builtin_amdgcn_s_setprio(1);
        product(fb0, ob0, faA, oaA, I0{}, I4{}, I0{}, I1{}, I0{}, [&](int ks) { if (cpy) { if (ks == 0) dmaA1(0, kt + 2, kt & 1, I0{}); else dmaA1(0, kt + 2, kt & 1, I1{}); } });
        if (nxt) rd_a(faA, a_img(kt + 1), I0{});     // (the MFMAs that read the old aA have been issued)
        product(fb0, ob0, faB, oaB, I0{}, I4{}, I2{}, I0{}, I0{}, [&](int ks) { if (cpy) { if (ks == 0) dmaA1(1, kt + 2, kt & 1, I0{}); else dmaA1(1, kt + 2, kt & 1, I1{}); } });
        bb = bb1;
        PST(3);                                      // last two products (32 MFMAs, 8 copies, 10 reads) issued
        __builtin_amdgcn_s_setprio(0);
      }
    } else {
    int ab = 0;                                   // kt % 3
#pragma clang loop unroll(disable)
    for (int kt = 0; kt < nk; ++kt) {
      const char* At = smem + ab * A_BYTES + (wm >> 1) * HT_BYTES;
      const char* Bt = smem + B_OFF + ((kt & 1) * 2 + wn) * HT_BYTES;
      const int ab2 = ab == 0 ? 2 : ab - 1;       // (kt + 2) % 3: the A image read one K-step ago

      // Fragment reads are asm with counted waits (gemm_common.h "LDS fragment reads the caller waits for").  Per K-tile:
      //   reads P1 (b0 x8, aA x4) + P2 (b1 x6) issued together; P1's MFMAs start group by group as their fragments land;
      //   reads P3 (aB x4: A rows 32..63, its own registers) go out before P2's MFMAs, so nothing is read behind barrier 1.
      constexpr int cb = lfrag_ops<B_KMAJOR>::value, ca = lfrag_ops<A_KMAJOR>::value;
      constexpr int T1 = 14 * cb + 4 * ca;           // LDS operations of the first block of reads

      // ---- phase 1 ---------------------------------------------------------------------------
      if (kt + 1 < nk) dmaB(0, kt + 1, (kt + 1) & 1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (B_KMAJOR) lfrag_read<B_KMAJOR>(fb0[j][ks], Bt, j * 16 + r16, ks * 4 + g, lane);
          else lfrag_read<B_KMAJOR>(fb0[j][ks], Bt, ks * 32 + 8 * g, j * 16, lane);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if constexpr (A_KMAJOR) lfrag_read<A_KMAJOR>(faA[i][ks], At, arow + i * 16 + r16, ks * 4 + g, lane);
          else lfrag_read<A_KMAJOR>(faA[i][ks], At, ks * 32 + 8 * g, arow + i * 16, lane);
        }
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          if constexpr (B_KMAJOR) lfrag_read<B_KMAJOR>(fb1[j][ks], Bt, 64 + j * 16 + r16, ks * 4 + g, lane);
          else lfrag_read<B_KMAJOR>(fb1[j][ks], Bt, ks * 32 + 8 * g, 64 + j * 16, lane);
        }
      __builtin_amdgcn_s_setprio(1);
      {
        bf16x8 vb[4], va;
        // ks = 0, rows 0..15: needs b0[0..3][0], aA[0][0]
        lgkm_wait<T1 - 4 * cb - ca>();
#pragma unroll
        for (int j = 0; j < 4; ++j) vb[j] = lfrag_get<B_KMAJOR>(fb0[j][0]);
        va = lfrag_get<A_KMAJOR>(faA[0][0]);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb[j], va, acc[0][j], 0, 0, 0);
        lgkm_wait<T1 - 4 * cb - 2 * ca>();
        va = lfrag_get<A_KMAJOR>(faA[1][0]);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb[j], va, acc[1][j], 0, 0, 0);
        // ks = 1
        lgkm_wait<T1 - 8 * cb - 3 * ca>();
#pragma unroll
        for (int j = 0; j < 4; ++j) vb[j] = lfrag_get<B_KMAJOR>(fb0[j][1]);
        va = lfrag_get<A_KMAJOR>(faA[0][1]);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb[j], va, acc[0][j], 0, 0, 0);
        lgkm_wait<T1 - 8 * cb - 4 * ca>();
        va = lfrag_get<A_KMAJOR>(faA[1][1]);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb[j], va, acc[1][j], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);

      // ---- phase 2 ---------------------------------------------------------------------------
      if (kt + 1 < nk) dmaB(1, kt + 1, (kt + 1) & 1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i) {              // phase 3's A fragments (rows 32..63): read ahead of barrier 1
          if constexpr (A_KMAJOR) lfrag_read<A_KMAJOR>(faB[i][ks], At, arow + 32 + i * 16 + r16, ks * 4 + g, lane);
          else lfrag_read<A_KMAJOR>(faB[i][ks], At, ks * 32 + 8 * g, arow + 32 + i * 16, lane);
        }
      __builtin_amdgcn_s_setprio(1);
      {
        bf16x8 vb1[3][2], va[2][2];
        lgkm_wait<3 * cb + 4 * ca>();              // b1[*][0]
#pragma unroll
        for (int j = 0; j < 3; ++j) vb1[j][0] = lfrag_get<B_KMAJOR>(fb1[j][0]);
#pragma unroll
        for (int i = 0; i < 2; ++i) va[i][0] = lfrag_get<A_KMAJOR>(faA[i][0]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) acc[i][4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb1[j][0], va[i][0], acc[i][4 + j], 0, 0, 0);
        lgkm_wait<4 * ca>();                       // b1[*][1]
#pragma unroll
        for (int j = 0; j < 3; ++j) vb1[j][1] = lfrag_get<B_KMAJOR>(fb1[j][1]);
#pragma unroll
        for (int i = 0; i < 2; ++i) va[i][1] = lfrag_get<A_KMAJOR>(faA[i][1]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j) acc[i][4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb1[j][1], va[i][1], acc[i][4 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        lgkm_wait<0>();

        // ---- phase 3 ---------------------------------------------------------------------------
        if (kt + 2 < nk) dmaA(0, kt + 2, ab2);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int i = 0; i < 2; ++i) va[i][ks] = lfrag_get<A_KMAJOR>(faB[i][ks]);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[2 + i][4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb1[j][ks], va[i][ks], acc[2 + i][4 + j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);

        // ---- phase 4 ---------------------------------------------------------------------------
        if (kt + 2 < nk) dmaA(1, kt + 2, ab2);
        bf16x8 vb0[4][2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int j = 0; j < 4; ++j) vb0[j][ks] = lfrag_get<B_KMAJOR>(fb0[j][ks]);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[2 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb0[j][ks], va[i][ks], acc[2 + i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
      }
      // everything the NEXT K-tile reads has landed: all copies but the youngest 4 of this iteration (A(kt+2)); near the end of
      // K fewer were issued, so wait for all
      if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      ab = ab == 2 ? 0 : ab + 1;
    }

    }
    STAMP(1);
#ifdef G224_PHASES
    if (a.stamps && lane == 0 && round == 0)
      for (int i = 0; i < 8; ++i) a.stamps[((size_t)(blockIdx.x + gridDim.x * blockIdx.z) * 8 + wave_u) * 8 + i] = pacc[i];
#endif
    const int m0 = bm0 + wm * 64, n0 = bn0 + wn * HN224;
    bool fast = false;
    if constexpr (sizeof(TC) == 2 && !PERSIST) fast = (a.variant != 1) && epi224_fast_ok(a, bm0);
    t += gridDim.x;
    const bool more = PERSIST && t < nt;
    if (fast) {
      // one tile per workgroup.  Issue order (in-order counter: every wait is a count of YOUNGER operations):
      //   bias(7) res0(7) | res1(7) res2(7) | stores p0(4) | res3(7) | stores p1(4) p2(4) p3(4)
      Epi224 e;
      e.init(a, m0, n0, first_slice, lane);
      float4 bias[7];
      uint2 r0[7], r1[7], r2[7];
      e.load_bias(bias);
      e.load_res(0, r0);
      STAMP(2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      e.fold_bias(acc, bias);
      e.load_res(1, r1);
      e.load_res(2, r2);
      e.pass(a, acc[0], r0, 0, m0, n0, stage, lane);
      e.load_res(3, r0);                                               // r0 is free again
      asm volatile("s_waitcnt vmcnt(11)" ::: "memory");                // res1, res2: older than stores p0 (4) + res3 (7)
      e.pass(a, acc[1], r1, 1, m0, n0, stage, lane);
      e.pass(a, acc[2], r2, 2, m0, n0, stage, lane);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                 // res3: older than stores p1, p2
      e.pass(a, acc[3], r0, 3, m0, n0, stage, lane);
      STAMP(3);
      break;                                                           // the stores drain behind the workgroup's end
    } else {
      if (more) {
        tile_origin(t, bm0, bn0);
        prologue();
      }
      STAMP(2);
      store_tile_224<TC>(a, acc, m0, n0, first_slice, stage, lane);
      STAMP(3);
      if (!more) break;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // stores are younger than the DMAs: wait for everything
      __builtin_amdgcn_s_barrier();
    }
    STAMP(4);
    ++round;
  }
}

template <int LAYOUT, typename TC, bool PERSIST, bool MIDBAR>
static int launch_one_p(const GemmArgs& a, int splits, hipStream_t st) {
  auto k = gemm_bf16_224_kernel<LAYOUT, TC, PERSIST, MIDBAR>;
  static std::atomic<bool> attr_set{false};   // idempotent process-wide attribute: a race sets it twice
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    if (e != hipSuccess) { mts_set_error("gemm224: hipFuncSetAttribute: %s", hipGetErrorString(e)); return MTS_ERR_LAUNCH; }
    attr_set = true;
  }
  const int nt = ceil_div(a.M, 256) * (a.N / BN224);
  const int gx = (PERSIST && splits == 1) ? std::min(nt, 256) : nt;
#ifdef MTS_GEMM_STAMPS
  GemmArgs b = a;
  b.stamps = g_gemm_stamps;
  hipLaunchKernelGGL(k, dim3(gx, 1, splits), dim3(512), LDS_TOTAL, st, b);
  return MTS_OK;
#endif
  hipLaunchKernelGGL(k, dim3(gx, 1, splits), dim3(512), LDS_TOTAL, st, a);
  return MTS_OK;
}

// bf16 C: one tile per workgroup (the stores of a tile drain while the CU's next workgroup runs its K loop); fp32 C (weight
// gradients, split-K slabs): persistent.  gemm_variant 5 forces the end-of-tile barrier schedule (A/B).
template <int LAYOUT, typename TC>
static int launch_one(const GemmArgs& a, int splits, hipStream_t st) {
  constexpr bool persist = sizeof(TC) != 2;
  // Schedule per layout, from the in-process A/B on the BASELINE shapes (tools/gemm_ab_libs.py lib@0 lib@5): the mid-tile barrier
  // wins 4-7 % with at least one K-major operand (NT forward, NN data gradient, TT).  With two strided ones (TN weight gradient:
  // two transposing reads per fragment and a register copy where the halves are joined) it measured 4-8 % slower and, with the
  // persistent fp32-C state on top, no longer fits the register file without spilling fragment registers in flight: TN is only
  // built with the end-of-tile barrier.  gemm_variant 5 selects that schedule for every layout (A/B).
  if constexpr (LAYOUT == MTS_TN) {
    return launch_one_p<LAYOUT, TC, persist, false>(a, splits, st);
  } else {
    if (a.variant == 5) return launch_one_p<LAYOUT, TC, persist, false>(a, splits, st);
    return launch_one_p<LAYOUT, TC, persist, true>(a, splits, st);
  }
}

int mts_launch_gemm224r(const GemmArgs& a, int layout, bool c_is_f32, int splits, hipStream_t st);   // gemm224r.hip: four-wave NT form, one tile per workgroup (gemm_bf16_224d_kernel)
int mts_launch_gemm224t(const GemmArgs& a, int layout, bool c_is_f32, int splits, hipStream_t st);   // gemm224t.hip: four-wave weight-gradient (TN) form
int mts_launch_gemm224p(const GemmArgs& a, int layout, bool c_is_f32, int splits, hipStream_t st);   // gemm224p.hip: four-wave persistent forward (NT) form
int mts_launch_gemm224n(const GemmArgs& a, int layout, bool c_is_f32, int splits, hipStream_t st);   // gemm224n.hip: four-wave data-gradient (NN) form

// which kernel the calling thread's last mts_launch_gemm224 ran, as mts_gemm_last_plan reports it: 224, or 225 = gemm_bf16_224n_kernel, 226 =
// gemm_bf16_224d_kernel (symbols of their own in a kernel trace: per-symbol averages of a bench line and of a rocprofv3 CSV must mean the same launches)
static thread_local int g_last224 = 224;
int mts_gemm224_last_kernel() { return g_last224; }

// called from mts_gemm (gemm.hip) when N is a multiple of 224 and the 224-wide tiling fills the CUs better
int mts_launch_gemm224(const GemmArgs& a, int layout, bool c_is_f32, int splits, hipStream_t st) {
  g_last224 = 224;
  // gemm_variant 0 = production.  A/B switches that keep the four-wave kernels: 11 = NN four-wave only where the epilogue has no residual, 12 = NT with a
  // residual on the persistent kernel (both: the round's earlier dispatch); 10 = NN four-wave forced.  6 = the eight-wave kernels, 9 = NT one tile per workgroup.
  const bool dflt = a.variant == 0 || a.variant == 11 || a.variant == 12;
  // bf16 C, NT (the forward projections): the four-wave kernels with buffer-load LDS-DMA where they apply -- persistent (gemm224p.hip), or one tile per
  // workgroup (gemm224r.hip) where the epilogue has a RESIDUAL: that kernel's stages are free at the end of the K loop and take the residual tile,
  // the persistent one's are busy with the next tile and it fetches the residual into registers behind a full wait.  Bitwise the eight-wave kernel.
  if (dflt && !c_is_f32 && layout == MTS_NT && splits == 1) {
    GemmArgs b = a;
    b.variant = 9;
    if ((a.epi & MTS_EPI_RESIDUAL) && a.variant != 12) {
      const int rc1 = mts_launch_gemm224r(b, layout, c_is_f32, splits, st);
      if (rc1 >= 0) { g_last224 = 226; return rc1; }
    }
    int rc = mts_launch_gemm224p(a, layout, c_is_f32, splits, st);
    if (rc >= 0) return rc;
    rc = mts_launch_gemm224r(b, layout, c_is_f32, splits, st);
    if (rc >= 0) { g_last224 = 226; return rc; }
  }
  // bf16 C, NN (the data gradients): the four-wave kernel with a k-strided B (gemm224n.hip) where it applies
  if ((dflt || a.variant == 10) && !c_is_f32 && layout == MTS_NN && splits == 1) {
    const int rc = mts_launch_gemm224n(a, layout, c_is_f32, splits, st);
    if (rc >= 0) { g_last224 = 225; return rc; }
  }
  // fp32 C, TN (the weight gradients): the four-wave unit-pipelined kernel (gemm224t.hip) where it applies
  if (dflt && c_is_f32 && layout == MTS_TN) {
    const int rc = mts_launch_gemm224t(a, layout, c_is_f32, splits, st);
    if (rc >= 0) return rc;
  }
  if (a.variant == 9) {                                   // A/B: the one-tile-per-workgroup four-wave kernel where it applies
    const int rc = mts_launch_gemm224r(a, layout, c_is_f32, splits, st);
    if (rc >= 0) { g_last224 = 226; return rc; }
  }
  if (c_is_f32) {
    if (layout == MTS_NT) return launch_one<MTS_NT, float>(a, splits, st);
    if (layout == MTS_NN) return launch_one<MTS_NN, float>(a, splits, st);
    if (layout == MTS_TT) return launch_one<MTS_TT, float>(a, splits, st);
    return launch_one<MTS_TN, float>(a, splits, st);
  }
  if (layout == MTS_NT) return launch_one<MTS_NT, bf16_t>(a, splits, st);
  if (layout == MTS_NN) return launch_one<MTS_NN, bf16_t>(a, splits, st);
  if (layout == MTS_TT) return launch_one<MTS_TT, bf16_t>(a, splits, st);
  return launch_one<MTS_TN, bf16_t>(a, splits, st);
}
