// 256x256x64 bf16 MFMA GEMM for gfx950: one 512-thread workgroup (8 waves = 2 per SIMD) per CU, 128 KiB of LDS.
//
// Why a second kernel: the 128x128 form (gemm.hip) keeps only ONE K-tile in flight behind the MFMAs, so ~40 % of
// its wave-cycles are spent at the barrier waiting for the LDS-DMA (SQ_WAIT_ANY, profiles/).  Here the tile is
// split into four half-tile images (A0, A1: 128 rows x 64 k; B0, B1: 128 cols x 64 k; 16 KiB each, the same two
// swizzled images as gemm.hip) in two K-tile buffers, and the LDS-DMA runs 1 K-tile ahead for A and 2 K-tiles
// ahead for B behind a COUNTED s_waitcnt vmcnt(4) that is issued once per K-tile, with raw s_barrier (a
// __syncthreads() would drain the DMA queue):
//
//   K-tile t (buffer t&1), wave (wm, wn) owns C rows wm*128..+127 (all of A[wm]) and cols wn*64..+63 (half of B[wn>>1]):
//     phase 1: DMA A0(t+1) ; read A[rows 0..63], B[cols 0..31]  ; 16 MFMA  quadrant (0,0)
//     phase 2: DMA A1(t+1) ; read B[cols 32..63]                ; 16 MFMA  quadrant (0,1) ; barrier   (B of this buffer is now free)
//     phase 3: DMA B0(t+2) ; read A[rows 64..127]               ; 16 MFMA  quadrant (1,1)
//     phase 4: DMA B1(t+2) ;                                      16 MFMA  quadrant (1,0) ; vmcnt(4) ; barrier
//
//   RAW: every wave waits for its own A(t+1)/B(t+1) pieces (all but the 4 youngest = B(t+2)) before the K-tile's
//        last barrier; reads of K-tile t+1 come after that barrier.
//   WAR: A(t+1) lands in the other buffer, last read in phase 3 of K-tile t-1 (two barriers ago); B(t+2) lands in
//        this buffer's B images, last read in phase 2 (one barrier ago; the (1,0) quadrant reuses B fragments kept
//        in registers).
// 24 ds_read_b128 (or 48 ds_read_b64_tr_b16) feed 64 MFMAs per wave and K-tile, vs 32 in the 128x128 kernel.
#include <algorithm>
#include <math.h>
#include "gemm_common.h"

#define HT_BYTES 16384
#define BUF_BYTES (4 * HT_BYTES)
#define LDS_TOTAL (2 * BUF_BYTES + 8 * 4096)   // 160 KiB: two K-tile buffers + per-wave store staging

template <bool KMAJOR>
__device__ __forceinline__ void dma_half(const bf16_t* __restrict__ G, int ld, int row0, int dim, int k0, char* dst, int wave_u, int lane) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
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

// C tile of one wave (128 x 64) -> global.  bf16 output goes through a wave-private 4 KiB LDS staging area, 32 rows at
// a time, so that HBM sees full 128-byte row segments (16 B per lane, 8 lanes per row) instead of the 8-byte pieces
// the MFMA accumulator layout yields (at K = 1792 the C write is ~70 % of a GEMM's HBM bytes).
template <typename TC>
__device__ __forceinline__ void store_tile_256(const GemmArgs& a, f32x4 (&acc)[8][4], int m0, int n0, bool first_slice, char* stage,
                                               int lane) {
  const int r16 = lane & 15, g = lane >> 4;
  const bool vec_ok = (a.N % 8 == 0) && (a.ldc % 8 == 0);
  if constexpr (sizeof(TC) == 2) {
    if (!a.slab && vec_ok) {
      // the bias of this lane's 4 x 4 columns is fetched ONCE, ahead of every store (a load between the stores is waited for
      // with vmcnt(0), i.e. together with the stores before it), and the compiler is told -- with the builtin, which its wait
      // insertion understands -- that nothing is outstanding any more (vmcnt = 0; expcnt / lgkmcnt unconstrained)
      float4 bj[4];
      const bool pre_bias = (a.epi & MTS_EPI_BIAS) && first_slice && !(a.epi & MTS_EPI_RESIDUAL) && (((uintptr_t)a.bias & 15) == 0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        bj[j] = pre_bias ? *reinterpret_cast<const float4*>(a.bias + min(n0 + j * 16 + 4 * g, a.N - 4)) : make_float4(0.f, 0.f, 0.f, 0.f);
      GemmArgs a2 = a;
      if (pre_bias) a2.epi = a.epi & ~MTS_EPI_BIAS;
      __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
#pragma unroll
        for (int ii = 0; ii < 2; ++ii) {
          const int i = 2 * pass + ii;
          const int row = ii * 16 + r16;
          const int m = m0 + i * 16 + r16;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int n = n0 + j * 16 + 4 * g;
            float v[4] = {acc[i][j][0] + bj[j].x, acc[i][j][1] + bj[j].y, acc[i][j][2] + bj[j].z, acc[i][j][3] + bj[j].w};
            if (m < a.M && n < a.N) epi_math4<bf16_t>(a2, m, n, v, first_slice);
            const int chunk = (2 * j + (g >> 1)) ^ (row & 7);
            uint2 pk;
            pk.x = pack_bf16x2(v[0], v[1]);
            pk.y = pack_bf16x2(v[2], v[3]);
            *reinterpret_cast<uint2*>(stage + row * 128 + (chunk << 4) + ((g & 1) << 3)) = pk;
          }
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int row = rr * 8 + (lane >> 3), ch = lane & 7;
          const uint4 val = *reinterpret_cast<const uint4*>(stage + row * 128 + ((ch ^ (row & 7)) << 4));
          const int m = m0 + pass * 32 + row, n = n0 + ch * 8;
          if (m < a.M && n < a.N) *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.C) + (size_t)m * a.ldc + n) = val;
        }
      }
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int m = m0 + i * 16 + r16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + j * 16 + 4 * g;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      epilogue4<bf16_t, TC>(a, m, n, v, first_slice);
    }
  }
}

// PERSIST = false (bf16 C): one tile per workgroup, so that a tile's stores drain while the CU's next workgroup runs its K loop
template <int LAYOUT, typename TC, bool PERSIST>
__global__ __launch_bounds__(512, 2) void gemm_bf16_256_kernel(const GemmArgs a) {
  constexpr bool A_KMAJOR = (LAYOUT == MTS_NT || LAYOUT == MTS_NN);
  constexpr bool B_KMAJOR = (LAYOUT == MTS_NT || LAYOUT == MTS_TT);
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2 buffers][A0 | A1 | B0 | B1] + 8 x 4 KiB store staging

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave_u >> 2, wn = wave_u & 3;
  const int r16 = lane & 15, g = lane >> 4;
  char* stage = smem + 2 * BUF_BYTES + wave_u * 4096;

  const int ntn = (a.N + 255) / 256;
  const int ntm = (a.M + 255) / 256;
  const int nt = ntn * ntm;
  const int kbeg = blockIdx.z * a.ksplit;
  const int kend = min(a.K, kbeg + a.ksplit);
  const int nk = (kend - kbeg) / BK;
  const bool first_slice = (blockIdx.z == 0);

  const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(a.A);
  const bf16_t* __restrict__ B = reinterpret_cast<const bf16_t*>(a.B);

  // persistent over output tiles: workgroup w takes tiles w, w + gridDim.x, ...; the XCD remap keeps the tiles
  // that run at the same time on one XCD adjacent in N (they share the A panel in that XCD's L2)
  auto tile_origin = [&](int t, int& bm0, int& bn0) {
    const int q = nt >> 3, rr = nt & 7, xcd = t & 7, idx = t >> 3;
    const int id = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    bm0 = (id / ntn) * 256;
    bn0 = (id % ntn) * 256;
  };
  int bm0, bn0;
  auto dmaA = [&](int h, int kt) {
    dma_half<A_KMAJOR>(A, a.lda, bm0 + h * 128, a.M, kbeg + kt * BK, smem + (kt & 1) * BUF_BYTES + h * HT_BYTES, wave_u, lane);
  };
  auto dmaB = [&](int h, int kt) {
    dma_half<B_KMAJOR>(B, a.ldb, bn0 + h * 128, a.N, kbeg + kt * BK, smem + (kt & 1) * BUF_BYTES + (2 + h) * HT_BYTES, wave_u, lane);
  };
  auto prologue = [&]() {
    if (nk > 0) {
      dmaB(0, 0); dmaB(1, 0); dmaA(0, 0); dmaA(1, 0);
      if (nk > 1) { dmaB(0, 1); dmaB(1, 1); }
    }
  };

  f32x4 acc[8][4];
  bf16x8 af[4][2], b0[2][2], b1[2][2];
  frag_raw raf[4][2], rb0[2][2], rb1[2][2];      // transposed reads in flight (strided operands only; gemm_common.h)
  const int bcol = (wn & 1) * 64;

  int t = blockIdx.x;
  if (t >= nt) return;
  tile_origin(t, bm0, bn0);
  prologue();
  // the first K-tile needs everything but B(1): 4 youngest DMAs may stay in flight
  if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  for (;;) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int kt = 0; kt < nk; ++kt) {
      const char* At = smem + (kt & 1) * BUF_BYTES + wm * HT_BYTES;
      const char* Bt = smem + (kt & 1) * BUF_BYTES + (2 + (wn >> 1)) * HT_BYTES;

      // ---- phase 1 ---------------------------------------------------------------------------
      if (kt + 1 < nk) dmaA(0, kt + 1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (B_KMAJOR) b0[j][ks] = frag_kmajor(Bt, bcol + j * 16 + r16, ks * 4 + g);
          else rb0[j][ks] = frag_strided(Bt, ks * 32 + 8 * g, bcol + j * 16, lane);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if constexpr (A_KMAJOR) af[i][ks] = frag_kmajor(At, i * 16 + r16, ks * 4 + g);
          else raf[i][ks] = frag_strided(At, ks * 32 + 8 * g, i * 16, lane);
        }
      }
      if constexpr (!A_KMAJOR || !B_KMAJOR) {            // transposed reads are asm: wait for them by hand (gemm_common.h)
        lds_frags_wait();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int j = 0; j < 2; ++j) if constexpr (!B_KMAJOR) b0[j][ks] = frag_finish(rb0[j][ks]);
#pragma unroll
          for (int i = 0; i < 4; ++i) if constexpr (!A_KMAJOR) af[i][ks] = frag_finish(raf[i][ks]);
        }
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[j][ks], af[i][ks], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);

      // ---- phase 2 ---------------------------------------------------------------------------
      if (kt + 1 < nk) dmaA(1, kt + 1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr (B_KMAJOR) b1[j][ks] = frag_kmajor(Bt, bcol + 32 + j * 16 + r16, ks * 4 + g);
          else rb1[j][ks] = frag_strided(Bt, ks * 32 + 8 * g, bcol + 32 + j * 16, lane);
        }
      if constexpr (!B_KMAJOR) {
        lds_frags_wait();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int j = 0; j < 2; ++j) b1[j][ks] = frag_finish(rb1[j][ks]);
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[j][ks], af[i][ks], acc[i][2 + j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();          // every wave is done reading this buffer's B images

      // ---- phase 3 ---------------------------------------------------------------------------
      if (kt + 2 < nk) dmaB(0, kt + 2);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if constexpr (A_KMAJOR) af[i][ks] = frag_kmajor(At, 64 + i * 16 + r16, ks * 4 + g);
          else raf[i][ks] = frag_strided(At, ks * 32 + 8 * g, 64 + i * 16, lane);
        }
      if constexpr (!A_KMAJOR) {
        lds_frags_wait();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int i = 0; i < 4; ++i) af[i][ks] = frag_finish(raf[i][ks]);
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[4 + i][2 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[j][ks], af[i][ks], acc[4 + i][2 + j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);

      // ---- phase 4 ---------------------------------------------------------------------------
      if (kt + 2 < nk) dmaB(1, kt + 2);
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[j][ks], af[i][ks], acc[4 + i][j], 0, 0, 0);
      __builtin_amdgcn_s_setprio(0);
      if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");   // all but B(kt+2) have landed
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }

    // all waves are past the last barrier: both buffers are free, so the next tile's first DMAs fly under this
    // tile's epilogue stores
    const int m0 = bm0 + wm * 128, n0 = bn0 + wn * 64;
    t += gridDim.x;
    const bool more = PERSIST && t < nt;
    if (more) {
      tile_origin(t, bm0, bn0);
      prologue();
    }
    store_tile_256<TC>(a, acc, m0, n0, first_slice, stage, lane);
    if (!more) break;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // stores are younger than the DMAs: wait for everything
    __builtin_amdgcn_s_barrier();
  }
}

template <int LAYOUT, typename TC>
static int launch_one(const GemmArgs& a, int splits, hipStream_t st) {
  constexpr bool PERSIST = sizeof(TC) != 2;
  auto k = gemm_bf16_256_kernel<LAYOUT, TC, PERSIST>;
  static std::atomic<bool> attr_set{false};   // idempotent process-wide attribute: a race sets it twice
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    if (e != hipSuccess) { mts_set_error("gemm256: hipFuncSetAttribute: %s", hipGetErrorString(e)); return MTS_ERR_LAUNCH; }
    attr_set = true;
  }
  const int nt = ceil_div(a.M, 256) * ceil_div(a.N, 256);
  const int gx = (PERSIST && splits == 1) ? std::min(nt, 256) : nt;      // fp32 C: persistent over tiles when K is not split
  hipLaunchKernelGGL(k, dim3(gx, 1, splits), dim3(512), LDS_TOTAL, st, a);
  return MTS_OK;
}

// called from mts_gemm (gemm.hip) when the shape suits the big tile
int mts_launch_gemm256(const GemmArgs& a, int layout, bool c_is_f32, int splits, hipStream_t st) {
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
