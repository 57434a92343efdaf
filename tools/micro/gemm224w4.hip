// EXPERIMENT, NOT BUILT INTO libmts_hip.so (round 2): measured 0.82 PFLOP/s on the forward Q|K|V projection against 1.06-1.17 for the
// 8-wave kernel (gemm224.hip) and 1.35 for the vendor library in the same process, bitwise equal results.  What it taught: with ONE
// wave per SIMD every global_load_lds issue (m0 write + hazard nops + the instruction: 60-190 cycles each, 16 per K-tile and wave)
// stalls that SIMD's only MFMA stream -- ~2,800 of a K-tile's ~4,600 cycles were not MFMA.  A four-wave kernel has to stage its
// operands through registers (plain 16-byte loads cost a few issue cycles each) as the vendor kernel does; the LDS-DMA path needs a
// partner wave per SIMD to issue behind.  Kept as the record of that measurement (to build: add it to build.py's SOURCES and
// restore the mts_launch_gemm224w4 call in gemm.hip).
//
// 256x224x64 bf16 MFMA GEMM, FOUR waves per workgroup (one per SIMD), wave tile 128 x 112, fragments double-buffered in registers.
//
// Why a second form of gemm224.hip.  rocprofv3 counters on the BASELINE forward projection (profiles/r02_gemm_pmc_vs_vendor.txt) put the
// 8-wave kernel's matrix cores at ~51 % busy against ~70 % for the vendor library's 256x256x64 kernel (four waves, one per SIMD):
// its two waves per SIMD run the same program in lockstep, so both wait for LDS behind every workgroup barrier and both want the
// matrix pipe at the same moment.  Here a wave owns its SIMD and hides LDS latency INSIDE its own instruction stream:
//
//   K-tile kt = two 32-deep steps ("sets"); a wave's fragments of a set are 8 A + 7 B reads (60 registers), two sets live.
//     S1  MFMA x56 on set 0 of kt                      (set 1 of kt was requested before them: it lands underneath)
//     S2  all LDS reads of kt are done -> wait for my copies of kt+1 -> BARRIER -> A image kt%3 and B image kt&1 are free:
//         issue copies B(kt+2), A(kt+3); request set 0 of kt+1
//     S3  MFMA x56 on set 1 of kt                      (set 0 of kt+1 lands underneath); request set 1 of kt+1
//   One barrier per K-tile, and the only thing a wave ever waits for outside a barrier is a read issued ~56 MFMAs earlier.
//   A (activations: L2 misses) is copied three K-tiles ahead into three images, B (weights: L2 hits) two ahead into two.
//
// 512 registers per lane are available to a wave that is alone on its SIMD: 224 accumulators + 120 fragment registers.
#include <algorithm>
#include <type_traits>
#include <math.h>
#include "gemm_common.h"

#define W4_HT 16384                    // one half image: 128 rows x 64 k (K-major) or 64 k-rows x 128 columns (strided)
#define W4_A_BYTES (2 * W4_HT)
#define W4_B_BASE (3 * W4_A_BYTES)
#define W4_LDS (3 * W4_A_BYTES + 2 * 2 * W4_HT)      // 160 KiB
#define W4_BN 224
#define W4_HN 112

// 16 pieces of 1 KiB per half image, 4 per wave (same source-side swizzles as gemm224.hip / gemm256.hip)
template <bool KMAJOR>
__device__ __forceinline__ void w4_dma_half(const bf16_t* __restrict__ G, int ld, int row0, int dim, int k0, char* dst, int wave_u, int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int piece = wave_u * 4 + i;
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

__device__ __forceinline__ bool w4_fast_ok(const GemmArgs& a, int bm0) {
  const unsigned simple = MTS_EPI_BIAS | MTS_EPI_COLSCALE | MTS_EPI_RESIDUAL;
  return !a.slab && (a.epi & ~simple) == 0 && bm0 + 256 <= a.M && (a.ldc % 8 == 0) && (((uintptr_t)a.C & 15) == 0) &&
         (!(a.epi & MTS_EPI_COLSCALE) || a.ncols_scaled % 4 == 0) &&
         (!(a.epi & MTS_EPI_RESIDUAL) || (a.ldr % 4 == 0 && ((uintptr_t)a.residual & 7) == 0)) &&
         (!(a.epi & MTS_EPI_BIAS) || ((uintptr_t)a.bias & 15) == 0);
}

template <int LAYOUT, typename TC, bool PERSIST>
__global__ __launch_bounds__(256, 1) void gemm_bf16_224w4_kernel(const GemmArgs a) {
  constexpr bool A_KMAJOR = (LAYOUT == MTS_NT || LAYOUT == MTS_NN);
  constexpr bool B_KMAJOR = (LAYOUT == MTS_NT || LAYOUT == MTS_TT);
  constexpr int ca = lfrag_ops<A_KMAJOR>::value, cb = lfrag_ops<B_KMAJOR>::value;
  constexpr int SET_OPS = 8 * ca + 7 * cb;              // LDS operations of one fragment set
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [3][A0 | A1]  [2][B0 | B1]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave_u >> 1, wn = wave_u & 1;
  const int r16 = lane & 15, g = lane >> 4;

  const int ntn = a.N / W4_BN;
  const int ntm = (a.M + 255) / 256;
  const int nt = ntn * ntm;
  const int kbeg = blockIdx.z * a.ksplit;
  const int kend = min(a.K, kbeg + a.ksplit);
  const int nk = (kend - kbeg) / BK;
  const bool first_slice = (blockIdx.z == 0);

  const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(a.A);
  const bf16_t* __restrict__ B = reinterpret_cast<const bf16_t*>(a.B);

  auto tile_origin = [&](int t, int& bm0, int& bn0) {       // XCD-aware 4-row bands, as gemm224.hip
    const int q = nt >> 3, rr = nt & 7, xcd = t & 7, idx = t >> 3;
    const int id = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    if (a.order == 0) { bm0 = (id / ntn) * 256; bn0 = (id % ntn) * W4_BN; return; }
    const int band = id / (4 * ntn), within = id - band * 4 * ntn;
    const int rows = min(4, ntm - band * 4);
    bm0 = (band * 4 + within % rows) * 256;
    bn0 = (within / rows) * W4_BN;
  };
  int bm0, bn0;
  auto dmaA = [&](int kt) {                                  // 8 copy instructions per wave
    char* dst = smem + (kt % 3) * W4_A_BYTES;
    w4_dma_half<A_KMAJOR>(A, a.lda, bm0, a.M, kbeg + kt * BK, dst, wave_u, lane);
    w4_dma_half<A_KMAJOR>(A, a.lda, bm0 + 128, a.M, kbeg + kt * BK, dst + W4_HT, wave_u, lane);
  };
  auto dmaB = [&](int kt) {                                  // 8 copy instructions per wave
    char* dst = smem + W4_B_BASE + (kt & 1) * 2 * W4_HT;
    w4_dma_half<B_KMAJOR>(B, a.ldb, bn0, a.N, kbeg + kt * BK, dst, wave_u, lane);
    w4_dma_half<B_KMAJOR>(B, a.ldb, bn0 + W4_HN, a.N, kbeg + kt * BK, dst + W4_HT, wave_u, lane);
  };

  f32x4 acc[8][7];
  LFrag<A_KMAJOR> fa[2][8];
  LFrag<B_KMAJOR> fb[2][7];
  auto request = [&](int kt, int set) {                     // the 15 fragment reads of (K-tile, set) for this wave
    const char* At = smem + (kt % 3) * W4_A_BYTES + wm * W4_HT;
    const char* Bt = smem + W4_B_BASE + ((kt & 1) * 2 + wn) * W4_HT;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      if constexpr (B_KMAJOR) lfrag_read<B_KMAJOR>(fb[set][j], Bt, j * 16 + r16, set * 4 + g, lane);
      else lfrag_read<B_KMAJOR>(fb[set][j], Bt, set * 32 + 8 * g, j * 16, lane);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (A_KMAJOR) lfrag_read<A_KMAJOR>(fa[set][i], At, i * 16 + r16, set * 4 + g, lane);
      else lfrag_read<A_KMAJOR>(fa[set][i], At, set * 32 + 8 * g, i * 16, lane);
    }
  };
  auto mfma_set = [&](int set) {
    bf16x8 vb[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) vb[j] = lfrag_get<B_KMAJOR>(fb[set][j]);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bf16x8 va = lfrag_get<A_KMAJOR>(fa[set][i]);
#pragma unroll
      for (int j = 0; j < 7; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb[j], va, acc[i][j], 0, 0, 0);
    }
  };

  int t = blockIdx.x;
  if (t >= nt) return;
  tile_origin(t, bm0, bn0);
  for (;;) {
    // ---- prologue of a tile: B(0) A(0) | B(1) A(1) | A(2); every image is free (first tile, or behind the barrier of the epilogue)
    if (nk > 0) { dmaB(0); dmaA(0); }
    if (nk > 1) { dmaB(1); dmaA(1); }
    if (nk > 2) dmaA(2);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 7; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // K-tile 0 has landed when everything but the copies of K-tiles 1, 2 is done
    if (nk > 2) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if (nk > 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    request(0, 0);
    request(0, 1);

#pragma clang loop unroll(disable)
    for (int kt = 0; kt < nk; ++kt) {
      // S1 ------------------------------------------------------------------------------------------------------------
      lgkm_wait<SET_OPS>();                        // set 0 of kt (set 1 of kt was requested after it and may still be in flight)
      mfma_set(0);
      // S2 ------------------------------------------------------------------------------------------------------------
      lgkm_wait<0>();                              // set 1 of kt: this wave will not read K-tile kt's images again
      if (kt + 1 < nk) {
        // my copies of K-tile kt+1: everything but the 8 copies of A(kt+2) issued behind them (none near the end of K)
        if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) dmaB(kt + 2);             // into the B image K-tile kt has just left
        if (kt + 3 < nk) dmaA(kt + 3);             // into the A image K-tile kt has just left
        request(kt + 1, 0);
      }
      // S3 ------------------------------------------------------------------------------------------------------------
      mfma_set(1);
      if (kt + 1 < nk) request(kt + 1, 1);
    }

    // ---- epilogue ------------------------------------------------------------------------------------------------------
    const int m0 = bm0 + wm * 128, n0 = bn0 + wn * W4_HN;
    bool fast = false;
    if constexpr (sizeof(TC) == 2) fast = w4_fast_ok(a, bm0);
    t += gridDim.x;
    const bool more = PERSIST && t < nt;
    __builtin_amdgcn_s_barrier();                  // every wave is out of the K loop: the A images may serve as store staging
    char* stage = smem + wave_u * 4096;            // wave-private 16 rows x 240 B
    if (fast) {
      // every global load of the epilogue ahead of every store (in-order counter: a load behind a store is waited for with it)
      const bool has_bias = (a.epi & MTS_EPI_BIAS) && first_slice, has_res = (a.epi & MTS_EPI_RESIDUAL) && first_slice;
      const float* bias_p = has_bias ? a.bias + n0 + 4 * g : reinterpret_cast<const float*>(a.A) + 4 * g;
      const size_t res_ld = has_res ? (size_t)a.ldr : 0;
      const bf16_t* res_p = has_res ? reinterpret_cast<const bf16_t*>(a.residual) + (size_t)(m0 + r16) * a.ldr + n0 + 4 * g
                                    : reinterpret_cast<const bf16_t*>(a.A) + 4 * g;
      float4 bias[7];
      uint2 res[8][7];
#pragma unroll
      for (int j = 0; j < 7; ++j) bias[j] = *reinterpret_cast<const float4*>(bias_p + j * 16);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) res[i][j] = *reinterpret_cast<const uint2*>(res_p + (size_t)(i * 16) * res_ld + j * 16);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const float colscale = (a.epi & MTS_EPI_COLSCALE) ? a.colscale : 1.0f;
      const int nsc = (a.epi & MTS_EPI_COLSCALE) ? a.ncols_scaled - n0 - 4 * g : 0;
      bf16_t* __restrict__ C = reinterpret_cast<bf16_t*>(a.C);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
#pragma unroll
        for (int j = 0; j < 7; ++j) {
          const float sc = (j * 16 < nsc) ? colscale : 1.0f;
          const float4 bb = has_bias ? bias[j] : make_float4(0.f, 0.f, 0.f, 0.f);
          const uint2 rr = has_res ? res[i][j] : make_uint2(0u, 0u);
          uint2 pk;
          pk.x = pack_bf16x2((acc[i][j][0] + bb.x) * sc + bf16_lo(rr.x), (acc[i][j][1] + bb.y) * sc + bf16_hi(rr.x));
          pk.y = pack_bf16x2((acc[i][j][2] + bb.z) * sc + bf16_lo(rr.y), (acc[i][j][3] + bb.w) * sc + bf16_hi(rr.y));
          *reinterpret_cast<uint2*>(stage + r16 * 240 + (j * 16 + 4 * g) * 2) = pk;
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
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int m = m0 + i * 16 + r16;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
          const int n = n0 + j * 16 + 4 * g;
          float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
          epilogue4<bf16_t, TC>(a, m, n, v, first_slice);
        }
      }
    }
    if (!more) break;
    tile_origin(t, bm0, bn0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                  // staging reads are done: the images may be filled again
  }
}

template <int LAYOUT, typename TC>
static int w4_launch(const GemmArgs& a, int splits, hipStream_t st) {
  constexpr bool PERSIST = sizeof(TC) != 2;        // bf16 C: one tile per workgroup (stores drain behind the workgroup's end)
  auto k = gemm_bf16_224w4_kernel<LAYOUT, TC, PERSIST>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS);
    if (e != hipSuccess) { mts_set_error("gemm224w4: hipFuncSetAttribute: %s", hipGetErrorString(e)); return MTS_ERR_LAUNCH; }
    attr_set = true;
  }
  const int nt = ceil_div(a.M, 256) * (a.N / W4_BN);
  const int gx = (PERSIST && splits == 1) ? std::min(nt, 256) : nt;
  hipLaunchKernelGGL(k, dim3(gx, 1, splits), dim3(256), W4_LDS, st, a);
  return MTS_OK;
}

// called from mts_gemm (gemm.hip): the four-wave form of the 256x224 tile
int mts_launch_gemm224w4(const GemmArgs& a, int layout, bool c_is_f32, int splits, hipStream_t st) {
  if (c_is_f32) {
    if (layout == MTS_NT) return w4_launch<MTS_NT, float>(a, splits, st);
    if (layout == MTS_NN) return w4_launch<MTS_NN, float>(a, splits, st);
    if (layout == MTS_TT) return w4_launch<MTS_TT, float>(a, splits, st);
    return w4_launch<MTS_TN, float>(a, splits, st);
  }
  if (layout == MTS_NT) return w4_launch<MTS_NT, bf16_t>(a, splits, st);
  if (layout == MTS_NN) return w4_launch<MTS_NN, bf16_t>(a, splits, st);
  if (layout == MTS_TT) return w4_launch<MTS_TT, bf16_t>(a, splits, st);
  return w4_launch<MTS_TN, bf16_t>(a, splits, st);
}
