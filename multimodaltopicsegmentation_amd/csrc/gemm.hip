// Dense projection GEMMs for gfx950.
//
//   gemm_bf16_kernel : bf16 operands, fp32 accumulate on the matrix cores (v_mfma_f32_16x16x32_bf16),
//                      128x128x64 block tile, 4 waves (2x2) of 64x64, register-staged global->LDS with
//                      the next tile's loads in flight behind the current tile's MFMAs, double-buffered
//                      LDS (one barrier per K-step), XOR-swizzled LDS images so that both the row-wise
//                      ds_read_b128 fragment reads and the transposed ds_read_b64_tr_b16 reads are
//                      bank-conflict free.  All three layouts of mts.h (NT/NN/TN) share the kernel:
//                      a K-contiguous operand is read row-wise, an M-/N-contiguous operand through the
//                      hardware transpose read, so neither activations nor weights are ever transposed
//                      in HBM.
//   gemm_f32_kernel  : exact-fp32 VALU kernel with arbitrary strides (parity mode; deterministic).
//
// The accumulator tile is computed TRANSPOSED (the W fragment is the MFMA A operand) so that each lane
// ends up with 4 consecutive output columns of one row: bias/residual/aux/C are accessed as 8/16-byte
// vectors in the epilogue.
#include <atomic>
#include <algorithm>
#include <math.h>
#include <stdlib.h>
#include "common.h"

#include "gemm_common.h"


// GLDS = true: both operands go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging VGPRs, no ds_write
// instructions -- the ds_write_b128 pass of the register-staged form costs about as many LDS cycles as all the
// fragment reads of a K-step).  A wave-instruction writes 1 KiB lane-linear, so the XOR swizzles of the two LDS
// images are applied to the per-lane SOURCE address (same involution as on the read side).  Needs K % 64 == 0;
// rows/columns past M/N are clamped to valid memory (they only feed outputs that are never stored).
template <int LAYOUT, typename TC, bool GLDS>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const GemmArgs a) {
  constexpr bool A_KMAJOR = (LAYOUT == MTS_NT || LAYOUT == MTS_NN);
  constexpr bool B_KMAJOR = (LAYOUT == MTS_NT || LAYOUT == MTS_TT);
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][A stage | B stage]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r16 = lane & 15, g = lane >> 4;

  // XCD-aware tile order: blocks b, b+8, b+16.. share an XCD (round-robin dispatch), so give each XCD a
  // contiguous run of tiles; consecutive tiles walk N first, sharing the same A panel in that XCD's L2.
  const int ntn = (a.N + BN - 1) / BN;
  const int ntm = (a.M + BM - 1) / BM;
  const int nt = ntn * ntm;
  int bid = blockIdx.x;
  {
    const int q = nt >> 3, rr = nt & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
  }
  // ... walking the SHORTER side of the tile grid first: with 2 x 14 tiles (feed-forward weight gradient, M = 256) the two tiles of a column
  // run side by side on one XCD and the wide operand's panel is fetched once, not once per tile row
  const int bm0 = (ntm < ntn ? bid % ntm : bid / ntn) * BM;
  const int bn0 = (ntm < ntn ? bid / ntm : bid % ntn) * BN;
  const int kbeg = blockIdx.z * a.ksplit;
  const int kend = min(a.K, kbeg + a.ksplit);
  const int nk = (kend - kbeg + BK - 1) / BK;

  const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(a.A);
  const bf16_t* __restrict__ B = reinterpret_cast<const bf16_t*>(a.B);

  uint4 ra[4], rb[4];
  auto load_tile = [&](int kt) {
    const int k0 = kbeg + kt * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * i;
      if constexpr (A_KMAJOR) {
        const int row = c >> 3, ch = c & 7;
        const int gm = bm0 + row, gk = k0 + ch * 8;
        ra[i] = (gm < a.M && gk < kend) ? *reinterpret_cast<const uint4*>(A + (size_t)gm * a.lda + gk) : make_uint4(0, 0, 0, 0);
      } else {
        const int kr = c >> 4, ch = c & 15;
        const int gk = k0 + kr, gm = bm0 + ch * 8;
        ra[i] = (gk < kend && gm < a.M) ? *reinterpret_cast<const uint4*>(A + (size_t)gk * a.lda + gm) : make_uint4(0, 0, 0, 0);
      }
      if constexpr (B_KMAJOR) {
        const int row = c >> 3, ch = c & 7;
        const int gn = bn0 + row, gk = k0 + ch * 8;
        rb[i] = (gn < a.N && gk < kend) ? *reinterpret_cast<const uint4*>(B + (size_t)gn * a.ldb + gk) : make_uint4(0, 0, 0, 0);
      } else {
        const int kr = c >> 4, ch = c & 15;
        const int gk = k0 + kr, gn = bn0 + ch * 8;
        rb[i] = (gk < kend && gn < a.N) ? *reinterpret_cast<const uint4*>(B + (size_t)gk * a.ldb + gn) : make_uint4(0, 0, 0, 0);
      }
    }
  };
  auto store_tile = [&](int buf) {
    char* sa = smem + buf * (2 * TILE_BYTES);
    char* sb = sa + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + 256 * i;
      if constexpr (A_KMAJOR) {
        *reinterpret_cast<uint4*>(sa + kmajor_off(c >> 3, c & 7)) = ra[i];
      } else {
        const int kr = c >> 4, ch = c & 15;
        *reinterpret_cast<uint4*>(sa + kr * 256 + ((((ch >> 1) ^ strided_key(kr))) << 5) + ((ch & 1) << 4)) = ra[i];
      }
      if constexpr (B_KMAJOR) {
        *reinterpret_cast<uint4*>(sb + kmajor_off(c >> 3, c & 7)) = rb[i];
      } else {
        const int kr = c >> 4, ch = c & 15;
        *reinterpret_cast<uint4*>(sb + kr * 256 + ((((ch >> 1) ^ strided_key(kr))) << 5) + ((ch & 1) << 4)) = rb[i];
      }
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  auto dma_tile = [&](int kt, int buf) {
    const int k0 = kbeg + kt * BK;
    char* sa = smem + buf * (2 * TILE_BYTES);
    char* sb = sa + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int piece = wave_u * 4 + i;            // 1 KiB piece of the 16 KiB image, wave-uniform
      const bf16_t* srca;
      if constexpr (A_KMAJOR) {
        const int row = piece * 8 + (lane >> 3), pos = lane & 7;
        const int ch = pos ^ ((row >> 1) & 7);
        srca = A + (size_t)min(bm0 + row, a.M - 1) * a.lda + k0 + ch * 8;
      } else {
        const int kr = piece * 4 + (lane >> 4), c16 = lane & 15;
        const int ch = ((((c16 >> 1) ^ strided_key(kr))) << 1) | (c16 & 1);
        srca = A + (size_t)(k0 + kr) * a.lda + min(bm0 + ch * 8, a.M - 8);
      }
      __builtin_amdgcn_global_load_lds((gptr_t*)srca, (lptr_t*)(sa + piece * 1024), 16, 0, 0);
      const bf16_t* srcb;
      if constexpr (B_KMAJOR) {
        const int row = piece * 8 + (lane >> 3), pos = lane & 7;
        const int ch = pos ^ ((row >> 1) & 7);
        srcb = B + (size_t)min(bn0 + row, a.N - 1) * a.ldb + k0 + ch * 8;
      } else {
        const int kr = piece * 4 + (lane >> 4), c16 = lane & 15;
        const int ch = ((((c16 >> 1) ^ strided_key(kr))) << 1) | (c16 & 1);
        srcb = B + (size_t)(k0 + kr) * a.ldb + min(bn0 + ch * 8, a.N - 8);
      }
      __builtin_amdgcn_global_load_lds((gptr_t*)srcb, (lptr_t*)(sb + piece * 1024), 16, 0, 0);
    }
  };

  // staged bf16 epilogue (see below): its residual chunks are fetched NOW so that they arrive under the K loop
  bool vec_ok = false;
  uint4 rres[4][2];
  float4 bias_lo = make_float4(0.f, 0.f, 0.f, 0.f), bias_hi = bias_lo;      // columns bn0 + wn*64 + (lane & 7)*8 .. +7: the same for every row
  if constexpr (sizeof(TC) == 2) {
    vec_ok = !a.slab && (a.N % 8 == 0) && (a.ldc % 8 == 0) && (((uintptr_t)a.C & 15) == 0) &&
             (!(a.epi & MTS_EPI_RESIDUAL) || ((a.ldr % 8 == 0) && (((uintptr_t)a.residual & 15) == 0))) &&
             (!a.aux || ((a.ldaux % 8 == 0) && (((uintptr_t)a.aux & 15) == 0)));
    vec_ok = vec_ok && (!(a.epi & MTS_EPI_BIAS) || (((uintptr_t)a.bias & 15) == 0));
    if (vec_ok && (a.epi & MTS_EPI_BIAS) && blockIdx.z == 0) {
      const int n = min(bn0 + wn * 64 + (lane & 7) * 8, a.N - 8);
      bias_lo = *reinterpret_cast<const float4*>(a.bias + n);
      bias_hi = *reinterpret_cast<const float4*>(a.bias + n + 4);
    }
    if (vec_ok && (a.epi & MTS_EPI_RESIDUAL) && blockIdx.z == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int idx = h * 64 + lane, row = idx >> 3, ch = idx & 7;
          const int m = min(bm0 + wm * 64 + i * 16 + row, a.M - 1), n = min(bn0 + wn * 64 + ch * 8, a.N - 8);
          rres[i][h] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(a.residual) + (size_t)m * a.ldr + n);
        }
    }
  }

  // LDS-DMA form: NB K-tile buffers, copies run NB-1 tiles ahead; a wave issues 8 DMA instructions per K-tile and the vector-
  // memory counter retires in order, so "tile kt+1 has landed" = at most 8 x (tiles issued after it) instructions outstanding
  const int NB = GLDS ? a.nbuf : 2;
  auto wait_tiles_outstanding = [&](int n) {
    if (n >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (n == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  if (nk > 0) {
    if constexpr (GLDS) {
      const int pre = min(NB - 1, nk);
      for (int t = 0; t < pre; ++t) dma_tile(t, t);
      wait_tiles_outstanding(pre - 1);
    } else {
      load_tile(0);
      store_tile(0);
    }
  }
  if constexpr (GLDS) __builtin_amdgcn_s_barrier();      // (not __syncthreads(): its fence would wait for every copy in flight)
  else __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int buf = GLDS ? (kt & (NB - 1)) : (kt & 1);
    if constexpr (GLDS) {
      // lands in the buffer read one iteration ago (every wave is past that iteration's barrier) while this one feeds the MFMAs
      if (kt + NB - 1 < nk) dma_tile(kt + NB - 1, (kt + NB - 1) & (NB - 1));
    } else {
      if (kt + 1 < nk) load_tile(kt + 1);              // global loads stay in flight behind the MFMAs below
    }
    const char* sa = smem + buf * (2 * TILE_BYTES);
    const char* sb = sa + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
      frag_raw ra4[4], rb4[4];                           // transposed reads in flight (strided operands only)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (A_KMAJOR) af[i] = frag_kmajor(sa, wm * 64 + i * 16 + r16, ks * 4 + g);
        else ra4[i] = frag_strided(sa, ks * 32 + 8 * g, wm * 64 + i * 16, lane);
        if constexpr (B_KMAJOR) bfr[i] = frag_kmajor(sb, wn * 64 + i * 16 + r16, ks * 4 + g);
        else rb4[i] = frag_strided(sb, ks * 32 + 8 * g, wn * 64 + i * 16, lane);
      }
      if constexpr (!A_KMAJOR || !B_KMAJOR) {            // transposed reads are asm: wait for them by hand (gemm_common.h)
        lds_frags_wait();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if constexpr (!A_KMAJOR) af[i] = frag_finish(ra4[i]);
          if constexpr (!B_KMAJOR) bfr[i] = frag_finish(rb4[i]);
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);  // C^T tile
    }
    if constexpr (GLDS) {
      wait_tiles_outstanding(max(0, min(nk - 1, kt + NB - 1) - (kt + 1)));   // this wave's pieces of tile kt+1 have landed
      // raw barrier: __syncthreads() carries a fence that the compiler turns into "s_waitcnt vmcnt(0)", i.e. a wait for the
      // copies of the tiles further ahead as well
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    } else {
      if (kt + 1 < nk) store_tile(buf ^ 1);
      __syncthreads();
    }
  }

  const bool first_slice = (blockIdx.z == 0);
  if constexpr (sizeof(TC) == 2) {
    // bf16 output: the accumulator layout gives every lane 4 columns of one row (8-byte pieces of 16 different rows per
    // instruction).  Stage the wave's 64x64 fp32 sub-tile through LDS 16 rows at a time (the K loop's buffers are free
    // after its last barrier) and run the epilogue on 8-column chunks of contiguous rows instead: bias / residual / aux /
    // C move as 16-byte pieces of 128-byte row segments, still one rounding from the fp32 accumulator.
    if (vec_ok) {
      // every vector-memory load of this kernel (operand copies, the residual / bias chunks fetched before the K loop) is complete
      // from here on -- said with the BUILTIN so that the compiler's wait insertion knows it too: behind an inline-asm wait it
      // still puts "s_waitcnt vmcnt(0)" in front of the first use of a prefetched register on every control-flow path of the
      // loop below, i.e. between the stores, where it waits for the stores (vmcnt = 0, expcnt / lgkmcnt unconstrained)
      __builtin_amdgcn_s_waitcnt(0x0F70);
      constexpr int RS = 64 * 4 + 16;                       // fp32 row of 64 columns + 16 B padding
      char* stage = smem + wave * (16 * RS);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<float4*>(stage + r16 * RS + (j * 16 + 4 * g) * 4) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int idx = h * 64 + lane, row = idx >> 3, ch = idx & 7;
          const float4 lo = *reinterpret_cast<const float4*>(stage + row * RS + ch * 32);
          const float4 hi = *reinterpret_cast<const float4*>(stage + row * RS + ch * 32 + 16);
          float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
          const int m = bm0 + wm * 64 + i * 16 + row, n = bn0 + wn * 64 + ch * 8;
          if (m < a.M && n < a.N) {
            epi_math8(a, m, n, v, first_slice, rres[i][h], bias_lo, bias_hi);
            uint4 pk;
            pk.x = pack_bf16x2(v[0], v[1]); pk.y = pack_bf16x2(v[2], v[3]); pk.z = pack_bf16x2(v[4], v[5]); pk.w = pack_bf16x2(v[6], v[7]);
            *reinterpret_cast<uint4*>(reinterpret_cast<bf16_t*>(a.C) + (size_t)m * a.ldc + n) = pk;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      return;
    }
  }
  if constexpr (sizeof(TC) == 4) {
    if (a.chain) {
      // Split-K without slabs: the K-slices of a tile add their partial sums into C one after the other, slice 0 first.
      // Workgroups are dispatched x-fastest, z-slowest, so slice z-1 of a tile was dispatched before slice z and always makes
      // progress; by the time slice z has finished its own K range the wait is normally already over.  Hand-off = the
      // agent-scope release/acquire recipe of the CDNA guide (plain C stores, one release fence + counter store by one lane;
      // consumer polls relaxed, fences once, then plain loads).
      const unsigned z = blockIdx.z;
      unsigned* turn = a.chain + bid;
      if (z > 0) {
        if (tid == 0) {
          while (__hip_atomic_load(turn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != z) __builtin_amdgcn_s_sleep(16);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
      }
      GemmArgs a3 = a;
      a3.slab = nullptr;
      if (z > 0) a3.epi |= MTS_EPI_ACCUM;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = bm0 + wm * 64 + i * 16 + r16;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = bn0 + wn * 64 + j * 16 + 4 * g;
          float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
          epilogue4<bf16_t, TC>(a3, m, n, v, first_slice);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(turn, z + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      return;
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = bm0 + wm * 64 + i * 16 + r16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = bn0 + wn * 64 + j * 16 + 4 * g;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      epilogue4<bf16_t, TC>(a, m, n, v, first_slice);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// fp32 VALU kernel, arbitrary strides: C[m,n] = sum_k A[m*sam + k*sak] * B[n*sbn + k*sbk]
// 64x64 tile, BK = 16, 256 threads, 4x4 outputs per thread (columns n0 + tx*4 .. +3, rows m0 + ty + 16*i)
// ------------------------------------------------------------------------------------------------
struct StrideArgs { long sam, sak, sbn, sbk; };

__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmArgs a, const StrideArgs s) {
  __shared__ float As[16][64 + 4];
  __shared__ float Bs[16][64 + 4];
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int ntn = (a.N + 63) / 64;
  const int bm0 = (blockIdx.x / ntn) * 64, bn0 = (blockIdx.x % ntn) * 64;
  const float* __restrict__ A = reinterpret_cast<const float*>(a.A);
  const float* __restrict__ B = reinterpret_cast<const float*>(a.B);
  float acc[4][4] = {};
  for (int k0 = 0; k0 < a.K; k0 += 16) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + 256 * i;
      int m, k;
      if (s.sak == 1) { m = e >> 4; k = e & 15; } else { m = e & 63; k = e >> 6; }
      As[k][m] = (bm0 + m < a.M && k0 + k < a.K) ? A[(long)(bm0 + m) * s.sam + (long)(k0 + k) * s.sak] : 0.f;
      int n, kb;
      if (s.sbk == 1) { n = e >> 4; kb = e & 15; } else { n = e & 63; kb = e >> 6; }
      Bs[kb][n] = (bn0 + n < a.N && k0 + kb < a.K) ? B[(long)(bn0 + n) * s.sbn + (long)(k0 + kb) * s.sbk] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      float av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) av[i] = As[k][ty + 16 * i];
      const float4 b4 = *reinterpret_cast<const float4*>(&Bs[k][tx * 4]);
      bv[0] = b4.x; bv[1] = b4.y; bv[2] = b4.z; bv[3] = b4.w;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float v[4] = {acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
    epilogue4<float, float>(a, bm0 + ty + 16 * i, bn0 + tx * 4, v, true);
  }
}

// ------------------------------------------------------------------------------------------------
// fp32 on the matrix cores: v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate; exact fp32 -- per output element the same
// k-ordered fma chain as the VALU kernel above, so parity mode keeps its bars; 64 FLOP/clk/SIMD = the fp32 vector rate, but one
// instruction per 16x16x4 block instead of 1024 v_fma lane-operations).  128x128x16 tile, 4 waves x (64x64 = 4x4 MFMA tiles),
// operands staged through registers (arbitrary strides, 16-byte loads along whichever of m / k is contiguous) into k-major LDS
// rows of 128 + 16 floats (the 4 k-rows a wave-instruction reads then sit 16 banks apart: conflict-free ds_read_b32).
// The accumulator is computed transposed (B fragment as the first operand) so that a lane owns 4 consecutive columns of one row.
// ------------------------------------------------------------------------------------------------
#define F32_BM 128
#define F32_BK 16
#define F32_PITCH 144

__global__ __launch_bounds__(256) void gemm_f32_mfma_kernel(const GemmArgs a, const StrideArgs s) {
  __shared__ float As[2][F32_BK][F32_PITCH];
  __shared__ float Bs[2][F32_BK][F32_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r16 = lane & 15, g = lane >> 4;
  const int ntn = (a.N + F32_BM - 1) / F32_BM;
  const int bm0 = (blockIdx.x / ntn) * F32_BM, bn0 = (blockIdx.x % ntn) * F32_BM;
  const float* __restrict__ A = reinterpret_cast<const float*>(a.A);
  const float* __restrict__ B = reinterpret_cast<const float*>(a.B);
  // operand X (rows r of the tile, k): element (r, k) at X[(r0 + r) * sr + (k0 + k) * sk]; one of sr / sk is 1
  const bool a_kc = (s.sak == 1), b_kc = (s.sbk == 1);
  // split-K (weight gradients: few output tiles, K = all tokens): slice blockIdx.z covers [kbeg, kend) and writes its partial
  // tile to a.slab; splitk_reduce_kernel adds the slices in a fixed order
  const int kbeg = blockIdx.z * a.ksplit, kend = min(a.K, kbeg + a.ksplit);
  const bool a_vec = (((uintptr_t)A & 15) == 0) && ((a_kc ? s.sam : s.sak) % 4 == 0);
  const bool b_vec = (((uintptr_t)B & 15) == 0) && ((b_kc ? s.sbn : s.sbk) % 4 == 0);
  float4 ra[2], rb[2];

  // 128 x 16 floats per operand and K-tile = 512 float4: two per thread.  k-contiguous: float4 = 4 consecutive k of one row
  // (f = tid + 256 i: row f >> 2, k group f & 3); row-contiguous: 4 consecutive rows of one k (k = f >> 5, row group f & 31).
  auto load_op = [&](const float* __restrict__ X, long sr, long sk, bool kc, bool vec, int r0, int rdim, int k0, float4 (&r)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int f = tid + 256 * i;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      if (kc) {
        const int row = r0 + (f >> 2), k = k0 + (f & 3) * 4;
        if (row < rdim) {
          const float* p = X + (long)row * sr + k;
          if (vec && k + 3 < kend) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
          else for (int j = 0; j < 4; ++j) if (k + j < kend) v[j] = p[j];
        }
      } else {
        const int k = k0 + (f >> 5), row = r0 + (f & 31) * 4;
        if (k < kend) {
          const float* p = X + (long)k * sk + row;
          if (vec && row + 3 < rdim) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
          else for (int j = 0; j < 4; ++j) if (row + j < rdim) v[j] = p[j];
        }
      }
      r[i] = make_float4(v[0], v[1], v[2], v[3]);
    }
  };
  auto store_op = [&](float (&S)[F32_BK][F32_PITCH], bool kc, const float4 (&r)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int f = tid + 256 * i;
      if (kc) {
        const int row = f >> 2, k = (f & 3) * 4;
        S[k][row] = r[i].x; S[k + 1][row] = r[i].y; S[k + 2][row] = r[i].z; S[k + 3][row] = r[i].w;
      } else {
        *reinterpret_cast<float4*>(&S[f >> 5][(f & 31) * 4]) = r[i];
      }
    }
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = (kend - kbeg + F32_BK - 1) / F32_BK;
  load_op(A, s.sam, s.sak, a_kc, a_vec, bm0, a.M, kbeg, ra);
  load_op(B, s.sbn, s.sbk, b_kc, b_vec, bn0, a.N, kbeg, rb);
  store_op(As[0], a_kc, ra);
  store_op(Bs[0], b_kc, rb);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) {                                  // the next K-tile's global loads fly under this tile's MFMAs
      load_op(A, s.sam, s.sak, a_kc, a_vec, bm0, a.M, kbeg + (kt + 1) * F32_BK, ra);
      load_op(B, s.sbn, s.sbk, b_kc, b_vec, bn0, a.N, kbeg + (kt + 1) * F32_BK, rb);
    }
#pragma unroll
    for (int ks = 0; ks < F32_BK / 4; ++ks) {
      float av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        av[i] = As[buf][ks * 4 + g][wm * 64 + i * 16 + r16];
        bv[i] = Bs[buf][ks * 4 + g][wn * 64 + i * 16 + r16];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[j], av[i], acc[i][j], 0, 0, 0);   // C^T tile
    }
    if (kt + 1 < nk) {
      store_op(As[buf ^ 1], a_kc, ra);
      store_op(Bs[buf ^ 1], b_kc, rb);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = bm0 + wm * 64 + i * 16 + r16;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = bn0 + wn * 64 + j * 16 + 4 * g;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      epilogue4<float, float>(a, m, n, v, true);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// column sums (bias gradients): two deterministic stages
// ------------------------------------------------------------------------------------------------
#define COLSUM_RS 128
// workgroup = 64 columns (16 lanes x 4) x 16 row lanes over one slice of rows; 4 independent loads in flight per lane,
// the 16 row lanes are combined through LDS in a fixed order
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ X, int M, int N, int ldx,
                                                             float* __restrict__ partial) {
  __shared__ float red[16][64];
  const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int n = blockIdx.x * 64 + cx * 4;
  const int rs = blockIdx.y;
  const int rows_per = (M + COLSUM_RS - 1) / COLSUM_RS;
  const int m0 = rs * rows_per, m1 = min(M, m0 + rows_per);
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (n + 3 < N) {
    int m = m0 + ry;
    for (; m + 48 < m1; m += 64) {
      float v[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) load4<T>(X + (size_t)(m + 16 * u) * ldx + n, v[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) s[i] += v[u][i];
    }
    for (; m < m1; m += 16) {
      float v[4];
      load4<T>(X + (size_t)m * ldx + n, v);
#pragma unroll
      for (int i = 0; i < 4; ++i) s[i] += v[i];
    }
  } else if (n < N) {
    for (int m = m0 + ry; m < m1; m += 16)
      for (int i = 0; i < 4 && n + i < N; ++i) s[i] += to_f32(X[(size_t)m * ldx + n + i]);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) red[ry][cx * 4 + i] = s[i];
  __syncthreads();
  if (threadIdx.x < 64) {
    const int c = threadIdx.x, nn = blockIdx.x * 64 + c;
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) t += red[r][c];
    if (nn < N) partial[(size_t)rs * N + nn] = t;
  }
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, int N, int nparts,
                                                           float* __restrict__ out, int accumulate) {
  // 64 columns x 4 groups of partial rows per workgroup; fixed-order LDS combine
  __shared__ float red[4][64];
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + col;
  float s = 0.f;
  if (n < N) {
    // eight loads in flight, the adds in their old order (one dependent L2 round trip per partial made this launch 9 us for 128 x 256 floats)
    int r = grp;
    for (; r + 4 * 7 < nparts; r += 4 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(r + 4 * u) * N + n];
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; r < nparts; r += 4) s += partial[(size_t)r * N + n];
  }
  red[grp][col] = s;
  __syncthreads();
  if (grp == 0 && n < N) {
    const float t = (red[0][col] + red[1][col]) + (red[2][col] + red[3][col]);
    out[n] = accumulate ? out[n] + t : t;
  }
}

// C[m,n] (+)= sum_z slab[z][m][n]  -- fixed summation order, so split-K weight gradients are bitwise reproducible
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slab, int splits, int M, int N, float* __restrict__ C,
                                                            int ldc, int accumulate) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int n4 = N / 4;
  if (idx >= (size_t)M * n4) return;
  const int m = (int)(idx / n4), n = 4 * (int)(idx % n4);
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (accumulate) load4<float>(C + (size_t)m * ldc + n, s);
  const size_t plane = (size_t)M * N;
  for (int z = 0; z < splits; ++z) {
    float v[4];
    load4<float>(slab + z * plane + (size_t)m * N + n, v);
#pragma unroll
    for (int i = 0; i < 4; ++i) s[i] += v[i];
  }
  store4<float>(C + (size_t)m * ldc + n, s);
}

// Ct[n][m] (+)= sum_z slab[z][m][n]: the same fixed-order sum, stored TRANSPOSED (32 x 32 tiles through LDS) -- the second problem of
// mts_wgrad_pair is computed as the transpose of the gradient it belongs to
__global__ __launch_bounds__(256) void splitk_reduce_t_kernel(const float* __restrict__ slab, int splits, int M, int N, float* __restrict__ Ct,
                                                              int ldc, int accumulate) {
  __shared__ float tile[32][33];
  const int n0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const size_t plane = (size_t)M * N;
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int m = m0 + r, n = n0 + tx;
    float s = 0.f;
    if (m < M && n < N) {
      const float* p = slab + (size_t)m * N + n;
      int z = 0;
      for (; z + 7 < splits; z += 8) {                  // eight loads in flight, the adds in slice order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(z + u) * plane];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
      }
      for (; z < splits; ++z) s += p[(size_t)z * plane];
    }
    tile[r][tx] = s;
  }
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int n = n0 + r, m = m0 + tx;
    if (n < N && m < M) {
      float* o = Ct + (size_t)n * ldc + m;
      *o = accumulate ? *o + tile[tx][r] : tile[tx][r];
    }
  }
}

template <typename T> __global__ void cast_kernel(const float* __restrict__ src, T* __restrict__ dst, size_t n) {
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
  for (; i + 3 < n; i += stride) {
    float v[4];
    load4<float>(src + i, v);
    store4<T>(dst + i, v);
  }
  // tail (n % 4 elements) handled by the thread that lands on it
  if (i < n && i + 3 >= n)
    for (size_t j = i; j < n; ++j) dst[j] = from_f32<T>(src[j]);
}

// dst[r, :] = bf16/f32( src1[r, 0:D1] | src2[r, 0:D2] ): the early-fusion concat and the fp32 -> act-dtype cast in one pass over two
// separate fp32 matrices (the recurrent taggers' K-split input; D1, D2 multiples of 4)
template <typename T> __global__ void cast_concat_kernel(const float* __restrict__ s1, const float* __restrict__ s2, T* __restrict__ dst,
                                                          size_t rows, int D1, int D2) {
  const int D = D1 + D2, q = D / 4;
  const size_t total = rows * q;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = i / q;
    const int e = (int)(i - r * q) * 4;
    float v[4];
    load4<float>(e < D1 ? s1 + r * D1 + e : s2 + r * D2 + (e - D1), v);
    store4<T>(dst + r * D + e, v);
  }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
extern "C" int mts_cast_concat(void* stream, int dst_dtype, size_t rows, int D1, int D2, const float* src1, const float* src2, void* dst) {
  MTS_CHECK_ARG(src1 && src2 && dst && D1 > 0 && D2 > 0 && D1 % 4 == 0 && D2 % 4 == 0, "mts_cast_concat: bad arguments (widths must be multiples of 4)");
  if (rows == 0) return MTS_OK;
  const int blocks = (int)std::min<size_t>(4096, (rows * ((D1 + D2) / 4) + 255) / 256);
  if (dst_dtype == MTS_BF16) hipLaunchKernelGGL(cast_concat_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src1, src2, (bf16_t*)dst, rows, D1, D2);
  else if (dst_dtype == MTS_F32) hipLaunchKernelGGL(cast_concat_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src1, src2, (float*)dst, rows, D1, D2);
  else { mts_set_error("mts_cast_concat: bad dtype %d", dst_dtype); return MTS_ERR_INVALID; }
  MTS_LAUNCH_CHECK("mts_cast_concat");
  return MTS_OK;
}

extern "C" size_t mts_colsum_workspace(int N) { return (size_t)COLSUM_RS * (size_t)N * sizeof(float); }

extern "C" int mts_colsum(void* stream, int dtype, int M, int N, const void* X, int ldx, float* out, int accumulate,
                          void* partial) {
  MTS_CHECK_ARG(M > 0 && N > 0 && X && out && partial, "mts_colsum: bad arguments");
  MTS_CHECK_ARG(dtype == MTS_F32 || dtype == MTS_BF16, "mts_colsum: bad dtype %d", dtype);
  MTS_CHECK_ARG(ldx % 4 == 0, "mts_colsum: ldx must be a multiple of 4");
  hipStream_t st = (hipStream_t)stream;
  const int nparts = min(COLSUM_RS, M);
  dim3 grid(ceil_div(N, 64), COLSUM_RS);
  // rows_per = ceil(M / RS): slices past M write zeros (m0 >= m1), so all RS partials are defined
  (void)nparts;
  if (dtype == MTS_F32)
    hipLaunchKernelGGL(colsum_partial_kernel<float>, grid, dim3(256), 0, st, (const float*)X, M, N, ldx, (float*)partial);
  else
    hipLaunchKernelGGL(colsum_partial_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)X, M, N, ldx, (float*)partial);
  hipLaunchKernelGGL(colsum_final_kernel, dim3(ceil_div(N, 64)), dim3(256), 0, st, (const float*)partial, N, COLSUM_RS, out,
                     accumulate);
  MTS_LAUNCH_CHECK("mts_colsum");
  return MTS_OK;
}

extern "C" int mts_cast(void* stream, int dst_dtype, const float* src, void* dst, size_t n) {
  MTS_CHECK_ARG(src && dst, "mts_cast: null pointer");
  if (n == 0) return MTS_OK;
  hipStream_t st = (hipStream_t)stream;
  const int blocks = (int)std::min<size_t>(2048, (n / 4 + 255) / 256 + 1);
  if (dst_dtype == MTS_BF16)
    hipLaunchKernelGGL(cast_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, src, (bf16_t*)dst, n);
  else if (dst_dtype == MTS_F32)
    hipLaunchKernelGGL(cast_kernel<float>, dim3(blocks), dim3(256), 0, st, src, (float*)dst, n);
  else {
    mts_set_error("mts_cast: bad dtype %d", dst_dtype);
    return MTS_ERR_INVALID;
  }
  MTS_LAUNCH_CHECK("mts_cast");
  return MTS_OK;
}

int mts_launch_gemm256(const GemmArgs& a, int layout, bool c_is_f32, int splits, hipStream_t st);   // gemm256.hip
int mts_launch_gemm224(const GemmArgs& a, int layout, bool c_is_f32, int splits, hipStream_t st);   // gemm224.hip
bool mts_gemm224t_applies(const GemmArgs& a, int layout, bool c_is_f32, int splits);                 // gemm224t.hip

// Tuning / A-B switches and the last-plan record are PER HOST THREAD (include/mts.h "Threads"): mts_set_option from one thread never
// changes the plan of a GEMM another thread is issuing, and mts_gemm_last_plan reports the calling thread's own most recent call.
static thread_local int g_gemm_glds = -1;
static thread_local int g_tile_mode = -1;
static thread_local int g_force_splits = 0;
static thread_local int g_tile_order = 1;
static thread_local int g_gemm_deep = 1;      // "gemm_deep" = 0: never use the four-buffer copy pipeline of the 128x128 kernel (A/B testing)
static thread_local int g_chain = 0;          // "gemm_chain" = 1: split-K of the 128x128 kernel accumulates in place (no slabs, no reduce launch); measured
                                 // SLOWER (532 vs 465 us on the QKV weight gradient: one agent-scope release per workgroup writes the L2 back)
static thread_local int g_combine = 0;        // "gemm_combine" = 1: the slice of a tile that arrives last adds the split-K planes of the 256x224 weight-gradient kernel inside
                                              // the launch (no reduce launch).  Bitwise the same and measured SLOWER (profiles/r04_tn_sweep.txt: 346 vs 276 us on the q|k|v
                                              // gradient at 3 slices, 164 vs 98 us on the attention-output gradient at 4): 229 KB of fp32 per plane and tile is an order of
                                              // magnitude past what the guide's recipe pays for, and the agent-scope release of 504 workgroups writes the XCDs' L2s back
#define MTS_GEMM_WS_HEAD 8192                 // bytes at the start of mts_gemm's workspace reserved for the arrival tickets of the in-launch combine
static thread_local int g_last_tile = 0, g_last_splits = 0;   // what the planner chose for the most recent bf16 mts_gemm (bench.py labels)

// Measurement aid: a host callback run between the GEMM launch and the split-K reduce launch of a call (bench.py records the END event of its
// HIP-event bracket there, so that the bracket times the GEMM kernel alone -- the figure a rocprofv3 kernel trace reports for that symbol).
static thread_local void (*g_mid_hook)(void) = nullptr;
extern "C" int mts_gemm_set_mid_hook(void (*fn)(void)) { g_mid_hook = fn; return MTS_OK; }

extern "C" int mts_gemm_last_plan(int* tile, int* splits) {
  if (tile) *tile = g_last_tile;
  if (splits) *splits = g_last_splits;
  return MTS_OK;
}

void mts_band_set_mfma(int on);   // band_attn.hip
void mts_band_set_fused(int on);  // band_attn_mfma.hip
static thread_local int g_f32_mfma = 1;                   // fp32 (parity mode) GEMM on v_mfma_f32_16x16x4_f32; 0 = VALU kernel
static thread_local int g_big_min_k = 256;                // smallest K the big-tile kernels are considered for ("gemm_big_min_k")
static thread_local int g_gemm_variant = 0;               // A/B switch of the big-tile kernels (GemmArgs::variant)
void mts_lstm_pair_set_spin_limit(int n);   // lstm_pair.hip
void mts_lstm_pair_set_max_pairs(int n);
void mts_lstm_pair_set_parts(int n);
extern "C" int mts_set_option(const char* key, int value) {
  if (!key) return MTS_ERR_INVALID;
  if (!strcmp(key, "gemm_tile")) { g_tile_mode = value; return MTS_OK; }
  if (!strcmp(key, "gemm_glds")) { g_gemm_glds = value; return MTS_OK; }
  if (!strcmp(key, "gemm_splits")) { g_force_splits = value; return MTS_OK; }
  if (!strcmp(key, "gemm_order")) { g_tile_order = value; return MTS_OK; }
  if (!strcmp(key, "gemm_chain")) { g_chain = value; return MTS_OK; }
  if (!strcmp(key, "gemm_combine")) { g_combine = value; return MTS_OK; }
  if (!strcmp(key, "gemm_deep")) { g_gemm_deep = value; return MTS_OK; }
  if (!strcmp(key, "gemm_variant")) { g_gemm_variant = value; return MTS_OK; }
  if (!strcmp(key, "gemm_big_min_k")) { g_big_min_k = value; return MTS_OK; }
  if (!strcmp(key, "gemm_f32_mfma")) { g_f32_mfma = value; return MTS_OK; }
  if (!strcmp(key, "band_mfma")) { mts_band_set_mfma(value); return MTS_OK; }
  if (!strcmp(key, "band_fused_bwd")) { mts_band_set_fused(value); return MTS_OK; }
  if (!strcmp(key, "lstm_pair_spin_limit")) { mts_lstm_pair_set_spin_limit(value); return MTS_OK; }
  if (!strcmp(key, "lstm_pair_max_pairs")) { mts_lstm_pair_set_max_pairs(value); return MTS_OK; }
  if (!strcmp(key, "lstm_parts")) { mts_lstm_pair_set_parts(value); return MTS_OK; }
  mts_set_error("mts_set_option: unknown key %s", key);
  return MTS_ERR_INVALID;
}
   // MTS_GEMM_GLDS=0 forces the register-staged form (A/B testing)

template <int LAYOUT, typename TC>
static void launch_bf16(const GemmArgs& a, int splits, hipStream_t st) {
  const int nt = ceil_div(a.M, BM) * ceil_div(a.N, BN);
  if (g_gemm_glds < 0) {
    const char* e = getenv("MTS_GEMM_GLDS");
    g_gemm_glds = (e && e[0] == '0') ? 0 : 1;
  }
  const bool glds = g_gemm_glds && (a.K % BK == 0) && (a.ksplit % BK == 0) && a.M >= 8 && a.N >= 8;
  if (glds) {
    // at most one workgroup per CU: nothing but a deeper copy pipeline hides the DMA latency of a K-tile (the feed-forward
    // projections, N or K = 256: 256 tiles, 28 K-tiles each) -- four buffers, copies three tiles ahead
    GemmArgs b = a;
    const bool deep = g_gemm_deep && (size_t)nt * splits <= 256 && a.ksplit >= 4 * BK;
    b.nbuf = deep ? 4 : 2;
    auto k = gemm_bf16_kernel<LAYOUT, TC, true>;
    static std::atomic<bool> attr_set{false};      // idempotent, process-wide: two threads racing here set the same attribute twice
    if (deep && !attr_set.load(std::memory_order_acquire)) {
      if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * TILE_BYTES) != hipSuccess) b.nbuf = 2;
      else attr_set.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(k, dim3(nt, 1, splits), dim3(256), (b.nbuf == 4 ? 8 : 4) * TILE_BYTES, st, b);
  } else {
    hipLaunchKernelGGL((gemm_bf16_kernel<LAYOUT, TC, false>), dim3(nt, 1, splits), dim3(256), 4 * TILE_BYTES, st, a);
  }
}

// Planner of the bf16 path (pure host code, no device call): which kernel and which K split mts_gemm uses for a shape under the
// CALLING THREAD's options.  use256: 0 = 128x128 kernel, 1 = 256x256, 2 = 256x224 (N a multiple of 224: d = 1792 projections).
static void plan_bf16(int c_dtype, int layout, int M, int N, int K, unsigned epilogue, bool has_workspace, size_t workspace_bytes,
                      int* use256_out, int* splits_out) {
  // Plan: tile size (128x128, 2 workgroups per CU  vs  256x256, 1 per CU, deeper DMA pipeline) and, for weight-gradient
  // shapes (fp32 C, no epilogue, K = all tokens), the K split.  Cost model = rounds of workgroups x K per slice x
  // measured time per K element + the slab round trip; partial tiles go to `workspace` with plain stores and a second
  // kernel sums them in a fixed order (no float atomics: deterministic, and several times the atomic byte rate).
  if (g_tile_mode < 0) {       // MTS_GEMM_TILE=128|256 forces one kernel (A/B testing); default: cost model
    const char* e = getenv("MTS_GEMM_TILE");
    g_tile_mode = e ? atoi(e) : 0;
  }
  const int tile_mode = g_tile_mode;
  const unsigned plain = epilogue & ~MTS_EPI_ACCUM;
  const bool can_split = (c_dtype == MTS_F32 && plain == 0 && has_workspace && N % 4 == 0 && K >= 2048);
  // (until the transposed LDS reads became inline asm -- gemm_common.h -- the big tiles' TN form was slower than the 128
  // kernel: one workgroup per CU had nothing to hide the exposed DMA wait behind; now it is the fastest weight-gradient form)
  const bool can256 = (K % BK == 0) && K >= g_big_min_k && M >= 8 && N >= 8 && tile_mode != 128;
  const double bw = 3500.0;     // slab MB per us
  double best = 1e30;
  int splits = 1;
  int use256 = 0;               // 0: 128x128 kernel, 1: 256x256, 2: 256x224 (N a multiple of 224: d = 1792 projections)
  const bool can224 = can256 && (N % 224 == 0) && tile_mode != 256;
  for (int big = 0; big <= (can256 ? 2 : 0); ++big) {
    if (big == 2 && !can224) continue;
    if (tile_mode == 256 && can256 && big != 1) continue;
    if (tile_mode == 224 && can224 && big != 2) continue;
    const int tile = big ? 256 : 128;
    // measured on MI355X (tools/gemm_ksweep.py, gemm_sweep.py): us per K element per round of workgroups, and per-round fixed cost
    const double slots = big ? 256.0 : 512.0;
    const double t_k = layout == MTS_TN ? (big == 2 ? (M % 256 == 0 ? 0.0160 : 0.0200) : big == 1 ? 0.0218 : 0.0145)   // tools/tn_sweep.py (the four-wave kernel of
                                                                                                                       // gemm224t.hip takes M % 256 == 0), gemm_ab.py
                                        : (big ? 0.0232 * (big == 2 ? 0.875 : 1.0) : 0.0180);
    const double t_0 = big ? 7.7 : 6.5;
    const int nt = ceil_div(M, tile) * (big == 2 ? N / 224 : ceil_div(N, tile));
    for (int sp = 1; sp <= (can_split ? 32 : 1); ++sp) {
      const int ks = ceil_div(ceil_div(K, sp), BK) * BK;
      if (ceil_div(K, ks) != sp) continue;
      if (sp > 1 && (size_t)sp * M * N * sizeof(float) > workspace_bytes) break;
      if (g_force_splits > 0 && can_split && sp != g_force_splits) continue;
      const double rounds = ceil((double)nt * sp / slots);
      const double cost = rounds * (ks * t_k + t_0) + (sp > 1 ? (2.0 * sp + 1.0) * M * N * 4.0 / 1e6 / bw : 0.0);
      if (cost < best) { best = cost; splits = sp; use256 = big; }
    }
  }
  *use256_out = use256;
  *splits_out = splits;
}

// ---- two weight gradients of one shape in one launch ------------------------------------------------------------------------------------
int mts_gemm224_last_kernel();   // gemm224.hip
int mts_launch_gemm224t_pair(const GemmArgs& a, int splits, const void* A2, const void* B2, float* slab2, hipStream_t st);   // gemm224t.hip
bool mts_gemm224t_applies(const GemmArgs& a, int layout, bool c_is_f32, int splits);

static int wgrad_pair_splits(int M, int N, int K, size_t workspace_bytes) {
  // one round of workgroups: 2 problems x tiles x slices ~ the CUs (slices of at least 128 k, multiples of 64; 32 at most); -1: not covered
  if (M % 256 || N % 224 || K % 64 || K < 256) return -1;
  static int cus = 0;
  if (!cus) { int dev = 0; hipDeviceProp_t pr; cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 256; }
  const int nt = (M / 256) * (N / 224);
  int sp = std::max(1, std::min(32, cus / (2 * nt)));
  for (; sp >= 1; --sp) {
    const int ks = ceil_div(ceil_div(K, sp), BK) * BK;
    if (ceil_div(K, ks) != sp) continue;
    if (ks >= 128 && K - (sp - 1) * ks >= 128 && (size_t)2 * sp * M * N * sizeof(float) <= workspace_bytes) return sp;
  }
  return -1;
}
extern "C" size_t mts_wgrad_pair_workspace(int M, int N, int K) {
  const int sp = wgrad_pair_splits(M, N, K, (size_t)-1);
  return sp < 1 ? 0 : (size_t)2 * sp * M * N * sizeof(float);
}
extern "C" int mts_wgrad_pair(void* stream, int M, int N, int K, const void* A1, const void* B1, float* C1, int ldc1, const void* A2, const void* B2,
                              float* C2t, int ldc2t, int lda, int ldb, int accumulate, void* workspace, size_t workspace_bytes) {
  MTS_CHECK_ARG(M > 0 && N > 0 && K > 0 && A1 && B1 && C1 && A2 && B2 && C2t && workspace, "mts_wgrad_pair: bad arguments");
  MTS_CHECK_ARG(ldc1 >= N && ldc2t >= M && ldc1 % 4 == 0, "mts_wgrad_pair: bad output strides");
  const int sp = wgrad_pair_splits(M, N, K, workspace_bytes);
  MTS_UNSUPPORTED(sp >= 1, "mts_wgrad_pair: M=%d N=%d K=%d not covered (M %% 256, N %% 224, K %% 64) or workspace too small", M, N, K);
  hipStream_t st = (hipStream_t)stream;
  GemmArgs a;
  memset(&a, 0, sizeof(a));
  a.A = A1; a.B = B1; a.C = C1;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc1;
  a.epi = 0; a.colscale = 1.0f;
  a.ksplit = ceil_div(ceil_div(K, sp), BK) * BK;
  a.order = 1; a.nbuf = 2; a.variant = 0;
  float* const planes = (float*)workspace;
  a.slab = planes;
  float* const planes2 = planes + (size_t)sp * M * N;
  MTS_UNSUPPORTED(mts_gemm224t_applies(a, MTS_TN, true, sp), "mts_wgrad_pair: operands not covered by the four-wave weight-gradient kernel (alignment / strides)");
  const int rc = mts_launch_gemm224t_pair(a, sp, A2, B2, planes2, st);
  MTS_UNSUPPORTED(rc == MTS_OK, "mts_wgrad_pair: launch refused");
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(((size_t)M * (N / 4) + 255) / 256)), dim3(256), 0, st, (const float*)planes, sp, M, N, C1, ldc1,
                     accumulate ? 1 : 0);
  hipLaunchKernelGGL(splitk_reduce_t_kernel, dim3(ceil_div(N, 32), ceil_div(M, 32)), dim3(256), 0, st, (const float*)planes2, sp, M, N, C2t, ldc2t,
                     accumulate ? 1 : 0);
  MTS_LAUNCH_CHECK("mts_wgrad_pair");
  return MTS_OK;
}

extern "C" int mts_gemm_plan(int a_dtype, int c_dtype, int layout, int M, int N, int K, unsigned epilogue, size_t workspace_bytes,
                             int* tile, int* splits) {
  MTS_CHECK_ARG(M > 0 && N > 0 && K > 0, "mts_gemm_plan: bad shape M=%d N=%d K=%d", M, N, K);
  MTS_CHECK_ARG(layout == MTS_NT || layout == MTS_NN || layout == MTS_TN || layout == MTS_TT, "mts_gemm_plan: bad layout %d", layout);
  MTS_CHECK_ARG(a_dtype == MTS_F32 || a_dtype == MTS_BF16, "mts_gemm_plan: bad a_dtype %d", a_dtype);
  int use256 = 0, sp = 1;
  const size_t ws_planes = workspace_bytes > MTS_GEMM_WS_HEAD ? workspace_bytes - MTS_GEMM_WS_HEAD : 0;
  if (a_dtype == MTS_BF16) plan_bf16(c_dtype, layout, M, N, K, epilogue, ws_planes > 0, ws_planes, &use256, &sp);
  if (tile) *tile = a_dtype == MTS_F32 ? (g_f32_mfma ? 128 : 64) : use256 == 2 ? 224 : use256 == 1 ? 256 : 128;
  if (splits) *splits = sp;
  return MTS_OK;
}

extern "C" int mts_gemm(void* stream, int a_dtype, int c_dtype, int layout, int M, int N, int K, const void* A, int lda,
                        const void* B, int ldb, void* C, int ldc, const float* bias, const void* residual, int ldr,
                        void* aux, int ldaux, unsigned epilogue, float colscale, int ncols_scaled, void* workspace, size_t workspace_bytes) {
  MTS_CHECK_ARG(M > 0 && N > 0 && K > 0, "mts_gemm: bad shape M=%d N=%d K=%d", M, N, K);
  MTS_CHECK_ARG(A && B && C, "mts_gemm: null operand");
  MTS_CHECK_ARG(layout == MTS_NT || layout == MTS_NN || layout == MTS_TN || layout == MTS_TT, "mts_gemm: bad layout %d", layout);
  MTS_CHECK_ARG(a_dtype == MTS_F32 || a_dtype == MTS_BF16, "mts_gemm: bad a_dtype %d", a_dtype);
  MTS_CHECK_ARG(c_dtype == MTS_F32 || c_dtype == a_dtype, "mts_gemm: c_dtype must be f32 or a_dtype");
  MTS_CHECK_ARG(!(epilogue & MTS_EPI_BIAS) || bias, "mts_gemm: MTS_EPI_BIAS without bias");
  MTS_CHECK_ARG(!(epilogue & MTS_EPI_RESIDUAL) || residual, "mts_gemm: MTS_EPI_RESIDUAL without residual");
  MTS_CHECK_ARG(!(epilogue & MTS_EPI_ACCUM) || c_dtype == MTS_F32, "mts_gemm: MTS_EPI_ACCUM needs fp32 C");
  MTS_CHECK_ARG((epilogue & (MTS_EPI_GELU | MTS_EPI_RELU)) != (MTS_EPI_GELU | MTS_EPI_RELU), "mts_gemm: MTS_EPI_GELU and MTS_EPI_RELU are exclusive");
  hipStream_t st = (hipStream_t)stream;
  GemmArgs a;
  a.A = A; a.B = B; a.C = C; a.bias = bias; a.residual = residual; a.aux = aux;
  a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.ldr = ldr; a.ldaux = ldaux;
  a.epi = epilogue; a.colscale = colscale; a.ncols_scaled = ncols_scaled;
  a.variant = g_gemm_variant;
#ifdef MTS_GEMM_STAMPS
  a.stamps = nullptr;
#endif
  a.ksplit = K; a.slab = nullptr; a.chain = nullptr; a.order = g_tile_order; a.nbuf = 2;

  if (a_dtype == MTS_F32) {
    StrideArgs s;
    if (layout == MTS_NT) { s.sam = lda; s.sak = 1; s.sbn = ldb; s.sbk = 1; }
    else if (layout == MTS_NN) { s.sam = lda; s.sak = 1; s.sbn = 1; s.sbk = ldb; }
    else if (layout == MTS_TT) { s.sam = 1; s.sak = lda; s.sbn = ldb; s.sbk = 1; }
    else { s.sam = 1; s.sak = lda; s.sbn = 1; s.sbk = ldb; }
    MTS_CHECK_ARG(ldc % 4 == 0 && ((uintptr_t)C % 16) == 0, "mts_gemm(f32): C must be 16-byte aligned with ldc %% 4 == 0");
    MTS_CHECK_ARG(!(epilogue & MTS_EPI_RESIDUAL) || (ldr % 4 == 0), "mts_gemm(f32): ldr %% 4");
    // "gemm_f32_mfma" = 1 (default): exact-fp32 matrix-core kernel; 0: the VALU kernel (A/B, and the form the round-1 fixtures ran on)
    // split-K for the weight-gradient shapes (K = all tokens, a handful of 128 x 128 output tiles: dW_hh of one LSTM direction is 16
    // tiles on 256 CUs): up to 16 slices of >= 512 k each while the launch stays within one wave of workgroups; plain or
    // accumulating epilogue only (bias / residual / activation belong to the first slice and forward GEMMs have tiles enough)
    const int nt = ceil_div(M, F32_BM) * ceil_div(N, F32_BM);
    int splits = 1;
    if (g_f32_mfma && workspace && !(epilogue & ~MTS_EPI_ACCUM) && N % 4 == 0 && nt <= 128 && K >= 1024) {
      splits = std::min(std::min(16, 256 / nt), K / 512);
      while (splits > 1 && (size_t)splits * M * N * sizeof(float) > workspace_bytes) --splits;
    }
    g_last_tile = 128;
    g_last_splits = splits;
    if (splits > 1) {
      a.ksplit = ceil_div(ceil_div(K, splits), F32_BK) * F32_BK;
      splits = ceil_div(K, a.ksplit);
      a.slab = (float*)workspace;
    }
    if (g_f32_mfma) hipLaunchKernelGGL(gemm_f32_mfma_kernel, dim3(nt, 1, splits), dim3(256), 0, st, a, s);
    else hipLaunchKernelGGL(gemm_f32_kernel, dim3(ceil_div(M, 64) * ceil_div(N, 64)), dim3(256), 0, st, a, s);
    if (splits > 1)
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(((size_t)M * (N / 4) + 255) / 256)), dim3(256), 0, st, (const float*)workspace,
                         splits, M, N, (float*)C, ldc, (epilogue & MTS_EPI_ACCUM) ? 1 : 0);
    MTS_LAUNCH_CHECK("mts_gemm(f32)");
    return MTS_OK;
  }

  // bf16 MFMA path: 16-byte vector accesses along the contiguous dimension of every operand
  MTS_UNSUPPORTED(lda % 8 == 0 && ldb % 8 == 0 && ((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0,
                  "mts_gemm(bf16): A/B must be 16-byte aligned with ld %% 8 == 0 (lda=%d ldb=%d)", lda, ldb);
  if (layout == MTS_NT) MTS_UNSUPPORTED(K % 8 == 0, "mts_gemm(bf16,NT): K %% 8 != 0 (K=%d)", K);
  if (layout == MTS_NN) MTS_UNSUPPORTED(K % 8 == 0 && N % 8 == 0, "mts_gemm(bf16,NN): K,N %% 8 (K=%d N=%d)", K, N);
  if (layout == MTS_TN) MTS_UNSUPPORTED(M % 8 == 0 && N % 8 == 0, "mts_gemm(bf16,TN): M,N %% 8 (M=%d N=%d)", M, N);
  if (layout == MTS_TT) MTS_UNSUPPORTED(M % 8 == 0 && K % 8 == 0, "mts_gemm(bf16,TT): M,K %% 8 (M=%d K=%d)", M, K);
  MTS_UNSUPPORTED(ldc % 4 == 0 && ((uintptr_t)C % (c_dtype == MTS_F32 ? 16 : 8)) == 0, "mts_gemm(bf16): C alignment/ldc");
  MTS_UNSUPPORTED(!(epilogue & MTS_EPI_RESIDUAL) || (ldr % 4 == 0 && ((uintptr_t)residual % 8) == 0), "mts_gemm(bf16): residual alignment");
  MTS_UNSUPPORTED(!aux || (ldaux % 4 == 0 && ((uintptr_t)aux % 8) == 0), "mts_gemm(bf16): aux alignment");

  // workspace = [MTS_GEMM_WS_HEAD bytes of arrival tickets][split-K planes]
  int splits = 1, use256 = 0;
  const size_t ws_planes = workspace_bytes > MTS_GEMM_WS_HEAD ? workspace_bytes - MTS_GEMM_WS_HEAD : 0;
  float* const planes = workspace ? (float*)((char*)workspace + MTS_GEMM_WS_HEAD) : nullptr;
  plan_bf16(c_dtype, layout, M, N, K, epilogue, workspace != nullptr && ws_planes > 0, ws_planes, &use256, &splits);
  g_last_tile = use256 == 2 ? 224 : use256 == 1 ? 256 : 128;
  g_last_splits = splits;
  a.slab = nullptr;
  bool chained = false;
  if (splits > 1) {
    a.ksplit = ceil_div(ceil_div(K, splits), BK) * BK;
    chained = g_chain && !use256;
    if (chained) {
      const size_t nt128 = (size_t)ceil_div(M, BM) * ceil_div(N, BN);
      a.chain = (unsigned*)planes;
      hipError_t e = hipMemsetAsync(planes, 0, nt128 * sizeof(unsigned), st);
      if (e != hipSuccess) { mts_set_error("mts_gemm: hipMemsetAsync: %s", hipGetErrorString(e)); return MTS_ERR_LAUNCH; }
    } else {
      a.slab = planes;
      // weight gradients on the 256x224 four-wave kernel: the slice of a tile that arrives last adds the planes inside the launch
      // ("gemm_combine" = 0: the reduce launch, A/B); the kernel's launcher falls back to the eight-wave kernel + reduce launch by itself
      if (use256 == 2 && g_combine && (a.variant == 0 || a.variant == 11 || a.variant == 12) && (size_t)ceil_div(M, 256) * (N / 224) * sizeof(unsigned) <= MTS_GEMM_WS_HEAD &&
          mts_gemm224t_applies(a, layout, c_dtype == MTS_F32, splits)) {
        a.chain = (unsigned*)workspace;
        chained = true;
      }
    }
  }
  if (use256) {
    int rc = use256 == 2 ? mts_launch_gemm224(a, layout, c_dtype == MTS_F32, splits, st) : mts_launch_gemm256(a, layout, c_dtype == MTS_F32, splits, st);
    if (rc) return rc;
    // (mts_gemm_last_plan: 225 / 226 = the 224-wide tile on gemm_bf16_224n_kernel / gemm_bf16_224d_kernel -- other symbols in a kernel trace)
    if (use256 == 2) g_last_tile = mts_gemm224_last_kernel();
  } else if (c_dtype == MTS_F32) {
    if (layout == MTS_NT) launch_bf16<MTS_NT, float>(a, splits, st);
    else if (layout == MTS_NN) launch_bf16<MTS_NN, float>(a, splits, st);
    else if (layout == MTS_TT) launch_bf16<MTS_TT, float>(a, splits, st);
    else launch_bf16<MTS_TN, float>(a, splits, st);
  } else {
    if (layout == MTS_NT) launch_bf16<MTS_NT, bf16_t>(a, splits, st);
    else if (layout == MTS_NN) launch_bf16<MTS_NN, bf16_t>(a, splits, st);
    else if (layout == MTS_TT) launch_bf16<MTS_TT, bf16_t>(a, splits, st);
    else launch_bf16<MTS_TN, bf16_t>(a, splits, st);
  }
  if (g_mid_hook) g_mid_hook();
  if (splits > 1 && !chained)
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)(((size_t)M * (N / 4) + 255) / 256)), dim3(256), 0, st, (const float*)planes,
                       splits, M, N, (float*)C, ldc, (epilogue & MTS_EPI_ACCUM) ? 1 : 0);
  MTS_LAUNCH_CHECK("mts_gemm(bf16)");
  return MTS_OK;
}
