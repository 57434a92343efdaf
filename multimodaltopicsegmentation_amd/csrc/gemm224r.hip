// 256x224x64 bf16 MFMA GEMM, NT, bf16 C, FOUR waves per workgroup, ONE tile per workgroup: gemm_bf16_224d_kernel.
//
// The round-3 forward-projection kernel, kept as the in-process A/B partner (mts_set_option("gemm_variant", 9)) of its persistent successor
// gemm_bf16_224p_kernel (gemm224p.hip), which is the default.  Structure: four waves of 128 x 112 (224 accumulators each,
// one wave per SIMD), operands HBM -> LDS by buffer-load LDS-DMA (resource descriptor + one of two lane offsets + an SGPR offset per copy), two
// 64-KiB stages, every fragment of a K-tile in registers, two barriers per K-tile, copies of K-tile kt + 2 two per block behind barrier 1.
// Two earlier four-wave forms lived in this file through round 3 (register-staged operands: gemm_bf16_224r_kernel, and the vendor kernel's loop
// structure on register-staged operands: gemm_bf16_224v_kernel); both measured slower than this one (profiles/r03_gemm224r_ab.txt) and were
// removed in round 4 with their timing-experiment variants.
#include <algorithm>
#include <type_traits>
#include "gemm_common.h"

#define R_A_BYTES 32768                       // 256 rows x 128 B (K-major)
#define R_B_BYTES 32768                       // 224 rows x 128 B K-major image (28 KiB used)
#define R_STAGE (R_A_BYTES + R_B_BYTES)
#define R_EPI (2 * R_STAGE)                   // the epilogue's output staging: 4 x 4 KiB behind the stages (the stages take the residual tile)
#define R_LDS (2 * R_STAGE + 16384)           // 144 KiB
#define R_BN 224
#define R_HN 112

// (free function templates, not generic lambdas: clang rejects inline-asm operands that name variables captured by a generic lambda)
// one B fragment (K-major only: one LDS operation) -- the k-step-1 fragments are requested one per block
template <int KS, int J>
__device__ __forceinline__ void r_rd_b1(LFrag<true> (&fb)[7], unsigned base, const unsigned (&lk)[2]) {
  const unsigned ad = base + lk[KS];
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[J].k) : "v"(ad), "n"(J * 2048));
}
template <int KS, int I>
__device__ __forceinline__ void v_rd_a(s16x8 (&fa)[2][8], unsigned base, const unsigned (&lk)[2]) {
  const unsigned ad = base + lk[KS];
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[KS][I]) : "v"(ad), "n"(I * 2048));
}
template <int N>
__device__ __forceinline__ void v_block(f32x4 (&acc)[8][7], s16x8 (&fa)[2][8], LFrag<true> (&fb)[2][7]) {
  constexpr int ks = N >> 3, i = N & 7;
  asm volatile("" : "+v"(fa[ks][i]));
  const bf16x8 va = __builtin_bit_cast(bf16x8, fa[ks][i]);
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    asm volatile("" : "+v"(fb[ks][j].k));
    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[ks][j].k), va, acc[i][j], 0, 0, 0);
  }
}
// request number R (0..14) of a k-step's 15 fragments: 0..6 = B fragment R, 7..14 = A fragment R - 7
template <int KS, int R>
__device__ __forceinline__ void v_rd(s16x8 (&fa)[2][8], LFrag<true> (&fb)[2][7], unsigned fA, unsigned fB, const unsigned (&lk)[2]) {
  if constexpr (R < 7) r_rd_b1<KS, R>(fb[KS], fB, lk);
  else v_rd_a<KS, R - 7>(fa, fA, lk);
}

__global__ __launch_bounds__(256, 1) void gemm_bf16_224d_kernel(const GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave_u >> 1, wn = wave_u & 1;
  const int r16 = lane & 15, g = lane >> 4;
  const int ntn = a.N / R_BN, ntm = a.M / 256, nt = ntn * ntm;
  const int nk = a.K / BK;
  const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(a.A);
  const bf16_t* __restrict__ B = reinterpret_cast<const bf16_t*>(a.B);
  int bm0, bn0;
  {
    const int t = blockIdx.x;
    const int q = nt >> 3, rr = nt & 7, xcd = t & 7, idx = t >> 3;
    const int id = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
    if (a.order == 0) { bm0 = (id / ntn) * 256; bn0 = (id % ntn) * R_BN; }
    else {
      const int band = id / (4 * ntn), within = id - band * 4 * ntn;
      const int rows = min(4, ntm - band * 4);
      bm0 = (band * 4 + within % rows) * 256;
      bn0 = (within / rows) * R_BN;
    }
  }
  // ---- HBM -> LDS by buffer-load LDS-DMA: 32 A pieces + 28 B pieces of 1 KiB (8 rows x 128 B) per K-tile, 8 + 7 per wave.  A piece is
  // lane-contiguous in the LDS (16 B per lane); the K-major swizzle (chunk position = source chunk ^ key(row)) is applied on the SOURCE side:
  // lane (row8 = lane >> 3, pos = lane & 7) fetches chunk pos ^ ((row >> 1) & 7) of its row, row = 8 piece + row8, i.e. one of two lane offsets
  // by the parity of the piece.  Everything else of a copy's address is scalar: resource (tile origin), soffset = piece rows + K offset.
  typedef __attribute__((address_space(3))) void dlptr_t;
  const int row8 = lane >> 3, pos = lane & 7;
  const unsigned keyE = (unsigned)((row8 >> 1) & 7), keyO = (unsigned)(((row8 >> 1) + 4) & 7);
  const unsigned voA[2] = {(unsigned)(row8 * a.lda + (int)((pos ^ keyE) * 8)) * 2u, (unsigned)(row8 * a.lda + (int)((pos ^ keyO) * 8)) * 2u};
  const unsigned voB[2] = {(unsigned)(row8 * a.ldb + (int)((pos ^ keyE) * 8)) * 2u, (unsigned)(row8 * a.ldb + (int)((pos ^ keyO) * 8)) * 2u};
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)bm0 * a.lda), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(B + (size_t)bn0 * a.ldb), 0, 0x7ffffff0, 0x00020000);
  const int rowsA = 16 * a.lda, rowsB = 16 * a.ldb;          // bytes per 8 rows
  // copy number c (0..14) of this wave for K-tile kt into stage st: 0..7 = A piece 8 wave + c, 8..14 = B piece 7 wave + c - 8
  auto copy1 = [&](auto C, int kt, int st) {
    constexpr int c = decltype(C)::value;
    char* dst0 = smem + st * R_STAGE;
    if constexpr (c < 8) {
      const int piece = wave_u * 8 + c;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (dlptr_t*)(dst0 + piece * 1024), 16, voA[c & 1], piece * rowsA + kt * (BK * 2), 0, 0);
    } else {
      const int piece = wave_u * 7 + (c - 8);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (dlptr_t*)(dst0 + R_A_BYTES + piece * 1024), 16, voB[piece & 1], piece * rowsB + kt * (BK * 2), 0, 0);
    }
  };
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  unsigned lk[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) lk[ks] = r16 * 128 + (((ks * 4 + g) ^ ((r16 >> 1) & 7)) << 4);
  const unsigned fA = lds0 + wm * 16384;
  const unsigned fB = lds0 + R_A_BYTES + wn * (R_HN * 128);

  s16x8 fa[2][8];
  LFrag<true> fb[2][7];
  f32x4 acc[8][7];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 7; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#define VBLOCK(N_) do { __builtin_amdgcn_sched_barrier(0); v_block<N_>(acc, fa, fb); __builtin_amdgcn_sched_barrier(0); } while (0)
#define VRD(ST_, KS_, R_) v_rd<KS_, R_>(fa, fb, fA + (ST_) * R_STAGE, fB + (ST_) * R_STAGE, lk)
#define CP(C_, KT_, ST_) copy1(std::integral_constant<int, C_>{}, KT_, ST_)
  // the epilogue's bias (7 x 16 bytes per lane) is fetched HERE, ahead of every copy: after the K loop it cost a memory round trip per tile
  const int m0 = bm0 + wm * 128, n0 = bn0 + wn * R_HN;
  const bool has_bias = (a.epi & MTS_EPI_BIAS) != 0, has_res = (a.epi & MTS_EPI_RESIDUAL) != 0;
  const float* bias_p = has_bias ? a.bias + n0 + 4 * g : reinterpret_cast<const float*>(a.A) + 4 * g;
  float4 bias[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) bias[j] = *reinterpret_cast<const float4*>(bias_p + j * 16);
  // ---- the residual tile goes through the LDS (as in gemm_bf16_224n_kernel, gemm224n.hip): fetched into registers in the epilogue it is 56 loads per
  // lane and a memory round trip per tile.  The wave's rows 0..63 (x 14 chunks of 16 B) are copied into the stage of K-tile nk - 2 behind that K-tile's
  // barrier 2, rows 64..127 into the stage of K-tile nk - 1 behind its last fragment read: 16 LDS-DMA each, four rows per copy (56 of 64 lanes).
  const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(
      has_res ? (void*)(reinterpret_cast<const bf16_t*>(a.residual) + (size_t)m0 * a.ldr + n0) : (void*)A, 0, 0x7ffffff0, 0x00020000);
  const unsigned voR = (unsigned)((lane / 14) * a.ldr + (lane % 14) * 8) * 2u;
  const int rowR4 = 8 * a.ldr;                                     // bytes per 4 rows of the residual
  auto res_copy = [&](int c, int h, int st) {
    if (lane < 56)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsR, (dlptr_t*)(smem + st * R_STAGE + wave_u * 16384 + c * 1024), 16, voR, (h * 16 + c) * rowR4, 0, 0);
  };
  // ---- prologue: K-tiles 0 and 1 -> stages 0 and 1; fragments of K-tile 0 / k-step 0 requested -------------------------------------------
  CP(0, 0, 0); CP(1, 0, 0); CP(2, 0, 0); CP(3, 0, 0); CP(4, 0, 0); CP(5, 0, 0); CP(6, 0, 0); CP(7, 0, 0);
  CP(8, 0, 0); CP(9, 0, 0); CP(10, 0, 0); CP(11, 0, 0); CP(12, 0, 0); CP(13, 0, 0); CP(14, 0, 0);
  CP(0, 1, 1); CP(1, 1, 1); CP(2, 1, 1); CP(3, 1, 1); CP(4, 1, 1); CP(5, 1, 1); CP(6, 1, 1); CP(7, 1, 1);
  CP(8, 1, 1); CP(9, 1, 1); CP(10, 1, 1); CP(11, 1, 1); CP(12, 1, 1); CP(13, 1, 1); CP(14, 1, 1);
  asm volatile("s_waitcnt vmcnt(15)" ::: "memory");      // K-tile 0 has landed (in-order counter), K-tile 1 may still fly
  __builtin_amdgcn_s_barrier();
  VRD(0, 0, 0); VRD(0, 0, 1); VRD(0, 0, 2); VRD(0, 0, 3); VRD(0, 0, 4); VRD(0, 0, 5); VRD(0, 0, 6); VRD(0, 0, 7); VRD(0, 0, 8); VRD(0, 0, 9);
  VRD(0, 0, 10); VRD(0, 0, 11); VRD(0, 0, 12); VRD(0, 0, 13); VRD(0, 0, 14);

  // One K-tile (stage st = kt & 1).  NXT: K-tile kt + 1 exists (it was copied a K-tile ago into the other stage); LD: K-tile kt + 2 exists -- its 15
  // copies go into THIS stage once everybody has finished reading it (barrier 1), two per block from block 8 on.
  auto ktile = [&](int kt, auto NXT, auto LD) {
    constexpr bool nxt = decltype(NXT)::value, ld = decltype(LD)::value;
    const int st = kt & 1;
#define KA(n)                                                               \
    lgkm_wait<(7 - (n)) + 2 * (n)>();                                       \
    VBLOCK(n);                                                              \
    if constexpr (2 * (n) < 15) VRD(st, 1, (2 * (n) < 15 ? 2 * (n) : 0));       \
    if constexpr (2 * (n) + 1 < 15) VRD(st, 1, (2 * (n) + 1 < 15 ? 2 * (n) + 1 : 0));
    KA(0) KA(1) KA(2) KA(3) KA(4) KA(5) KA(6) KA(7)
#undef KA
    lgkm_wait<0>();                                         // every fragment of this K-tile is in my registers
    if constexpr (ld) __builtin_amdgcn_s_barrier();         // ... and in everybody's: this stage may be overwritten
    if constexpr (!nxt) { if (has_res) __builtin_amdgcn_s_barrier(); }     // (last K-tile: by the second half of the residual)
#define KC(c) if constexpr (ld) CP(c, kt + 2, st);
    // K-tile nk - 2 (nxt && !ld): behind barrier 2 nobody reads this stage any more -> the first half of the residual (6 + 5 + 5 copies)
    // K-tile nk - 1 (!nxt): the barrier above says the same of the last stage -> the second half, two copies per block
#define KR(c0, n_) if constexpr (nxt && !ld) { if (has_res) { for (int c_ = (c0); c_ < (c0) + (n_); ++c_) res_copy(c_, 0, st); } }
#define KS(c0) if constexpr (!nxt) { if (has_res) { res_copy((c0), 1, st); res_copy((c0) + 1, 1, st); } }
    VBLOCK(8);  KC(0) KC(1) KS(0)
    VBLOCK(9);  KC(2) KC(3) KS(2)
    VBLOCK(10); KC(4) KC(5) KS(4)
    VBLOCK(11); KC(6) KC(7) KS(6)
    VBLOCK(12); KC(8) KC(9) KS(8)
    if constexpr (nxt) {
      // K-tile kt + 1 (copied during K-tile kt - 1, or in the prologue) has landed: only this K-tile's ten copies may still fly
      if constexpr (ld) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    VBLOCK(13); KC(10) KC(11) KR(0, 6) KS(10)
    if constexpr (nxt) { VRD(st ^ 1, 0, 0); VRD(st ^ 1, 0, 1); VRD(st ^ 1, 0, 2); VRD(st ^ 1, 0, 3); VRD(st ^ 1, 0, 4); }
    VBLOCK(14); KC(12) KC(13) KR(6, 5) KS(12)
    if constexpr (nxt) { VRD(st ^ 1, 0, 5); VRD(st ^ 1, 0, 6); VRD(st ^ 1, 0, 7); VRD(st ^ 1, 0, 8); VRD(st ^ 1, 0, 9); }
    VBLOCK(15); KC(14) KR(11, 5) KS(14)
    if constexpr (nxt) { VRD(st ^ 1, 0, 10); VRD(st ^ 1, 0, 11); VRD(st ^ 1, 0, 12); VRD(st ^ 1, 0, 13); VRD(st ^ 1, 0, 14); }
#undef KC
#undef KR
#undef KS
  };
  {
    using T = std::true_type; using F = std::false_type;
    int kt = 0;
#pragma clang loop unroll(disable)
    for (; kt + 2 < nk; ++kt) ktile(kt, T{}, T{});
    ktile(kt, T{}, F{});
    ktile(kt + 1, F{}, F{});
  }
#undef CP
#undef VGLOAD
#undef VBLOCK
#undef VRD

  // ---- epilogue: bias / column scale / residual, 16 rows at a time through a wave-private LDS area; the residual out of the stages ------------
  char* stage = smem + R_EPI + wave_u * 4096;
  const int stA = (nk - 2) & 1;                              // the stage that took the first half of the residual
  if (has_res) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");     // the first half has landed (in-order counter): the second half's 16 copies may still fly
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const float colscale = (a.epi & MTS_EPI_COLSCALE) ? a.colscale : 1.0f;
  const int nsc = (a.epi & MTS_EPI_COLSCALE) ? a.ncols_scaled - n0 - 4 * g : 0;
  bf16_t* __restrict__ C = reinterpret_cast<bf16_t*>(a.C);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    // (younger than the second half's copies are the 16 store instructions of groups 0..3)
    if (i == 4 && has_res) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    uint2 res[7];
    if (has_res) {
      const int rr = (i & 3) * 16 + r16;
      const char* rp = smem + ((i < 4) ? stA : (stA ^ 1)) * R_STAGE + wave_u * 16384 + (rr >> 2) * 1024 + ((rr & 3) * 14 + (g >> 1)) * 16 + (g & 1) * 8;
#pragma unroll
      for (int j = 0; j < 7; ++j) res[j] = *reinterpret_cast<const uint2*>(rp + j * 32);
    } else {
#pragma unroll
      for (int j = 0; j < 7; ++j) res[j] = make_uint2(0u, 0u);
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const float sc = (j * 16 < nsc) ? colscale : 1.0f;
      const float4 bb = has_bias ? bias[j] : make_float4(0.f, 0.f, 0.f, 0.f);
      const uint2 rr2 = has_res ? res[j] : make_uint2(0u, 0u);
      uint2 pk;
      pk.x = pack_bf16x2((acc[i][j][0] + bb.x) * sc + bf16_lo(rr2.x), (acc[i][j][1] + bb.y) * sc + bf16_hi(rr2.x));
      pk.y = pack_bf16x2((acc[i][j][2] + bb.z) * sc + bf16_lo(rr2.y), (acc[i][j][3] + bb.w) * sc + bf16_hi(rr2.y));
      *reinterpret_cast<uint2*>(stage + r16 * 240 + (j * 16 + 4 * g) * 2) = pk;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int idx = it * 64 + lane;
      const int row = idx / 14, chn = idx - row * 14;
      if (idx < 16 * 14) {
        const uint4 val = *reinterpret_cast<const uint4*>(stage + row * 240 + chn * 16);
        *reinterpret_cast<uint4*>(C + (size_t)(m0 + i * 16 + row) * a.ldc + n0 + chn * 8) = val;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

static int d_launch(const GemmArgs& a, hipStream_t st) {
  auto k = gemm_bf16_224d_kernel;
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, R_LDS);
    if (e != hipSuccess) { mts_set_error("gemm224d: hipFuncSetAttribute: %s", hipGetErrorString(e)); return MTS_ERR_LAUNCH; }
    attr_set = true;
  }
  const int nt = (a.M / 256) * (a.N / R_BN);
  hipLaunchKernelGGL(k, dim3(nt), dim3(256), R_LDS, st, a);
  return MTS_OK;
}

// called from mts_launch_gemm224 (gemm224.hip); -1: shape / epilogue not covered here
int mts_launch_gemm224r(const GemmArgs& a, int layout, bool c_is_f32, int splits, hipStream_t st) {
  if (c_is_f32 || splits != 1 || layout != MTS_NT) return -1;
  const unsigned simple = MTS_EPI_BIAS | MTS_EPI_COLSCALE | MTS_EPI_RESIDUAL;
  const bool ok = !a.slab && (a.epi & ~simple) == 0 && (a.M % 256 == 0) && (a.N % R_BN == 0) && (a.K % BK == 0) && a.K % (2 * BK) == 0 && a.K >= 4 * BK && a.ksplit == a.K &&
                  (a.ldc % 8 == 0) && (((uintptr_t)a.C & 15) == 0) && (a.lda % 8 == 0) && (a.ldb % 8 == 0) &&
                  (!(a.epi & MTS_EPI_COLSCALE) || a.ncols_scaled % 4 == 0) &&
                  (!(a.epi & MTS_EPI_RESIDUAL) || (a.ldr % 4 == 0 && ((uintptr_t)a.residual & 7) == 0)) &&
                  (!(a.epi & MTS_EPI_BIAS) || ((uintptr_t)a.bias & 15) == 0);
  if (!ok) return -1;
  return d_launch(a, st);
}
