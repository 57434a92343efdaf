// 256x224x64 bf16 MFMA GEMM for the FORWARD PROJECTIONS (layout NT, bf16 C), four waves per workgroup, PERSISTENT: one workgroup per CU walks its
// tiles and the K loop never drains between them.
//
//   Y[M, N] (bf16) = X[M, K] W[N, K]^T (+ bias, column scale, residual)        (modeling_longformer.py:504-506 q|k|v, :1069 attention output)
//
// gemm_bf16_224d_kernel (the round-3 form: the same four waves, buffer-load LDS-DMA copies, whole-K-tile fragments in registers, two barriers
// per K-tile) computes ONE tile per workgroup.  With 128 KiB of LDS a CU holds one workgroup, so nothing overlaps a tile's edges: the next
// workgroup is dispatched when this one has retired, issues its first 30 copies, waits a memory round trip for them, reads 15 fragments -- and
// 28 K-tiles later (K = 1792) stores its tile and leaves: ~3.7 us of a 45-us tile, six times per CU on the q|k|v projection.  Here the stream
// of K-tiles runs on across the tile boundary:
//   * during the last two K-tiles of a tile the copies that the steady state would issue for "K-tiles nk and nk + 1" fetch K-tiles 0 and 1 of
//     the NEXT tile (same stages, same counted waits: only the buffer resources differ);
//   * the epilogue stores the tile from a staging area of its own (16 KiB behind the two 64-KiB stages) while those copies fly; the bias row
//     of a tile is itself copied into LDS behind its K-tile 0 (no registers held across the K loop, no memory round trip in the epilogue);
//   * accumulators are BORN in a tile's first K-tile (the MFMAs of K-tile 0, k-step 0 take C = 0) and die in its epilogue: nothing but scalars
//     lives across the tile loop, which is what lets the compiler keep the 224 accumulators where they are.
// Accumulation order per output element is that of every other bf16 kernel here (K-tiles ascending, k-step 0 then 1): bitwise the results of
// gemm_bf16_224d_kernel (tests/test_gpu_kernels.py::test_gemm_224_barrier_schedules_agree_bitwise, variants 0 / 9).
#include <algorithm>
#include <type_traits>
#include "gemm_common.h"

#define P_A_BYTES 32768                       // 256 rows x 128 B (K-major)
#define P_STAGE 65536                         // A image | B image (224 rows x 128 B = 28 KiB used)
#define P_EPI (2 * P_STAGE)                   // store staging: 4 waves x 4 KiB
#define P_BIAS (P_EPI + 16384)                // the tile's bias row: 224 floats
#define P_LDS (P_BIAS + 1024)
#define P_BN 224
#define P_HN 112

typedef __attribute__((address_space(3))) void p_dlptr;

// request R (0..14) of k-step KS: 0..6 = B fragment R, 7..14 = A fragment R - 7 (fA / fB: the wave's image bases incl. the stage)
template <int KS, int R>
__device__ __forceinline__ void p_rd(s16x8 (&fa)[2][8], s16x8 (&fb)[2][7], unsigned fA, unsigned fB, const unsigned (&lk)[2]) {
  if constexpr (R < 7) {
    const unsigned ad = fB + lk[KS];
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[KS][R]) : "v"(ad), "n"(R * 2048));
  } else {
    const unsigned ad = fA + lk[KS];
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[KS][R - 7]) : "v"(ad), "n"((R - 7) * 2048));
  }
}
// block N = 8 ks + i: A fragment i of k-step ks x the 7 B fragments.  ZERO: the accumulators of row block i start here (C = 0)
template <int N, bool ZERO>
__device__ __forceinline__ void p_block(f32x4 (&acc)[8][7], s16x8 (&fa)[2][8], s16x8 (&fb)[2][7]) {
  constexpr int ks = N >> 3, i = N & 7;
  asm volatile("" : "+v"(fa[ks][i]));
  const bf16x8 va = __builtin_bit_cast(bf16x8, fa[ks][i]);
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    asm volatile("" : "+v"(fb[ks][j]));
    if constexpr (ZERO) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[ks][j]), va, (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[ks][j]), va, acc[i][j], 0, 0, 0);
  }
}

__global__ __launch_bounds__(256, 1) void gemm_bf16_224p_kernel(const GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave_u >> 1, wn = wave_u & 1;
  const int r16 = lane & 15, g = lane >> 4;
  const int ntn = a.N / P_BN, ntm = a.M / 256, nt = ntn * ntm;
  const int nk = a.K / BK;
  const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(a.A);
  const bf16_t* __restrict__ B = reinterpret_cast<const bf16_t*>(a.B);
  // ---- this workgroup's tiles.  Workgroups b, b + 8, .. run on one XCD (own L2): XCD x owns a contiguous run of tile ids and its wx workgroups
  // take ids first + idx, first + idx + wx, ..: at any time the XCD works on wx consecutive ids = (with the band order) a 4 x 8 block of tiles
  const int G = gridDim.x, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  const int wx = (G - xcd + 7) >> 3;                          // workgroups of this launch on my XCD
  const int q = nt >> 3, rr = nt & 7;
  const int first = xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q, cnt = xcd < rr ? q + 1 : q;
  auto origin = [&](int j, int& bm0, int& bn0) {              // j: index inside my XCD's run
    const int id = first + j;
    if (a.order == 0) { bm0 = (id / ntn) * 256; bn0 = (id % ntn) * P_BN; return; }
    const int band = id / (4 * ntn), within = id - band * 4 * ntn;
    const int rows = min(4, ntm - band * 4);
    bm0 = (band * 4 + within % rows) * 256;
    bn0 = (within / rows) * P_BN;
  };
  int j = idx;
  if (j >= cnt) return;

  // ---- copies: 32 A pieces + 28 B pieces of 1 KiB (8 rows x 128 B) per K-tile, 8 + 7 per wave; the K-major swizzle on the SOURCE side (two lane
  // offsets by the parity of the piece), everything else of a copy's address scalar (gemm_bf16_224d_kernel)
  const int row8 = lane >> 3, pos = lane & 7;
  const unsigned keyE = (unsigned)((row8 >> 1) & 7), keyO = (unsigned)(((row8 >> 1) + 4) & 7);
  const unsigned voA[2] = {(unsigned)(row8 * a.lda + (int)((pos ^ keyE) * 8)) * 2u, (unsigned)(row8 * a.lda + (int)((pos ^ keyO) * 8)) * 2u};
  const unsigned voB[2] = {(unsigned)(row8 * a.ldb + (int)((pos ^ keyE) * 8)) * 2u, (unsigned)(row8 * a.ldb + (int)((pos ^ keyO) * 8)) * 2u};
  const int rowsA = 16 * a.lda, rowsB = 16 * a.ldb;          // bytes per 8 rows
  auto rsrc_of = [&](const bf16_t* p) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, 0x7ffffff0, 0x00020000); };
  // copy c (0..14) of this wave for K-tile kt of the tile whose resources are (ra, rb), into stage st
  auto copy1 = [&](auto C, const __amdgpu_buffer_rsrc_t& ra, const __amdgpu_buffer_rsrc_t& rb, int kt, int st) {
    constexpr int c = decltype(C)::value;
    char* dst0 = smem + st * P_STAGE;
    if constexpr (c < 8) {
      const int piece = wave_u * 8 + c;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (p_dlptr*)(dst0 + piece * 1024), 16, voA[c & 1], piece * rowsA + kt * (BK * 2), 0, 0);
    } else {
      const int piece = wave_u * 7 + (c - 8);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (p_dlptr*)(dst0 + P_A_BYTES + piece * 1024), 16, voB[piece & 1], piece * rowsB + kt * (BK * 2), 0, 0);
    }
  };
  const bool has_bias = (a.epi & MTS_EPI_BIAS) != 0, has_res = (a.epi & MTS_EPI_RESIDUAL) != 0;
  // the tile's bias row -> LDS: 224 floats = 56 lanes x 16 B, one copy by wave 0 (behind the tile's K-tile-0 copies in the counter)
  const unsigned voBias = (unsigned)min(lane, 55) * 16u;
  auto copy_bias = [&](int bn0) {
    if (has_bias && wave_u == 0)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc((void*)(a.bias + bn0), 0, 0x7ffffff0, 0x00020000),
                                               (p_dlptr*)(smem + P_BIAS), 16, voBias, 0, 0, 0);
  };
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  unsigned lk[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) lk[ks] = r16 * 128 + (((ks * 4 + g) ^ ((r16 >> 1) & 7)) << 4);
  const unsigned fA = lds0 + wm * 16384;
  const unsigned fB = lds0 + P_A_BYTES + wn * (P_HN * 128);

  s16x8 fa[2][8], fb[2][7];
  f32x4 acc[8][7];

#define PBLOCK(N_, Z_) do { __builtin_amdgcn_sched_barrier(0); p_block<N_, Z_>(acc, fa, fb); __builtin_amdgcn_sched_barrier(0); } while (0)
#define PRD(ST_, KS_, R_) p_rd<KS_, R_>(fa, fb, fA + (ST_) * P_STAGE, fB + (ST_) * P_STAGE, lk)
#define PCP(C_, RA_, RB_, KT_, ST_) copy1(std::integral_constant<int, C_>{}, RA_, RB_, KT_, ST_)
  int bm0, bn0;
  origin(j, bm0, bn0);
  __amdgpu_buffer_rsrc_t rsA = rsrc_of(A + (size_t)bm0 * a.lda), rsB = rsrc_of(B + (size_t)bn0 * a.ldb);
  // ---- prologue of the FIRST tile: K-tiles 0 and 1 -> stages 0 and 1, its bias row, the fragments of K-tile 0 / k-step 0 -----------------------
  PCP(0, rsA, rsB, 0, 0); PCP(1, rsA, rsB, 0, 0); PCP(2, rsA, rsB, 0, 0); PCP(3, rsA, rsB, 0, 0); PCP(4, rsA, rsB, 0, 0); PCP(5, rsA, rsB, 0, 0);
  PCP(6, rsA, rsB, 0, 0); PCP(7, rsA, rsB, 0, 0); PCP(8, rsA, rsB, 0, 0); PCP(9, rsA, rsB, 0, 0); PCP(10, rsA, rsB, 0, 0); PCP(11, rsA, rsB, 0, 0);
  PCP(12, rsA, rsB, 0, 0); PCP(13, rsA, rsB, 0, 0); PCP(14, rsA, rsB, 0, 0);
  PCP(0, rsA, rsB, 1, 1); PCP(1, rsA, rsB, 1, 1); PCP(2, rsA, rsB, 1, 1); PCP(3, rsA, rsB, 1, 1); PCP(4, rsA, rsB, 1, 1); PCP(5, rsA, rsB, 1, 1);
  PCP(6, rsA, rsB, 1, 1); PCP(7, rsA, rsB, 1, 1); PCP(8, rsA, rsB, 1, 1); PCP(9, rsA, rsB, 1, 1); PCP(10, rsA, rsB, 1, 1); PCP(11, rsA, rsB, 1, 1);
  PCP(12, rsA, rsB, 1, 1); PCP(13, rsA, rsB, 1, 1); PCP(14, rsA, rsB, 1, 1);
  copy_bias(bn0);
  asm volatile("s_waitcnt vmcnt(15)" ::: "memory");      // K-tile 0 has landed (in-order counter); K-tile 1 (wave 0: all but its first copy, + the bias row) may still fly
  __builtin_amdgcn_s_barrier();

  // One K-tile (stage st = kt & 1).  ZERO: the tile's first K-tile (accumulators start).  NXT: the stream's next K-tile is read behind barrier 2
  // (inside a tile: kt + 1).  CPY: 15 copies into THIS stage once everybody has read it (barrier 1), two per block from block 8 on -- K-tile
  // (ckt) of the tile with resources (ra, rb): inside a tile kt + 2 of the same tile, in its last two K-tiles K-tiles 0 / 1 of the next tile.
  auto ktile = [&](int kt, auto ZERO, auto NXT, auto CPY, const __amdgpu_buffer_rsrc_t& ra, const __amdgpu_buffer_rsrc_t& rb, int ckt) {
    constexpr bool zero = decltype(ZERO)::value, nxt = decltype(NXT)::value, cpy = decltype(CPY)::value;
    const int st = kt & 1;
#define KA(n)                                                               \
    lgkm_wait<(7 - (n)) + 2 * (n)>();                                       \
    PBLOCK(n, zero);                                                        \
    if constexpr (2 * (n) < 15) PRD(st, 1, (2 * (n) < 15 ? 2 * (n) : 0));       \
    if constexpr (2 * (n) + 1 < 15) PRD(st, 1, (2 * (n) + 1 < 15 ? 2 * (n) + 1 : 0));
    KA(0) KA(1) KA(2) KA(3) KA(4) KA(5) KA(6) KA(7)
#undef KA
    lgkm_wait<0>();                                         // every fragment of this K-tile is in my registers
    if constexpr (cpy) __builtin_amdgcn_s_barrier();        // ... and in everybody's: this stage may be overwritten
#define KC(c) if constexpr (cpy) PCP(c, ra, rb, ckt, st);
    PBLOCK(8, false);  KC(0) KC(1)
    PBLOCK(9, false);  KC(2) KC(3)
    PBLOCK(10, false); KC(4) KC(5)
    PBLOCK(11, false); KC(6) KC(7)
    PBLOCK(12, false); KC(8) KC(9)
    if constexpr (nxt) {
      // the next K-tile (copied a K-tile ago) has landed: only this K-tile's ten copies may still fly
      if constexpr (cpy) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    PBLOCK(13, false); KC(10) KC(11)
    if constexpr (nxt) { PRD(st ^ 1, 0, 0); PRD(st ^ 1, 0, 1); PRD(st ^ 1, 0, 2); PRD(st ^ 1, 0, 3); PRD(st ^ 1, 0, 4); }
    PBLOCK(14, false); KC(12) KC(13)
    if constexpr (nxt) { PRD(st ^ 1, 0, 5); PRD(st ^ 1, 0, 6); PRD(st ^ 1, 0, 7); PRD(st ^ 1, 0, 8); PRD(st ^ 1, 0, 9); }
    PBLOCK(15, false); KC(14)
    if constexpr (nxt) { PRD(st ^ 1, 0, 10); PRD(st ^ 1, 0, 11); PRD(st ^ 1, 0, 12); PRD(st ^ 1, 0, 13); PRD(st ^ 1, 0, 14); }
#undef KC
  };
  using T = std::true_type; using F = std::false_type;
  char* const stage = smem + P_EPI + wave_u * 4096;
  bf16_t* __restrict__ C = reinterpret_cast<bf16_t*>(a.C);
  const float colscale = (a.epi & MTS_EPI_COLSCALE) ? a.colscale : 1.0f;

#pragma clang loop unroll(disable)
  for (;;) {
    // fragments of this tile's K-tile 0 / k-step 0 (it has landed: barrier above / at the end of the previous round)
    PRD(0, 0, 0); PRD(0, 0, 1); PRD(0, 0, 2); PRD(0, 0, 3); PRD(0, 0, 4); PRD(0, 0, 5); PRD(0, 0, 6); PRD(0, 0, 7); PRD(0, 0, 8); PRD(0, 0, 9);
    PRD(0, 0, 10); PRD(0, 0, 11); PRD(0, 0, 12); PRD(0, 0, 13); PRD(0, 0, 14);
    // the next tile of this workgroup (its own again when there is none: the copies then fetch bytes nobody uses)
    const int jn = j + wx;
    const bool more = jn < cnt;
    int bm0n, bn0n;
    origin(more ? jn : j, bm0n, bn0n);
    const __amdgpu_buffer_rsrc_t rsAn = rsrc_of(A + (size_t)bm0n * a.lda), rsBn = rsrc_of(B + (size_t)bn0n * a.ldb);
    const int m0 = bm0 + wm * 128, n0 = bn0 + wn * P_HN;

    ktile(0, T{}, T{}, T{}, rsA, rsB, 2);
    int kt = 1;
#pragma clang loop unroll(disable)
    for (; kt + 2 < nk; ++kt) ktile(kt, F{}, T{}, T{}, rsA, rsB, kt + 2);
    ktile(kt, F{}, T{}, T{}, rsAn, rsBn, 0);             // kt = nk - 2: its stage takes the NEXT tile's K-tile 0
    // kt = nk - 1: ... K-tile 1; no next K-tile to read behind barrier 2
    ktile(kt + 1, F{}, F{}, T{}, rsAn, rsBn, 1);
    // (pins the tile in the accumulator registers up to here: left alone the compiler starts moving it to VGPRs inside the last K-tile -- the
    // fragment registers are free there -- one MFMA, its full latency, four moves at a time)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int jj = 0; jj < 7; ++jj) asm volatile("" : "+a"(acc[i][jj]));

    // ---- epilogue: (acc + bias) * scale + residual -> bf16 -> staging -> 16-byte stores; the next tile's K-tiles 0 / 1 are landing meanwhile ----
    // (a register vector, not a float4 struct: the compiler copies a struct's members out of the asm's result registers right behind the
    // read, i.e. before the data has arrived)
    f32x4 bias[7];
    {
      const unsigned bad = lds0 + P_BIAS + (unsigned)(wn * P_HN + 4 * g) * 4u;
#pragma unroll
      for (int jj = 0; jj < 7; ++jj) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(bias[jj]) : "v"(bad), "n"(jj * 64));
    }
    const size_t res_ld = has_res ? (size_t)a.ldr : 0;
    const bf16_t* res_p = has_res ? reinterpret_cast<const bf16_t*>(a.residual) + (size_t)(m0 + r16) * a.ldr + n0 + 4 * g
                                  : reinterpret_cast<const bf16_t*>(a.A) + 4 * g;
    uint2 res[8][7];
    if (has_res) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int jj = 0; jj < 7; ++jj) res[i][jj] = *reinterpret_cast<const uint2*>(res_p + (size_t)(i * 16) * res_ld + jj * 16);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int jj = 0; jj < 7; ++jj) res[i][jj] = make_uint2(0u, 0u);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the bias row is in my registers
#pragma unroll
    for (int jj = 0; jj < 7; ++jj) asm volatile("" : "+v"(bias[jj]));
    const int nsc = (a.epi & MTS_EPI_COLSCALE) ? a.ncols_scaled - n0 - 4 * g : 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int jj = 0; jj < 7; ++jj) {
        const float sc = (jj * 16 < nsc) ? colscale : 1.0f;
        const float4 bb = has_bias ? make_float4(bias[jj][0], bias[jj][1], bias[jj][2], bias[jj][3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        const uint2 rv = has_res ? res[i][jj] : make_uint2(0u, 0u);
        uint2 pk;
        pk.x = pack_bf16x2((acc[i][jj][0] + bb.x) * sc + bf16_lo(rv.x), (acc[i][jj][1] + bb.y) * sc + bf16_hi(rv.x));
        pk.y = pack_bf16x2((acc[i][jj][2] + bb.z) * sc + bf16_lo(rv.y), (acc[i][jj][3] + bb.w) * sc + bf16_hi(rv.y));
        *reinterpret_cast<uint2*>(stage + r16 * 240 + (jj * 16 + 4 * g) * 2) = pk;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int ix = it * 64 + lane;
        const int row = ix / 14, chn = ix - row * 14;
        if (ix < 16 * 14) {
          const uint4 val = *reinterpret_cast<const uint4*>(stage + row * 240 + chn * 16);
          *reinterpret_cast<uint4*>(C + (size_t)(m0 + i * 16 + row) * a.ldc + n0 + chn * 8) = val;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (!more) break;
    // ---- next round.  Every wave has read the bias row (its own epilogue is behind it; the others': the barrier below) -- but the row's copy
    // must not land before they have: it goes out AFTER the barrier, and nobody reads it before the next epilogue, ~28 K-tiles away.
    j = jn; bm0 = bm0n; bn0 = bn0n; rsA = rsAn; rsB = rsBn;
    // K-tile 0 of the new tile has landed: younger in the counter are its K-tile 1 (15 copies) and this epilogue's 32 stores
    asm volatile("s_waitcnt vmcnt(47)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    copy_bias(bn0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the surplus copies of the last round land in my LDS before the workgroup ends)
#undef PCP
#undef PRD
#undef PBLOCK
}

// called from mts_launch_gemm224 (gemm224.hip); -1: shape / epilogue not covered here
int mts_launch_gemm224p(const GemmArgs& a, int layout, bool c_is_f32, int splits, hipStream_t st) {
  if (c_is_f32 || splits != 1 || layout != MTS_NT) return -1;
  const unsigned simple = MTS_EPI_BIAS | MTS_EPI_COLSCALE | MTS_EPI_RESIDUAL;
  const size_t spanA = (size_t)256 * a.lda * 2 + (size_t)a.K * 2, spanB = (size_t)224 * a.ldb * 2 + (size_t)a.K * 2;    // byte offsets inside a tile's panels
  const bool ok = !a.slab && (a.epi & ~simple) == 0 && (a.M % 256 == 0) && (a.N % P_BN == 0) && (a.K % (2 * BK) == 0) && a.K >= 4 * BK && a.ksplit == a.K &&
                  (a.ldc % 8 == 0) && (((uintptr_t)a.C & 15) == 0) && (a.lda % 8 == 0) && (a.ldb % 8 == 0) && (((uintptr_t)a.A & 15) == 0) &&
                  (((uintptr_t)a.B & 15) == 0) && spanA < 0x7ff00000u && spanB < 0x7ff00000u &&
                  (!(a.epi & MTS_EPI_COLSCALE) || a.ncols_scaled % 4 == 0) &&
                  (!(a.epi & MTS_EPI_RESIDUAL) || (a.ldr % 4 == 0 && ((uintptr_t)a.residual & 7) == 0)) &&
                  (!(a.epi & MTS_EPI_BIAS) || ((uintptr_t)a.bias & 15) == 0);
  if (!ok) return -1;
  auto k = gemm_bf16_224p_kernel;
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS);
    if (e != hipSuccess) { mts_set_error("gemm224p: hipFuncSetAttribute: %s", hipGetErrorString(e)); return MTS_ERR_LAUNCH; }
    attr_set = true;
  }
  const int nt = (a.M / 256) * (a.N / P_BN);
  static std::atomic<int> ncu{0};
  if (!ncu) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    ncu = n;
  }
  hipLaunchKernelGGL(k, dim3(std::min(nt, (int)ncu)), dim3(256), P_LDS, st, a);
  return MTS_OK;
}
