// 256x224x64 bf16 MFMA GEMM, FOUR waves per workgroup (one per SIMD, 512 registers each), operands staged through REGISTERS.
//
// Why a second form of gemm224.hip.  That kernel runs 8 waves of 64 x 112 output each and copies its operands HBM -> LDS by LDS-DMA.
// Per 64-deep K-tile its waves read (64 + 112) rows x 128 B x 8 = 180 KB of fragments out of the LDS and the DMA writes 60 KB into it:
// 240 KB per K-tile against 1792 cycles of matrix-pipe time = 134 B/clk/CU -- MORE than the 128 B/clk the LDS delivers.  The matrix
// cores were busy 51 % of the time (profiles/r02_v2_pmc_sq_counters.txt) because the kernel is LDS-bandwidth-bound, not because of
// any one stall.  The fix is geometry: four waves of 128 x 112 read (128 + 112) x 128 B x 4 = 123 KB per K-tile, + 60 KB of writes =
// 183 KB = 102 B/clk.  A wave of that size needs 224 accumulator registers, i.e. the whole register file of its SIMD (one wave per
// SIMD), and then LDS-DMA is out: with nobody else to issue behind, every global_load_lds (m0 write + hazard wait states + the copy
// queue) stalls the SIMD's only MFMA stream (tools/micro/gemm224w4.hip measured 0.82 PFLOP/s that way).  So the operands go
// HBM -> registers (plain 16-byte loads: a few issue cycles each, 15 per thread and K-tile) -> ds_write_b128 -> LDS, one K-tile
// ahead in registers and one in the second LDS stage.
//
// Schedule of one K-tile (all LDS traffic is inline asm with COUNTED lgkmcnt waits; LDS returns in order):
//   on entry:   B fragments of k-step 0 (7) and A fragments 0..2 of this K-tile are already requested (behind the previous barrier)
//   k-step 0:   for A block i = 0..7: request A fragment i+3 (a ring of 8), [i = 0: request the 7 B fragments of k-step 1],
//               wait for fragment i, 7 MFMAs
//   k-step 1:   same on the second half of K; under blocks 0..2 the NEXT K-tile goes registers -> other LDS stage (15 ds_write_b128,
//               its global loads were issued a K-tile ago); before block 5: all my LDS operations done -> ONE barrier -> request the
//               next K-tile's first fragments, which land under the 21 MFMAs of blocks 5..7; the global loads of the K-tile after
//               next go out between those MFMAs.
// One barrier per K-tile, no LDS latency exposed at the K-tile boundary, accumulation order per output element identical to
// gemm224.hip (k-step 0 then 1 of every K-tile): results are bitwise equal.
//
// Covers what the step launches: bf16 C, A K-major (NT and NN), M % 256 == 0, N % 224 == 0, K % 64 == 0, bias / column scale /
// residual epilogue.  Anything else stays on gemm224.hip (mts_launch_gemm224r returns -1).
#include <algorithm>
#include <type_traits>
#include "gemm_common.h"

#define R_A_BYTES 32768                       // 256 rows x 128 B (K-major) -- NN: the same
#define R_B_BYTES 32768                       // NT: 224 rows x 128 B K-major image (28 KiB used); NN: two strided half images of 16 KiB
#define R_STAGE (R_A_BYTES + R_B_BYTES)
#define R_LDS (2 * R_STAGE)                   // 128 KiB
#define R_BN 224
#define R_HN 112

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ bool r_shape_ok(const GemmArgs& a) {
  const unsigned simple = MTS_EPI_BIAS | MTS_EPI_COLSCALE | MTS_EPI_RESIDUAL;
  return !a.slab && (a.epi & ~simple) == 0 && (a.M % 256 == 0) && (a.N % R_BN == 0) && (a.K % BK == 0) && a.ksplit == a.K && (a.ldc % 8 == 0) &&
         (((uintptr_t)a.C & 15) == 0) && (!(a.epi & MTS_EPI_COLSCALE) || a.ncols_scaled % 4 == 0) &&
         (!(a.epi & MTS_EPI_RESIDUAL) || (a.ldr % 4 == 0 && ((uintptr_t)a.residual & 7) == 0)) &&
         (!(a.epi & MTS_EPI_BIAS) || ((uintptr_t)a.bias & 15) == 0);
}

// (free function templates, not generic lambdas: clang rejects inline-asm operands that name variables captured by a generic lambda)
template <bool B_KMAJOR, int I0, int N, int NGB>
__device__ __forceinline__ void r_lds_write(unsigned sb, unsigned wA, const unsigned (&wB)[B_KMAJOR ? 1 : 4], const u32x4 (&ga)[8], const u32x4 (&gb)[NGB]) {
#pragma unroll
  for (int w = I0; w < I0 + N; ++w) {
    if (w < 8) {
      asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(sb + wA), "v"(ga[w]), "n"(w * 4096) : "memory");
    } else if constexpr (B_KMAJOR) {
      asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(sb + wB[0]), "v"(gb[w - 8]), "n"((w - 8) * 4096) : "memory");
    } else {
      asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(sb + wB[(w - 8) & 3]), "v"(gb[w - 8]), "n"(((w - 8) >> 2) * 16384) : "memory");
    }
  }
}
// global -> registers, loads number I0 .. I0 + N - 1 of the 8 + NGB of one K-tile.  pA / pB: wave-uniform addresses of the K-tile's
// first A row / B row (NT: B rows = output columns; NN: B row = k), strides in bytes.  Scalar base + 32-bit lane offset form.
template <bool B_KMAJOR, int I0, int N, int NGB>
__device__ __forceinline__ void r_gload(u32x4 (&ga)[8], u32x4 (&gb)[NGB], const char* pA, size_t strideA32, const char* pB, size_t strideB32,
                                        unsigned voffA, unsigned voffB, unsigned voffB7) {
#pragma unroll
  for (int w = I0; w < I0 + N; ++w) {
    if (w < 8) {
      const char* sb = pA + (size_t)w * strideA32;
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(ga[w]) : "v"(voffA), "s"(sb));
    } else if constexpr (B_KMAJOR) {
      const char* sb = pB + (size_t)(w - 8) * strideB32;
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(gb[w - 8]) : "v"(voffB), "s"(sb));
    } else {
      // i = 4 h + q: half h (columns 112 h ..), chunk (tid & 3) + 4 q; the last chunk group of a half (q = 3) reaches half-local
      // columns 96 .. 127 of which 96 .. 111 are used.  For h = 0 the surplus is the first columns of half 1 (in the tile); for h = 1
      // it would lie beyond the tile -- beyond the MATRIX in the last tile column -- so those lanes read 16 columns further left
      // (voffB7); what they fetch lands in LDS columns nobody reads
      const int i = w - 8, h = i >> 2, q = i & 3;
      const char* sb = pB + (size_t)(h * R_HN + 32 * q) * 2;
      if (i == 7) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(gb[i]) : "v"(voffB7), "s"(sb));
      else asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(gb[i]) : "v"(voffB), "s"(sb));
    }
  }
}

// A fragment n = 8 ks + i of the K-tile whose A image starts at `base` (incl. the wave's rows) -> ring slot n & 7
template <int N>
__device__ __forceinline__ void r_rd_a(s16x8 (&fa)[8], unsigned base, const unsigned (&lk)[2]) {
  constexpr int ks = N >> 3, i = N & 7;
  const unsigned ad = base + lk[ks];
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[N & 7]) : "v"(ad), "n"(i * 2048));
}
template <bool B_KMAJOR, int KS>
__device__ __forceinline__ void r_rd_b(LFrag<B_KMAJOR> (&fb)[7], unsigned base, const unsigned (&lk)[2], unsigned sx) {
  if constexpr (B_KMAJOR) {
    const unsigned ad = base + lk[KS];
#pragma unroll
    for (int j = 0; j < 7; ++j) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[j].k) : "v"(ad), "n"(j * 2048));
  } else {
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const unsigned ad = base + (sx ^ (j << 5));
      if (KS == 0) {
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fb[j].s.lo) : "v"(ad));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(fb[j].s.hi) : "v"(ad));
      } else {
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8192" : "=v"(fb[j].s.lo) : "v"(ad));
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:9216" : "=v"(fb[j].s.hi) : "v"(ad));
      }
    }
  }
}
// one B fragment (K-major only: one LDS operation) -- the k-step-1 fragments are requested one per block
template <int KS, int J>
__device__ __forceinline__ void r_rd_b1(LFrag<true> (&fb)[7], unsigned base, const unsigned (&lk)[2]) {
  const unsigned ad = base + lk[KS];
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[J].k) : "v"(ad), "n"(J * 2048));
}
// ---- LDS operation schedule of one K-tile (NT) and the counted waits that go with it -------------------------------------------------
// lgkmcnt is a 4-bit counter: with more than 15 LDS operations of a wave in flight "s_waitcnt lgkmcnt(15)" no longer guarantees
// anything (measured: A block 4 of every K-tile read its fragment too early in 3 of 4 waves) -- so the schedule never has more than
// 15 outstanding.  Issue order inside block n (0..11): [b1 fragment n, n < 7] [wcnt(n) writes] [A fragment n + 4]; before block 12:
// everything done, barrier, then b0 x 7 + a0..a3 of the next K-tile (11).  At block 0: 11 + 4 = 15 outstanding at most.
struct RSched {
  static constexpr int wcnt(int n, bool nxt) { return (nxt && n < 12) ? ((n % 4 == 0) ? 2 : 1) : 0; }       // 2 1 1 1 | 2 1 1 1 | 2 1 1 1 = 15
  static constexpr int woff(int n) { return (n / 4) * 5 + (n % 4 == 0 ? 0 : (n % 4) + 1); }
  static constexpr int ops(int m, bool nxt) { return (m < 7 ? 1 : 0) + wcnt(m, nxt) + (m + 4 <= 15 ? 1 : 0); }
  // operations that may still be outstanding when block n's MFMAs start = those issued after A fragment n
  static constexpr int younger(int n, bool nxt) {
    int y = 0;
    if (n < 4) { y = 3 - n; for (int m = 0; m <= n; ++m) y += ops(m, nxt); }
    else { for (int m = n - 3; m <= n; ++m) y += ops(m, nxt); }
    if (n == 8) {                                  // block 8 also needs the last b1 fragment (issued first in block 6)
      const int after_b1 = (ops(6, nxt) - 1) + ops(7, nxt) + ops(8, nxt);
      if (after_b1 < y) y = after_b1;
    }
    return y;
  }
};

// the 7 MFMAs of A fragment n (ring slot n & 7) against the B fragments of its k-step; at i = 0 the B operands are formed
template <bool B_KMAJOR, int N>
__device__ __forceinline__ void r_block(f32x4 (&acc)[8][7], s16x8 (&fa)[8], LFrag<B_KMAJOR> (&fb)[7], bf16x8 (&ob)[7]) {
  constexpr int i = N & 7;
  if constexpr (i == 0) {
#pragma unroll
    for (int j = 0; j < 7; ++j) ob[j] = lfrag_get<B_KMAJOR>(fb[j]);
  }
  asm volatile("" : "+v"(fa[N & 7]));
  const bf16x8 va = __builtin_bit_cast(bf16x8, fa[N & 7]);
#pragma unroll
  for (int j = 0; j < 7; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ob[j], va, acc[i][j], 0, 0, 0);
}

// EXP: timing experiments (results wrong on purpose; gemm_variant 71 / 72 / 73): 1 = no global loads and no LDS writes in the K loop,
// 2 = additionally no fragment reads (matrix pipe + barrier only), 3 = no global loads only
template <int LAYOUT, int EXP = 0>
__global__ __launch_bounds__(256, 1) void gemm_bf16_224r_kernel(const GemmArgs a) {
  static_assert(LAYOUT == MTS_NT || LAYOUT == MTS_NN, "A is K-major in this kernel");
  constexpr bool B_KMAJOR = (LAYOUT == MTS_NT);
  constexpr int cb = B_KMAJOR ? 1 : 2;           // LDS operations per B fragment
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

  // tile order: XCD-aware 4-row bands, as gemm224.hip
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

  // ---- global -> registers: 8 A chunks + 7 B chunks of 16 bytes per thread and K-tile --------------------------------------------
  // K-major operand: thread (row r = tid >> 3 (+ 32 i), 16-byte chunk tid & 7): a wave reads 8 whole 128-byte lines per instruction.
  // The row part of the address is wave-uniform per i (scalar base), the thread part a 32-bit offset that never changes.
  const int ra = tid >> 3, ch = tid & 7;
  const unsigned voffA = (unsigned)(ra * a.lda + ch * 8) * 2u;
  const char* baseA = reinterpret_cast<const char*>(A + (size_t)bm0 * a.lda);
  // NT: B rows = output columns, same shape of access.  NN: B is [K][N]: thread (k-row kr = tid >> 2 (0..63), chunk group tid & 3);
  // chunk c8 = (tid & 3) + 4 i, i < 4, covers columns 8 c8 .. 8 c8 + 7 of each 128-column half (112 used): 8 loads, one of them
  // mostly padding columns -- the price of addresses that are affine in i.
  unsigned voffB;
  const char* baseB;
  if constexpr (B_KMAJOR) {
    voffB = (unsigned)(ra * a.ldb + ch * 8) * 2u;
    baseB = reinterpret_cast<const char*>(B + (size_t)bn0 * a.ldb);
  } else {
    voffB = (unsigned)((tid >> 2) * a.ldb + (tid & 3) * 8) * 2u;
    baseB = reinterpret_cast<const char*>(B + bn0);
  }
  const unsigned voffB7 = voffB - (((tid & 3) >= 2) ? 32u : 0u);
  constexpr int NGB = B_KMAJOR ? 7 : 8;
  u32x4 ga[2][8], gb[2][NGB];          // two K-tiles in flight in registers (set = K-tile parity)
  // The loads are inline asm in the scalar-base + 32-bit-lane-offset form: one VGPR of address per operand for the whole kernel (as
  // plain C++ loads the compiler kept fifteen 64-bit lane pointers per register set and spilled fragment registers around them).
  // Being asm they are invisible to the compiler's vmcnt bookkeeping: the K loop waits for them itself (R_VMWAIT below), and NOTHING
  // may touch ga / gb between a load and that wait -- the build must show zero spills (tests check results, the Makefile check greps).

  // ---- registers -> LDS -------------------------------------------------------------------------------------------------------------
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned wA = (unsigned)kmajor_off(ra, ch);                  // + 4096 i (rows 32 i further: the swizzle key repeats every 16 rows)
  unsigned wB[B_KMAJOR ? 1 : 4];
  if constexpr (B_KMAJOR) wB[0] = R_A_BYTES + (unsigned)kmajor_off(ra, ch);
  else {
#pragma unroll
    for (int q = 0; q < 4; ++q) wB[q] = R_A_BYTES + (unsigned)strided_off(tid >> 2, 32 * q + 8 * (tid & 3));   // + 16384 h
  }

  // ---- LDS -> fragments ---------------------------------------------------------------------------------------------------------------
  unsigned lk[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) lk[ks] = r16 * 128 + (((ks * 4 + g) ^ ((r16 >> 1) & 7)) << 4);
  const int q_ = r16 >> 2, p_ = r16 & 3;
  const unsigned sx = (8 * g + q_) * 256 + 8 * p_ + ((((8 * g + q_) & 3) | ((g & 1) << 2)) << 5);    // strided_off(8g + q, 4p), column block 0
  const unsigned fA = lds0 + wm * 16384;                               // + stage * R_STAGE + lk[ks] + 2048 i
  const unsigned fB = lds0 + R_A_BYTES + (B_KMAJOR ? wn * (R_HN * 128) : wn * 16384);

  s16x8 fa[8];                        // ring of A fragments: fragment n of the K-tile (n = 8 ks + i) lives in slot n & 7
  LFrag<B_KMAJOR> fb[2][7];           // B fragments of k-step ks

  f32x4 acc[8][7];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 7; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  bf16x8 ob[7];
  constexpr int NW = 8 + NGB;                                         // LDS writes (= global loads) per thread and K-tile
  using S0 = std::integral_constant<int, 0>; using S1 = std::integral_constant<int, 1>;
#define LDSW(STAGE, SET, I0_, N_) do { if constexpr (EXP != 1 && EXP != 2) r_lds_write<B_KMAJOR, I0_, N_, NGB>(lds0 + (STAGE) * R_STAGE, wA, wB, ga[SET], gb[SET]); } while (0)
#define GLOAD(KT, SET, I0_, N_)                                                                                                             \
  r_gload<B_KMAJOR, I0_, N_, NGB>(ga[SET], gb[SET], baseA + (size_t)(KT) * BK * 2, (size_t)32 * a.lda * 2,                                    \
                                  B_KMAJOR ? baseB + (size_t)(KT) * BK * 2 : baseB + (size_t)(KT) * BK * a.ldb * 2, (size_t)32 * a.ldb * 2, \
                                  voffA, voffB, voffB7)
#define RDA(STAGE, N_) do { if constexpr (EXP != 2) r_rd_a<N_>(fa, fA + (STAGE) * R_STAGE, lk); } while (0)
#define RDB(STAGE, KS_) do { if constexpr (EXP != 2) r_rd_b<B_KMAJOR, KS_>(fb[KS_], fB + (STAGE) * R_STAGE, lk, sx); } while (0)
// (sched_barrier: the MFMA builtins are not volatile -- without a fence the scheduler sinks them below the counted waits of later
// blocks and bunches the LDS writes together with their vmcnt waits)
#define BLOCK(N_) do { __builtin_amdgcn_sched_barrier(0); r_block<B_KMAJOR, N_>(acc, fa, fb[(N_) >> 3], ob); __builtin_amdgcn_sched_barrier(0); } while (0)
  // ---- prologue: K-tile 0 -> stage 0; K-tiles 1, 2 -> registers ------------------------------------------------------------------------
  GLOAD(0, 0, 0, NW);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  r_lds_write<B_KMAJOR, 0, NW, NGB>(lds0, wA, wB, ga[0], gb[0]);
  GLOAD(1, 1, 0, NW);                              // nk >= 4 (launcher)
  GLOAD(2, 0, 0, NW);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  RDB(0, 0);
  RDA(0, 0); RDA(0, 1); RDA(0, 2); RDA(0, 3);

  // One K-tile.  NXT: a next K-tile exists (registers -> LDS, the barrier, its first fragment requests).  PAR = kt & 1 (compile time:
  // which register set holds K-tile kt + 1, into which K-tile kt + 3 is then loaded).
  // LDS operations are issued per block n in the order [a(n+4)] [b1 x 7cb at n = 0] [W(n) writes]; WPB = writes per block 0..11.
  // The wait before block n's MFMAs allows as many outstanding operations as were issued AFTER a(n) -- computed below at compile time.
  // LD: issue the loads of K-tile kt + 3 (false near the end of K: a load nobody consumes would still land -- in registers the
  // compiler has long given to fragments).  VMW: loads that may still be outstanding once K-tile kt + 1 has landed (NW, or 0 when
  // nothing was issued behind it).
  auto ktile = [&](int kt, auto NXT, auto PAR, auto LD, auto VMW) {
    constexpr bool nxt = decltype(NXT)::value;
    constexpr int par = decltype(PAR)::value, oth = par ^ 1;
    const int s = kt & 1;
    if constexpr (nxt && (EXP == 0 || EXP == 4)) {
      // K-tile kt + 1 has landed in its registers: everything but the loads of K-tile kt + 2 issued behind it (in-order counter)
      if constexpr (EXP == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(decltype(VMW)::value) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_setprio(1);
    constexpr bool ld = nxt && decltype(LD)::value && (EXP == 0 || EXP == 4);
    const int kl = kt + 3;
    // global load number n - 1 of K-tile kt + 3 goes out under block n: ONE per block (four in a row fill the memory pipeline's
    // queue and the wave -- alone on its SIMD -- stalls at the issue), each behind the LDS write that read its registers
    // (write w is issued in block <= w)
#define LOAD1(n) if constexpr (ld && (n) >= 1 && (n) <= NW) GLOAD(kl, oth, ((n) >= 1 ? (n) - 1 : 0), 1);
#define STEP(n)                                                                                                       \
    if constexpr (B_KMAJOR && (n) < 7) { if constexpr (EXP != 2) r_rd_b1<1, ((n) < 7 ? (n) : 0)>(fb[1], fB + s * R_STAGE, lk); }  \
    if constexpr (RSched::wcnt(n, nxt) > 0) LDSW(s ^ 1, oth, RSched::woff(n), RSched::wcnt(n, nxt));                  \
    if constexpr ((n) + 4 <= 15) RDA(s, ((n) + 4 <= 15 ? (n) + 4 : 15));                                              \
    lgkm_wait<RSched::younger(n, nxt)>();                                                                             \
    BLOCK(n);                                                                                                         \
    LOAD1(n)
    STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7) STEP(8) STEP(9) STEP(10) STEP(11)
    // before block 12: everything of mine on the LDS is done (a12..a15 were requested at blocks 8..11, the last write at block 11)
    // -> barrier -> the next K-tile's first fragments land under the 28 MFMAs of blocks 12..15; the global loads of K-tile kt + 3 go
    // out between them, into the registers the writes have just read
    lgkm_wait<0>();
    if constexpr (nxt) {
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_s_setprio(1);
      RDB(s ^ 1, 0);
      RDA(s ^ 1, 0); RDA(s ^ 1, 1); RDA(s ^ 1, 2); RDA(s ^ 1, 3);
    }
    BLOCK(12);
    LOAD1(12)
    BLOCK(13);
    LOAD1(13)
    BLOCK(14);
    LOAD1(14)
    BLOCK(15);
    LOAD1(15)
    __builtin_amdgcn_s_setprio(0);
#undef STEP
#undef LOAD1
  };
  {   // nk is even (checked by the launcher): pairs of K-tiles, register sets by parity; the last two pairs are peeled because the
      // loads stop there (no K-tile kt + 3) and with them the number of loads behind the one being waited for
    using T = std::true_type; using F = std::false_type;
    using VN = std::integral_constant<int, NW>; using V0 = std::integral_constant<int, 0>;
    {                                             // nk >= 4 (launcher)
      int kt = 0;
#pragma clang loop unroll(disable)
      for (; kt + 4 < nk; kt += 2) { ktile(kt, T{}, S0{}, T{}, VN{}); ktile(kt + 1, T{}, S1{}, T{}, VN{}); }
      ktile(kt, T{}, S0{}, T{}, VN{});             // kt = nk - 4: loads K-tile nk - 1
      ktile(kt + 1, T{}, S1{}, F{}, VN{});         // behind K-tile nk - 2: the loads of nk - 1
      ktile(kt + 2, T{}, S0{}, F{}, V0{});         // K-tile nk - 1 is the last load in flight
      ktile(kt + 3, F{}, S1{}, F{}, V0{});
    }
  }
#undef LDSW
#undef GLOAD
#undef RDA
#undef RDB
#undef BLOCK

  // ---- epilogue: bias / column scale / residual, bf16 C through a wave-private LDS staging area -------------------------------------
  const int m0 = bm0 + wm * 128, n0 = bn0 + wn * R_HN;
  __builtin_amdgcn_s_barrier();                    // every wave is out of the K loop: the stages may serve as store staging
  char* stage = smem + wave_u * 4096;              // wave-private 16 rows x 240 B
  const bool has_bias = (a.epi & MTS_EPI_BIAS) != 0, has_res = (a.epi & MTS_EPI_RESIDUAL) != 0;
  // every global load of the epilogue ahead of every store (in-order counter: a load behind a store is waited for with it); without a
  // bias / residual the loads still run, from the start of A (always mapped), and a select discards them: no branch, counted waits
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
      const int row = idx / 14, chn = idx - row * 14;
      if (idx < 16 * 14) {
        const uint4 val = *reinterpret_cast<const uint4*>(stage + row * 240 + chn * 16);
        *reinterpret_cast<uint4*>(C + (size_t)(m0 + i * 16 + row) * a.ldc + n0 + chn * 8) = val;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

// =====================================================================================================================================
// gemm_variant 8: the same four waves and the same operand paths with the LOOP STRUCTURE of the vendor library's kernel for these shapes
// (DESIGN.md section 8): ONE LDS stage; every fragment of a K-tile in registers (2 x 8 A + 2 x 7 B); per K-tile
//   blocks 0..7  (k-step 0): the 15 fragments of k-step 1 are requested two per block;  -> all my reads done, BARRIER 1
//   blocks 8..12 (k-step 1): the next K-tile goes registers -> LDS, three ds_write_b128 per block, each followed by the global load that
//                            refills its register with the K-tile after next (ONE register set: a load is in flight for one K-tile);
//                            -> all my writes done, BARRIER 2
//   blocks 13..15:           the 15 fragments of the next K-tile's k-step 0 are requested, five per block.
// Never more than 15 LDS operations of a wave in flight (4-bit lgkmcnt).  Accumulation order as gemm224.hip: bitwise equal results.
// NT only.
// =====================================================================================================================================
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

__global__ __launch_bounds__(256, 1) void gemm_bf16_224v_kernel(const GemmArgs a) {
  constexpr bool B_KMAJOR = true;
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
  const int ra = tid >> 3, ch = tid & 7;
  const unsigned voffA = (unsigned)(ra * a.lda + ch * 8) * 2u;
  const char* baseA = reinterpret_cast<const char*>(A + (size_t)bm0 * a.lda);
  const unsigned voffB = (unsigned)(ra * a.ldb + ch * 8) * 2u;
  const char* baseB = reinterpret_cast<const char*>(B + (size_t)bn0 * a.ldb);
  constexpr int NGB = 7, NW = 15;
  u32x4 ga[8], gb[NGB];               // ONE register set: K-tile kt + 1 while K-tile kt is multiplied; refilled write by write
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned wA = (unsigned)kmajor_off(ra, ch);
  unsigned wB[1];
  wB[0] = R_A_BYTES + (unsigned)kmajor_off(ra, ch);
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

#define VGLOAD(KT, I0_, N_) r_gload<true, I0_, N_, NGB>(ga, gb, baseA + (size_t)(KT) * BK * 2, (size_t)32 * a.lda * 2, baseB + (size_t)(KT) * BK * 2, (size_t)32 * a.ldb * 2, voffA, voffB, voffB)
#define VBLOCK(N_) do { __builtin_amdgcn_sched_barrier(0); v_block<N_>(acc, fa, fb); __builtin_amdgcn_sched_barrier(0); } while (0)
#define VRD(KS_, R_) v_rd<KS_, R_>(fa, fb, fA, fB, lk)
  // ---- prologue: K-tile 0 -> LDS, K-tile 1 -> registers, fragments of K-tile 0 / k-step 0 requested ------------------------------------
  VGLOAD(0, 0, NW);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  r_lds_write<true, 0, NW, NGB>(lds0, wA, wB, ga, gb);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the writes have read their registers
  VGLOAD(1, 0, NW);                                       // nk >= 3 (launcher)
  __builtin_amdgcn_s_barrier();
  VRD(0, 0); VRD(0, 1); VRD(0, 2); VRD(0, 3); VRD(0, 4); VRD(0, 5); VRD(0, 6); VRD(0, 7); VRD(0, 8); VRD(0, 9); VRD(0, 10); VRD(0, 11); VRD(0, 12);
  VRD(0, 13); VRD(0, 14);

  // One K-tile.  NXT: K-tile kt + 1 exists (its LDS writes, barrier 2, its first fragment requests); LD: K-tile kt + 2 exists (the refills).
  auto ktile = [&](int kt, auto NXT, auto LD) {
    constexpr bool nxt = decltype(NXT)::value, ld = decltype(LD)::value;
    // k-step 0.  On entry 15 requests are (at most) in flight: B0 x 7, then A0[0..7].  Block n needs B0 and A0[n]: allowed outstanding =
    // the 7 - n younger A0 requests + the k-step-1 requests issued so far (two per block)
#define KA(n)                                                               \
    lgkm_wait<(7 - (n)) + 2 * (n)>();                                       \
    VBLOCK(n);                                                              \
    if constexpr (2 * (n) < 15) VRD(1, (2 * (n) < 15 ? 2 * (n) : 0));       \
    if constexpr (2 * (n) + 1 < 15) VRD(1, (2 * (n) + 1 < 15 ? 2 * (n) + 1 : 0));
    KA(0) KA(1) KA(2) KA(3) KA(4) KA(5) KA(6) KA(7)
#undef KA
    lgkm_wait<0>();                                         // every fragment of this K-tile is in my registers
    if constexpr (nxt) __builtin_amdgcn_s_barrier();        // ... and in everybody's: the stage may be overwritten
    // k-step 1: blocks 8..12 carry the next K-tile's 15 writes (3 per block), each write behind the load that filled its register
    // (in-order vector-memory counter: with refills running, 14 younger loads are allowed; at the end of K only the rest of this set)
#define KW(w)                                                                                                              \
    if constexpr (nxt) {                                                                                                   \
      if constexpr (ld) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");                                                  \
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(14 - (w)) : "memory");                                                 \
      r_lds_write<true, (w), 1, NGB>(lds0, wA, wB, ga, gb);                                                                \
      if constexpr (ld) VGLOAD(kt + 2, (w), 1);                                                                            \
    }
    VBLOCK(8);  KW(0) KW(1) KW(2)
    VBLOCK(9);  KW(3) KW(4) KW(5)
    VBLOCK(10); KW(6) KW(7) KW(8)
    VBLOCK(11); KW(9) KW(10) KW(11)
    VBLOCK(12); KW(12) KW(13) KW(14)
#undef KW
    if constexpr (nxt) {
      lgkm_wait<0>();                                       // my writes are in the LDS
      __builtin_amdgcn_s_barrier();
    }
    VBLOCK(13);
    if constexpr (nxt) { VRD(0, 0); VRD(0, 1); VRD(0, 2); VRD(0, 3); VRD(0, 4); }
    VBLOCK(14);
    if constexpr (nxt) { VRD(0, 5); VRD(0, 6); VRD(0, 7); VRD(0, 8); VRD(0, 9); }
    VBLOCK(15);
    if constexpr (nxt) { VRD(0, 10); VRD(0, 11); VRD(0, 12); VRD(0, 13); VRD(0, 14); }
  };
  {
    using T = std::true_type; using F = std::false_type;
    int kt = 0;
#pragma clang loop unroll(disable)
    for (; kt + 2 < nk; ++kt) ktile(kt, T{}, T{});
    ktile(kt, T{}, F{});                                    // kt = nk - 2: writes the last K-tile, no refills
    ktile(kt + 1, F{}, F{});
  }
#undef VGLOAD
#undef VBLOCK
#undef VRD

  // ---- epilogue (as gemm_bf16_224r_kernel) ----------------------------------------------------------------------------------------------
  const int m0 = bm0 + wm * 128, n0 = bn0 + wn * R_HN;
  __builtin_amdgcn_s_barrier();
  char* stage = smem + wave_u * 4096;
  const bool has_bias = (a.epi & MTS_EPI_BIAS) != 0, has_res = (a.epi & MTS_EPI_RESIDUAL) != 0;
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
      const int row = idx / 14, chn = idx - row * 14;
      if (idx < 16 * 14) {
        const uint4 val = *reinterpret_cast<const uint4*>(stage + row * 240 + chn * 16);
        *reinterpret_cast<uint4*>(C + (size_t)(m0 + i * 16 + row) * a.ldc + n0 + chn * 8) = val;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
}

__global__ __launch_bounds__(256, 1) void gemm_bf16_224d_kernel(const GemmArgs a) {
  constexpr bool B_KMAJOR = true;
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
#define KC(c) if constexpr (ld) CP(c, kt + 2, st);
    VBLOCK(8);  KC(0) KC(1)
    VBLOCK(9);  KC(2) KC(3)
    VBLOCK(10); KC(4) KC(5)
    VBLOCK(11); KC(6) KC(7)
    VBLOCK(12); KC(8) KC(9)
    if constexpr (nxt) {
      // K-tile kt + 1 (copied during K-tile kt - 1, or in the prologue) has landed: only this K-tile's ten copies may still fly
      if constexpr (ld) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    VBLOCK(13); KC(10) KC(11)
    if constexpr (nxt) { VRD(st ^ 1, 0, 0); VRD(st ^ 1, 0, 1); VRD(st ^ 1, 0, 2); VRD(st ^ 1, 0, 3); VRD(st ^ 1, 0, 4); }
    VBLOCK(14); KC(12) KC(13)
    if constexpr (nxt) { VRD(st ^ 1, 0, 5); VRD(st ^ 1, 0, 6); VRD(st ^ 1, 0, 7); VRD(st ^ 1, 0, 8); VRD(st ^ 1, 0, 9); }
    VBLOCK(15); KC(14)
    if constexpr (nxt) { VRD(st ^ 1, 0, 10); VRD(st ^ 1, 0, 11); VRD(st ^ 1, 0, 12); VRD(st ^ 1, 0, 13); VRD(st ^ 1, 0, 14); }
#undef KC
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

  // ---- epilogue (as gemm_bf16_224r_kernel) ----------------------------------------------------------------------------------------------
  __builtin_amdgcn_s_barrier();
  char* stage = smem + wave_u * 4096;
  const size_t res_ld = has_res ? (size_t)a.ldr : 0;
  const bf16_t* res_p = has_res ? reinterpret_cast<const bf16_t*>(a.residual) + (size_t)(m0 + r16) * a.ldr + n0 + 4 * g
                                : reinterpret_cast<const bf16_t*>(a.A) + 4 * g;
  // the bias is wave-uniform work of 7 loads; the residual is 56 loads per lane and a wait of a full memory round trip per tile: only where the
  // epilogue has one (the forward Q|K|V projection has not: six tiles per CU there)
  uint2 res[8][7];
  if (has_res) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 7; ++j) res[i][j] = *reinterpret_cast<const uint2*>(res_p + (size_t)(i * 16) * res_ld + j * 16);
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 7; ++j) res[i][j] = make_uint2(0u, 0u);
  }
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

static int v_launch(const GemmArgs& a, hipStream_t st) {
  auto k = gemm_bf16_224v_kernel;
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, R_STAGE);
    if (e != hipSuccess) { mts_set_error("gemm224v: hipFuncSetAttribute: %s", hipGetErrorString(e)); return MTS_ERR_LAUNCH; }
    attr_set = true;
  }
  const int nt = (a.M / 256) * (a.N / R_BN);
  hipLaunchKernelGGL(k, dim3(nt), dim3(256), R_STAGE, st, a);
  return MTS_OK;
}

template <int LAYOUT, int EXP = 0>
static int r_launch(const GemmArgs& a, hipStream_t st) {
  auto k = gemm_bf16_224r_kernel<LAYOUT, EXP>;
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, R_LDS);
    if (e != hipSuccess) { mts_set_error("gemm224r: hipFuncSetAttribute: %s", hipGetErrorString(e)); return MTS_ERR_LAUNCH; }
    attr_set = true;
  }
  const int nt = (a.M / 256) * (a.N / R_BN);
  hipLaunchKernelGGL(k, dim3(nt), dim3(256), R_LDS, st, a);
  return MTS_OK;
}

// called from mts_launch_gemm224 (gemm224.hip); -1: shape / epilogue not covered here
int mts_launch_gemm224r(const GemmArgs& a, int layout, bool c_is_f32, int splits, hipStream_t st) {
  if (c_is_f32 || splits != 1 || layout != MTS_NT) return -1;      // (NN: its B fragments are two LDS operations each -- needs its own schedule)
  const unsigned simple = MTS_EPI_BIAS | MTS_EPI_COLSCALE | MTS_EPI_RESIDUAL;
  const bool ok = !a.slab && (a.epi & ~simple) == 0 && (a.M % 256 == 0) && (a.N % R_BN == 0) && (a.K % BK == 0) && a.K % (2 * BK) == 0 && a.K >= 4 * BK && a.ksplit == a.K &&
                  (a.ldc % 8 == 0) && (((uintptr_t)a.C & 15) == 0) && (a.lda % 8 == 0) && (a.ldb % 8 == 0) &&
                  (!(a.epi & MTS_EPI_COLSCALE) || a.ncols_scaled % 4 == 0) &&
                  (!(a.epi & MTS_EPI_RESIDUAL) || (a.ldr % 4 == 0 && ((uintptr_t)a.residual & 7) == 0)) &&
                  (!(a.epi & MTS_EPI_BIAS) || ((uintptr_t)a.bias & 15) == 0) &&
                  (layout == MTS_NT || a.N - 0 >= 32);
  if (!ok) return -1;
  if (a.variant == 8) return v_launch(a, st);
  if (a.variant == 9) return d_launch(a, st);
  if (a.variant == 71) return layout == MTS_NT ? r_launch<MTS_NT, 1>(a, st) : r_launch<MTS_NN, 1>(a, st);
  if (a.variant == 72) return layout == MTS_NT ? r_launch<MTS_NT, 2>(a, st) : r_launch<MTS_NN, 2>(a, st);
  if (a.variant == 73) return layout == MTS_NT ? r_launch<MTS_NT, 3>(a, st) : r_launch<MTS_NN, 3>(a, st);
  if (a.variant == 74) return layout == MTS_NT ? r_launch<MTS_NT, 4>(a, st) : r_launch<MTS_NN, 4>(a, st);
  return layout == MTS_NT ? r_launch<MTS_NT>(a, st) : r_launch<MTS_NN>(a, st);
}
