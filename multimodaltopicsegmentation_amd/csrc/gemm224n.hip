// 256x224x64 bf16 MFMA GEMM for the DATA GRADIENTS (layout NN: A K-major, B k-strided, bf16 C), four waves per workgroup.
//
//   dX[M, N] (bf16) = dY[M, K] W[K, N] (+ residual)      (the backward of modeling_longformer.py:504-506,1069 and of every nn.Linear of the
//   taggers: d(input) of the Q|K|V projection, K = 3 D, and of the attention output projection, K = D)
//
// The loop of gemm_bf16_224d_kernel (gemm224r.hip: four waves of 128 x 112, buffer-load LDS-DMA, two 64-KiB stages, every fragment of a
// K-tile in registers, two barriers per K-tile, copies of K-tile kt + 2 two per block behind barrier 1) with the B side of
// gemm_bf16_224t_kernel (gemm224t.hip): W is k-strided, so its LDS image is k-row-major -- per k-step two half images [32 k][128 column
// slots, 112 used], copied as 1-KiB pieces of 4 k-rows x 256 B with the swizzle on the source side -- and a B fragment is two transposing
// reads.  A K-tile therefore costs a wave 8 + 8 copies and 2 x (14 + 8) = 44 LDS reads against 15 and 30 in the NT kernel; the reads of
// k-step 1 go out three per block during the blocks of k-step 0, those of the next K-tile's k-step 0 in the last four blocks behind
// barrier 2, every group behind a counted wait that keeps at most 15 LDS operations of the wave in flight (the counter has 4 bits).
//
// Per output element the products are accumulated in ascending k, 32 at a time, as in every other bf16 kernel of this library: results are
// bitwise those of gemm_bf16_224_kernel<NN> (tests/test_gpu_kernels.py::test_gemm_224n_matches_the_eight_wave_kernel_bitwise).
#include <algorithm>
#include <type_traits>
#include "gemm_common.h"

#define N_A_BYTES 32768                       // 256 rows x 128 B (K-major)
#define N_B_BYTES 32768                       // 2 k-steps x 2 halves x [32 k][256 B]
#define N_STAGE (N_A_BYTES + N_B_BYTES)
#define N_EPI (2 * N_STAGE)                   // the epilogue's output staging: 4 x 4 KiB behind the stages (the stages take the residual tile)
#define N_LDS (2 * N_STAGE + 16384)           // 144 KiB
#define N_BN 224
#define N_HN 112

typedef __attribute__((address_space(3))) void n_dlptr;

struct NFrag { s16x4 lo, hi; };

// (free function templates, not generic lambdas: clang rejects inline-asm operands that name variables captured by a generic lambda)
// read operation R (0..21) of k-step KS: 2 j, 2 j + 1 = the halves of B fragment j; 14 + i = A fragment i.  sA / sB: stage base (+ the wave's
// A rows) and stage base + B region + k-step + the wave's half image
template <int KS, int R>
__device__ __forceinline__ void n_rd(s16x8 (&fa)[2][8], NFrag (&fb)[2][7], unsigned sA, unsigned sB, const unsigned (&lk)[2], const unsigned (&ax)[7]) {
  if constexpr (R < 14) {
    constexpr int j = R >> 1;
    const unsigned ad = ax[j] + sB;
    if constexpr (R & 1) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fb[KS][j].hi) : "v"(ad), "n"(N_A_BYTES + KS * 16384 + 1024));
    else asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(fb[KS][j].lo) : "v"(ad), "n"(N_A_BYTES + KS * 16384));
  } else {
    const unsigned ad = sA + lk[KS];
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[KS][R - 14]) : "v"(ad), "n"((R - 14) * 2048));
  }
}
// block N of a K-tile: A fragment N & 7 of k-step N >> 3 x the 7 B fragments (joined once per k-step, in its first block)
template <int N>
__device__ __forceinline__ void n_block(f32x4 (&acc)[8][7], s16x8 (&fa)[2][8], NFrag (&fb)[2][7], bf16x8 (&ob)[7]) {
  constexpr int ks = N >> 3, i = N & 7;
  if constexpr (i == 0) {
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      asm volatile("" : "+v"(fb[ks][j].lo), "+v"(fb[ks][j].hi));
      ob[j] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(fb[ks][j].lo, fb[ks][j].hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
  }
  asm volatile("" : "+v"(fa[ks][i]));
  const bf16x8 va = __builtin_bit_cast(bf16x8, fa[ks][i]);
#pragma unroll
  for (int j = 0; j < 7; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ob[j], va, acc[i][j], 0, 0, 0);
}

__global__ __launch_bounds__(256, 1) void gemm_bf16_224n_kernel(const GemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave_u >> 1, wn = wave_u & 1;
  const int r16 = lane & 15, g = lane >> 4, q = r16 >> 2, p = r16 & 3;
  const int ntn = a.N / N_BN, ntm = a.M / 256, nt = ntn * ntm;
  const int nk = a.K / BK;
  const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(a.A);
  const bf16_t* __restrict__ B = reinterpret_cast<const bf16_t*>(a.B);
  int bm0, bn0;
  {
    const int t = blockIdx.x;
    const int qq = nt >> 3, rr = nt & 7, xcd = t & 7, idx = t >> 3;
    const int id = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    if (a.order == 0) { bm0 = (id / ntn) * 256; bn0 = (id % ntn) * N_BN; }
    else {
      const int band = id / (4 * ntn), within = id - band * 4 * ntn;
      const int rows = min(4, ntm - band * 4);
      bm0 = (band * 4 + within % rows) * 256;
      bn0 = (within / rows) * N_BN;
    }
  }
  // ---- copies.  A: 32 pieces of 1 KiB (8 rows x 128 B) per K-tile, 8 per wave, as in gemm_bf16_224d_kernel (lane (row8, pos) fetches chunk
  // pos ^ key(row) of its row: one of two lane offsets by the parity of the piece).  B: per k-step two half images, each 8 pieces of 4 k-rows x
  // 256 B; lane (lr = lane >> 4, c16 = lane & 15) fills k-row lr, 32-B slot c16 >> 1, half c16 & 1, and slot s of k-row r holds source column
  // block s ^ key(r), key(r) = (r & 3) | (((r >> 3) & 1) << 2) (gemm_common.h strided_off): r = 4 piece + lr, so the key is lr | (bit 1 of the
  // piece number << 2) -- two lane offsets.  A half image holds 112 used columns of 128: the lanes of the unused 16 fetch 16 columns further
  // left (nobody reads what they bring).  Wave w copies pieces 4 (w & 1) .. + 3 of half image w >> 1 of both k-steps: 8 copies.
  const int row8 = lane >> 3, pos = lane & 7;
  const unsigned keyE = (unsigned)((row8 >> 1) & 7), keyO = (unsigned)(((row8 >> 1) + 4) & 7);
  const unsigned voA[2] = {(unsigned)(row8 * a.lda + (int)((pos ^ keyE) * 8)) * 2u, (unsigned)(row8 * a.lda + (int)((pos ^ keyO) * 8)) * 2u};
  const int lr = lane >> 4, c16 = lane & 15;
  unsigned voB[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const int key = lr | (par << 2);
    const int col = (((c16 >> 1) ^ key) << 4) + ((c16 & 1) << 3);
    voB[par] = (unsigned)(lr * a.ldb + (col >= N_HN ? col - 16 : col)) * 2u;
  }
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)bm0 * a.lda), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(B + bn0), 0, 0x7ffffff0, 0x00020000);
  const int rowsA = 16 * a.lda;                                    // bytes per 8 rows of A
  const int h_w = wave_u >> 1, pi0 = (wave_u & 1) * 4;
  const int sB0 = (4 * pi0 * a.ldb + h_w * N_HN) * 2;
  const int rowB4 = 8 * a.ldb, stepB = 64 * a.ldb, tileB = 128 * a.ldb;   // bytes per 4 / 32 / 64 k-rows of B
  // copy number c (0..15) of this wave for K-tile kt into stage st: 0..7 = A piece 8 wave + c; 8..15 = B: k-step (c - 8) >> 2, piece pi0 + (c & 3)
  auto copy1 = [&](auto C, int kt, int st) {
    constexpr int c = decltype(C)::value;
    char* dst0 = smem + st * N_STAGE;
    if constexpr (c < 8) {
      const int piece = wave_u * 8 + c;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (n_dlptr*)(dst0 + piece * 1024), 16, voA[c & 1], piece * rowsA + kt * (BK * 2), 0, 0);
    } else {
      constexpr int ks = (c - 8) >> 2, cc = c & 3;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (n_dlptr*)(dst0 + N_A_BYTES + ks * 16384 + h_w * 8192 + (pi0 + cc) * 1024), 16, voB[(cc >> 1) & 1],
                                               sB0 + cc * rowB4 + ks * stepB + kt * tileB, 0, 0);
    }
  };
  // ---- fragment reads.  A: row r16 of the 16-row block, 16-byte chunk ks * 4 + g (K-major swizzle).  B: X[k = 8 g + q (+ 4), column block j,
  // columns 4 p ..] through two transposing reads (gemm_common.h frag_strided)
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  unsigned lk[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) lk[ks] = r16 * 128 + (((ks * 4 + g) ^ ((r16 >> 1) & 7)) << 4);
  const unsigned sx = (unsigned)((8 * g + q) * 256 + 8 * p + ((q | ((g & 1) << 2)) << 5));
  unsigned ax[7];
#pragma unroll
  for (int c = 0; c < 7; ++c) ax[c] = sx ^ (unsigned)(c << 5);
  const unsigned fA = lds0 + wm * 16384;
  const unsigned fB = lds0 + wn * 8192;

  s16x8 fa[2][8];
  NFrag fb[2][7];
  bf16x8 ob[7];
  f32x4 acc[8][7];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 7; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

#define NBLOCK(N_) do { __builtin_amdgcn_sched_barrier(0); n_block<N_>(acc, fa, fb, ob); __builtin_amdgcn_sched_barrier(0); } while (0)
#define NRD(ST_, KS_, R_) n_rd<KS_, R_>(fa, fb, fA + (ST_) * N_STAGE, fB + (ST_) * N_STAGE, lk, ax)
#define CP(C_, KT_, ST_) copy1(std::integral_constant<int, C_>{}, KT_, ST_)
#define CP16(KT_, ST_) CP(0, KT_, ST_); CP(1, KT_, ST_); CP(2, KT_, ST_); CP(3, KT_, ST_); CP(4, KT_, ST_); CP(5, KT_, ST_); CP(6, KT_, ST_); CP(7, KT_, ST_); \
                       CP(8, KT_, ST_); CP(9, KT_, ST_); CP(10, KT_, ST_); CP(11, KT_, ST_); CP(12, KT_, ST_); CP(13, KT_, ST_); CP(14, KT_, ST_); CP(15, KT_, ST_)
  // the epilogue's bias (7 x 16 bytes per lane) is fetched ahead of every copy (a data gradient rarely has one)
  const int m0 = bm0 + wm * 128, n0 = bn0 + wn * N_HN;
  const bool has_bias = (a.epi & MTS_EPI_BIAS) != 0, has_res = (a.epi & MTS_EPI_RESIDUAL) != 0;
  const float* bias_p = has_bias ? a.bias + n0 + 4 * g : reinterpret_cast<const float*>(a.A) + 4 * g;
  float4 bias[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) bias[j] = *reinterpret_cast<const float4*>(bias_p + j * 16);
  // ---- the residual tile goes through the LDS.  Fetched into registers in the epilogue it is 56 loads per lane and a memory round trip per tile
  // with nothing to hide it (one tile per workgroup): 278 against 246 us on the Q|K|V data gradient.  The two stages are dead a K-tile apart at the
  // end of the K loop: the wave's rows 0..63 (x 112 columns = 14 chunks of 16 B per row) are copied into the stage of K-tile nk - 2 behind that
  // K-tile's barrier 2, rows 64..127 into the stage of K-tile nk - 1 behind its last fragment read -- 16 LDS-DMA each, four rows per copy (lane =
  // (row in four, chunk): 56 of 64 lanes), 16 KiB per wave and half; the epilogue reads 8 bytes per (row, 16-column block) back.
  const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc(
      has_res ? (void*)(reinterpret_cast<const bf16_t*>(a.residual) + (size_t)m0 * a.ldr + n0) : (void*)A, 0, 0x7ffffff0, 0x00020000);
  const unsigned voR = (unsigned)((lane / 14) * a.ldr + (lane % 14) * 8) * 2u;
  const int rowR4 = 8 * a.ldr;                                     // bytes per 4 rows of the residual
  // copy c (0..15) of half h: rows 64 h + 4 c .. + 3 -> stage st, 16 KiB of this wave
  auto res_copy = [&](int c, int h, int st) {
    if (lane < 56)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsR, (n_dlptr*)(smem + st * N_STAGE + wave_u * 16384 + c * 1024), 16, voR, (h * 16 + c) * rowR4, 0, 0);
  };
  // ---- prologue: K-tiles 0 and 1 -> stages 0 and 1; the fragments of K-tile 0 / k-step 0 requested ------------------------------------------
  CP16(0, 0);
  CP16(1, 1);
  asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // K-tile 0 (and the bias) has landed (in-order counter), K-tile 1 may still fly
  __builtin_amdgcn_s_barrier();
  NRD(0, 0, 0); NRD(0, 0, 1); NRD(0, 0, 2); NRD(0, 0, 3); NRD(0, 0, 4); NRD(0, 0, 5); NRD(0, 0, 6); NRD(0, 0, 7); NRD(0, 0, 8); NRD(0, 0, 9);
  NRD(0, 0, 10); NRD(0, 0, 11); NRD(0, 0, 12); NRD(0, 0, 13); NRD(0, 0, 14);
  lgkm_wait<8>();
  NRD(0, 0, 15); NRD(0, 0, 16); NRD(0, 0, 17); NRD(0, 0, 18); NRD(0, 0, 19); NRD(0, 0, 20); NRD(0, 0, 21);

  // One K-tile (stage st = kt & 1).  NXT: K-tile kt + 1 exists (copied a K-tile ago into the other stage); LD: K-tile kt + 2 exists -- its 16
  // copies go into THIS stage once everybody has finished reading it (barrier 1), two per block in blocks 8..15.
  // Reads in flight: on entry the 22 of k-step 0 were requested in the order B (14), A0..A7, and at most 15 of them are outstanding.  Block n
  // (0..7) needs B and A_n of k-step 0: at most (7 - n) + (k-step-1 reads requested so far) outstanding, and never more than 15 - (what the
  // block is about to request).
  auto ktile = [&](int kt, auto NXT, auto LD) {
    constexpr bool nxt = decltype(NXT)::value, ld = decltype(LD)::value;
    const int st = kt & 1;
    lgkm_wait<7>();  NBLOCK(0); NRD(st, 1, 0);  NRD(st, 1, 1);  NRD(st, 1, 2);                 // 7 -> 10 in flight at most
    lgkm_wait<9>();  NBLOCK(1); NRD(st, 1, 3);  NRD(st, 1, 4);  NRD(st, 1, 5);                 // 6 + 3 -> 12
    lgkm_wait<11>(); NBLOCK(2); NRD(st, 1, 6);  NRD(st, 1, 7);  NRD(st, 1, 8);                 // 5 + 6 -> 14
    lgkm_wait<12>(); NBLOCK(3); NRD(st, 1, 9);  NRD(st, 1, 10); NRD(st, 1, 11);                // (4 + 9 = 13 would do) -> 15
    lgkm_wait<12>(); NBLOCK(4); NRD(st, 1, 12); NRD(st, 1, 13); NRD(st, 1, 14);                // (3 + 12) -> 15
    lgkm_wait<12>(); NBLOCK(5); NRD(st, 1, 15); NRD(st, 1, 16); NRD(st, 1, 17);                // (2 + 15) -> 15
    lgkm_wait<13>(); NBLOCK(6); NRD(st, 1, 18); NRD(st, 1, 19);                                // (1 + 18) -> 15
    lgkm_wait<13>(); NBLOCK(7); NRD(st, 1, 20); NRD(st, 1, 21);                                // (0 + 20) -> 15
    lgkm_wait<0>();                                         // every fragment of this K-tile is in my registers
    if constexpr (ld) __builtin_amdgcn_s_barrier();         // ... and in everybody's: this stage may be overwritten
    if constexpr (!nxt) { if (has_res) __builtin_amdgcn_s_barrier(); }     // (last K-tile: by the second half of the residual)
#define KC(c) if constexpr (ld) CP(c, kt + 2, st);
#define KS0(c0) if constexpr (!nxt) { if (has_res) { res_copy((c0), 1, st); res_copy((c0) + 1, 1, st); } }
    NBLOCK(8);  KC(0) KC(1) KS0(0)
    NBLOCK(9);  KC(2) KC(3) KS0(2)
    NBLOCK(10); KC(4) KC(5) KS0(4)
    NBLOCK(11); KC(6) KC(7) KS0(6)
#undef KS0
    if constexpr (nxt) {
      // K-tile kt + 1 (copied during K-tile kt - 1, or in the prologue) has landed: only this K-tile's eight copies may still fly
      if constexpr (ld) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    // K-tile nk - 2 (nxt && !ld): behind barrier 2 nobody reads this stage any more -> the first half of the residual, four copies per block
    // K-tile nk - 1 (!nxt): the barrier of block 8 below says the same of the last stage -> the second half, two copies per block
#define KR(c0, n_) if constexpr (nxt && !ld) { if (has_res) { for (int c_ = (c0); c_ < (c0) + (n_); ++c_) res_copy(c_, 0, st); } }
#define KS(c0) if constexpr (!nxt) { if (has_res) { res_copy((c0), 1, st); res_copy((c0) + 1, 1, st); } }
    NBLOCK(12); KC(8) KC(9) KR(0, 4) KS(8)
    if constexpr (nxt) { NRD(st ^ 1, 0, 0); NRD(st ^ 1, 0, 1); NRD(st ^ 1, 0, 2); NRD(st ^ 1, 0, 3); NRD(st ^ 1, 0, 4); NRD(st ^ 1, 0, 5); }
    NBLOCK(13); KC(10) KC(11) KR(4, 4) KS(10)
    if constexpr (nxt) { lgkm_wait<9>(); NRD(st ^ 1, 0, 6); NRD(st ^ 1, 0, 7); NRD(st ^ 1, 0, 8); NRD(st ^ 1, 0, 9); NRD(st ^ 1, 0, 10); NRD(st ^ 1, 0, 11); }
    NBLOCK(14); KC(12) KC(13) KR(8, 4) KS(12)
    if constexpr (nxt) { lgkm_wait<10>(); NRD(st ^ 1, 0, 12); NRD(st ^ 1, 0, 13); NRD(st ^ 1, 0, 14); NRD(st ^ 1, 0, 15); NRD(st ^ 1, 0, 16); }
    NBLOCK(15); KC(14) KC(15) KR(12, 4) KS(14)
    if constexpr (nxt) { lgkm_wait<10>(); NRD(st ^ 1, 0, 17); NRD(st ^ 1, 0, 18); NRD(st ^ 1, 0, 19); NRD(st ^ 1, 0, 20); NRD(st ^ 1, 0, 21); }
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
#undef CP16
#undef CP
#undef NBLOCK
#undef NRD

  // ---- epilogue: bias / column scale / residual, 16 rows at a time through a wave-private LDS area; the residual out of the stages ------------
  char* stage = smem + N_EPI + wave_u * 4096;
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
      // row i * 16 + r16 of the wave's 128 = row (i & 3) * 16 + r16 of its half: copy (row >> 2), lane slot (row & 3) * 14 + chunk
      const int rr = (i & 3) * 16 + r16;
      const char* rp = smem + ((i < 4) ? stA : (stA ^ 1)) * N_STAGE + wave_u * 16384 + (rr >> 2) * 1024 + ((rr & 3) * 14 + (g >> 1)) * 16 + (g & 1) * 8;
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

// does this kernel take the call?  Measured against the eight-wave kernel at steady state, one process (tools/nn_ab.py, profiles/r04_nn_ab.txt): without a
// residual 89.8 against 92.6 us (16384 x 1792 x 1792, the attention-output data gradient), 240 against 254 (K = 5376), 725 against 765 (8192 x 7168 x
// 8192, = the vendor library's 723); with a residual 95.1 against 100.6 and 266 against 271 since the residual tile goes through the LDS stages (fetched
// into registers in the epilogue -- 56 loads per lane, a memory round trip per tile -- it only tied: 106 / 102, 278 / 275).  gemm_variant 11: the
// round's earlier dispatch (four waves only where the epilogue has no residual), for the in-step A/B; 6: the eight-wave kernel everywhere.
bool mts_gemm224n_applies(const GemmArgs& a, int layout, bool c_is_f32, int splits) {
  if (c_is_f32 || splits != 1 || layout != MTS_NN) return false;
  if ((a.epi & MTS_EPI_RESIDUAL) && a.variant == 11) return false;
  const unsigned simple = MTS_EPI_BIAS | MTS_EPI_COLSCALE | MTS_EPI_RESIDUAL;
  const size_t spanB = ((size_t)a.K + 64) * a.ldb * 2;     // byte offsets of the B copies stay below 2^31
  return !a.slab && (a.epi & ~simple) == 0 && (a.M % 256 == 0) && (a.N % N_BN == 0) && (a.K % BK == 0) && a.K >= 2 * BK && a.ksplit == a.K &&
         (a.ldc % 8 == 0) && (((uintptr_t)a.C & 15) == 0) && (a.lda % 8 == 0) && (a.ldb % 8 == 0) && (((uintptr_t)a.A & 15) == 0) &&
         (((uintptr_t)a.B & 15) == 0) && spanB < 0x7ff00000u && (size_t)256 * a.lda * 2 + (size_t)a.K * 2 < 0x7ff00000u &&
         (!(a.epi & MTS_EPI_COLSCALE) || a.ncols_scaled % 4 == 0) &&
         (!(a.epi & MTS_EPI_RESIDUAL) || (a.ldr % 4 == 0 && ((uintptr_t)a.residual & 7) == 0)) &&
         (!(a.epi & MTS_EPI_BIAS) || ((uintptr_t)a.bias & 15) == 0);
}

// called from mts_launch_gemm224 (gemm224.hip); -1: shape / epilogue not covered here
int mts_launch_gemm224n(const GemmArgs& a, int layout, bool c_is_f32, int splits, hipStream_t st) {
  if (!mts_gemm224n_applies(a, layout, c_is_f32, splits)) return -1;
  auto k = gemm_bf16_224n_kernel;
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, N_LDS);
    if (e != hipSuccess) { mts_set_error("gemm224n: hipFuncSetAttribute: %s", hipGetErrorString(e)); return MTS_ERR_LAUNCH; }
    attr_set = true;
  }
  const int nt = (a.M / 256) * (a.N / N_BN);
  hipLaunchKernelGGL(k, dim3(nt), dim3(256), N_LDS, st, a);
  return MTS_OK;
}
