// Fused feed-forward block of the restricted-window encoder layer, hidden width F = 256, bf16 (BASELINE: d = 1792, ff = 256).
//
//   forward  (modeling_longformer.py:1113-1131):  u = a1 W1^T + b1 ; f = act(u) ; s2 = f W2^T + b2 + a1
//   backward (data gradient of the same block):   du = (ds2 W2) * act'(u) ;      da1 = du W1 + ds2
//
// As two GEMM launches each direction moves the [M, d] activation through HBM twice plus the [M, F] intermediate out and back, and
// both launches are latency-bound: N = 256 (or K = 256) leaves ONE 128-row tile per CU with 28 dependent K-tiles (up / down
// projection: 40-46 us each inside the step, 0.33 PFLOP/s).  Here a workgroup owns 64 rows: phase A reduces over the wide
// dimension into a 64 x 256 tile that stays in LDS (as four K-major MFMA operand images), phase B streams the second weight matrix
// through the same LDS ring and writes 64 x 256 output chunks.  Traffic: X in (+ the residual re-read of the same rows), Y out, the
// 64 x 256 intermediates out once (saved for the backward / weight gradients).  Results are BITWISE those of the two-launch path:
// same k order per output element, same single rounding of f / du to bf16, same epilogue order (bias, residual).
//
// Geometry: 512 threads = 8 waves as 2 (M: 32 rows) x 4 (N: 64 columns), wave tile 32 x 64 = 2 x 4 v_mfma_f32_16x16x32_bf16.
// LDS: [0, 32 KiB) the 64 x 256 intermediate; [32 KiB, 160 KiB) a ring of copy stages -- phase A: 3 stages of (X tile 64 rows x 64 k =
// 8 KiB | first-weight tile 256 x 64 = 32 KiB), two K-tiles ahead; phase B: 4 stages of 32 KiB (second-weight tile), three ahead.
// One workgroup barrier per K-tile; every vector-memory wait is COUNTED (the counter retires in order: loads, copies and stores
// share it), including the lazily drained output stores of phase B -- see the table at the waits.
#include <algorithm>
#include "gemm_common.h"

#define FF_BM 64
#define FF_F 256
#define FF_IMG 32768                  // intermediate: 4 images [64 rows][64 k]
#define FF_RING (FF_IMG)              // ring base
#define FF_A_STAGE (8192 + 32768)     // phase A stage: X tile | W tile
#define FF_B_STAGE 32768
#define FF_LDS (FF_IMG + 4 * FF_B_STAGE)      // 160 KiB (phase A uses 3 x 40 KiB of the ring)
#ifdef FFN_STAMPS
#define FSTAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tacc[i] += t_ - tprev; tprev = t_; } while (0)
#else
#define FSTAMP(i) do { } while (0)
#endif

struct FfnArgs {
  const bf16_t* X;      // [M, D]   forward: a1;  backward: ds2
  const bf16_t* Wa;     // phase A weight.  forward: W1 [F, D] (K-major rows = F columns of u);  backward: W2 [D, F] (k rows, F contiguous)
  const bf16_t* Wb;     // phase B weight.  forward: W2 [D, F] (K-major rows = output columns);   backward: W1 [F, D] (k rows, D contiguous)
  const float* ba;      // forward: b1 [F];  backward: unused
  const float* bb;      // forward: b2 [D];  backward: unused
  const bf16_t* U;      // backward: saved pre-activation u [M, F]
  bf16_t* Uout;         // forward: u out;  backward: unused        } OUTPUTS have room for ceil(M / 64) * 64 rows: rows past M are
  bf16_t* T;            // forward: f out;  backward: du out  [., F] } written (garbage) instead of masked, so that every wave issues
  bf16_t* Y;            // forward: s2;  backward: da1        [., D] } the same number of stores
  int M, D, relu;
#ifdef FFN_STAMPS
  unsigned long long* stamps;      // [workgroup][wave][16] accumulated cycles (tools/micro/ffn_variants.sh STAMPS)
#endif
};

// Output-column order.  An MFMA tile leaves lane (r16, g) with 4 consecutive output columns (n = 4g .. 4g+3 of the tile's 16): stored
// as they fall, a wave instruction writes 16 rows x 32 B.  The weight columns are therefore dealt to the four tiles j of a wave's
// 64 columns so that a lane's values of tiles 2s and 2s+1 are 8 CONSECUTIVE columns and the four g-lanes of a row are adjacent:
//     column(j, n) = (j >> 1) * 32 + (n >> 2) * 8 + (j & 1) * 4 + (n & 3)
// -> 16-byte accesses, 64 contiguous bytes per row and instruction, half as many vector-memory instructions and row segments for
// every epilogue load and store.  K-major weight tiles take the order while they are copied (the copy's per-lane source row),
// strided ones while they are read (the transposing read's per-lane column group).  Which column an accumulator holds does not
// change its value: results stay bitwise those of the two-launch path.
__device__ __forceinline__ int ff_col(int j, int n) { return (j >> 1) * 32 + (n >> 2) * 8 + (j & 1) * 4 + (n & 3); }

// one 1-KiB copy piece of a K-major image: image rows piece*8 .. +7 <- rows of `G` (row pitch ld), 64 k starting at k0.
// PERM: image row R (wave R >> 6, tile (R >> 4) & 3, n = R & 15) holds source row (R & ~63) + ff_col(tile, n).
template <bool PERM>
__device__ __forceinline__ void ff_piece_kmajor(const bf16_t* __restrict__ G, int ld, int row0, int dim, int k0, char* img, int piece, int lane) {
  const int row = piece * 8 + (lane >> 3), pos = lane & 7;
  const int ch = pos ^ ((row >> 1) & 7);
  const int srow = PERM ? (row & ~63) + ff_col((row >> 4) & 3, row & 15) : row;
  const bf16_t* src = G + (size_t)min(row0 + srow, dim - 1) * ld + k0 + ch * 8;
  __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(img + piece * 1024), 16, 0, 0);
}
// one 1-KiB piece of a strided half image [64 k-rows][128 columns]: k-rows piece*4 .. +3, columns col0 .. col0+127
__device__ __forceinline__ void ff_piece_strided(const bf16_t* __restrict__ G, int ld, int k0, int col0, char* img, int piece, int lane) {
  const int kr = piece * 4 + (lane >> 4), c16 = lane & 15;
  const int ch = ((((c16 >> 1) ^ strided_key(kr))) << 1) | (c16 & 1);
  const bf16_t* src = G + (size_t)(k0 + kr) * ld + col0 + ch * 8;
  __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(img + piece * 1024), 16, 0, 0);
}

// fragment of a strided weight image for tile j of the wave's 64 columns, in the ff_col order: the transposing read hands lane r the
// column its 16-lane group's lane r >> 2 addressed, + (r & 3) -- so the four addresses of a group are the four 4-column runs of the
// tile (8 columns apart; frag_strided's are adjacent).  Two of the eight k-rows of a read cycle now meet in a bank (2-way).
__device__ __forceinline__ frag_raw ff_frag_strided(const char* tile, int kbase, int wcol0, int j, int lane) {
  const int r = lane & 15, q = r >> 2, p = r & 3;
  const int col = wcol0 + (j >> 1) * 32 + (j & 1) * 4 + 8 * p;
  const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)(tile + strided_off(kbase + q, col));
  frag_raw f;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.lo) : "v"(addr));
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(f.hi) : "v"(addr));
  return f;
}

#define FF_CWAVES 8                     // compute waves: 2 (M: 32 rows) x 4 (N: 64 columns)
#define FF_LWAVES 4                     // copy waves
#define FF_THREADS ((FF_CWAVES + FF_LWAVES) * 64)

template <bool BWD>
__global__ __launch_bounds__(FF_THREADS) void ffn_fused_kernel(const FfnArgs a) {
  constexpr bool WKM = !BWD;            // weight tiles K-major (forward) or strided (backward)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* fimg = smem;
  char* ring = smem + FF_RING;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bm0 = blockIdx.x * FF_BM;
  const int D = a.D, M = a.M;
  const int nkA = D / 64, nch = D / 256, nB = nch * 4;
#ifdef FFN_STAMPS
  unsigned long long tacc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tprev = __builtin_amdgcn_s_memtime();
#define FF_STAMPS_OUT() do { if (a.stamps && lane == 0) for (int i_ = 0; i_ < 16; ++i_) a.stamps[((size_t)blockIdx.x * 12 + wave) * 16 + i_] = tacc[i_]; } while (0)
#else
#define FF_STAMPS_OUT() do { } while (0)
#endif

  // ======================================================================================================================
  // Copy waves (8 .. 11).  Every LDS-DMA of the kernel is issued here and only here: the vector-memory front end of a CU takes one
  // 1-KiB copy instruction per ~18 cycles, so the 40 (phase A) / 32 (phase B) copies of a step keep it busy for most of the step
  // -- and the wave that issues them is stalled for as long.  Issued by the compute waves (as in the first version of this kernel)
  // that stall sat in front of every step's MFMAs: 690 of a step's 1600 cycles.  Here the compute waves never touch the copy
  // queue; their own loads and stores (epilogues) have their counter to themselves, the copy waves' counter holds copies only.
  // Both kinds meet at the one workgroup barrier per step: a copy wave arrives after the NEXT step's stage has landed.
  // ======================================================================================================================
  if (wave >= FF_CWAVES) {
    const int lw = wave - FF_CWAVES;
    // phase A stage: 8 X pieces + 32 weight pieces = 40, 10 per copy wave.  Weight tile: forward rows = the 256 columns of u
    // (K-major, 32 pieces of 8 rows); backward k-rows of W2 (two strided half images of 128 columns, 16 pieces each).
    auto dmaA = [&](int kt) {
      char* st = ring + (kt % 3) * FF_A_STAGE;
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        const int q = lw * 10 + i;
        if (q < 8) ff_piece_kmajor<false>(a.X, D, bm0, M, kt * 64, st, q, lane);
        else if constexpr (WKM) ff_piece_kmajor<true>(a.Wa, D, 0, FF_F, kt * 64, st + 8192, q - 8, lane);
        else ff_piece_strided(a.Wa, FF_F, kt * 64, ((q - 8) >> 4) * 128, st + 8192 + ((q - 8) >> 4) * 16384, (q - 8) & 15, lane);
      }
    };
    // phase B stage gi = chunk * 4 + kk: second-weight tile for output columns chunk*256 .. +255, k = kk*64 .. +63; 32 pieces, 8 each
    auto dmaB = [&](int gi) {
      char* st = ring + (gi & 3) * FF_B_STAGE;
      const int c = gi >> 2, kk = gi & 3;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int q = lw * 8 + i;
        if constexpr (WKM) ff_piece_kmajor<true>(a.Wb, FF_F, c * 256, D, kk * 64, st, q, lane);
        else ff_piece_strided(a.Wb, D, kk * 64, c * 256 + (q >> 4) * 128, st + (q >> 4) * 16384, q & 15, lane);
      }
    };
    dmaA(0);
    if (nkA > 1) { dmaA(1); asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); }      // K-tile 0 landed
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#pragma clang loop unroll(disable)
    for (int kt = 0; kt < nkA; ++kt) {
      // into the stage read during K-tile kt-1 (every compute wave is past that step's barrier)
      if (kt + 2 < nkA) { dmaA(kt + 2); FSTAMP(0); asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); }     // all but K-tile kt+2: kt+1 landed
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      FSTAMP(3);
      __builtin_amdgcn_s_barrier();
      FSTAMP(4);
    }
    // the ring is free (every wave is past the last barrier of phase A): phase B's first three stages, while the compute waves run
    // phase A's epilogue
    dmaB(0);
    dmaB(1);
    dmaB(2);
    dmaB(3);                                                                  // nB >= 4 always (D >= 256)
    FSTAMP(5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // (the compute waves' epilogue takes far longer than these)
    __builtin_amdgcn_s_barrier();
    FSTAMP(6);
#pragma clang loop unroll(disable)
    for (int gi = 0; gi < nB; ++gi) {
      // During step gi the compute waves read slot gi & 3 and slots of stages gi+1 .. gi+3 are full or filling: the one free slot
      // is that of stage gi-1 (everyone is past that step's barrier), and stage gi+3 goes there.  The step's barrier needs stage
      // gi+1 landed: stages gi+2 and gi+3 (8 copies each from this wave) may still be in flight.
      if (gi >= 1 && gi + 3 < nB) { dmaB(gi + 3); FSTAMP(8); asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      FSTAMP(11);
      __builtin_amdgcn_s_barrier();
      FSTAMP(12);
    }
    FF_STAMPS_OUT();
    return;
  }

  // ======================================================================================================================
  // Compute waves (0 .. 7)
  // ======================================================================================================================
  const int wm = wave >> 2, wn = wave & 3;
  const int r16 = lane & 15, g = lane >> 4;
#ifdef FFN_STAMPS
  int tb = 0;                            // 0: phase A slots 0..4, 8: phase B slots 8..12
#endif
  f32x4 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // one K-tile of MFMAs: A fragments from a K-major image (rows wm*32 + i*16 + r16), B fragments from the weight tile.
  // K-major reads as  lane part (VGPR) + wave part (added once per tile) + immediate offset: kmajor_off(row, chunk) with
  // row = 16 t + r16 is  t * 2048  +  r16 * 128 + ((chunk ^ key(r16)) << 4), so the tile index t of a fragment is an instruction
  // immediate and a step needs four address registers instead of one per fragment (which the compiler hoisted out of the loops
  // for every stage of the ring, then spilled).
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  unsigned lk[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) lk[ks] = lds0 + r16 * 128 + (((ks * 4 + g) ^ ((r16 >> 1) & 7)) << 4);
  auto mma_tile = [&](int Aoff, int Boff) {                // byte offsets of the two tiles from the start of the LDS
    LFrag<true> fa[2][2];
    LFrag<WKM> fb[4][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const unsigned ab = lk[ks] + Aoff + wm * 4096, bb = lk[ks] + Boff + wn * 8192;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if constexpr (WKM) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fb[j][ks].k) : "v"(bb), "n"(j * 2048));
        else fb[j][ks].s = ff_frag_strided(smem + Boff + (wn >> 1) * 16384, ks * 32 + 8 * g, (wn & 1) * 64, j, lane);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fa[i][ks].k) : "v"(ab), "n"(i * 2048));
    }
    lgkm_wait<0>();
    FSTAMP(tb + 1);                      // operand reads landed
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 vb[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) vb[j] = lfrag_get<WKM>(fb[j][ks]);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bf16x8 va = lfrag_get<true>(fa[i][ks]);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vb[j], va, acc[i][j], 0, 0, 0);   // C^T tile
      }
    }
    __builtin_amdgcn_s_setprio(0);
    FSTAMP(tb + 2);                      // MFMAs issued
  };

  // =============================== phase A: T[64, 256] = X[64, D] . Wa ====================================================
  // phase A's epilogue operands first (needed 28 steps from now): backward the saved pre-activations, forward the first bias
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  uint4 upre[2][2];                     // [row block i][column half h]: 8 columns wn*64 + h*32 + g*8 .. +7
  float4 b1[2][2];
  if constexpr (BWD) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        upre[i][h] = *reinterpret_cast<const uint4*>(a.U + (size_t)min(bm0 + wm * 32 + i * 16 + r16, M - 1) * FF_F + wn * 64 + h * 32 + g * 8);
  } else {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int e = 0; e < 2; ++e) b1[h][e] = *reinterpret_cast<const float4*>(a.ba + wn * 64 + h * 32 + g * 8 + e * 4);
  }
  __builtin_amdgcn_s_barrier();                            // K-tile 0 landed (the copy waves waited for it)
#pragma clang loop unroll(disable)
  for (int kt = 0; kt < nkA; ++kt) {
    FSTAMP(0);
    const int st = FF_RING + (kt % 3) * FF_A_STAGE;
    mma_tile(st, st + 8192);
    FSTAMP(3);
    __builtin_amdgcn_s_barrier();
    FSTAMP(4);                                             // barrier: K-tile kt+1 landed, everyone done with K-tile kt
  }
  // ---- phase A epilogue: bias + activation (forward) / times act'(u) (backward); intermediate -> LDS images + global ---------
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = wm * 32 + i * 16 + r16;
#pragma unroll
    for (int h = 0; h < 2; ++h) {                        // tiles 2h, 2h+1 = 8 consecutive columns
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = acc[i][2 * h][e]; v[4 + e] = acc[i][2 * h + 1][e]; }
      const size_t go = (size_t)(bm0 + row) * FF_F + wn * 64 + h * 32 + g * 8;
      if constexpr (!BWD) {
        v[0] += b1[h][0].x; v[1] += b1[h][0].y; v[2] += b1[h][0].z; v[3] += b1[h][0].w;
        v[4] += b1[h][1].x; v[5] += b1[h][1].y; v[6] += b1[h][1].z; v[7] += b1[h][1].w;
        uint4 pre;
        pre.x = pack_bf16x2(v[0], v[1]); pre.y = pack_bf16x2(v[2], v[3]); pre.z = pack_bf16x2(v[4], v[5]); pre.w = pack_bf16x2(v[6], v[7]);
        *reinterpret_cast<uint4*>(a.Uout + go) = pre;     // rows past M: padding rows of the caller's buffer
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = a.relu ? fmaxf(v[r], 0.0f) : gelu_erf_f(v[r]);
      } else {
        const unsigned uw[4] = {upre[i][h].x, upre[i][h].y, upre[i][h].z, upre[i][h].w};
        // the two-launch path rounds the data gradient to bf16 first and then multiplies (mts_gelu_bwd works on the stored tensor)
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float uu = (r & 1) ? bf16_hi(uw[r >> 1]) : bf16_lo(uw[r >> 1]);
          const float dr = to_f32(from_f32<bf16_t>(v[r]));
          v[r] = a.relu ? (uu > 0.0f ? dr : 0.0f) : dr * gelu_erf_grad_f(uu);
        }
      }
      uint4 t;
      t.x = pack_bf16x2(v[0], v[1]); t.y = pack_bf16x2(v[2], v[3]); t.z = pack_bf16x2(v[4], v[5]); t.w = pack_bf16x2(v[6], v[7]);
      *reinterpret_cast<uint4*>(a.T + go) = t;
      // K-major image wn (k = the wave's 64 columns): the 8 columns are 16-byte chunk h*4 + g of the row
      *reinterpret_cast<uint4*>(fimg + wn * 8192 + kmajor_off(row, h * 4 + g)) = t;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  FSTAMP(5);                                               // phase A epilogue issued
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                            // the intermediate is complete in LDS; stage 0 of phase B has landed
  FSTAMP(6);
#ifdef FFN_STAMPS
  tb = 8;
#endif

  // =============================== phase B: Y[64, D] = T[64, 256] . Wb (+ bias) + residual ====================================
  // Per output chunk c (256 columns): 4 K-steps kk.  The chunk's epilogue operands are requested at its first step and used at
  // its last; its stores drain during the next chunk.
  float4 b2[2][2];
  uint4 res[2][2];
#pragma clang loop unroll(disable)
  for (int c = 0; c < nch; ++c) {
    const int colw = c * 256 + wn * 64 + g * 8;            // + h * 32: this lane's 8 columns of half h
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int gi = c * 4 + kk;
      if (kk == 3) {                                       // (the bias one step before its use only: registers)
        if constexpr (!BWD) {
#pragma unroll
          for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 2; ++e) b2[h][e] = *reinterpret_cast<const float4*>(a.bb + colw + h * 32 + e * 4);
        }
      }
      if (kk == 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int h = 0; h < 2; ++h)
            res[i][h] = *reinterpret_cast<const uint4*>(a.X + (size_t)min(bm0 + wm * 32 + i * 16 + r16, M - 1) * D + colw + h * 32);
      }
      FSTAMP(8);
      mma_tile(kk * 8192, FF_RING + kk * FF_B_STAGE);      // stage gi & 3 = kk
      if (kk == 3) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int row = bm0 + wm * 32 + i * 16 + r16;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[e] = acc[i][2 * h][e]; v[4 + e] = acc[i][2 * h + 1][e]; }
            if constexpr (!BWD) {
              v[0] += b2[h][0].x; v[1] += b2[h][0].y; v[2] += b2[h][0].z; v[3] += b2[h][0].w;
              v[4] += b2[h][1].x; v[5] += b2[h][1].y; v[6] += b2[h][1].z; v[7] += b2[h][1].w;
            }
            const unsigned rw[4] = {res[i][h].x, res[i][h].y, res[i][h].z, res[i][h].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[2 * e] += bf16_lo(rw[e]); v[2 * e + 1] += bf16_hi(rw[e]); }
            uint4 o;
            o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]); o.z = pack_bf16x2(v[4], v[5]); o.w = pack_bf16x2(v[6], v[7]);
            *reinterpret_cast<uint4*>(a.Y + (size_t)row * D + colw + h * 32) = o;      // rows past M: the caller's padding rows
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
      }
      FSTAMP(11);
      __builtin_amdgcn_s_barrier();
      FSTAMP(12);
    }
  }
  FF_STAMPS_OUT();
}

#ifdef FFN_STAMPS
static unsigned long long* g_ffn_stamps = nullptr;
extern "C" void mts_ffn_set_stamps(void* p) { g_ffn_stamps = (unsigned long long*)p; }
#endif
static int ffn_launch(bool bwd, hipStream_t st, const FfnArgs& a0) {
  FfnArgs a = a0;
#ifdef FFN_STAMPS
  a.stamps = g_ffn_stamps;
#endif
  const int grid = ceil_div(a.M, FF_BM);
  if (bwd) {
    auto k = ffn_fused_kernel<true>;
    static std::atomic<bool> attr{false};
    if (!attr) { if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS) != hipSuccess) { mts_set_error("ffn_fused: cannot reserve %d bytes of LDS", FF_LDS); return MTS_ERR_LAUNCH; } attr = true; }
    hipLaunchKernelGGL(k, dim3(grid), dim3(FF_THREADS), FF_LDS, st, a);
  } else {
    auto k = ffn_fused_kernel<false>;
    static std::atomic<bool> attr{false};
    if (!attr) { if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, FF_LDS) != hipSuccess) { mts_set_error("ffn_fused: cannot reserve %d bytes of LDS", FF_LDS); return MTS_ERR_LAUNCH; } attr = true; }
    hipLaunchKernelGGL(k, dim3(grid), dim3(FF_THREADS), FF_LDS, st, a);
  }
  MTS_LAUNCH_CHECK("mts_ffn_fused");
  return MTS_OK;
}

static int ffn_check(const char* who, int M, int D, int F, const void* p0, const void* p1, const void* p2, const void* p3, const void* p4) {
  MTS_CHECK_ARG(M > 0 && p0 && p1 && p2 && p3 && p4, "%s: bad arguments", who);
  MTS_UNSUPPORTED(F == FF_F && D >= 256 && D % 256 == 0, "%s: the fused block covers F = 256 and D a multiple of 256 (F=%d, D=%d)", who, F, D);
  return MTS_OK;
}

extern "C" int mts_ffn_supported(int dtype, int M, int D, int F) { return dtype == MTS_BF16 && M > 0 && F == FF_F && D >= 256 && D % 256 == 0; }

extern "C" int mts_ffn_fwd(void* stream, int M, int D, int F, const void* a1, const void* w1, const float* b1, const void* w2, const float* b2,
                           int relu, void* u, void* f, void* s2) {
  if (int rc = ffn_check("mts_ffn_fwd", M, D, F, a1, w1, w2, s2, f)) return rc;
  MTS_CHECK_ARG(b1 && b2 && u, "mts_ffn_fwd: null bias / u");
  FfnArgs a{(const bf16_t*)a1, (const bf16_t*)w1, (const bf16_t*)w2, b1, b2, nullptr, (bf16_t*)u, (bf16_t*)f, (bf16_t*)s2, M, D, relu};
  return ffn_launch(false, (hipStream_t)stream, a);
}

extern "C" int mts_ffn_bwd_data(void* stream, int M, int D, int F, const void* ds2, const void* w1, const void* w2, const void* u, int relu,
                                void* du, void* da1) {
  if (int rc = ffn_check("mts_ffn_bwd_data", M, D, F, ds2, w1, w2, du, da1)) return rc;
  MTS_CHECK_ARG(u, "mts_ffn_bwd_data: null u");
  FfnArgs a{(const bf16_t*)ds2, (const bf16_t*)w2, (const bf16_t*)w1, nullptr, nullptr, (const bf16_t*)u, nullptr, (bf16_t*)du, (bf16_t*)da1, M, D, relu};
  return ffn_launch(true, (hipStream_t)stream, a);
}
