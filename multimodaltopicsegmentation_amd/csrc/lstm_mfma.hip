// Persistent MFMA LSTM recurrence for gfx950 (bf16 activations/weights, fp32 state): the fast path of mts_lstm_fwd/bwd.
//
// One workgroup = 16 documents x one direction, H/32 waves; wave w owns hidden units [32w, 32w+32) and ALL FOUR gates
// of those units, so after the per-step GEMM  gates[16 docs, 4H] = h[16, H] . W_hh^T  (v_mfma_f32_16x16x32_bf16 with
// the weight fragment as the A operand and h as the B operand) every lane already holds i,f,g,o of its own
// (document, 4 units) cells: the cell update is lane-local, c stays in registers for the whole sequence and the only
// exchange per time step is the new h (bf16, 16 x H) through a double-buffered LDS tile -> ONE barrier per step.
//
// The step is latency-bound, so the schedule of every memory operation is explicit:
//   * W_hh (512 KiB bf16 at H = 256, more than a CU's LDS): each wave keeps RT of its 8 weight tiles in VGPRs and LT
//     in a wave-private LDS area for the whole sequence and streams the rest from L2 through two register buffers,
//     the first of them loaded at the END of the previous step;
//   * the x-projection rows (forward) / saved gates, cells, dOut (backward) of step s+1 are loaded during step s;
//   * vmcnt retires in issue order and counts stores, so every load is issued BEFORE the step's stores, never
//     right after them (a load behind a store waits for the store's write acknowledgement);
//   * read-only weight loads are loop-invariant to the compiler, which would hoist ALL of them out of the time loop
//     and spill: the per-lane offset is laundered through an empty asm each step.
// Backward mirrors forward with W_hh^T (dh_prev = da . W_hh) and the gate gradients exchanged through LDS.
// Packed-sequence semantics as lstm.hip (NeuralArchitectures.py:98-115).
#include <stdlib.h>
#include "common.h"

#define LM_DOCS 16

// v_exp/v_rcp based activations (bf16 path: ~1e-6 relative, far below the storage rounding)
__device__ __forceinline__ float fsig(float x) { return __frcp_rn(1.0f + __expf(-x)); }
__device__ __forceinline__ float ftanh(float x) { return 2.0f * __frcp_rn(1.0f + __expf(-2.0f * x)) - 1.0f; }

__device__ __forceinline__ uint2 pack4(const float (&v)[4]) {
  uint2 r;
  r.x = pack_bf16x2(v[0], v[1]);
  r.y = pack_bf16x2(v[2], v[3]);
  return r;
}
__device__ __forceinline__ void unpack4(const uint2& u, float (&v)[4]) {
  v[0] = bf16_lo(u.x); v[1] = bf16_hi(u.x); v[2] = bf16_lo(u.y); v[3] = bf16_hi(u.y);
}
#define SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

// KS = H/32 (k-steps = waves); RT weight tiles in registers, LT in LDS, NS = 8-RT-LT streamed per step (NS >= 1)
template <int KS, int RT, int LT>
__global__ __launch_bounds__(KS * 64, 2) void lstm_fwd_mfma_kernel(int B, int L, int ndir, const bf16_t* __restrict__ xproj,
                                                                   const bf16_t* __restrict__ whh /*[ndir][4H][H] bf16*/,
                                                                   const float* __restrict__ bhh, const int32_t* __restrict__ lengths,
                                                                   bf16_t* __restrict__ out, bf16_t* __restrict__ gates, float* __restrict__ cells,
                                                                   int xflags) {
  constexpr int H = KS * 32;
  constexpr int NS = 8 - RT - LT;
  static_assert(NS >= 1, "at least one streamed tile");
  // xflags: timing diagnostics only (MTS_LSTM_EXP): 1 no global stores, 2 no x loads, 4 no streamed weight loads, 8 no barrier
  const bool x_nostore = xflags & 1, x_nox = xflags & 2, x_nostream = xflags & 4, x_nobar = xflags & 8;
  constexpr int HROW = (H + 8) * 2;                    // bytes per h row in LDS (16 B x odd: conflict-free b128)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* hbuf = smem;                                   // [2][16][HROW]
  char* wlds = smem + 2 * LM_DOCS * HROW;              // [waves][LT][KS][1024]
  float* blds = reinterpret_cast<float*>(wlds + (size_t)KS * LT * KS * 1024);   // [4H] recurrent bias
  const int d = blockIdx.y;
  const int b0 = blockIdx.x * LM_DOCS;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int doc = lane & 15, g4 = lane >> 4;
  const int u0 = w * 32;
  const int ldx = ndir * 4 * H, ldo = ndir * H;
  const bf16_t* W = whh + (size_t)d * 4 * H * H;

  const int bdoc = b0 + doc;
  const int len = (bdoc < B) ? (lengths ? min(lengths[bdoc], L) : L) : 0;
  int maxlen = len;
#pragma unroll
  for (int off = 1; off < 16; off <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off, 64));

  // tile q = gate*2 + t2 covers gate columns gate*H + u0 + t2*16 .. +15 ; A fragment: row = that column + (lane&15)
  unsigned lane_off = (unsigned)((lane & 15) * H + 8 * g4);
  auto wptr = [&](int q, int ks) { return W + (size_t)((q >> 1) * H + u0 + (q & 1) * 16) * H + ks * 32 + lane_off; };
  bf16x8 wreg[RT > 0 ? RT : 1][KS];
#pragma unroll
  for (int q = 0; q < RT; ++q)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) wreg[q][ks] = *reinterpret_cast<const bf16x8*>(wptr(q, ks));
  char* myw = wlds + (size_t)w * LT * KS * 1024;
#pragma unroll
  for (int q = 0; q < LT; ++q)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      *reinterpret_cast<bf16x8*>(myw + (q * KS + ks) * 1024 + lane * 16) = *reinterpret_cast<const bf16x8*>(wptr(RT + q, ks));
  for (int i = threadIdx.x; i < 4 * H; i += KS * 64) blds[i] = bhh ? bhh[(size_t)d * 4 * H + i] : 0.f;

  float c[2][4];
  uint2 hq[2];
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2) {
    hq[t2] = make_uint2(0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) c[t2][r] = 0.f;
    *reinterpret_cast<uint2*>(hbuf + doc * HROW + (u0 + t2 * 16 + 4 * g4) * 2) = hq[t2];
  }

  auto xrow = [&](int s) -> long {
    if (s >= len) return -1;
    const int t = (d == 0) ? s : (len - 1 - s);
    return (long)bdoc * L + t;
  };
  auto load_x = [&](int s, uint2 (&x)[4][2]) {
    const long row = x_nox ? -1 : xrow(s);
#pragma unroll
    for (int gt = 0; gt < 4; ++gt)
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
        x[gt][t2] = (row >= 0) ? *reinterpret_cast<const uint2*>(xproj + (size_t)row * ldx + (size_t)d * 4 * H + gt * H + u0 + t2 * 16 + 4 * g4)
                               : make_uint2(0, 0);
  };
  uint2 xc[4][2];
  load_x(0, xc);
  bf16x8 wa[KS], wb[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) wa[ks] = *reinterpret_cast<const bf16x8*>(wptr(RT + LT, ks));   // first streamed tile of step 0
  __syncthreads();

  for (int s = 0; s < maxlen; ++s) {
    asm volatile("" : "+v"(lane_off));                   // keep the streamed weight loads inside the time loop
    const char* hcur = hbuf + (s & 1) * LM_DOCS * HROW;
    char* hnext = hbuf + ((s + 1) & 1) * LM_DOCS * HROW;

    bf16x8 hf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) hf[ks] = *reinterpret_cast<const bf16x8*>(hcur + doc * HROW + (ks * 32 + 8 * g4) * 2);
    if constexpr (NS > 1) {
      if (!x_nostream) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wb[ks] = *reinterpret_cast<const bf16x8*>(wptr(RT + LT + 1, ks));
      }
    }
    SCHED_FENCE();

    f32x4 acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < RT; ++q)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[q][ks], hf[ks], acc[q], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < LT; ++q)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const bf16x8 wv = *reinterpret_cast<const bf16x8*>(myw + (q * KS + ks) * 1024 + lane * 16);
        acc[RT + q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, hf[ks], acc[RT + q], 0, 0, 0);
      }
    SCHED_FENCE();
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      const int tq = RT + LT + q;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
        acc[tq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((q & 1) ? wb[ks] : wa[ks], hf[ks], acc[tq], 0, 0, 0);
      SCHED_FENCE();
      if (q + 2 < NS && !x_nostream) {                   // refill the buffer just consumed with tile q+2
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          if (q & 1) wb[ks] = *reinterpret_cast<const bf16x8*>(wptr(tq + 2, ks));
          else wa[ks] = *reinterpret_cast<const bf16x8*>(wptr(tq + 2, ks));
        }
        SCHED_FENCE();
      }
    }
    // loads for the NEXT step go out before this step's stores: x rows, then the first streamed tile
    uint2 xn[4][2];
    load_x(s + 1, xn);
    if (!x_nostream) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) wa[ks] = *reinterpret_cast<const bf16x8*>(wptr(RT + LT, ks));
    }
    SCHED_FENCE();

    const bool active = s < len;
    const long row = xrow(s);
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      if (active) {
        float xi[4], xf[4], xg[4], xo[4], gi[4], gf[4], gg[4], go[4], hn[4];
        unpack4(xc[0][t2], xi); unpack4(xc[1][t2], xf); unpack4(xc[2][t2], xg); unpack4(xc[3][t2], xo);
        const int ub = u0 + t2 * 16 + 4 * g4;
        const float4 bi = *reinterpret_cast<const float4*>(blds + ub), bf = *reinterpret_cast<const float4*>(blds + H + ub);
        const float4 bg = *reinterpret_cast<const float4*>(blds + 2 * H + ub), bo = *reinterpret_cast<const float4*>(blds + 3 * H + ub);
        const float bia[4][4] = {{bi.x, bi.y, bi.z, bi.w}, {bf.x, bf.y, bf.z, bf.w}, {bg.x, bg.y, bg.z, bg.w}, {bo.x, bo.y, bo.z, bo.w}};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          gi[r] = fsig((xi[r] + bia[0][r]) + acc[0 + t2][r]);
          gf[r] = fsig((xf[r] + bia[1][r]) + acc[2 + t2][r]);
          gg[r] = ftanh((xg[r] + bia[2][r]) + acc[4 + t2][r]);
          go[r] = fsig((xo[r] + bia[3][r]) + acc[6 + t2][r]);
          c[t2][r] = gf[r] * c[t2][r] + gi[r] * gg[r];
          hn[r] = go[r] * ftanh(c[t2][r]);
        }
        hq[t2] = pack4(hn);
        if (x_nostore) continue;
        bf16_t* gp = gates + (size_t)row * ldx + (size_t)d * 4 * H + ub;
        *reinterpret_cast<uint2*>(gp) = pack4(gi);
        *reinterpret_cast<uint2*>(gp + H) = pack4(gf);
        *reinterpret_cast<uint2*>(gp + 2 * H) = pack4(gg);
        *reinterpret_cast<uint2*>(gp + 3 * H) = pack4(go);
        *reinterpret_cast<float4*>(cells + (size_t)row * ldo + (size_t)d * H + ub) = make_float4(c[t2][0], c[t2][1], c[t2][2], c[t2][3]);
        *reinterpret_cast<uint2*>(out + (size_t)row * ldo + (size_t)d * H + ub) = hq[t2];
      }
      *reinterpret_cast<uint2*>(hnext + doc * HROW + (u0 + t2 * 16 + 4 * g4) * 2) = hq[t2];   // inactive documents carry h
    }
#pragma unroll
    for (int gt = 0; gt < 4; ++gt)
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2) xc[gt][t2] = xn[gt][t2];
    if (!x_nobar) __syncthreads();
  }
  // rows >= len are exactly zero
  if (bdoc < B) {
    for (int t = len + g4; t < L; t += 4)
      for (int e = 0; e < 32; e += 4) *reinterpret_cast<uint2*>(out + ((size_t)bdoc * L + t) * ldo + (size_t)d * H + u0 + e) = make_uint2(0, 0);
  }
}

// everything the elementwise part of one backward step reads from global memory, for one (document, 2 x 4 units) lane
struct BwdIn {
  uint2 gi[2], gf[2], gg[2], go[2], dov[2], hp[2];
  float4 ct[2], cp[2];
};

// backward: dh_prev[16, H] = da[16, 4H] . W_hh  ->  A operand = W_hh^T tile (rows = hidden unit, K = gate column), B = da (LDS)
template <int KS, int RT, int LT>
__global__ __launch_bounds__(KS * 64, 2) void lstm_bwd_mfma_kernel(int B, int L, int ndir, const bf16_t* __restrict__ whhT /*[ndir][H][4H] bf16*/,
                                                                   const int32_t* __restrict__ lengths, const bf16_t* __restrict__ out,
                                                                   const bf16_t* __restrict__ gates, const float* __restrict__ cells,
                                                                   const bf16_t* __restrict__ dout, bf16_t* __restrict__ dxproj,
                                                                   bf16_t* __restrict__ hprev) {
  constexpr int H = KS * 32;
  constexpr int NS = 8 - RT - LT;
  static_assert(NS >= 1, "at least one streamed slice");
  constexpr int DROW = (4 * H + 8) * 2;                // bytes per da row in LDS
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* dabuf = smem;                                  // [2][16][DROW]
  char* wlds = smem + 2 * LM_DOCS * DROW;              // [waves][LT][KS][1024]
  const int d = blockIdx.y;
  const int b0 = blockIdx.x * LM_DOCS;
  const int lane = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int doc = lane & 15, g4 = lane >> 4;
  const int u0 = w * 32;
  const int ldx = ndir * 4 * H, ldo = ndir * H;
  const bf16_t* WT = whhT + (size_t)d * 4 * H * H;

  const int bdoc = b0 + doc;
  const int len = (bdoc < B) ? (lengths ? min(lengths[bdoc], L) : L) : 0;
  int maxlen = len;
#pragma unroll
  for (int off = 1; off < 16; off <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off, 64));

  // output tile t2 (units u0 + t2*16 ..): A fragment row = unit + (lane&15), k over the 4H gate columns.
  // Residency is counted in slices of KS k-steps: slice sl = t2*4 + gate, 8 slices per wave.
  unsigned lane_off = (unsigned)((lane & 15) * 4 * H + 8 * g4);
  auto sptr = [&](int sl, int k) { return WT + (size_t)(u0 + (sl >> 2) * 16) * 4 * H + ((sl & 3) * KS + k) * 32 + lane_off; };
  bf16x8 wreg[RT > 0 ? RT : 1][KS];
#pragma unroll
  for (int q = 0; q < RT; ++q)
#pragma unroll
    for (int k = 0; k < KS; ++k) wreg[q][k] = *reinterpret_cast<const bf16x8*>(sptr(q, k));
  char* myw = wlds + (size_t)w * LT * KS * 1024;
#pragma unroll
  for (int q = 0; q < LT; ++q)
#pragma unroll
    for (int k = 0; k < KS; ++k)
      *reinterpret_cast<bf16x8*>(myw + (q * KS + k) * 1024 + lane * 16) = *reinterpret_cast<const bf16x8*>(sptr(RT + q, k));

  auto load_in = [&](int s, BwdIn& in) {
    const bool act = (s >= 0) && (s < len);
    const int t = (d == 0) ? s : (len - 1 - s);
    const int tp = (d == 0) ? t - 1 : t + 1;
    const bool has_prev = s > 0;
    const size_t row = (size_t)bdoc * L + (act ? t : 0);
    const size_t prow = (size_t)bdoc * L + ((act && has_prev) ? tp : 0);
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      const int u = u0 + t2 * 16 + 4 * g4;
      const uint2 z2 = make_uint2(0, 0);
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      const bf16_t* gp = gates + row * ldx + (size_t)d * 4 * H + u;
      in.gi[t2] = act ? *reinterpret_cast<const uint2*>(gp) : z2;
      in.gf[t2] = act ? *reinterpret_cast<const uint2*>(gp + H) : z2;
      in.gg[t2] = act ? *reinterpret_cast<const uint2*>(gp + 2 * H) : z2;
      in.go[t2] = act ? *reinterpret_cast<const uint2*>(gp + 3 * H) : z2;
      in.dov[t2] = act ? *reinterpret_cast<const uint2*>(dout + row * ldo + (size_t)d * H + u) : z2;
      in.ct[t2] = act ? *reinterpret_cast<const float4*>(cells + row * ldo + (size_t)d * H + u) : z4;
      in.cp[t2] = (act && has_prev) ? *reinterpret_cast<const float4*>(cells + prow * ldo + (size_t)d * H + u) : z4;
      in.hp[t2] = (act && has_prev) ? *reinterpret_cast<const uint2*>(out + prow * ldo + (size_t)d * H + u) : z2;
    }
  };

  float dh[2][4], dc[2][4];
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
    for (int r = 0; r < 4; ++r) dh[t2][r] = dc[t2][r] = 0.f;
  BwdIn cur;
  load_in(maxlen - 1, cur);
  bf16x8 wa[KS], wb[KS];
#pragma unroll
  for (int k = 0; k < KS; ++k) wa[k] = *reinterpret_cast<const bf16x8*>(sptr(RT + LT, k));
  __syncthreads();

  for (int s = maxlen - 1; s >= 0; --s) {
    asm volatile("" : "+v"(lane_off));                  // keep the streamed weight loads inside the time loop
    char* da = dabuf + (s & 1) * LM_DOCS * DROW;
    const bool active = s < len;
    const int t = (d == 0) ? s : (len - 1 - s);
    const size_t row = (size_t)bdoc * L + (active ? t : 0);
    // next step's inputs and this step's second streamed slice go out before any store of this step
    BwdIn nxt;
    load_in(s - 1, nxt);
    if constexpr (NS > 1) {
#pragma unroll
      for (int k = 0; k < KS; ++k) wb[k] = *reinterpret_cast<const bf16x8*>(sptr(RT + LT + 1, k));
    }
    SCHED_FENCE();
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      const int u = u0 + t2 * 16 + 4 * g4;
      float ai[4] = {0.f, 0.f, 0.f, 0.f}, af[4] = {0.f, 0.f, 0.f, 0.f}, ag[4] = {0.f, 0.f, 0.f, 0.f}, ao[4] = {0.f, 0.f, 0.f, 0.f};
      if (active) {
        float gi[4], gf[4], gg[4], go[4], dov[4];
        unpack4(cur.gi[t2], gi); unpack4(cur.gf[t2], gf); unpack4(cur.gg[t2], gg); unpack4(cur.go[t2], go); unpack4(cur.dov[t2], dov);
        const float ct[4] = {cur.ct[t2].x, cur.ct[t2].y, cur.ct[t2].z, cur.ct[t2].w};
        const float cp[4] = {cur.cp[t2].x, cur.cp[t2].y, cur.cp[t2].z, cur.cp[t2].w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float tc = ftanh(ct[r]);
          const float dht = dov[r] + dh[t2][r];
          const float dct = dc[t2][r] + dht * go[r] * (1.f - tc * tc);
          ai[r] = dct * gg[r] * gi[r] * (1.f - gi[r]);
          af[r] = dct * cp[r] * gf[r] * (1.f - gf[r]);
          ag[r] = dct * gi[r] * (1.f - gg[r] * gg[r]);
          ao[r] = dht * tc * go[r] * (1.f - go[r]);
          dc[t2][r] = dct * gf[r];
        }
        bf16_t* dx = dxproj + row * ldx + (size_t)d * 4 * H + u;
        *reinterpret_cast<uint2*>(dx) = pack4(ai);
        *reinterpret_cast<uint2*>(dx + H) = pack4(af);
        *reinterpret_cast<uint2*>(dx + 2 * H) = pack4(ag);
        *reinterpret_cast<uint2*>(dx + 3 * H) = pack4(ao);
        *reinterpret_cast<uint2*>(hprev + row * ldo + (size_t)d * H + u) = cur.hp[t2];
      }
      char* dr = da + doc * DROW + u * 2;
      *reinterpret_cast<uint2*>(dr) = pack4(ai);
      *reinterpret_cast<uint2*>(dr + H * 2) = pack4(af);
      *reinterpret_cast<uint2*>(dr + 2 * H * 2) = pack4(ag);
      *reinterpret_cast<uint2*>(dr + 3 * H * 2) = pack4(ao);
    }
    __syncthreads();
    // dh_prev = da . W : 2 output tiles x 4*KS k-steps, weights by residency class
    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
    auto dafrag = [&](int sl, int k) {
      const int kk = (sl & 3) * KS + k;
      return *reinterpret_cast<const bf16x8*>(da + doc * DROW + (kk * 32 + 8 * g4) * 2);
    };
#pragma unroll
    for (int q = 0; q < RT; ++q)
#pragma unroll
      for (int k = 0; k < KS; ++k) acc[q >> 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[q][k], dafrag(q, k), acc[q >> 2], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < LT; ++q)
#pragma unroll
      for (int k = 0; k < KS; ++k) {
        const bf16x8 wv = *reinterpret_cast<const bf16x8*>(myw + (q * KS + k) * 1024 + lane * 16);
        acc[(RT + q) >> 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, dafrag(RT + q, k), acc[(RT + q) >> 2], 0, 0, 0);
      }
    SCHED_FENCE();
#pragma unroll
    for (int q = 0; q < NS; ++q) {
      const int sl = RT + LT + q;
#pragma unroll
      for (int k = 0; k < KS; ++k)
        acc[sl >> 2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16((q & 1) ? wb[k] : wa[k], dafrag(sl, k), acc[sl >> 2], 0, 0, 0);
      SCHED_FENCE();
      if (q + 2 < NS) {
#pragma unroll
        for (int k = 0; k < KS; ++k) {
          if (q & 1) wb[k] = *reinterpret_cast<const bf16x8*>(sptr(sl + 2, k));
          else wa[k] = *reinterpret_cast<const bf16x8*>(sptr(sl + 2, k));
        }
        SCHED_FENCE();
      }
    }
#pragma unroll
    for (int k = 0; k < KS; ++k) wa[k] = *reinterpret_cast<const bf16x8*>(sptr(RT + LT, k));   // first streamed slice of the next step
    if (active) {
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int r = 0; r < 4; ++r) dh[t2][r] = acc[t2][r];
    }
    cur = nxt;
    // da is double-buffered: the next step writes the other buffer, so one barrier per step suffices
  }
  // rows >= len: zero gradients (the GEMMs that follow read every row)
  if (bdoc < B) {
    for (int t = len + g4; t < L; t += 4) {
      for (int e = 0; e < 32; e += 4) {
        const size_t r0 = (size_t)bdoc * L + t;
#pragma unroll
        for (int gt = 0; gt < 4; ++gt) *reinterpret_cast<uint2*>(dxproj + r0 * ldx + (size_t)d * 4 * H + gt * H + u0 + e) = make_uint2(0, 0);
        *reinterpret_cast<uint2*>(hprev + r0 * ldo + (size_t)d * H + u0 + e) = make_uint2(0, 0);
      }
    }
  }
}

// ---- host side ---------------------------------------------------------------------------------------
__global__ void cast_transpose_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, bf16_t* __restrict__ dstT, int rows, int cols) {
  // src [z][rows][cols] fp32 -> dst same layout bf16, dstT [z][cols][rows] bf16
  __shared__ float tile[32][33];
  const float* s = src + (size_t)blockIdx.z * rows * cols;
  int x = blockIdx.x * 32 + threadIdx.x, y = blockIdx.y * 32 + threadIdx.y;
  for (int i = 0; i < 32; i += 8)
    if (x < cols && y + i < rows) {
      const float v = s[(size_t)(y + i) * cols + x];
      tile[threadIdx.y + i][threadIdx.x] = v;
      if (dst) dst[(size_t)blockIdx.z * rows * cols + (size_t)(y + i) * cols + x] = (bf16_t)v;
    }
  __syncthreads();
  x = blockIdx.y * 32 + threadIdx.x; y = blockIdx.x * 32 + threadIdx.y;
  for (int i = 0; i < 32; i += 8)
    if (dstT && x < rows && y + i < cols) dstT[(size_t)blockIdx.z * rows * cols + (size_t)(y + i) * rows + x] = (bf16_t)tile[threadIdx.x][threadIdx.y + i];
}

// residency: forward tiles / backward slices are 8 KiB per wave each; LDS budget 160 KiB
#ifndef LM_FWD_RT
#define LM_FWD_RT 2
#endif
#ifndef LM_FWD_LT
#define LM_FWD_LT 2
#endif
#ifndef LM_BWD_RT
#define LM_BWD_RT 1
#endif
#ifndef LM_BWD_LT
#define LM_BWD_LT 1
#endif

bool mts_lstm_mfma_supported(int dtype, int H) { return dtype == MTS_BF16 && H == 256; }

size_t mts_lstm_mfma_workspace(int H, int ndir) { return align_up((size_t)ndir * 4 * H * H * 2, 256) * 2; }   // W bf16 + W^T bf16

int mts_lstm_mfma_fwd(hipStream_t st, int B, int L, int H, int ndir, const void* xproj, const float* w_hh, const float* b_hh,
                      const int32_t* lengths, void* out, void* gates, float* cells, void* ws) {
  bf16_t* wb = (bf16_t*)ws;
  hipLaunchKernelGGL(cast_transpose_bf16_kernel, dim3(ceil_div(H, 32), ceil_div(4 * H, 32), ndir), dim3(32, 8), 0, st, w_hh, wb, (bf16_t*)nullptr,
                     4 * H, H);
  constexpr int KS = 8;
  const size_t lds = 2 * LM_DOCS * (H + 8) * 2 + (size_t)KS * LM_FWD_LT * KS * 1024 + (size_t)4 * H * sizeof(float);
  auto k = lstm_fwd_mfma_kernel<KS, LM_FWD_RT, LM_FWD_LT>;
  static std::atomic<bool> attr{false};
  if (!attr) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      mts_set_error("lstm_mfma_fwd: cannot reserve %zu bytes of LDS", lds);
      return MTS_ERR_LAUNCH;
    }
    attr = true;
  }
  static int xflags = -1;
  if (xflags < 0) { const char* e = getenv("MTS_LSTM_EXP"); xflags = e ? atoi(e) : 0; }
  hipLaunchKernelGGL(k, dim3(ceil_div(B, LM_DOCS), ndir), dim3(KS * 64), lds, st, B, L, ndir, (const bf16_t*)xproj, (const bf16_t*)wb, b_hh, lengths,
                     (bf16_t*)out, (bf16_t*)gates, cells, xflags);
  MTS_LAUNCH_CHECK("mts_lstm_fwd(mfma)");
  return MTS_OK;
}

int mts_lstm_mfma_bwd(hipStream_t st, int B, int L, int H, int ndir, const float* w_hh, const int32_t* lengths, const void* out, const void* gates,
                      const float* cells, const void* dout, void* dxproj, void* hprev, void* ws) {
  bf16_t* wT = (bf16_t*)((char*)ws + align_up((size_t)ndir * 4 * H * H * 2, 256));
  hipLaunchKernelGGL(cast_transpose_bf16_kernel, dim3(ceil_div(H, 32), ceil_div(4 * H, 32), ndir), dim3(32, 8), 0, st, w_hh, (bf16_t*)nullptr, wT,
                     4 * H, H);
  constexpr int KS = 8;
  const size_t lds = 2 * LM_DOCS * (4 * H + 8) * 2 + (size_t)KS * LM_BWD_LT * KS * 1024;
  auto k = lstm_bwd_mfma_kernel<KS, LM_BWD_RT, LM_BWD_LT>;
  static std::atomic<bool> attr{false};
  if (!attr) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      mts_set_error("lstm_mfma_bwd: cannot reserve %zu bytes of LDS", lds);
      return MTS_ERR_LAUNCH;
    }
    attr = true;
  }
  hipLaunchKernelGGL(k, dim3(ceil_div(B, LM_DOCS), ndir), dim3(KS * 64), lds, st, B, L, ndir, (const bf16_t*)wT, lengths, (const bf16_t*)out,
                     (const bf16_t*)gates, cells, (const bf16_t*)dout, (bf16_t*)dxproj, (bf16_t*)hprev);
  MTS_LAUNCH_CHECK("mts_lstm_bwd(mfma)");
  return MTS_OK;
}
