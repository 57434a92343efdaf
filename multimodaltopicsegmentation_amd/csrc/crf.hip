// CRF head: negative log-likelihood (forward algorithm + gold path) with its gradients, and Viterbi decode.
// models/CRF.py:98-240.  C = num_tags + 2 is tiny (4): one wavefront per document walks the L dependent
// steps with the C x C transition table in registers; the gradient is the exact reverse sweep (beta
// recursion -> marginals), so nothing but the alphas is stored.
#include <algorithm>
#include "common.h"

#define CRF_MAXC 8
#define CRF_IMPOSSIBLE (-1e4f)

__device__ __forceinline__ float lse_arr(const float* v, int n) {
  float m = v[0];
  for (int i = 1; i < n; ++i) m = fmaxf(m, v[i]);
  float s = 0.f;
  for (int i = 0; i < n; ++i) s += expf(v[i] - m);
  return m + logf(s);
}


// ------------------------------------------------------------------------------------------------
// C == 4 (binary tagset + START/STOP: every configuration of the reference): one QUAD of lanes per document, lane i
// owns tag i.  The C x C table lives in 8 registers per lane (its row and its column), the alpha/beta vectors are
// exchanged with DPP quad broadcasts, no memory on the dependent chain except the (prefetched) emission row.
// ------------------------------------------------------------------------------------------------
template <int J> __device__ __forceinline__ float quad_bcast(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), J * 0x55, 0xf, 0xf, false));
}
__device__ __forceinline__ float lse4(float v0, float v1, float v2, float v3) {
  const float m = fmaxf(fmaxf(fmaxf(v0, v1), v2), v3);
  const float s = ((expf(v0 - m) + expf(v1 - m)) + expf(v2 - m)) + expf(v3 - m);
  return m + logf(s);
}

// Forward-backward in the SCALED PROBABILITY domain.  The first version of this kernel walked the log domain: four expf and a logf on every
// forward step, five expf and an lse4 on every backward step, ~250 instructions per step of a lone wave -- 299 us per training step at
// 64 x 256 (8 % of the BiLSTM + CRF step).  With p_t = softmax(alpha_t) carried instead of alpha_t (the textbook scaled recursion):
//     u_i = (sum_j exp(T[i][j]) p_t(j)) * e_t(i),  e_t(i) = exp(f_t(i) - max_i f_t(i));   z_t = sum_i u_i;   p_{t+1} = u / z_t
//     log Z = sum_t (log z_t + max_i f_t(i)) + log(sum_i p_n(i) exp(T[stop][i]))
//     bhat_n(i) = exp(T[stop][i]) / sum_i p_n(i) exp(T[stop][i]);   bhat_t(j) = sum_i exp(T[i][j]) e_t(i) bhat_{t+1}(i) / z_t
//     P(y_t = i) = p_{t+1}(i) bhat_{t+1}(i);   P(y_{t-1} = j, y_t = i) = p_t(j) exp(T[i][j]) e_t(i) bhat_{t+1}(i) / z_t
// a step is four multiply-adds, one exp2 of the emission row (independent of the recursion), a quad sum and a reciprocal: ~50 instructions.
// exp(-1e4) is exactly 0, as it is (after the max shift) in the log-domain sums: IMPOSSIBLE transitions behave as before.  Same results to
// fp32 rounding (tests/test_gpu_kernels.py::test_crf_nll_viterbi, ::test_crf_long_documents; fixture g5).  The saved state is p_t (in the
// `alphas` buffer); z_t is recomputed from it in the reverse sweep.
__device__ __forceinline__ float quad_sum(float v) { return (quad_bcast<0>(v) + quad_bcast<1>(v)) + (quad_bcast<2>(v) + quad_bcast<3>(v)); }
__device__ __forceinline__ float quad_max(float v) { return fmaxf(fmaxf(quad_bcast<0>(v), quad_bcast<1>(v)), fmaxf(quad_bcast<2>(v), quad_bcast<3>(v))); }

__global__ __launch_bounds__(64) void crf_nll4_kernel(int B, int L, const float* __restrict__ feats, const float* __restrict__ tags, int Lt,
                                                      const int32_t* __restrict__ lengths, const float* __restrict__ trans,
                                                      float* __restrict__ dfeats, float* __restrict__ alphas, float* __restrict__ part) {
  constexpr int C = 4, start = 2, stop = 3;
  const int i = threadIdx.x & 3;
  const int b = blockIdx.x * 16 + (threadIdx.x >> 2);
  if (b >= B) return;                                   // whole quads leave together
  const int n = lengths ? max(min(lengths[b], L), 0) : L;
  float Er[4], Ec[4];                                   // exp of my row (to me from j) and my column (from me to j) of the transition table
#pragma unroll
  for (int j = 0; j < 4; ++j) { Er[j] = expf(trans[i * C + j]); Ec[j] = expf(trans[j * C + i]); }
  const float Estop = expf(trans[stop * C + i]);
  const float* f = feats + (size_t)b * L * C;
  float* al = alphas + (size_t)b * (L + 1) * C;
  const float* tg = tags + (size_t)b * Lt;
  float p = (i == start) ? 1.f : 0.f;
  al[i] = p;
  float logscale = 0.f;
  float fq[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) fq[d] = d < n ? f[d * C + i] : 0.f;
  for (int t = 0; t < n; ++t) {
    const float ft = fq[0];
    fq[0] = fq[1]; fq[1] = fq[2]; fq[2] = fq[3];
    fq[3] = (t + 4 < n) ? f[(t + 4) * C + i] : 0.f;      // emission rows four steps ahead
    const float sgm = ((Er[0] * quad_bcast<0>(p) + Er[1] * quad_bcast<1>(p)) + Er[2] * quad_bcast<2>(p)) + Er[3] * quad_bcast<3>(p);
    // the shift is the largest emission among the REACHABLE tags (a tag nobody can move to -- START -- may carry the row's maximum: shifted
    // by it every reachable term could underflow and z with them)
    const float m = quad_max(sgm > 0.f ? ft : -INFINITY);
    const float e = sgm > 0.f ? expf(ft - m) : 0.f;
    const float u = sgm * e;
    const float z = quad_sum(u);
    p = u / z;
    logscale += logf(z) + m;
    al[(t + 1) * C + i] = p;
  }
  const float zn = quad_sum(p * Estop);
  const float logZ = logscale + logf(zn);
  // gold path score (CRF.py:148-170): the 4 lanes take every 4th step, combined in a fixed order
  float gpart = 0.f;
  for (int t = i; t < n; t += 4) {
    const int y = (int)tg[t], prev = t > 0 ? (int)tg[t - 1] : start;
    gpart += trans[y * C + prev] + f[t * C + y];
  }
  const int last = n > 0 ? (int)tg[n - 1] : start;
  const float gold = ((quad_bcast<0>(gpart) + quad_bcast<1>(gpart)) + (quad_bcast<2>(gpart) + quad_bcast<3>(gpart))) + trans[stop * C + last];
  float* pp = part + (size_t)b * (C * C + 1);
  if (i == 0) pp[C * C] = logZ - gold;
  if (!dfeats) return;
  // reverse sweep: scaled beta recursion -> marginals; the gold path's one-hot counts are subtracted on the fly
  float dT[4] = {0.f, 0.f, 0.f, 0.f};
  float bh = Estop / zn;                                 // bhat_n(i)
  const float dstop = p * bh;                            // P(last tag = i): row `stop`, column i
  float* df = dfeats + (size_t)b * L * C;
  // operands of step t: emission row, p_t and p_{t+1}, the gold tags -- requested two steps ahead
  float f0 = n > 0 ? f[(n - 1) * C + i] : 0.f, f1 = n > 1 ? f[(n - 2) * C + i] : 0.f;
  float a0 = n > 0 ? al[(n - 1) * C + i] : 0.f, a1 = n > 1 ? al[(n - 2) * C + i] : 0.f;
  float pnext = p;                                       // p_{t+1}
  for (int t = n - 1; t >= 0; --t) {
    const float ft = f0, pt = a0;
    f0 = f1; a0 = a1;
    if (t >= 2) { f1 = f[(t - 2) * C + i]; a1 = al[(t - 2) * C + i]; }
    const int y = (int)tg[t], prev = t > 0 ? (int)tg[t - 1] : start;
    const float p0 = quad_bcast<0>(pt), p1 = quad_bcast<1>(pt), p2 = quad_bcast<2>(pt), p3 = quad_bcast<3>(pt);
    const float sgm = ((Er[0] * p0 + Er[1] * p1) + Er[2] * p2) + Er[3] * p3;
    const float m = quad_max(sgm > 0.f ? ft : -INFINITY);       // (the forward step's own arithmetic: the same z, bit for bit)
    const float e = sgm > 0.f ? expf(ft - m) : 0.f;
    const float z = quad_sum(sgm * e);
    const float w = e * bh / z;                          // e_t(i) bhat_{t+1}(i) / z_t
    float marg = pnext * bh;
    dT[0] += p0 * Er[0] * w; dT[1] += p1 * Er[1] * w; dT[2] += p2 * Er[2] * w; dT[3] += p3 * Er[3] * w;
    if (y == i) {
      marg -= 1.f;
      dT[0] -= (prev == 0) ? 1.f : 0.f; dT[1] -= (prev == 1) ? 1.f : 0.f; dT[2] -= (prev == 2) ? 1.f : 0.f; dT[3] -= (prev == 3) ? 1.f : 0.f;
    }
    df[t * C + i] = marg;
    // bhat = (smoothed / filtered probability of the tag): finite whenever the filtered one is not 0; the clamp keeps 0 * bhat = 0 there
    bh = fminf(((Ec[0] * quad_bcast<0>(w) + Ec[1] * quad_bcast<1>(w)) + Ec[2] * quad_bcast<2>(w)) + Ec[3] * quad_bcast<3>(w), 1e30f);
    pnext = pt;
  }
  for (int t = n; t < L; ++t) df[t * C + i] = 0.f;
  const float ds0 = quad_bcast<0>(dstop), ds1 = quad_bcast<1>(dstop), ds2 = quad_bcast<2>(dstop), ds3 = quad_bcast<3>(dstop);
  if (i == stop) {
    dT[0] += ds0; dT[1] += ds1; dT[2] += ds2; dT[3] += ds3;
    dT[0] -= (last == 0) ? 1.f : 0.f; dT[1] -= (last == 1) ? 1.f : 0.f; dT[2] -= (last == 2) ? 1.f : 0.f; dT[3] -= (last == 3) ? 1.f : 0.f;
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) pp[i * C + j] = dT[j];
}

// Viterbi for C == 4: the back-pointers stay in LDS, the back-trace is walked by the quad's first lane.  A step's four back-pointers are
// two bits each and share ONE byte (the quad ORs its lanes' fields together with DPP broadcasts, lane 0 stores it): 16 documents x L bytes, so
// a block of full-length documents fits the LDS up to L = 10 240 -- with a byte per pointer (the first version) a single 2 437-sentence
// document (predict.py: one document per call) needed 156 KB for its block, and the call fell back to the generic kernel below: one lane, four
// dependent global stores per step, 5.7 ms per document.
__global__ __launch_bounds__(64) void crf_viterbi4_kernel(int B, int L, const float* __restrict__ feats, const int32_t* __restrict__ lengths,
                                                          const float* __restrict__ trans, float* __restrict__ best_score, int32_t* __restrict__ paths) {
  extern __shared__ unsigned char bp_lds[];            // [16 documents][L]
  constexpr int C = 4, start = 2, stop = 3;
  const int i = threadIdx.x & 3, q = threadIdx.x >> 2;
  const int b = blockIdx.x * 16 + q;
  if (b >= B) return;
  const int n = lengths ? max(min(lengths[b], L), 0) : L;
  float Tr[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) Tr[j] = trans[i * C + j];
  const float* f = feats + (size_t)b * L * C;
  unsigned char* bp = bp_lds + (size_t)q * L;
  float m = (i == start) ? 0.f : CRF_IMPOSSIBLE;
  // emission rows four steps ahead of the step that needs them: the row is the only memory operand of a step, and a load issued one step
  // ahead (a step is ~20 instructions) had not landed when the step wanted it
  float fq[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) fq[d] = d < n ? f[d * C + i] : 0.f;
  for (int t = 0; t < n; ++t) {
    const float ft = fq[0];
    fq[0] = fq[1]; fq[1] = fq[2]; fq[2] = fq[3];
    fq[3] = (t + 4 < n) ? f[(t + 4) * C + i] : 0.f;
    float best = quad_bcast<0>(m) + Tr[0];
    int arg = 0;
    float v = quad_bcast<1>(m) + Tr[1]; if (v > best) { best = v; arg = 1; }
    v = quad_bcast<2>(m) + Tr[2]; if (v > best) { best = v; arg = 2; }
    v = quad_bcast<3>(m) + Tr[3]; if (v > best) { best = v; arg = 3; }
    const int fld = arg << (2 * i);
    const int packed = (__builtin_amdgcn_update_dpp(0, fld, 0 * 0x55, 0xf, 0xf, false) | __builtin_amdgcn_update_dpp(0, fld, 1 * 0x55, 0xf, 0xf, false)) |
                       (__builtin_amdgcn_update_dpp(0, fld, 2 * 0x55, 0xf, 0xf, false) | __builtin_amdgcn_update_dpp(0, fld, 3 * 0x55, 0xf, 0xf, false));
    if (i == 0) bp[t] = (unsigned char)packed;
    m = best + ft;
  }
  const float vs = m + trans[stop * C + i];
  float best = quad_bcast<0>(vs);
  int tag = 0;
  float v = quad_bcast<1>(vs); if (v > best) { best = v; tag = 1; }
  v = quad_bcast<2>(vs); if (v > best) { best = v; tag = 2; }
  v = quad_bcast<3>(vs); if (v > best) { best = v; tag = 3; }
  int32_t* p = paths + (size_t)b * L;
  for (int t = n + i; t < L; t += 4) p[t] = -1;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  if (i == 0) {
    best_score[b] = best;
    for (int t = n - 1; t >= 0; --t) {
      p[t] = tag;
      tag = (bp[t] >> (2 * tag)) & 3;
    }
  }
}

// generic C (3..8): one thread per document (C^2 = 16 flops per step: latency-bound on the dependent chain, not on lanes)
__global__ __launch_bounds__(64) void crf_nll_kernel(int B, int L, int C, const float* __restrict__ feats, const float* __restrict__ tags, int Lt,
                                                     const int32_t* __restrict__ lengths, const float* __restrict__ trans,
                                                     float* __restrict__ dfeats, float* __restrict__ alphas /*[B][L+1][C]*/,
                                                     float* __restrict__ part /*[B][C*C + 1]*/) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const int start = C - 2, stop = C - 1;
  const int n = lengths ? min(lengths[b], L) : L;
  float T[CRF_MAXC][CRF_MAXC];
  for (int i = 0; i < C; ++i)
    for (int j = 0; j < C; ++j) T[i][j] = trans[i * C + j];
  const float* f = feats + (size_t)b * L * C;
  float* al = alphas + (size_t)b * (L + 1) * C;
  float a[CRF_MAXC], tmp[CRF_MAXC];
  for (int i = 0; i < C; ++i) { a[i] = (i == start) ? 0.f : CRF_IMPOSSIBLE; al[i] = a[i]; }
  // forward algorithm (CRF.py:218-240); masked steps leave the scores untouched
  for (int t = 0; t < n; ++t) {
    float nw[CRF_MAXC];
    for (int i = 0; i < C; ++i) {
      for (int j = 0; j < C; ++j) tmp[j] = a[j] + T[i][j] + f[t * C + i];
      nw[i] = lse_arr(tmp, C);
    }
    for (int i = 0; i < C; ++i) { a[i] = nw[i]; al[(t + 1) * C + i] = nw[i]; }
  }
  for (int i = 0; i < C; ++i) tmp[i] = a[i] + T[stop][i];
  const float logZ = lse_arr(tmp, C);
  // gold path (CRF.py:148-170)
  float gold = 0.f;
  int prev = start;
  const float* tg = tags + (size_t)b * Lt;
  for (int t = 0; t < n; ++t) {
    const int y = (int)tg[t];
    gold += T[y][prev] + f[t * C + y];
    prev = y;
  }
  gold += T[stop][prev];
  float* pp = part + (size_t)b * (C * C + 1);
  pp[C * C] = logZ - gold;
  if (!dfeats) return;
  // reverse sweep: beta[t][i] = log-sum over continuations after being in tag i at step t
  float dT[CRF_MAXC][CRF_MAXC];
  for (int i = 0; i < C; ++i)
    for (int j = 0; j < C; ++j) dT[i][j] = 0.f;
  float beta[CRF_MAXC];
  for (int i = 0; i < C; ++i) {
    beta[i] = T[stop][i];
    dT[stop][i] += expf(a[i] + T[stop][i] - logZ);             // P(last tag = i)
  }
  float* df = dfeats + (size_t)b * L * C;
  for (int t = n - 1; t >= 0; --t) {
    const float* ap = al + t * C;                                // alpha before step t
    float nb[CRF_MAXC];
    for (int j = 0; j < C; ++j) nb[j] = 0.f;
    float marg[CRF_MAXC];
    for (int i = 0; i < C; ++i) {
      marg[i] = expf(al[(t + 1) * C + i] + beta[i] - logZ);      // P(y_t = i)
      for (int j = 0; j < C; ++j) dT[i][j] += expf(ap[j] + T[i][j] + f[t * C + i] + beta[i] - logZ);   // P(y_{t-1}=j, y_t=i)
    }
    for (int j = 0; j < C; ++j) {
      for (int i = 0; i < C; ++i) tmp[i] = T[i][j] + f[t * C + i] + beta[i];
      nb[j] = lse_arr(tmp, C);
    }
    for (int i = 0; i < C; ++i) { df[t * C + i] = marg[i]; beta[i] = nb[i]; }
  }
  for (int t = n; t < L; ++t)
    for (int i = 0; i < C; ++i) df[t * C + i] = 0.f;
  // subtract the gold path's one-hot counts
  prev = start;
  for (int t = 0; t < n; ++t) {
    const int y = (int)tg[t];
    df[t * C + y] -= 1.f;
    dT[y][prev] -= 1.f;
    prev = y;
  }
  dT[stop][prev] -= 1.f;
  for (int i = 0; i < C; ++i)
    for (int j = 0; j < C; ++j) pp[i * C + j] = dT[i][j];
}

// mean over documents; scales dfeats by 1/B
__global__ __launch_bounds__(256) void crf_finish_kernel(int B, int L, int C, const float* __restrict__ part, float* __restrict__ loss_out,
                                                         float* __restrict__ dfeats, float* __restrict__ dtrans) {
  const float inv = 1.f / (float)B;
  if (blockIdx.x == 0) {
    const int e = threadIdx.x;
    if (e <= C * C) {
      float s = 0.f;
      for (int b = 0; b < B; ++b) s += part[(size_t)b * (C * C + 1) + e];
      if (e == C * C) loss_out[0] = s * inv;
      else if (dtrans) dtrans[e] = s * inv;
    }
  }
  if (dfeats) {
    const size_t n = (size_t)B * L * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dfeats[i] *= inv;
  }
}

// Viterbi (CRF.py:172-216): first-max tie break as torch.max(dim=-1)
__global__ __launch_bounds__(64) void crf_viterbi_kernel(int B, int L, int C, const float* __restrict__ feats, const int32_t* __restrict__ lengths,
                                                         const float* __restrict__ trans, float* __restrict__ best_score, int32_t* __restrict__ paths,
                                                         int32_t* __restrict__ bps) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  const int start = C - 2, stop = C - 1;
  const int n = lengths ? min(lengths[b], L) : L;
  float T[CRF_MAXC][CRF_MAXC];
  for (int i = 0; i < C; ++i)
    for (int j = 0; j < C; ++j) T[i][j] = trans[i * C + j];
  const float* f = feats + (size_t)b * L * C;
  int32_t* bp = bps + (size_t)b * L * C;
  float m[CRF_MAXC];
  for (int i = 0; i < C; ++i) m[i] = (i == start) ? 0.f : CRF_IMPOSSIBLE;
  for (int t = 0; t < n; ++t) {
    float nw[CRF_MAXC];
    for (int i = 0; i < C; ++i) {
      float best = m[0] + T[i][0];
      int arg = 0;
      for (int j = 1; j < C; ++j) {
        const float v = m[j] + T[i][j];
        if (v > best) { best = v; arg = j; }
      }
      bp[t * C + i] = arg;
      nw[i] = best + f[t * C + i];
    }
    for (int i = 0; i < C; ++i) m[i] = nw[i];
  }
  float best = m[0] + T[stop][0];
  int tag = 0;
  for (int i = 1; i < C; ++i) {
    const float v = m[i] + T[stop][i];
    if (v > best) { best = v; tag = i; }
  }
  best_score[b] = best;
  int32_t* p = paths + (size_t)b * L;
  for (int t = n - 1; t >= 0; --t) {
    p[t] = tag;
    tag = bp[t * C + tag];
  }
  for (int t = n; t < L; ++t) p[t] = -1;
}

extern "C" size_t mts_crf_workspace(int B, int L, int C) {
  return ((size_t)B * (L + 1) * C + (size_t)B * (C * C + 1)) * sizeof(float);
}

extern "C" int mts_crf_nll(void* stream, int B, int L, int C, const float* feats, const float* tags, int Lt, const int32_t* lengths,
                           const float* trans, float* loss_out, float* dfeats, float* dtrans, float* workspace) {
  MTS_CHECK_ARG(B > 0 && L > 0 && C >= 3 && C <= CRF_MAXC && Lt >= L, "mts_crf_nll: bad shape (C must be in 3..8)");
  MTS_CHECK_ARG(feats && tags && trans && loss_out && workspace, "mts_crf_nll: null pointer");
  MTS_CHECK_ARG(!dtrans || dfeats, "mts_crf_nll: dtrans requires dfeats");
  hipStream_t st = (hipStream_t)stream;
  float* alphas = workspace;
  float* part = workspace + (size_t)B * (L + 1) * C;
  if (C == 4) hipLaunchKernelGGL(crf_nll4_kernel, dim3(ceil_div(B, 16)), dim3(64), 0, st, B, L, feats, tags, Lt, lengths, trans, dfeats, alphas, part);
  else hipLaunchKernelGGL(crf_nll_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, st, B, L, C, feats, tags, Lt, lengths, trans, dfeats, alphas, part);
  const int blocks = dfeats ? std::min(256, ceil_div(B * L * C, 256)) : 1;
  hipLaunchKernelGGL(crf_finish_kernel, dim3(blocks), dim3(256), 0, st, B, L, C, (const float*)part, loss_out, dfeats, dtrans);
  MTS_LAUNCH_CHECK("mts_crf_nll");
  return MTS_OK;
}

extern "C" int mts_crf_viterbi(void* stream, int B, int L, int C, const float* feats, const int32_t* lengths, const float* trans,
                               float* best_score, int32_t* paths, int32_t* bp_ws) {
  MTS_CHECK_ARG(B > 0 && L > 0 && C >= 3 && C <= CRF_MAXC, "mts_crf_viterbi: bad shape (C must be in 3..8)");
  MTS_CHECK_ARG(feats && trans && best_score && paths && bp_ws, "mts_crf_viterbi: null pointer");
  const size_t bp_bytes = (size_t)16 * L;                   // one byte per document and step (four 2-bit back-pointers)
  bool quad = C == 4 && bp_bytes <= 160 * 1024;
  if (quad && bp_bytes > 64 * 1024) {
    static std::atomic<bool> attr{false};
    if (!attr) {
      if (hipFuncSetAttribute((const void*)crf_viterbi4_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) quad = false;
      else attr = true;
    }
  }
  if (quad)
    hipLaunchKernelGGL(crf_viterbi4_kernel, dim3(ceil_div(B, 16)), dim3(64), bp_bytes, (hipStream_t)stream, B, L, feats, lengths, trans, best_score,
                       paths);
  else
    hipLaunchKernelGGL(crf_viterbi_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, B, L, C, feats, lengths, trans, best_score, paths, bp_ws);
  MTS_LAUNCH_CHECK("mts_crf_viterbi");
  return MTS_OK;
}
