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

// one thread per document (C^2 = 16 flops per step: latency-bound on the dependent chain, not on lanes)
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
  hipLaunchKernelGGL(crf_nll_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, st, B, L, C, feats, tags, Lt, lengths, trans, dfeats, alphas, part);
  const int blocks = dfeats ? std::min(256, ceil_div(B * L * C, 256)) : 1;
  hipLaunchKernelGGL(crf_finish_kernel, dim3(blocks), dim3(256), 0, st, B, L, C, (const float*)part, loss_out, dfeats, dtrans);
  MTS_LAUNCH_CHECK("mts_crf_nll");
  return MTS_OK;
}

extern "C" int mts_crf_viterbi(void* stream, int B, int L, int C, const float* feats, const int32_t* lengths, const float* trans,
                               float* best_score, int32_t* paths, int32_t* bp_ws) {
  MTS_CHECK_ARG(B > 0 && L > 0 && C >= 3 && C <= CRF_MAXC, "mts_crf_viterbi: bad shape (C must be in 3..8)");
  MTS_CHECK_ARG(feats && trans && best_score && paths && bp_ws, "mts_crf_viterbi: null pointer");
  hipLaunchKernelGGL(crf_viterbi_kernel, dim3(ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, B, L, C, feats, lengths, trans, best_score, paths, bp_ws);
  MTS_LAUNCH_CHECK("mts_crf_viterbi");
  return MTS_OK;
}
