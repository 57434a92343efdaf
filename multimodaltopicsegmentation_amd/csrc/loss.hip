// Tagger tail: masked loss (+ its gradient wrt the scores) and greedy decode.
// Tiny tensors ([B, L, n_out]): one workgroup, everything in one launch, deterministic tree reductions.
#include "common.h"

__device__ __forceinline__ float block_sum_1024(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float t = (threadIdx.x < (blockDim.x >> 6)) ? red[threadIdx.x] : 0.f;
  if (wave == 0) {
    t = wave_sum(t);
    if (lane == 0) red[0] = t;
  }
  __syncthreads();
  return red[0];
}

// models/focal_loss.py:38-57 for one element; returns loss, writes dloss/dx
__device__ __forceinline__ float focal_elem(float x, float y, float alpha, float gamma, float& grad) {
  const float p = sigmoid_f(x);
  // BCE with logits, stable: max(x,0) - x*y + log1p(exp(-|x|))
  const float ce = fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
  const float pt = p * y + (1.f - p) * (1.f - y);
  const float om = 1.f - pt;
  float mod, dmod;   // (1-pt)^gamma and its derivative wrt pt
  if (gamma == 2.f) { mod = om * om; dmod = -2.f * om; }
  else if (gamma == 0.f) { mod = 1.f; dmod = 0.f; }
  else { mod = powf(om, gamma); dmod = (om > 0.f) ? -gamma * powf(om, gamma - 1.f) : 0.f; }
  const float at = (alpha >= 0.f) ? alpha * y + (1.f - alpha) * (1.f - y) : 1.f;
  // d ce/dx = p - y ; d pt/dx = (2y-1) p (1-p)
  const float dpt = (2.f * y - 1.f) * p * (1.f - p);
  grad = at * ((p - y) * mod + ce * dmod * dpt);
  return at * ce * mod;
}

// nn.BCELoss(sigmoid(x), y) with the log clamp at -100 (models/CRF.py:303, :345-352)
__device__ __forceinline__ float bce_elem(float x, float y, float& grad) {
  const float p = sigmoid_f(x);
  const float lp = fmaxf(logf(p), -100.f), l1p = fmaxf(logf(1.f - p), -100.f);
  // gradient through the clamps as autograd sees them: d/dp [-y log p] = -y/p unless clamped
  const float dlp = (logf(p) > -100.f) ? 1.f / p : 0.f;
  const float dl1p = (logf(1.f - p) > -100.f) ? -1.f / (1.f - p) : 0.f;
  grad = -(y * dlp + (1.f - y) * dl1p) * p * (1.f - p);
  return -(y * lp + (1.f - y) * l1p);
}

__global__ __launch_bounds__(1024) void tagger_loss_kernel(int kind, int B, int L, int Lt, int n_out, const float* __restrict__ scores,
                                                           const float* __restrict__ targets, const int32_t* __restrict__ lengths,
                                                           float alpha, float gamma, float* __restrict__ loss_out, float* __restrict__ dscores) {
  __shared__ float red[16];
  const int N = B * L;
  // pass 1: number of rows that are averaged
  float cnt = 0.f;
  for (int r = threadIdx.x; r < N; r += blockDim.x) {
    const int b = r / L, i = r % L;
    if (kind == MTS_LOSS_CE) cnt += (targets[(size_t)b * Lt + i] != -1.f) ? 1.f : 0.f;            // ignore_index = -1 (CRF.py:298)
    else cnt += (i < (lengths ? lengths[b] : L)) ? 1.f : 0.f;                                        // un-pad loop (CRF.py:348-350)
  }
  cnt = block_sum_1024(cnt, red);
  const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
  float acc = 0.f;
  for (int r = threadIdx.x; r < N; r += blockDim.x) {
    const int b = r / L, i = r % L;
    const float y = targets[(size_t)b * Lt + i];
    if (kind == MTS_LOSS_CE) {
      const float x0 = scores[(size_t)r * 2], x1 = scores[(size_t)r * 2 + 1];
      float g0 = 0.f, g1 = 0.f;
      if (y != -1.f) {
        const float m = fmaxf(x0, x1);
        const float lse = m + logf(expf(x0 - m) + expf(x1 - m));
        const int t = (int)y;
        acc += lse - (t == 0 ? x0 : x1);
        const float p0 = expf(x0 - lse), p1 = expf(x1 - lse);
        g0 = (p0 - (t == 0 ? 1.f : 0.f)) * inv;
        g1 = (p1 - (t == 1 ? 1.f : 0.f)) * inv;
      }
      if (dscores) { dscores[(size_t)r * 2] = g0; dscores[(size_t)r * 2 + 1] = g1; }
    } else {
      const bool valid = i < (lengths ? lengths[b] : L);
      float g = 0.f;
      if (valid) {
        const float x = scores[r];
        float gr;
        acc += (kind == MTS_LOSS_FOCAL) ? focal_elem(x, y, alpha, gamma, gr) : bce_elem(x, y, gr);
        g = gr * inv;
      }
      if (dscores) dscores[r] = g;
    }
  }
  acc = block_sum_1024(acc, red);
  if (threadIdx.x == 0) { loss_out[0] = acc * inv; loss_out[1] = cnt; }
}

__global__ __launch_bounds__(256) void greedy_decode_kernel(int B, int L, int n_out, const float* __restrict__ scores,
                                                            const int32_t* __restrict__ lengths, float threshold, uint8_t* __restrict__ tags) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= B * L) return;
  const int b = r / L, i = r % L;
  float p;
  if (n_out == 1) p = sigmoid_f(scores[r]);                                   // CRF.py:365
  else {                                                                      // softmax(...)[..., 1]  CRF.py:367
    const float x0 = scores[(size_t)r * 2], x1 = scores[(size_t)r * 2 + 1];
    const float m = fmaxf(x0, x1);
    const float e0 = expf(x0 - m), e1 = expf(x1 - m);
    p = e1 / (e0 + e1);
  }
  const bool valid = i < (lengths ? lengths[b] : L);
  tags[r] = (valid && p > threshold) ? 1 : 0;
}

extern "C" int mts_tagger_loss(void* stream, int loss_kind, int B, int L, int Lt, int n_out, const float* scores, const float* targets,
                               const int32_t* lengths, float alpha, float gamma, float* loss_out, float* dscores) {
  MTS_CHECK_ARG(B > 0 && L > 0 && Lt >= L && scores && targets && loss_out, "mts_tagger_loss: bad arguments");
  MTS_CHECK_ARG(loss_kind == MTS_LOSS_CE || loss_kind == MTS_LOSS_BCE || loss_kind == MTS_LOSS_FOCAL,
                "Choose one of CrossEntropy or BinaryCrossEntropy as loss function");   /* models/CRF.py:312 */
  MTS_CHECK_ARG((loss_kind == MTS_LOSS_CE) ? n_out == 2 : n_out == 1, "mts_tagger_loss: n_out=%d does not match the loss kind", n_out);
  hipLaunchKernelGGL(tagger_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, loss_kind, B, L, Lt, n_out, scores, targets, lengths,
                     alpha, gamma, loss_out, dscores);
  MTS_LAUNCH_CHECK("mts_tagger_loss");
  return MTS_OK;
}

extern "C" int mts_greedy_decode(void* stream, int B, int L, int n_out, const float* scores, const int32_t* lengths, float threshold,
                                 uint8_t* tags_out) {
  MTS_CHECK_ARG(B > 0 && L > 0 && (n_out == 1 || n_out == 2) && scores && tags_out, "mts_greedy_decode: bad arguments");
  hipLaunchKernelGGL(greedy_decode_kernel, dim3(ceil_div(B * L, 256)), dim3(256), 0, (hipStream_t)stream, B, L, n_out, scores, lengths,
                     threshold, tags_out);
  MTS_LAUNCH_CHECK("mts_greedy_decode");
  return MTS_OK;
}
