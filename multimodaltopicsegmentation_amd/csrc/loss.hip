// Tagger tail: masked loss (+ its gradient wrt the scores) and greedy decode.
// Small tensors ([B, L, n_out]) but transcendental-heavy: 2048 rows per workgroup, per-workgroup partial sums added in a
// fixed order by a one-wave kernel (bitwise reproducible).
#include <algorithm>
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

#include "loss_elems.h"

__global__ __launch_bounds__(256) void tagger_loss_kernel(int kind, int B, int L, int Lt, int n_out, const float* __restrict__ scores,
                                                           const float* __restrict__ targets, const int32_t* __restrict__ lengths,
                                                           float alpha, float gamma, float* __restrict__ loss_out, float* __restrict__ dscores,
                                                           float* __restrict__ partial, const int32_t* __restrict__ row_src, int n_rows) {
  // gridDim.x > 1: every workgroup recomputes the (cheap) row count, handles an interleaved share of the rows and leaves
  // its partial loss sum in partial[blockIdx.x]; tagger_loss_final_kernel adds them in a fixed order.
  // packed batches (row_src != NULL): scores/dscores have n_rows rows, row r is sentence row_src[r] = b*L + i (always valid)
  __shared__ float red[16];
  const int N = row_src ? n_rows : B * L;
  constexpr int U = 2;                          // rows per thread per batch: their loads are issued together
  const int step = blockDim.x * U;
  // pass 1: number of rows that are averaged
  float cnt = 0.f;
  if (kind == MTS_LOSS_CE) {                    // ignore_index = -1 (CRF.py:298)
    for (int base = threadIdx.x; base < N; base += step) {
      float y[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int r = base + u * blockDim.x;
        const int src = (r < N && row_src) ? row_src[r] : r;
        y[u] = (r < N) ? targets[(size_t)(src / L) * Lt + src % L] : -1.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) cnt += (y[u] != -1.f) ? 1.f : 0.f;
    }
  } else {                                      // un-pad loop (CRF.py:348-350): rows i < len_b
    for (int b = threadIdx.x; b < B; b += blockDim.x) cnt += (float)(lengths ? min(max(lengths[b], 0), L) : L);
  }
  cnt = block_sum_1024(cnt, red);
  const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
  float acc = 0.f;
  for (int base = blockIdx.x * step + threadIdx.x; base < N; base += gridDim.x * step) {
    float y[U], x0[U], x1[U];
    int len[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = base + u * blockDim.x;
      const bool in = r < N;
      const int src = (in && row_src) ? row_src[r] : r;
      const int b = in ? src / L : 0, i = in ? src % L : 0;
      y[u] = in ? targets[(size_t)b * Lt + i] : -1.f;
      len[u] = in ? (lengths ? lengths[b] : L) : 0;
      if (kind == MTS_LOSS_CE) { x0[u] = (in && n_out == 2) ? scores[(size_t)r * 2] : 0.f; x1[u] = (in && n_out == 2) ? scores[(size_t)r * 2 + 1] : 0.f; }
      else { x0[u] = in ? scores[r] : 0.f; x1[u] = 0.f; }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int r = base + u * blockDim.x;
      if (r >= N) continue;
      const int i = (row_src ? row_src[r] : r) % L;
      if (kind == MTS_LOSS_CE && n_out > 2) {
        // tagset_size 3 or 4 (nn.CrossEntropyLoss over n_out classes, CRF.py:298,354): the general form, straight from memory
        float xs[4], gs[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < n_out; ++c) xs[c] = scores[(size_t)r * n_out + c];
        if (y[u] != -1.f) {
          float m = xs[0];
          for (int c = 1; c < n_out; ++c) m = fmaxf(m, xs[c]);
          float se = 0.f;
          for (int c = 0; c < n_out; ++c) se += expf(xs[c] - m);
          const float lse = m + logf(se);
          const int t = (int)y[u];
          for (int c = 0; c < n_out; ++c) {
            if (c == t) acc += lse - xs[c];
            gs[c] = (expf(xs[c] - lse) - (c == t ? 1.f : 0.f)) * inv;
          }
        }
        if (dscores) for (int c = 0; c < n_out; ++c) dscores[(size_t)r * n_out + c] = gs[c];
      } else if (kind == MTS_LOSS_CE) {
        float g0 = 0.f, g1 = 0.f;
        if (y[u] != -1.f) {
          const float m = fmaxf(x0[u], x1[u]);
          const float lse = m + logf(expf(x0[u] - m) + expf(x1[u] - m));
          const int t = (int)y[u];
          acc += lse - (t == 0 ? x0[u] : x1[u]);
          const float p0 = expf(x0[u] - lse), p1 = expf(x1[u] - lse);
          g0 = (p0 - (t == 0 ? 1.f : 0.f)) * inv;
          g1 = (p1 - (t == 1 ? 1.f : 0.f)) * inv;
        }
        if (dscores) { dscores[(size_t)r * 2] = g0; dscores[(size_t)r * 2 + 1] = g1; }
      } else {
        float g = 0.f;
        if (i < len[u]) {
          float gr;
          acc += (kind == MTS_LOSS_FOCAL) ? focal_elem(x0[u], y[u], alpha, gamma, gr) : bce_elem(x0[u], y[u], gr);
          g = gr * inv;
        }
        if (dscores) dscores[r] = g;
      }
    }
  }
  acc = block_sum_1024(acc, red);
  if (threadIdx.x == 0) {
    if (gridDim.x == 1) { loss_out[0] = acc * inv; loss_out[1] = cnt; }
    else { partial[blockIdx.x] = acc; if (blockIdx.x == 0) loss_out[1] = cnt; }
  }
}

__global__ __launch_bounds__(64) void tagger_loss_final_kernel(int nblocks, const float* __restrict__ partial, float* __restrict__ loss_out) {
  float s = 0.f;
  for (int b = threadIdx.x; b < nblocks; b += 64) s += partial[b];
  s = wave_sum(s);
  if (threadIdx.x == 0) { const float cnt = loss_out[1]; loss_out[0] = cnt > 0.f ? s * (1.f / cnt) : 0.f; }
}

__global__ __launch_bounds__(256) void greedy_decode_kernel(int B, int L, int n_out, const float* __restrict__ scores,
                                                            const int32_t* __restrict__ lengths, float threshold, uint8_t* __restrict__ tags) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= B * L) return;
  const int b = r / L, i = r % L;
  float p;
  if (n_out == 1) p = sigmoid_f(scores[r]);                                   // CRF.py:365
  else if (n_out > 2) {                                                       // softmax over 3 or 4 tags, class 1 (CRF.py:367)
    float m = scores[(size_t)r * n_out];
    for (int c = 1; c < n_out; ++c) m = fmaxf(m, scores[(size_t)r * n_out + c]);
    float se = 0.f;
    for (int c = 0; c < n_out; ++c) se += expf(scores[(size_t)r * n_out + c] - m);
    p = expf(scores[(size_t)r * n_out + 1] - m) / se;
  } else {                                                                    // softmax(...)[..., 1]  CRF.py:367
    const float x0 = scores[(size_t)r * 2], x1 = scores[(size_t)r * 2 + 1];
    const float m = fmaxf(x0, x1);
    const float e0 = expf(x0 - m), e1 = expf(x1 - m);
    p = e1 / (e0 + e1);
  }
  const bool valid = i < (lengths ? lengths[b] : L);
  tags[r] = (valid && p > threshold) ? 1 : 0;
}

extern "C" size_t mts_tagger_loss_workspace(int B, int L) { return (size_t)std::min(1024, ceil_div(B * L, 256 * 2)) * sizeof(float); }

extern "C" int mts_tagger_loss(void* stream, int loss_kind, int B, int L, int Lt, int n_out, const float* scores, const float* targets,
                               const int32_t* lengths, float alpha, float gamma, float* loss_out, float* dscores, void* workspace,
                               size_t workspace_bytes, const int32_t* row_src, int n_rows) {
  MTS_CHECK_ARG(B > 0 && L > 0 && Lt >= L && scores && targets && loss_out, "mts_tagger_loss: bad arguments");
  MTS_CHECK_ARG(loss_kind == MTS_LOSS_CE || loss_kind == MTS_LOSS_BCE || loss_kind == MTS_LOSS_FOCAL,
                "Choose one of CrossEntropy or BinaryCrossEntropy as loss function");   /* models/CRF.py:312 */
  MTS_CHECK_ARG((loss_kind == MTS_LOSS_CE) ? (n_out >= 2 && n_out <= 4) : n_out == 1, "mts_tagger_loss: n_out=%d does not match the loss kind", n_out);
  // 256 threads x 8 rows per workgroup; one workgroup (no workspace needed) up to 2048 rows
  MTS_CHECK_ARG(!row_src || (n_rows > 0 && n_rows <= B * L), "mts_tagger_loss: packed form needs 0 < n_rows <= B*L");
  const int nblocks = (int)std::min<size_t>(ceil_div(row_src ? n_rows : B * L, 256 * 2), workspace ? workspace_bytes / sizeof(float) : 1);
  const int grid = std::max(1, std::min(nblocks, 1024));
  hipLaunchKernelGGL(tagger_loss_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, loss_kind, B, L, Lt, n_out, scores, targets, lengths,
                     alpha, gamma, loss_out, dscores, (float*)workspace, row_src, n_rows);
  if (grid > 1) hipLaunchKernelGGL(tagger_loss_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, grid, (const float*)workspace, loss_out);
  MTS_LAUNCH_CHECK("mts_tagger_loss");
  return MTS_OK;
}

extern "C" int mts_greedy_decode(void* stream, int B, int L, int n_out, const float* scores, const int32_t* lengths, float threshold,
                                 uint8_t* tags_out) {
  MTS_CHECK_ARG(B > 0 && L > 0 && n_out >= 1 && n_out <= 4 && scores && tags_out, "mts_greedy_decode: bad arguments");
  hipLaunchKernelGGL(greedy_decode_kernel, dim3(ceil_div(B * L, 256)), dim3(256), 0, (hipStream_t)stream, B, L, n_out, scores, lengths,
                     threshold, tags_out);
  MTS_LAUNCH_CHECK("mts_greedy_decode");
  return MTS_OK;
}
