// Optimizer steps over the flat fp32 parameter buffer + error plumbing of the C ABI.
// HBM-bound streaming kernels: 16-byte accesses, grid-stride, optional fused bf16 weight copy so the next
// step's GEMMs never re-read the fp32 master weights.
#include <stdarg.h>
#include <algorithm>
#include <math.h>
#include "common.h"

static thread_local char g_err[512] = "";
void mts_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* mts_last_error(void) { return g_err; }
extern "C" const char* mts_version(void) { return "mts-hip 1 gfx950"; }

// torch.optim.Adam (no amsgrad, no weight decay): m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
__global__ __launch_bounds__(256) void adam_kernel(size_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt,
                                                   float gscale, bf16_t* __restrict__ copy) {
  const size_t stride = (size_t)gridDim.x * 256 * 4;
  const float step_size = lr / bc1;
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 3 < n) {
      float pv[4], gv[4], mv[4], vv[4];
      load4<float>(p + i, pv); load4<float>(g + i, gv); load4<float>(m + i, mv); load4<float>(v + i, vv);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float gr = gv[j] * gscale;
        mv[j] = mv[j] + (1.f - b1) * (gr - mv[j]);                 // lerp form used by torch
        vv[j] = b2 * vv[j] + (1.f - b2) * gr * gr;
        const float denom = sqrtf(vv[j]) / bc2_sqrt + eps;
        pv[j] = pv[j] - step_size * (mv[j] / denom);
      }
      store4<float>(p + i, pv); store4<float>(m + i, mv); store4<float>(v + i, vv);
      if (copy) store4<bf16_t>(copy + i, pv);
    } else {
      for (size_t k = i; k < n; ++k) {
        const float gr = g[k] * gscale;
        const float mm = m[k] + (1.f - b1) * (gr - m[k]);
        const float vv = b2 * v[k] + (1.f - b2) * gr * gr;
        m[k] = mm; v[k] = vv;
        const float pp = p[k] - step_size * (mm / (sqrtf(vv) / bc2_sqrt + eps));
        p[k] = pp;
        if (copy) copy[k] = (bf16_t)pp;
      }
    }
  }
}

// torch.optim.SGD(momentum, weight_decay, dampening 0, no nesterov)
__global__ __launch_bounds__(256) void sgd_kernel(size_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, float lr,
                                                  float mom, float wd, int first, float gscale, bf16_t* __restrict__ copy) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
    float gr = g[i] * gscale + wd * p[i];
    float b = first ? gr : mom * buf[i] + gr;
    buf[i] = b;
    const float pp = p[i] - lr * b;
    p[i] = pp;
    if (copy) copy[i] = (bf16_t)pp;
  }
}

extern "C" int mts_adam_step(void* stream, size_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float lr, float beta1,
                             float beta2, float eps, int step, float grad_scale, void* bf16_copy) {
  MTS_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && step >= 1, "mts_adam_step: bad arguments");
  if (n == 0) return MTS_OK;
  const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
  const int blocks = (int)std::min<size_t>(2048, (n / 4 + 255) / 256 + 1);
  hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps,
                     (float)bc1, (float)sqrt(bc2), grad_scale, (bf16_t*)bf16_copy);
  MTS_LAUNCH_CHECK("mts_adam_step");
  return MTS_OK;
}

extern "C" int mts_sgd_step(void* stream, size_t n, float* param, const float* grad, float* momentum_buf, float lr, float momentum,
                            float weight_decay, int first_step, float grad_scale, void* bf16_copy) {
  MTS_CHECK_ARG(param && grad && momentum_buf, "mts_sgd_step: bad arguments");
  if (n == 0) return MTS_OK;
  const int blocks = (int)std::min<size_t>(2048, (n + 255) / 256);
  hipLaunchKernelGGL(sgd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, param, grad, momentum_buf, lr, momentum, weight_decay,
                     first_step, grad_scale, (bf16_t*)bf16_copy);
  MTS_LAUNCH_CHECK("mts_sgd_step");
  return MTS_OK;
}

// x *= scale (fp32, any n): the loss-gradient weight of token-weighted data parallelism (trainer.py: a rank's d loss / d scores
// is multiplied by world * n_r / sum n_r before the SUM all-reduce)
__global__ __launch_bounds__(256) void scale_kernel(size_t n, float* __restrict__ x, float scale) {
  const size_t stride = (size_t)gridDim.x * 256;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) x[i] *= scale;
}
extern "C" int mts_scale(void* stream, size_t n, float* x, float scale) {
  MTS_CHECK_ARG(x || n == 0, "mts_scale: null pointer");
  if (n == 0 || scale == 1.0f) return MTS_OK;
  const int blocks = (int)std::min<size_t>(1024, (n + 255) / 256);
  hipLaunchKernelGGL(scale_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, x, scale);
  MTS_LAUNCH_CHECK("mts_scale");
  return MTS_OK;
}
