// Shared device/host helpers for the gfx950 kernels of libmts_hip.so.
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/mts.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) short s16x4;

#define MTS_WAVE 64

// ---- error plumbing ---------------------------------------------------------------------------
void mts_set_error(const char* fmt, ...);
#define MTS_CHECK_ARG(cond, ...)                 \
  do {                                           \
    if (!(cond)) {                               \
      mts_set_error(__VA_ARGS__);                \
      return MTS_ERR_INVALID;                    \
    }                                            \
  } while (0)
#define MTS_UNSUPPORTED(cond, ...)               \
  do {                                           \
    if (!(cond)) {                               \
      mts_set_error(__VA_ARGS__);                \
      return MTS_ERR_UNSUPPORTED;                \
    }                                            \
  } while (0)
#define MTS_LAUNCH_CHECK(name)                                                        \
  do {                                                                                \
    hipError_t e__ = hipGetLastError();                                               \
    if (e__ != hipSuccess) {                                                          \
      mts_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));           \
      return MTS_ERR_LAUNCH;                                                          \
    }                                                                                 \
  } while (0)

// ---- scalar conversions -----------------------------------------------------------------------
__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }  // RNE, NaN-safe

__device__ __forceinline__ float bf16_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }

// ---- vector access: VEC elements of T starting at p (p aligned to VEC*sizeof(T)) ----------------
template <typename T, int VEC> struct Pack;
template <> struct Pack<float, 4> {
  float4 v;
  __device__ __forceinline__ void load(const float* p) { v = *reinterpret_cast<const float4*>(p); }
  __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<float4*>(p) = v; }
  __device__ __forceinline__ float get(int i) const { return (&v.x)[i]; }
  __device__ __forceinline__ void set(int i, float f) { (&v.x)[i] = f; }
};
template <> struct Pack<bf16_t, 8> {
  uint4 v;
  __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void store(bf16_t* p) const { *reinterpret_cast<uint4*>(p) = v; }
  __device__ __forceinline__ float get(int i) const {
    uint32_t u = (&v.x)[i >> 1];
    return (i & 1) ? bf16_hi(u) : bf16_lo(u);
  }
};
template <> struct Pack<bf16_t, 4> {
  uint2 v;
  __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const uint2*>(p); }
  __device__ __forceinline__ void store(bf16_t* p) const { *reinterpret_cast<uint2*>(p) = v; }
  __device__ __forceinline__ float get(int i) const {
    uint32_t u = (&v.x)[i >> 1];
    return (i & 1) ? bf16_hi(u) : bf16_lo(u);
  }
};

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  bf16x2 t;
  t[0] = (bf16_t)lo;
  t[1] = (bf16_t)hi;
  return __builtin_bit_cast(uint32_t, t);
}

// 4 consecutive elements load/store as floats (T = float: 16 B, T = bf16: 8 B)
template <typename T> __device__ __forceinline__ void load4(const T* p, float (&o)[4]);
template <> __device__ __forceinline__ void load4<float>(const float* p, float (&o)[4]) {
  float4 v = *reinterpret_cast<const float4*>(p);
  o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float (&o)[4]) {
  uint2 v = *reinterpret_cast<const uint2*>(p);
  o[0] = bf16_lo(v.x); o[1] = bf16_hi(v.x); o[2] = bf16_lo(v.y); o[3] = bf16_hi(v.y);
}
template <typename T> __device__ __forceinline__ void store4(T* p, const float (&o)[4]);
template <> __device__ __forceinline__ void store4<float>(float* p, const float (&o)[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1], o[2], o[3]);
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float (&o)[4]) {
  uint2 v;
  v.x = pack_bf16x2(o[0], o[1]);
  v.y = pack_bf16x2(o[2], o[3]);
  *reinterpret_cast<uint2*>(p) = v;
}

// ---- wave / block reductions --------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// d/dx gelu_erf(x) = Phi(x) + x * phi(x)
__device__ __forceinline__ float gelu_erf_grad_f(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// keep/drop decision of the dropout kernels: counter-based (no state), reproducible on the host
__host__ __device__ inline uint32_t mts_hash32(uint64_t seed, uint64_t idx) {   // splitmix64 finaliser
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (idx + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (uint32_t)(z >> 32);
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }
