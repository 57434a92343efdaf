// Row-wise LayerNorm-like streaming probe: what access shape a one-wave-per-row kernel needs to approach HBM speed on
// [16384 x 1792] bf16 (read x, write y).  hipcc --offload-arch=gfx950 -O3 ln_probe.hip -o ln_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__device__ __forceinline__ float wsum(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64); return v; }
__device__ __forceinline__ float lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ uint32_t pk(float a, float b) { return (__float_as_uint(a) >> 16) | (__float_as_uint(b) & 0xffff0000u); }

// V0: 8 B per lane, 7 slots, one row per wave, grid = rows/4
__global__ __launch_bounds__(256) void v0(const uint2* x, uint2* y, int rows) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  uint2 v[7]; float s = 0.f;
  for (int i = 0; i < 7; ++i) { v[i] = x[(size_t)row * 448 + lane + 64 * i]; s += lo(v[i].x) + hi(v[i].x) + lo(v[i].y) + hi(v[i].y); }
  const float m = wsum(s) * (1.f / 1792.f);
  float q = 0.f;
  for (int i = 0; i < 7; ++i) { float a = lo(v[i].x) - m, b = hi(v[i].x) - m, c = lo(v[i].y) - m, d = hi(v[i].y) - m; q += a * a + b * b + c * c + d * d; }
  const float r = rsqrtf(wsum(q) * (1.f / 1792.f) + 1e-12f);
  for (int i = 0; i < 7; ++i) { uint2 o; o.x = pk((lo(v[i].x) - m) * r, (hi(v[i].x) - m) * r); o.y = pk((lo(v[i].y) - m) * r, (hi(v[i].y) - m) * r); y[(size_t)row * 448 + lane + 64 * i] = o; }
}
// V1: 16 B per lane, 4 slots (last half-empty)
__global__ __launch_bounds__(256) void v1(const uint4* x, uint4* y, int rows) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  uint4 v[4]; float s = 0.f;
  for (int i = 0; i < 4; ++i) { const int c = lane + 64 * i; v[i] = c < 224 ? x[(size_t)row * 224 + c] : make_uint4(0, 0, 0, 0);
    s += lo(v[i].x) + hi(v[i].x) + lo(v[i].y) + hi(v[i].y) + lo(v[i].z) + hi(v[i].z) + lo(v[i].w) + hi(v[i].w); }
  const float m = wsum(s) * (1.f / 1792.f);
  float q = 0.f;
  for (int i = 0; i < 4; ++i) { const int c = lane + 64 * i; if (c < 224) { const uint32_t u[4] = {v[i].x, v[i].y, v[i].z, v[i].w}; for (int k = 0; k < 4; ++k) { float a = lo(u[k]) - m, b = hi(u[k]) - m; q += a * a + b * b; } } }
  const float r = rsqrtf(wsum(q) * (1.f / 1792.f) + 1e-12f);
  for (int i = 0; i < 4; ++i) { const int c = lane + 64 * i; if (c < 224) { uint4 o; o.x = pk((lo(v[i].x) - m) * r, (hi(v[i].x) - m) * r); o.y = pk((lo(v[i].y) - m) * r, (hi(v[i].y) - m) * r);
      o.z = pk((lo(v[i].z) - m) * r, (hi(v[i].z) - m) * r); o.w = pk((lo(v[i].w) - m) * r, (hi(v[i].w) - m) * r); y[(size_t)row * 224 + c] = o; } }
}
// V2: V0 body, persistent waves with the next row prefetched
__global__ __launch_bounds__(256) void v2(const uint2* x, uint2* y, int rows) {
  const int lane = threadIdx.x & 63; const int stride = gridDim.x * 4;
  int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  uint2 nx[7];
  if (row < rows) for (int i = 0; i < 7; ++i) nx[i] = x[(size_t)row * 448 + lane + 64 * i];
  for (; row < rows; row += stride) {
    uint2 v[7]; float s = 0.f;
    for (int i = 0; i < 7; ++i) { v[i] = nx[i]; s += lo(v[i].x) + hi(v[i].x) + lo(v[i].y) + hi(v[i].y); }
    if (row + stride < rows) for (int i = 0; i < 7; ++i) nx[i] = x[(size_t)(row + stride) * 448 + lane + 64 * i];
    const float m = wsum(s) * (1.f / 1792.f);
    float q = 0.f;
    for (int i = 0; i < 7; ++i) { float a = lo(v[i].x) - m, b = hi(v[i].x) - m, c = lo(v[i].y) - m, d = hi(v[i].y) - m; q += a * a + b * b + c * c + d * d; }
    const float r = rsqrtf(wsum(q) * (1.f / 1792.f) + 1e-12f);
    for (int i = 0; i < 7; ++i) { uint2 o; o.x = pk((lo(v[i].x) - m) * r, (hi(v[i].x) - m) * r); o.y = pk((lo(v[i].y) - m) * r, (hi(v[i].y) - m) * r); y[(size_t)row * 448 + lane + 64 * i] = o; }
  }
}
// V3: plain copy, 16 B per lane, grid-stride (what the memory system gives a trivial kernel)
__global__ __launch_bounds__(256) void v3(const uint4* x, uint4* y, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = x[i];
}
int main() {
  const int rows = 16384; const size_t bytes = (size_t)rows * 1792 * 2;
  void *x, *y[6];
  hipMalloc(&x, bytes * 6); for (int i = 0; i < 6; ++i) hipMalloc(&y[i], bytes);
  hipMemset(x, 0x3c, bytes * 6);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 3; ++i) launch(i);
    hipDeviceSynchronize(); hipEventRecord(e0);
    for (int i = 0; i < 30; ++i) launch(i);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %7.1f us  %6.2f TB/s (read+write)\n", name, ms * 1e3 / 30, 2.0 * bytes / (ms / 30 * 1e-3) / 1e12);
  };
  // rotate over 6 input/output buffers (705 MB of traffic between reuses > the 256 MiB Infinity Cache)
  run("v0 8B/lane row-per-wave", [&](int i) { hipLaunchKernelGGL(v0, dim3(rows / 4), dim3(256), 0, 0, (const uint2*)((char*)x + bytes * (i % 6)), (uint2*)y[i % 6], rows); });
  run("v1 16B/lane row-per-wave", [&](int i) { hipLaunchKernelGGL(v1, dim3(rows / 4), dim3(256), 0, 0, (const uint4*)((char*)x + bytes * (i % 6)), (uint4*)y[i % 6], rows); });
  for (int g : {512, 1024, 2048}) {
    char nm[64]; snprintf(nm, 64, "v2 persistent grid=%d", g);
    run(nm, [&](int i) { hipLaunchKernelGGL(v2, dim3(g), dim3(256), 0, 0, (const uint2*)((char*)x + bytes * (i % 6)), (uint2*)y[i % 6], rows); });
  }
  run("v3 copy 16B grid=2048", [&](int i) { hipLaunchKernelGGL(v3, dim3(2048), dim3(256), 0, 0, (const uint4*)((char*)x + bytes * (i % 6)), (uint4*)y[i % 6], bytes / 16); });
  run("v3 copy 16B grid=8192", [&](int i) { hipLaunchKernelGGL(v3, dim3(8192), dim3(256), 0, 0, (const uint4*)((char*)x + bytes * (i % 6)), (uint4*)y[i % 6], bytes / 16); });
  return 0;
}
