// Sustained MFMA rate of the whole chip from registers only (no LDS, no memory): what the matrix cores deliver under the
// board's power management, to put the GEMM kernels' TFLOP/s in proportion.  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(512) void mfma_loop(float* out, int iters) {
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (threadIdx.x + i)); b[i] = (__bf16)(0.002f * (threadIdx.x - i)); }
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
  float* out;
  hipMalloc(&out, 4096 * 512 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {256, 512}) {
    for (int ms_target : {1, 20, 200}) {
      const int iters = 4000 * ms_target;
      hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(512), 0, 0, out, 1000);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(512), 0, 0, out, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double flops = (double)blocks * 8 * iters * 16 * 16384.0;
      printf("blocks=%d iters=%d  %.3f ms  %.1f TFLOP/s\n", blocks, iters, ms, flops / ms / 1e9);
    }
  }
  return 0;
}
