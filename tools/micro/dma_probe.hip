// L2 -> LDS copy rate of one workgroup per CU through LDS-DMA (global_load_lds_dwordx4) as a function of the bytes kept in
// flight: the question behind the big GEMMs' 1.37 us per 60-KiB K-tile -- is a CU bound by latency x bytes in flight or by an L2
// bandwidth ceiling?   hipcc --offload-arch=gfx950 -O3 dma_probe.hip -o dma_probe ; ./dma_probe
// Every workgroup of an XCD streams the same WS-byte window (L2-resident, shared like GEMM operand panels) in 1-KiB wave pieces;
// DEPTH = DMA instructions a wave keeps outstanding (8 waves x DEPTH KiB in flight per CU).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

template <int DEPTH>
__global__ __launch_bounds__(512, 2) void probe(const char* __restrict__ src, size_t ws, int iters, int* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const size_t base = (size_t)(blockIdx.x & 7) * ws;             // one window per XCD (workgroups b, b+8, .. share an XCD)
  size_t off = ((size_t)(blockIdx.x >> 3) * 8 + wave) * 1024;    // workgroups of an XCD start at different pieces
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const char* p = src + base + (off % ws) + lane * 16;
      __builtin_amdgcn_global_load_lds((gptr_t*)p, (lptr_t*)(smem + (wave * DEPTH + d) * 1024), 16, 0, 0);
      off += 8 * 1024 * 37;                                       // stride through the window
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (threadIdx.x == 0 && smem[0] == 123) *sink = 1;
}

template <int DEPTH>
static void run(const char* d, size_t ws, int* sink) {
  const int iters = 4096 / DEPTH;
  hipFuncSetAttribute((const void*)probe<DEPTH>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * DEPTH * 1024);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  probe<DEPTH><<<256, 512, 8 * DEPTH * 1024>>>(d, ws, iters, sink);
  hipEventRecord(a);
  probe<DEPTH><<<256, 512, 8 * DEPTH * 1024>>>(d, ws, iters, sink);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double bytes = 256.0 * 8 * (double)iters * DEPTH * 1024;
  printf("window %5zu KiB/XCD  depth %2d (%3d KiB in flight per CU): %7.1f us  %6.2f TB/s chip  %5.1f GB/s per CU  (%.2f us per %d-KiB batch)\n", ws >> 10, DEPTH,
         8 * DEPTH, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / 256, ms * 1e3 / iters, 8 * DEPTH);
}

int main() {
  char* d; int* sink;
  const size_t total = (size_t)8 * 64 << 20;
  hipMalloc(&d, total); hipMemset(d, 1, total); hipMalloc(&sink, 4);
  for (size_t ws : {(size_t)1 << 20, (size_t)3 << 20, (size_t)16 << 20, (size_t)64 << 20}) {
    run<1>(d, ws, sink); run<2>(d, ws, sink); run<4>(d, ws, sink); run<8>(d, ws, sink); run<15>(d, ws, sink);
  }
  return 0;
}
