// 256x224 bf16 MFMA GEMM for the WEIGHT GRADIENTS (layout TN: both operands k-strided, fp32 C), four waves per workgroup.
//
//   dW[M, N] (fp32) = dY[K, M]^T  X[K, N],   K = all tokens of the batch        (models/CRF.py:574-595 backward through
//   modeling_longformer.py:504-506,1069 and NeuralArchitectures.py:98-115: every nn.Linear / LSTM input projection of the taggers)
//
// The eight-wave kernel of gemm224.hip ran this layout with its round-1 schedule (barrier at the end of a 64-deep K-tile, 1.07 PFLOP/s,
// 3.6x the operand bytes read from the fabric) because a fragment of a k-strided operand costs two transposing LDS reads and neither the
// mid-tile barrier schedule nor the four-wave NT loop (gemm_bf16_224d_kernel, 30 reads per K-tile, all of a k-step in flight at once)
// carries 60 reads per K-tile through a 4-bit lgkmcnt.  What a k-strided operand offers instead: its LDS image is k-row-major, so HALF a
// K-tile -- one 32-deep k-step, a UNIT -- is a self-contained 32-KiB image (A: 2 x [32 k][128 columns], B: 2 x [32 k][112 (+16) columns]).
// The pipeline runs on units:
//
//   LDS      four 32-KiB slots, unit u in slot u & 3
//   copies   buffer-load LDS-DMA (1-KiB pieces = 4 k-rows x 256 B, the swizzle on the SOURCE side; resource + one of two lane offsets +
//            an SGPR offset per copy, as in gemm_bf16_224d_kernel); unit u + 4 is copied during unit u, one copy per block of 7 MFMAs
//            -> two whole units (1792 matrix-pipe cycles) for the youngest copy to land
//   reads    the 30 transposing reads of unit u + 1 are issued during unit u, five per block in blocks 0..5, each group behind
//            "s_waitcnt lgkmcnt(10)": never more than 15 LDS operations of a wave in flight (the counter has 4 bits), and a fragment is
//            requested at least two blocks (224 cycles) before the wait that covers it
//   MFMAs    unit = 8 blocks (A fragment i x 7 B fragments), fragments of unit u in registers (two sets, by unit parity)
//   boundary lgkmcnt(0) (nobody still reads the slot the next unit's copies overwrite), vmcnt(16) (unit u + 2 has landed: the copies of
//            units u + 3 and u + 4 may fly), ONE barrier per unit
//
// Per output element the products are accumulated in ascending k, 32 at a time, exactly as in every other bf16 kernel of this library:
// results are bitwise those of gemm_bf16_224_kernel<TN> (tests/test_gpu_kernels.py::test_gemm_224t_matches_the_eight_wave_kernel_bitwise).
// Split-K slices (blockIdx -> (slice, tile)) store fp32 partial tiles to the slab; the fixed-order reduce is splitk_reduce_kernel or, with
// GemmArgs::chain set, the LAST-ARRIVING slice of each tile (below).
#include <algorithm>
#include <type_traits>
#include "gemm_common.h"

#define T_SLOT 32768
#define T_LDS (4 * T_SLOT)
#define T_BN 224
#define T_HN 112

typedef __attribute__((address_space(3))) void t_dlptr;

struct TFrag { s16x4 lo, hi; };

// (free function templates, not generic lambdas: clang rejects inline-asm operands that name variables captured by a generic lambda)
// read operation K (0..29) of the unit whose A / B half images start at sbA / sbB into fragment set PS: 2 j, 2 j + 1 = the halves of B fragment j;
// 14 + 2 i, 15 + 2 i = those of A fragment i
template <int PS, int K>
__device__ __forceinline__ void t_rd(TFrag (&fa)[2][8], TFrag (&fb)[2][7], const unsigned (&ax)[8], unsigned sbA, unsigned sbB) {
  constexpr int req = K >> 1, hi = K & 1;
  if constexpr (req < 7) {
    const unsigned ad = ax[req] + sbB;
    if constexpr (hi) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(fb[PS][req].hi) : "v"(ad));
    else asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fb[PS][req].lo) : "v"(ad));
  } else {
    const unsigned ad = ax[req - 7] + sbA;
    if constexpr (hi) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:1024" : "=v"(fa[PS][req - 7].hi) : "v"(ad));
    else asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(fa[PS][req - 7].lo) : "v"(ad));
  }
}
// block I of a unit: A fragment I x the 7 B fragments (formed once, in block 0); behind each of the first five MFMAs one read of the NEXT unit
// (blocks 0..5: operations 5 I .. 5 I + 4, behind a wait that keeps at most 15 LDS operations of the wave in flight), behind the sixth the
// block's copy
template <int PS, int I, bool NXT, class Copy>
__device__ __forceinline__ void t_block(f32x4 (&acc)[8][7], TFrag (&fa)[2][8], TFrag (&fb)[2][7], bf16x8 (&ob)[7], const unsigned (&ax)[8],
                                        unsigned sbA, unsigned sbB, Copy&& cp) {
  if constexpr (NXT && I < 6) lgkm_wait<10>();
  if constexpr (I == 0) {
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      asm volatile("" : "+v"(fb[PS][j].lo), "+v"(fb[PS][j].hi));
      ob[j] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(fb[PS][j].lo, fb[PS][j].hi, 0, 1, 2, 3, 4, 5, 6, 7));
    }
  }
  asm volatile("" : "+v"(fa[PS][I].lo), "+v"(fa[PS][I].hi));
  const bf16x8 va = __builtin_bit_cast(bf16x8, __builtin_shufflevector(fa[PS][I].lo, fa[PS][I].hi, 0, 1, 2, 3, 4, 5, 6, 7));
#define T_SB __builtin_amdgcn_sched_barrier(0)
#define T_MF(j_) T_SB; acc[I][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ob[j_], va, acc[I][j_], 0, 0, 0); T_SB;
#define T_RD(k_) if constexpr (NXT && I < 6) t_rd<PS ^ 1, (I < 6 ? 5 * I + (k_) : 0)>(fa, fb, ax, sbA, sbB);
  T_MF(0) T_RD(0) T_MF(1) T_RD(1) T_MF(2) T_RD(2) T_MF(3) T_RD(3) T_MF(4) T_RD(4) T_MF(5)
  cp();
  T_MF(6)
#undef T_MF
#undef T_RD
#undef T_SB
}

// A2 != NULL: a PAIR of problems of one shape in one launch (the two weight gradients of the fused feed-forward block, 8 tiles each: alone a launch
// fills half the chip at 16 slices): workgroups [0, nt * splits) take (a.A, a.B) -> a.slab, the next nt * splits take (A2, B2) -> slab2.
// (PAIR is a template parameter so that the pair launch is a symbol of its own in a kernel trace: per-symbol averages stay per call site)
template <bool PAIR>
__global__ __launch_bounds__(256, 1) void gemm_bf16_224t_kernel(const GemmArgs a, const int splits, const void* __restrict__ A2, const void* __restrict__ B2,
                                                                float* __restrict__ slab2) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave_u >> 1, wn = wave_u & 1;
  const int r16 = lane & 15, g = lane >> 4, q = r16 >> 2, p = r16 & 3;
  const int ntn = a.N / T_BN, ntm = a.M / 256, nt = ntn * ntm, ntot = nt * splits;
  // ---- workgroup -> (slice, tile).  Workgroups b, b + 8, .. run on one XCD (own L2): an XCD gets a contiguous run of ids; ids walk one
  // slice after the other, inside a slice bands of four tile rows, column-major inside a band -- the 32 workgroups an XCD runs at a time
  // share 4 A panels and up to 8 B panels of ONE k range.
  int bm0, bn0, z;
  bool second;
  {
    const int t = blockIdx.x, ngrid = gridDim.x;
    const int qq = ngrid >> 3, rr = ngrid & 7, xcd = t & 7, idx = t >> 3;
    int id = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    second = PAIR && id >= ntot;                        // (pair launch: gridDim.x = 2 ntot)
    if (second) id -= ntot;
    z = id / nt;
    const int tl = id - z * nt;
    const int band = tl / (4 * ntn), within = tl - band * 4 * ntn;
    const int rows = min(4, ntm - band * 4);
    bm0 = (band * 4 + within % rows) * 256;
    bn0 = (within / rows) * T_BN;
  }
  const int kbeg = z * a.ksplit;
  const int kend = min(a.K, kbeg + a.ksplit);
  const int nu = (kend - kbeg) >> 5;                   // units of 32 k (launcher: >= 4 in every slice)
  const bf16_t* __restrict__ A = reinterpret_cast<const bf16_t*>(second ? A2 : a.A);
  const bf16_t* __restrict__ B = reinterpret_cast<const bf16_t*>(second ? B2 : a.B);

  // ---- copies.  A piece = 4 k-rows x 256 B of one half image, lane-contiguous in the LDS (16 B per lane): lane (lr = lane >> 4, c16 = lane & 15)
  // fills k-row lr, 32-B slot c16 >> 1, half c16 & 1; slot s of k-row r holds source column block s ^ key(r), key(r) = (r & 3) | (((r >> 3) & 1) << 2)
  // (gemm_common.h strided_off) -- r = 4 piece + lr, so the key is lr | (bit 1 of the piece number << 2): two lane offsets per operand.
  // B half images hold 112 used columns of 128: the lanes of the unused 16 fetch 16 columns further left (nobody reads what they bring, and
  // the last tile column never reaches past the matrix).
  const int lr = lane >> 4, c16 = lane & 15;
  unsigned voA[2], voB[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const int key = lr | (par << 2);
    const int col = (((c16 >> 1) ^ key) << 4) + ((c16 & 1) << 3);
    voA[par] = (unsigned)(lr * a.lda + col) * 2u;
    voB[par] = (unsigned)(lr * a.ldb + (col >= T_HN ? col - 16 : col)) * 2u;
  }
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)kbeg * a.lda + bm0), 0, 0x7ffffff0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(B + (size_t)kbeg * a.ldb + bn0), 0, 0x7ffffff0, 0x00020000);
  // wave w copies pieces 4 (w & 1) .. + 3 of half image w >> 1, of A (copies 0..3) and of B (copies 4..7)
  const int h_w = wave_u >> 1, pi0 = (wave_u & 1) * 4;
  const int sA0 = (4 * pi0 * a.lda + h_w * 128) * 2, sB0 = (4 * pi0 * a.ldb + h_w * T_HN) * 2;
  const int rowA4 = 8 * a.lda, rowB4 = 8 * a.ldb;                 // bytes per 4 k-rows
  const int unitA = 64 * a.lda, unitB = 64 * a.ldb;               // bytes per unit
  char* const dst_w = smem + h_w * 8192 + pi0 * 1024;
  auto copy1 = [&](auto C, int u) {
    constexpr int c = decltype(C)::value;
    char* d = dst_w + (u & 3) * T_SLOT + (c & 3) * 1024;
    if constexpr (c < 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (t_dlptr*)d, 16, voA[(c >> 1) & 1], sA0 + c * rowA4 + u * unitA, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (t_dlptr*)(d + 16384), 16, voB[(c >> 1) & 1], sB0 + (c - 4) * rowB4 + u * unitB, 0, 0);
  };
  // ---- fragment reads: X[k = 8 g + q (+ 4), column block c, columns 4 p ..] through two transposing reads (gemm_common.h frag_strided)
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned sx = (unsigned)((8 * g + q) * 256 + 8 * p + ((q | ((g & 1) << 2)) << 5));
  unsigned ax[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) ax[c] = lds0 + (sx ^ (unsigned)(c << 5));
  const unsigned offA = wm * 8192, offB = 16384 + wn * 8192;

  TFrag fa[2][8], fb[2][7];
  f32x4 acc[8][7];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 7; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ---- prologue: units 0..3 on their way, unit 0 in the registers -------------------------------------------------------------------
  {
#define CPU(u_) copy1(std::integral_constant<int, 0>{}, u_); copy1(std::integral_constant<int, 1>{}, u_); copy1(std::integral_constant<int, 2>{}, u_); \
                copy1(std::integral_constant<int, 3>{}, u_); copy1(std::integral_constant<int, 4>{}, u_); copy1(std::integral_constant<int, 5>{}, u_); \
                copy1(std::integral_constant<int, 6>{}, u_); copy1(std::integral_constant<int, 7>{}, u_);
    CPU(0) CPU(1) CPU(2) CPU(3)
#undef CPU
    asm volatile("s_waitcnt vmcnt(24)" ::: "memory");      // unit 0 has landed (in-order counter)
    __builtin_amdgcn_s_barrier();
#define RD5(b_) lgkm_wait<10>(); t_rd<0, 5 * (b_)>(fa, fb, ax, offA, offB); t_rd<0, 5 * (b_) + 1>(fa, fb, ax, offA, offB); t_rd<0, 5 * (b_) + 2>(fa, fb, ax, offA, offB); \
                t_rd<0, 5 * (b_) + 3>(fa, fb, ax, offA, offB); t_rd<0, 5 * (b_) + 4>(fa, fb, ax, offA, offB);
    RD5(0) RD5(1) RD5(2) RD5(3) RD5(4) RD5(5)
#undef RD5
    lgkm_wait<0>();
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // unit 1 too
    __builtin_amdgcn_s_barrier();
  }

  // ---- one unit.  PS: fragment set of unit u (= u & 1); NXT: unit u + 1 exists (read into set PS ^ 1); LD: unit u + 4 exists (copied into
  // slot u & 3, which nobody reads any more); VM: copies that may still fly at the boundary behind this unit (-1: no boundary) ---------------
  auto unit = [&](int u, auto PS_, auto NXT_, auto LD_, auto VM_) {
    constexpr int PS = decltype(PS_)::value, VM = decltype(VM_)::value;
    constexpr bool nxt = decltype(NXT_)::value, ld = decltype(LD_)::value;
    const unsigned sb = (unsigned)((u + 1) & 3) * T_SLOT;
    const unsigned sbA = sb + offA, sbB = sb + offB;
    bf16x8 ob[7];
    auto nocp = [] {};
#define BLK(i_) if constexpr (ld) t_block<PS, i_, nxt>(acc, fa, fb, ob, ax, sbA, sbB, [&] { copy1(std::integral_constant<int, i_>{}, u + 4); }); \
                else t_block<PS, i_, nxt>(acc, fa, fb, ob, ax, sbA, sbB, nocp);
    BLK(0) BLK(1) BLK(2) BLK(3) BLK(4) BLK(5) BLK(6) BLK(7)
#undef BLK
    if constexpr (VM >= 0) {
      lgkm_wait<0>();                                        // my reads of unit u + 1 are done: its slot is the one unit u + 1's copies overwrite
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM >= 0 ? VM : 0) : "memory");
      __builtin_amdgcn_s_barrier();
    }
  };
  {
    using T = std::true_type; using F = std::false_type;
    using P0 = std::integral_constant<int, 0>; using P1 = std::integral_constant<int, 1>;
    using V16 = std::integral_constant<int, 16>; using V8 = std::integral_constant<int, 8>; using V0 = std::integral_constant<int, 0>;
    using VN = std::integral_constant<int, -1>;
    int u = 0;
#pragma clang loop unroll(disable)
    for (; u + 5 < nu; u += 2) { unit(u, P0{}, T{}, T{}, V16{}); unit(u + 1, P1{}, T{}, T{}, V16{}); }
    // nu is even (slices are multiples of 64 k) and >= 4: four units are left, the last copies (of unit nu - 1) went out during unit nu - 5
    unit(u, P0{}, T{}, F{}, V8{}); unit(u + 1, P1{}, T{}, F{}, V0{}); unit(u + 2, P0{}, T{}, F{}, V0{}); unit(u + 3, P1{}, F{}, F{}, VN{});
  }

  // ---- epilogue: fp32 tile -> C, or -> this slice's plane of the slab ------------------------------------------------------------------
  const int m0 = bm0 + wm * 128, n0 = bn0 + wn * T_HN;
  float* __restrict__ Cb = a.slab ? (second ? slab2 : a.slab) + (size_t)z * a.M * a.N : reinterpret_cast<float*>(a.C);
  const size_t ldo = a.slab ? (size_t)a.N : (size_t)a.ldc;
  const bool accum = !a.slab && (a.epi & MTS_EPI_ACCUM);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    float* row = Cb + (size_t)(m0 + i * 16 + r16) * ldo + n0 + 4 * g;
    if (accum) {
      float4 o[7];
#pragma unroll
      for (int j = 0; j < 7; ++j) o[j] = *reinterpret_cast<const float4*>(row + j * 16);
#pragma unroll
      for (int j = 0; j < 7; ++j) { acc[i][j][0] += o[j].x; acc[i][j][1] += o[j].y; acc[i][j][2] += o[j].z; acc[i][j][3] += o[j].w; }
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) *reinterpret_cast<float4*>(row + j * 16) = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
  }
  if (!a.slab || !a.chain) return;

  // ---- split-K combine INSIDE the launch: the slice of a tile that arrives last adds the planes, in slice order (what splitk_reduce_kernel
  // would do in a launch of its own: same order, same bits).  Agent-scope release by every slice, ticket, agent-scope acquire by the last one
  // (the XCDs' L2s are not coherent with each other; /opt/skills guide, "In-launch split-K reduction"); a.chain[tile] was zeroed by the launcher.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int tile = (bm0 / 256) * ntn + bn0 / T_BN;
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned ticket = __hip_atomic_fetch_add(a.chain + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *reinterpret_cast<volatile __attribute__((address_space(3))) unsigned*>(lds0) = ticket;        // (the one LDS array: no second __shared__ object)
  }
  __syncthreads();
  if (*reinterpret_cast<volatile __attribute__((address_space(3))) unsigned*>(lds0) != (unsigned)(splits - 1)) return;
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  {
    const size_t plane = (size_t)a.M * a.N;
    const bool acc_c = (a.epi & MTS_EPI_ACCUM) != 0;
    float* __restrict__ C = reinterpret_cast<float*>(a.C);
    // 256 rows x 56 float4: thread t takes float4 number it * 256 + t of the tile (row-major): 56 consecutive threads walk one 896-byte row
#pragma unroll 2
    for (int it = 0; it < 56; ++it) {
      const int idx = it * 256 + tid;
      const int row = idx / 56, c4 = idx - row * 56;
      const float* sp = a.slab + (size_t)(bm0 + row) * a.N + bn0 + 4 * c4;
      float* cp = C + (size_t)(bm0 + row) * a.ldc + bn0 + 4 * c4;
      float4 s = acc_c ? *reinterpret_cast<const float4*>(cp) : make_float4(0.f, 0.f, 0.f, 0.f);
      for (int zz = 0; zz < splits; ++zz) {
        const float4 v = *reinterpret_cast<const float4*>(sp + (size_t)zz * plane);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      *reinterpret_cast<float4*>(cp) = s;
    }
  }
}

// does this kernel take the call?  (mts_gemm asks before it promises the in-launch combine)
bool mts_gemm224t_applies(const GemmArgs& a, int layout, bool c_is_f32, int splits) {
  if (!c_is_f32 || layout != MTS_TN) return false;
  const int last = a.K - (splits - 1) * a.ksplit;           // k range of the last slice
  const size_t spanA = ((size_t)a.ksplit + 64) * a.lda * 2, spanB = ((size_t)a.ksplit + 64) * a.ldb * 2;   // byte offsets inside a slice stay below 2^31
  return (a.epi & ~MTS_EPI_ACCUM) == 0 && (a.M % 256 == 0) && (a.N % T_BN == 0) && (a.K % 64 == 0) && (a.ksplit % 64 == 0) &&
         a.ksplit >= 128 && last >= 128 && (a.lda % 8 == 0) && (a.ldb % 8 == 0) && (((uintptr_t)a.A & 15) == 0) && (((uintptr_t)a.B & 15) == 0) &&
         (a.ldc % 4 == 0) && (((uintptr_t)a.C & 15) == 0) && spanA < 0x7ff00000u && spanB < 0x7ff00000u && (splits == 1 || a.slab);
}

// called from mts_launch_gemm224 (gemm224.hip); -1: shape not covered here
int mts_launch_gemm224t(const GemmArgs& a, int layout, bool c_is_f32, int splits, hipStream_t st) {
  if (!mts_gemm224t_applies(a, layout, c_is_f32, splits)) return -1;
  auto k = gemm_bf16_224t_kernel<false>;
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS);
    if (e != hipSuccess) { mts_set_error("gemm224t: hipFuncSetAttribute: %s", hipGetErrorString(e)); return MTS_ERR_LAUNCH; }
    attr_set = true;
  }
  const int nt = (a.M / 256) * (a.N / T_BN);
  if (a.chain) {                                       // the arrival tickets of the in-launch combine: zeroed on the stream, every call
    hipError_t e = hipMemsetAsync(a.chain, 0, ((size_t)nt * sizeof(unsigned) + 15) & ~(size_t)15, st);
    if (e != hipSuccess) { mts_set_error("gemm224t: hipMemsetAsync: %s", hipGetErrorString(e)); return MTS_ERR_LAUNCH; }
  }
  hipLaunchKernelGGL(k, dim3(nt * splits), dim3(256), T_LDS, st, a, splits, (const void*)nullptr, (const void*)nullptr, (float*)nullptr);
  return MTS_OK;
}

// two problems of one shape, both into slabs (a.slab / slab2, `splits` planes each; a.chain must be NULL): see the kernel's header
int mts_launch_gemm224t_pair(const GemmArgs& a, int splits, const void* A2, const void* B2, float* slab2, hipStream_t st) {
  if (!mts_gemm224t_applies(a, MTS_TN, true, splits) || !a.slab || a.chain || !A2 || !B2 || !slab2 || (((uintptr_t)A2 | (uintptr_t)B2) & 15)) return -1;
  auto k = gemm_bf16_224t_kernel<true>;
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, T_LDS);
    if (e != hipSuccess) { mts_set_error("gemm224t: hipFuncSetAttribute: %s", hipGetErrorString(e)); return MTS_ERR_LAUNCH; }
    attr_set = true;
  }
  const int nt = (a.M / 256) * (a.N / T_BN);
  hipLaunchKernelGGL(k, dim3(2 * nt * splits), dim3(256), T_LDS, st, a, splits, A2, B2, slab2);
  return MTS_OK;
}
