// Row-wise HBM-bound kernels: embedding-sum + LayerNorm, LayerNorm fwd/bwd, tagger head fwd/bwd.
//
// Mapping: one 64-lane wave per row, the row lives in registers as NV vectors of 4 elements per lane
// (lane l owns elements 4*(l + 64*v) .. +3, v < NV), so every global access is a coalesced 8/16-byte
// vector and each row is read exactly once; statistics are fp32 wave reductions (DPP/shuffle), never LDS.
// Column reductions (dgamma, dbeta, bias gradients, head weight gradient) are accumulated in registers
// over the rows a wave visits, combined per workgroup through LDS and finished by a second tiny kernel
// over the per-workgroup slabs -> bitwise reproducible (no float atomics).
#include <algorithm>
#include <type_traits>
#include "common.h"
#include "loss_elems.h"

#define ROW_WAVES 4            // waves (= rows in flight) per workgroup
#define BWD_MAX_BLOCKS 512

template <typename T, int NV> struct RowRegs {
  float v[NV][4];
  __device__ __forceinline__ void load(const T* row, int D, int lane) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = 4 * (lane + 64 * i);
      if (e < D) load4<T>(row + e, v[i]);
      else { v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f; }
    }
  }
  __device__ __forceinline__ void store(T* row, int D, int lane) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = 4 * (lane + 64 * i);
      if (e < D) store4<T>(row + e, v[i]);
    }
  }
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i][0] = v[i][1] = v[i][2] = v[i][3] = 0.f;
  }
};

template <int NV> __device__ __forceinline__ void row_stats(const float (&x)[NV][4], int D, int lane, float& mean, float& rstd, float eps) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) s += (x[i][0] + x[i][1]) + (x[i][2] + x[i][3]);
  mean = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    if (4 * (lane + 64 * i) < D) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { const float d = x[i][j] - mean; q += d * d; }
    }
  }
  rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
}

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// FULL (D == NV * 256): no per-slot bounds checks, hence no divergent branches around the loads -- behind such a branch the
// compiler waits for every outstanding load before the next slot (seven serial HBM round trips per row at D = 1792).
// EMBED: 0 plain LayerNorm; 1 embedding block on fp32 inputs (the collater's batch); 2 the same on bf16 inputs (a batch that crossed PCIe in bf16:
// prefetch.DevicePrefetcher / AudioPortionDataset(wire_dtype='bf16') -- read as it is, no fp32 copy of it is ever made)
template <typename T, int NV, int EMBED, bool FULL = false>
__global__ __launch_bounds__(64 * ROW_WAVES) void ln_fwd_kernel(
    const void* __restrict__ xin, const float* __restrict__ pos, int pos_offset, int L, const float* __restrict__ type0,
    const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int rows, int D,
    T* __restrict__ y, T* __restrict__ pre, float* __restrict__ mean_out, float* __restrict__ rstd_out,
    const float* __restrict__ head_w, const float* __restrict__ head_b, int n_out, float* __restrict__ scores,
    const int32_t* __restrict__ row_src, const float* __restrict__ xin2 = nullptr, int D1 = 0) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * ROW_WAVES + (threadIdx.x >> 6);
  if (row >= rows) return;
  // gamma / beta for every slot are requested FIRST: before the row statistics are reduced (their latency hides behind it) and, in the embedding
  // form, before the stores of `pre` -- the vector-memory counter retires in order, so a load issued behind a store is waited for together with
  // the store's acknowledgement (the embedding block ran at 4.0 TB/s with the loads behind those stores, the plain form at 5.1)
  float gv[NV][4], bv[NV][4];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int e = 4 * (lane + 64 * i);
    if (FULL || e < D) { load4<float>(gamma + e, gv[i]); load4<float>(beta + e, bv[i]); }
  }
  float x[NV][4];
  if constexpr (EMBED) {
    // packed batches: output row `row` is sentence row_src[row] = b*L + i of the padded batch (position = i)
    const int src = row_src ? row_src[row] : row;
    // K-split input (xin2 != NULL): the row is text[src, 0:D1] | audio[src, 0:D-D1], two separate matrices -- the early-fusion
    // concat of utils/load_datasets_precomputed.py:158-161 is never materialised; D1 % 4 == 0, so no 16-byte chunk straddles
    const int Dx = xin2 ? D1 : D;
    const float* xr = reinterpret_cast<const float*>(xin) + (size_t)src * Dx;
    const float* xr2 = xin2 ? xin2 + (size_t)src * (D - D1) - D1 : xr;
    const float* pr = pos + (size_t)(pos_offset + src % L) * D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = 4 * (lane + 64 * i);
      if (FULL || e < D) {
        float a[4], b[4], c[4];
        if constexpr (EMBED == 2) load4<bf16_t>(reinterpret_cast<const bf16_t*>(xin) + (size_t)src * D + e, a);
        else load4<float>((e < Dx ? xr : xr2) + e, a);
        load4<float>(pr + e, b); load4<float>(type0 + e, c);
#pragma unroll
        for (int j = 0; j < 4; ++j) x[i][j] = (a[j] + b[j]) + c[j];   // same association as HF:421
        if constexpr (sizeof(T) == 2) {
          // the backward re-reads `pre` in storage precision: normalise exactly what was stored
#pragma unroll
          for (int j = 0; j < 4; ++j) x[i][j] = to_f32(from_f32<T>(x[i][j]));
        }
      } else { x[i][0] = x[i][1] = x[i][2] = x[i][3] = 0.f; }
    }
    if (pre) {
#pragma unroll
      for (int i = 0; i < NV; ++i) { const int e = 4 * (lane + 64 * i); if (FULL || e < D) store4<T>(pre + (size_t)row * D + e, x[i]); }
    }
  } else {
    const T* xr = reinterpret_cast<const T*>(xin) + (size_t)row * D;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = 4 * (lane + 64 * i);
      if (FULL || e < D) load4<T>(xr + e, x[i]);
      else { x[i][0] = x[i][1] = x[i][2] = x[i][3] = 0.f; }
    }
  }
  float mean, rstd;
  row_stats<NV>(x, D, lane, mean, rstd, eps);
  // Two passes: every load (gamma, beta, head weights) comes before the first store.  The vector-memory counter retires in
  // order, so a load issued after a store cannot be waited for without also waiting for the store's acknowledgement.
  float hs[4] = {0.f, 0.f, 0.f, 0.f};
  const float* hwp = head_w ? head_w : gamma;                 // always-valid pointer: the head loads are unconditional
  const int hw_max = head_w ? n_out - 1 : 0;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int e = 4 * (lane + 64 * i);
    if (FULL || e < D) {
#pragma unroll
      for (int j = 0; j < 4; ++j) x[i][j] = (x[i][j] - mean) * rstd * gv[i][j] + bv[i][j];
      if (head_w) {  // fused tagger head on the value that will be stored (storage precision)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          float w[4];
          load4<float>(hwp + (size_t)min(c, hw_max) * D + e, w);   // rows past n_out repeat the last one; their sums are not used
#pragma unroll
          for (int j = 0; j < 4; ++j) hs[c] += to_f32(from_f32<T>(x[i][j])) * w[j];
        }
      }
    }
  }
  if (y) {          // NULL: the caller only wants the fused head's scores and the statistics (the backward recomputes y, see ln_bwd_kernel NHG)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = 4 * (lane + 64 * i);
      if (FULL || e < D) store4<T>(y + (size_t)row * D + e, x[i]);
    }
  }
  if (lane == 0 && mean_out) { mean_out[row] = mean; rstd_out[row] = rstd; }
  if (head_w) {
    for (int c = 0; c < n_out; ++c) {
      const float s = wave_sum(hs[c]);
      if (lane == 0) scores[(size_t)row * n_out + c] = s + head_b[c];
    }
  }
}

// scores = x w^T + b for small n_out (1, 2 or 4): one wave per row
template <typename T>
__global__ __launch_bounds__(64 * ROW_WAVES) void head_fwd_kernel(const T* __restrict__ x, int ldx, int rows, int D, int n_out,
                                                                const float* __restrict__ w, const float* __restrict__ b,
                                                                float* __restrict__ scores) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * ROW_WAVES + (threadIdx.x >> 6);
  if (row >= rows) return;
  float hs[4] = {0.f, 0.f, 0.f, 0.f};
  for (int e = 4 * lane; e < D; e += 256) {
    float xv[4];
    load4<T>(x + (size_t)row * ldx + e, xv);
    for (int c = 0; c < n_out; ++c) {
      float wv[4];
      load4<float>(w + (size_t)c * D + e, wv);
#pragma unroll
      for (int j = 0; j < 4; ++j) hs[c] += xv[j] * wv[j];
    }
  }
  for (int c = 0; c < n_out; ++c) {
    const float s = wave_sum(hs[c]);
    if (lane == 0) scores[(size_t)row * n_out + c] = s + b[c];
  }
}

// dx[r,:] (+)= sum_c ds[r,c] w[c,:]
template <typename T>
__global__ __launch_bounds__(256) void head_bwd_data_kernel(const float* __restrict__ ds, const float* __restrict__ w, int rows, int D,
                                                            int n_out, T* __restrict__ dx, int lddx, int accumulate) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int per_row = D / 4;
  const size_t total = (size_t)rows * per_row;
  if (idx >= total) return;
  const int row = (int)(idx / per_row);
  const int e = 4 * (int)(idx % per_row);
  float o[4] = {0.f, 0.f, 0.f, 0.f};
  if (accumulate) load4<T>(dx + (size_t)row * lddx + e, o);
  for (int c = 0; c < n_out; ++c) {
    const float s = ds[(size_t)row * n_out + c];
    float wv[4];
    load4<float>(w + (size_t)c * D + e, wv);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] += s * wv[j];
  }
  store4<T>(dx + (size_t)row * lddx + e, o);
}

// ------------------------------------------------------------------------------------------------
// backward: per-row dx + register-accumulated column reductions
//   slab layout (fp32): partial[block][slot][D], slots: 0 dgamma, 1 dbeta, 2 colsum(dx)
//   MODE 0: LayerNorm backward.  MODE 1: head parameter gradient (slots 0..n_out-1 = dw rows; db via slot 4.. see below)
// ------------------------------------------------------------------------------------------------
// WIDE (D > 2048): blockIdx.y picks a chunk of NV*256 columns whose dx / column sums this workgroup produces; the row
// statistics s1, s2 need the whole row, so the other chunks are read once more for them.
// FULL (D == NV * 256, e.g. 1792 = 7 x 256): every lane owns a valid column in every slot, so the per-slot bounds checks go away --
// with them the divergent branches around the loads and stores, behind which the compiler waits for ALL outstanding memory
// operations (the next row's prefetch, the previous slot's store acknowledgement) before every slot.
// NHG > 0 (head_w given, dy == NULL: the LAST layer of a tagger, whose only incoming gradient is the fused head's): the head's
// PARAMETER gradients come out of this pass too, and the forward of that layer need not store its output y = xhat * gamma + beta.
// With dv[r,e] = sum_c dl[r,c] w[c,e] everything the pass owes is a function of  S_c[e] = sum_r dl[r,c] xhat[r,e]  and the
// scalars  T_c = sum_r dl[r,c]:
//     dgamma[e] = sum_c w[c,e] S_c[e]      dbeta[e] = sum_c w[c,e] T_c      dw[c,e] = gamma[e] S_c[e] + beta[e] T_c      db[c] = T_c
// so the wave accumulates S_c (NHG x NV x 4 registers instead of dgamma + dbeta) and ln_head_final_kernel forms the four outputs
// (slab slots 0 .. NHG-1: S_c, slot NHG: colsum(dx), slot NHG+1: T_c in its first NHG entries).  dw is taken on the UNROUNDED y
// (the forward fed the head the act-dtype rounding of it): the gradient of the function the reference computes.
// EMB: backward of the EMBEDDING LayerNorm (modeling_longformer.py:402-426).  Its dx is only ever summed: over every row into the
// token-type row and over the documents of each position i into the position table.  A wave therefore walks ONE position through a
// chunk of documents (task = chunk * L + i; rows b*L + i, or row0[b] + i of a packed batch), keeps the running sum in registers and
// writes it once per task to dpos_part[chunk][i][:] (fp32; pos_sum_kernel adds the chunks and forms the token-type row) -- dx
// itself is never stored and never re-read (was: 58.7 MB written + read back by embed_bwd_kernel at the BASELINE shape).
// Slab slots: 0 dgamma, 1 dbeta.
struct EmbArgs { int B, L, Bc, nchunks; const int32_t* row0; const int32_t* lengths; float* dpos_part; };

template <typename T, int NV, bool WIDE, bool FULL = false, int NHG = 0, bool EMB = false>
__global__ __launch_bounds__(64 * ROW_WAVES, 2) void ln_bwd_kernel(
    const T* __restrict__ x, const T* __restrict__ dy_, const float* __restrict__ dlogit, const float* __restrict__ head_w, int n_out,
    const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ rstd, int rows, int D,
    T* __restrict__ dx, float* __restrict__ partial, const EmbArgs ea) {
  static_assert(!(EMB && (WIDE || NHG > 0)), "the embedding form has no head and D <= 2048");
  constexpr bool HG = NHG > 0;
  constexpr int NR = HG ? NHG + 1 : EMB ? 2 : 3;       // full-width slots per workgroup slab
  constexpr int NS = HG ? NHG + 2 : NR;                // + the short slot of the T_c scalars
  constexpr int NA = HG ? NHG : 2;                     // per-column accumulator sets besides the dx sum: S_c, or dgamma + dbeta
  const T* __restrict__ dy = HG ? nullptr : dy_;
  __shared__ float red[ROW_WAVES][NR][64 * 4];         // one vector slot at a time is combined through LDS
  // gamma and the head weights of this workgroup's columns live in LDS for the whole kernel: row-invariant, too many for the
  // register file, and as global loads inside the row loop every one of them made the compiler wait for the next row's prefetch
  __shared__ __attribute__((aligned(16))) float gam_s[NV * 256];
  __shared__ __attribute__((aligned(16))) float hw_s[EMB ? 1 : 4][EMB ? 4 : NV * 256];
  // EMB: the running sum of a wave's current task (one position, a chunk of documents) lives in LDS, not in 4 NV registers: with them the
  // D = 1792 instance spilled 16 registers to scratch, and a scratch reload waits (vmcnt is in order) for the next row's prefetch as well
  __shared__ __attribute__((aligned(16))) float task_s[EMB ? ROW_WAVES : 1][EMB ? NV * 256 : 4];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: row / task bookkeeping lives in scalar registers
  const int col0 = WIDE ? (int)blockIdx.y * (NV * 256) : 0;
  for (int idx = threadIdx.x; idx < NV * 256; idx += 64 * ROW_WAVES) {
    const int e = col0 + idx;
    gam_s[idx] = (e < D) ? gamma[e] : 0.f;
    if constexpr (!EMB) {
      if (head_w)
        for (int c = 0; c < n_out; ++c) hw_s[c][idx] = (e < D) ? head_w[(size_t)c * D + e] : 0.f;
    }
  }
  __syncthreads();
  float acc[NA][NV][4];          // HG: S_c;  else [0] = dgamma, [1] = dbeta
  float dxs[NV][4];              // colsum(dx) (EMB: unused, see task_s)
  float tsum[HG ? NHG : 1];
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      dxs[i][j] = 0.f;
#pragma unroll
      for (int c = 0; c < NA; ++c) acc[c][i][j] = 0.f;
    }
#pragma unroll
  for (int c = 0; c < (HG ? NHG : 1); ++c) tsum[c] = 0.f;
  if constexpr (EMB) {
#pragma unroll
    for (int i = 0; i < NV; ++i) *reinterpret_cast<float4*>(&task_s[wave][4 * (lane + 64 * i)]) = make_float4(0.f, 0.f, 0.f, 0.f);
  }

  const float invD = 1.0f / (float)D;
  // the next row's x / dy are fetched (raw, storage precision) before the current row is reduced, so a wave always has
  // a full row of loads in flight behind its arithmetic and its stores
  const int stride = gridDim.x * ROW_WAVES;
  Pack<T, 4> px[NV], pd[HG ? 1 : NV];
  // the row's statistics and head gradients travel with the prefetch (unconditional loads from a pointer that is always valid),
  // so that the wait for them at the top of the next iteration is a counted one that leaves this row's stores in flight
  float mu_n = 0.f, rs_n = 0.f, dl_n[4] = {0.f, 0.f, 0.f, 0.f};
  const float* dlp = head_w ? dlogit : mean;
  const int dl_mul = head_w ? n_out : 1, dl_max = head_w ? n_out - 1 : 0;
  auto fetch = [&](int r) {
    mu_n = mean[r];
    rs_n = rstd[r];
    if constexpr (!EMB) {
#pragma unroll
      for (int c = 0; c < 4; ++c) dl_n[c] = dlp[(size_t)r * dl_mul + min(c, dl_max)];
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = col0 + 4 * (lane + 64 * i);
      if (FULL || e < D) {
        px[i].load(x + (size_t)r * D + e);
        if constexpr (!HG) { if (dy) pd[i].load(dy + (size_t)r * D + e); }
      }
    }
  };
  // ---- which rows this wave visits ---------------------------------------------------------------------------------------
  //   plain: rows w, w + stride, ...          EMB: tasks w, w + stride, ...; inside a task the documents of its chunk, one position
  int task = blockIdx.x * ROW_WAVES + wave;              // EMB: task id; plain: the row itself
  int ti = 0, tb = 0, tbend = 0;
  const int ntasks = EMB ? ea.L * ea.nchunks : 0;
  auto row_at = [&](int b, int i) { return ea.row0 ? ea.row0[b] + i : b * ea.L + i; };
  auto valid = [&](int b, int i) { return !ea.row0 || i < ea.lengths[b]; };
  auto open_task = [&](int t) {                           // -> false when no document of the chunk reaches position ti
    ti = t % ea.L;
    tb = (t / ea.L) * ea.Bc;
    tbend = min(ea.B, tb + ea.Bc);
    while (tb < tbend && !valid(tb, ti)) ++tb;
    return tb < tbend;
  };
  auto write_task = [&](int t) {                          // the finished task's sum: LDS -> dpos_part, and the LDS slot back to zero
    float* o = ea.dpos_part + (size_t)t * D;             // [chunk][i][D] = [t][D]
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = 4 * (lane + 64 * i);
      float4* slot = reinterpret_cast<float4*>(&task_s[EMB ? wave : 0][EMB ? e : 0]);
      const float4 v = *slot;
      if (FULL || e < D) *reinterpret_cast<float4*>(o + e) = v;
      *slot = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto skip_empty_tasks = [&]() {                         // leaves `task` on a task with at least one row, or >= ntasks
    if constexpr (EMB) {
      while (task < ntasks && !open_task(task)) task += stride;    // (packed batches: the host zeroed dpos_part, an empty task writes nothing)
    }
  };
  int row;
  if constexpr (EMB) { skip_empty_tasks(); row = task < ntasks ? row_at(tb, ti) : -1; }
  else row = task < rows ? task : -1;
  if (row >= 0) fetch(row);
  while (row >= 0) {
    const float mu = mu_n, rs = rs_n;
    float xh[NV][4], gy[NV][4];
    float s1 = 0.f, s2 = 0.f;
    const float dl[4] = {dl_n[0], dl_n[1], dl_n[2], dl_n[3]};
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = col0 + 4 * (lane + 64 * i);
      if (FULL || e < D) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          xh[i][j] = px[i].get(j);
          if constexpr (HG) gy[i][j] = 0.f;
          else gy[i][j] = dy ? pd[i].get(j) : 0.f;
        }
      }
    }
    // next row (and, EMB, whether the current one closes its task)
    int nrow = -1, done_task = -1;
    if constexpr (EMB) {
      int b = tb + 1;
      while (b < tbend && !valid(b, ti)) ++b;
      if (b < tbend) { tb = b; nrow = row_at(tb, ti); }
      else {
        done_task = task;
        task += stride;
        skip_empty_tasks();
        if (task < ntasks) nrow = row_at(tb, ti);
      }
    } else {
      if (row + stride < rows) nrow = row + stride;
    }
    if (nrow >= 0) fetch(nrow);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = col0 + 4 * (lane + 64 * i);
      if (FULL || e < D) {
        float dv[4];
        const float4 g4v = *reinterpret_cast<const float4*>(gam_s + (e - col0));
        const float gv[4] = {g4v.x, g4v.y, g4v.z, g4v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) { dv[j] = gy[i][j]; xh[i][j] = (xh[i][j] - mu) * rs; }
        if constexpr (!EMB) {
          if (head_w) {
            for (int c = 0; c < n_out; ++c) {
              const float4 w4v = *reinterpret_cast<const float4*>(hw_s[c] + (e - col0));
              const float wv[4] = {w4v.x, w4v.y, w4v.z, w4v.w};
#pragma unroll
              for (int j = 0; j < 4; ++j) dv[j] += dl[c] * wv[j];
            }
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          gy[i][j] = dv[j] * gv[j];
          s1 += gy[i][j];
          s2 += gy[i][j] * xh[i][j];
          if constexpr (HG) {
#pragma unroll
            for (int c = 0; c < NHG; ++c) acc[c][i][j] += dl[c] * xh[i][j];
          } else {
            acc[0][i][j] += dv[j] * xh[i][j];
            acc[1][i][j] += dv[j];
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) { xh[i][j] = 0.f; gy[i][j] = 0.f; }
      }
    }
    if constexpr (HG) {
#pragma unroll
      for (int c = 0; c < NHG; ++c) tsum[c] += dl[c];
    }
    if constexpr (WIDE) {
      for (int oc = 0; oc < (int)gridDim.y; ++oc) {
        if (oc == (int)blockIdx.y) continue;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const int e = oc * (NV * 256) + 4 * (lane + 64 * i);
          if (FULL || e < D) {
            float xv[4], dv[4] = {0.f, 0.f, 0.f, 0.f}, gv[4];
            load4<T>(x + (size_t)row * D + e, xv);
            if (dy) load4<T>(dy + (size_t)row * D + e, dv);
            load4<float>(gamma + e, gv);
            if (head_w) {
              for (int c = 0; c < n_out; ++c) {
                float wv[4];
                load4<float>(head_w + (size_t)c * D + e, wv);
#pragma unroll
                for (int j = 0; j < 4; ++j) dv[j] += dl[c] * wv[j];
              }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float gyv = dv[j] * gv[j]; s1 += gyv; s2 += gyv * ((xv[j] - mu) * rs); }
          }
        }
      }
    }
    s1 = wave_sum(s1) * invD;
    s2 = wave_sum(s2) * invD;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = col0 + 4 * (lane + 64 * i);
      if (FULL || e < D) {
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          o[j] = rs * (gy[i][j] - s1 - xh[i][j] * s2);
          if constexpr (!EMB && sizeof(T) == 2) o[j] = to_f32(from_f32<T>(o[j]));   // sum what is stored (EMB: nothing is stored, fp32 throughout)
          if constexpr (!EMB) dxs[i][j] += o[j];
        }
        if constexpr (EMB) {
          float4* slot = reinterpret_cast<float4*>(&task_s[wave][e]);
          float4 t = *slot;
          t.x += o[0]; t.y += o[1]; t.z += o[2]; t.w += o[3];
          *slot = t;
        } else {
          store4<T>(dx + (size_t)row * D + e, o);
        }
      }
    }
    if constexpr (EMB) {
      if (done_task >= 0) write_task(done_task);
    }
    row = nrow;
  }
  // combine the ROW_WAVES waves of this workgroup, one vector slot (256 columns) at a time
  float* slab = partial + (size_t)blockIdx.x * NS * D;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int c = 0; c < NA; ++c) red[wave][c][lane * 4 + j] = acc[c][i][j];
      if constexpr (!EMB) red[wave][NA][lane * 4 + j] = dxs[i][j];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < NR * 256; t += 64 * ROW_WAVES) {
      const int slot = t / 256, col = t % 256;
      const int e = col0 + 256 * i + col;
      if (FULL || e < D) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < ROW_WAVES; ++w) s += red[w][slot][col];
        slab[(size_t)slot * D + e] = s;
      }
    }
  }
  if constexpr (HG) {
    __syncthreads();
    if (lane == 0)
      for (int c = 0; c < NHG; ++c) red[wave][c][0] = tsum[c];
    __syncthreads();
    if (threadIdx.x < NHG && blockIdx.y == 0) {
      float s = 0.f;
      for (int w = 0; w < ROW_WAVES; ++w) s += red[w][threadIdx.x][0];
      slab[(size_t)(NHG + 1) * D + threadIdx.x] = s;
    }
  }
}

// Final step of the NHG form of ln_bwd_kernel: reduces the slabs (S_c, colsum(dx), T_c) over the workgroups in a fixed order and forms
// dgamma, dbeta, dxsum, dw[c], db[c] (see the kernel's header).  Workgroup = 32 columns x 32 block-groups, as slab_reduce_kernel.
// The loads of a lane go out eight blocks at a time (the adds keep their order): one dependent L2 round trip per block made this launch 10 us
// for 11 MB.  loss_part (the one-pass tail, else NULL): the last workgroup's second wave also finishes the loss -- the sum of the workgroups'
// partials, lane-strided then wave_sum, times 1 / loss_out[1] -- which used to be a launch of its own.
template <int NHG>
__global__ __launch_bounds__(256) void ln_head_final_kernel(const float* __restrict__ partial, int nblocks, int D, const float* __restrict__ head_w,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dxsum,
                                                            float* __restrict__ dhead_w, float* __restrict__ dhead_b,
                                                            const float* __restrict__ loss_part, float* __restrict__ loss_out) {
  constexpr int NS = NHG + 2;
  __shared__ float4 red[NHG + 1][32][8];
  __shared__ float tred[NHG][256];
  const int c4 = threadIdx.x & 7, grp = threadIdx.x >> 3;
  const int e = blockIdx.x * 32 + 4 * c4;
  const size_t bs = (size_t)NS * D;
  float4 s[NHG + 1];
#pragma unroll
  for (int k = 0; k <= NHG; ++k) s[k] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e < D) {
    const float* p = partial + e;
    int b = grp;
    for (; b + 32 * 7 < nblocks; b += 32 * 8) {
      float4 v[8][NHG + 1];
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k <= NHG; ++k) v[u][k] = *reinterpret_cast<const float4*>(p + (size_t)(b + 32 * u) * bs + (size_t)k * D);
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int k = 0; k <= NHG; ++k) { s[k].x += v[u][k].x; s[k].y += v[u][k].y; s[k].z += v[u][k].z; s[k].w += v[u][k].w; }
    }
    for (; b < nblocks; b += 32) {
#pragma unroll
      for (int k = 0; k <= NHG; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(p + (size_t)b * bs + (size_t)k * D);
        s[k].x += v.x; s[k].y += v.y; s[k].z += v.z; s[k].w += v.w;
      }
    }
  }
#pragma unroll
  for (int k = 0; k <= NHG; ++k) red[k][grp][c4] = s[k];
  // T_c: every workgroup sums the nblocks scalars itself (a few hundred values), in the same fixed order
#pragma unroll
  for (int c = 0; c < NHG; ++c) {
    float t = 0.f;
    for (int b = threadIdx.x; b < nblocks; b += 256) t += partial[(size_t)b * bs + (size_t)(NHG + 1) * D + c];
    tred[c][threadIdx.x] = t;
  }
  __syncthreads();
  if (loss_part && blockIdx.x == gridDim.x - 1 && (threadIdx.x >> 6) == 1) {
    float ls = 0.f;
    for (int b = threadIdx.x & 63; b < nblocks; b += 64) ls += loss_part[b];
    ls = wave_sum(ls);
    if ((threadIdx.x & 63) == 0) { const float cnt = loss_out[1]; loss_out[0] = cnt > 0.f ? ls * (1.f / cnt) : 0.f; }
  }
  if (threadIdx.x < 64) {                                  // (wave 0 holds the eight lanes of grp 0)
    // T_c = sum of the 256 per-thread sums: four per lane in index order, then the wave's fixed-order butterfly -- the same value in every lane
    float T[NHG];
#pragma unroll
    for (int c = 0; c < NHG; ++c) {
      const float4 q = *reinterpret_cast<const float4*>(&tred[c][4 * threadIdx.x]);
      T[c] = wave_sum(((q.x + q.y) + q.z) + q.w);
    }
    if (grp != 0 || e >= D) return;
    float S[NHG + 1][4];
#pragma unroll
    for (int k = 0; k <= NHG; ++k) {
      float4 t = red[k][0][c4];
      for (int gi = 1; gi < 32; ++gi) { const float4 v = red[k][gi][c4]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
      S[k][0] = t.x; S[k][1] = t.y; S[k][2] = t.z; S[k][3] = t.w;
    }
    float gv[4], bv[4], dg[4] = {0.f, 0.f, 0.f, 0.f}, dbt[4] = {0.f, 0.f, 0.f, 0.f};
    load4<float>(gamma + e, gv);
    load4<float>(beta + e, bv);
#pragma unroll
    for (int c = 0; c < NHG; ++c) {
      float wv[4], dwv[4];
      load4<float>(head_w + (size_t)c * D + e, wv);
#pragma unroll
      for (int j = 0; j < 4; ++j) { dg[j] += wv[j] * S[c][j]; dbt[j] += wv[j] * T[c]; dwv[j] = gv[j] * S[c][j] + bv[j] * T[c]; }
      store4<float>(dhead_w + (size_t)c * D + e, dwv);
    }
    store4<float>(dgamma + e, dg);
    store4<float>(dbeta + e, dbt);
    if (dxsum) store4<float>(dxsum + e, S[NHG]);
    if (blockIdx.x == 0 && c4 == 0)
      for (int c = 0; c < NHG; ++c) dhead_b[c] = T[c];
  }
}

// ------------------------------------------------------------------------------------------------
// The LAST layer's tail in ONE pass (training): LayerNorm forward + tagger head + masked loss + loss gradient + head data gradient + LayerNorm
// backward, row by row.  (modeling_longformer.py:1127-1131 -> models/CRF.py:579-595 -> focal_loss.py:38-57, and their backward.)
// Unfused these are four launches around a 16 384-float tensor: ln_fwd_kernel (reads s2, writes scores), tagger_loss_kernel (+ final), a scale,
// ln_bwd_kernel<NHG> (reads s2 AGAIN, writes ds2).  Everything between the two reads of a row is a function of that row alone -- the loss
// normaliser 1 / (number of averaged rows) follows from the lengths (BCE / focal) or the targets (CE) without looking at a score -- so one wave
// can take a row through all of it while it holds the row in registers: s2 is read once, scores / loss partials / ds2 written once.
// Arithmetic, in order, is that of the kernels it replaces (row_stats; y = xhat * gamma + beta rounded to the act dtype for the head's dot product,
// summed per lane over i then j, wave_sum; focal_elem / bce_elem / ce2_elem; g * inv, then * grad_scale; the NHG branch of ln_bwd_kernel):
// scores and every gradient are bitwise those of the unfused path, the loss differs by the grouping of its partial sums
// (tests/test_gpu_norm_fused.py::test_last_layer_tail_in_one_pass).  D == NV * 256 only; slabs as the NHG form, finished by ln_head_final_kernel.
struct TailArgs {
  int kind, B, L, Lt, n_rows;
  const float* targets; const int32_t* lengths; const int32_t* row_src;
  float alpha, gamma_f, grad_scale;
  float* scores; float* loss_part; float* loss_out;
};

template <typename T, int NV, int NHG>
__global__ __launch_bounds__(64 * ROW_WAVES, 2) void ln_tail_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                                    const float* __restrict__ head_w, const float* __restrict__ head_b, int rows, int D,
                                                                    T* __restrict__ dx, float* __restrict__ partial, const TailArgs ta) {
  constexpr int NS = NHG + 2;
  __shared__ float red[ROW_WAVES][NHG + 1][64 * 4];
  __shared__ __attribute__((aligned(16))) float gam_s[NV * 256];
  __shared__ __attribute__((aligned(16))) float bet_s[NV * 256];
  __shared__ __attribute__((aligned(16))) float hw_s[NHG][NV * 256];
  __shared__ float cred[ROW_WAVES];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int idx = threadIdx.x; idx < NV * 256; idx += 64 * ROW_WAVES) {
    gam_s[idx] = gamma[idx];
    bet_s[idx] = beta[idx];
#pragma unroll
    for (int c = 0; c < NHG; ++c) hw_s[c][idx] = head_w[(size_t)c * D + idx];
  }
  // how many rows the loss averages over (tagger_loss_kernel's pass 1): every workgroup counts for itself, in the same order
  float cnt = 0.f;
  if (ta.kind == MTS_LOSS_CE) {                   // ignore_index = -1 (CRF.py:298)
    for (int r = threadIdx.x; r < rows; r += 64 * ROW_WAVES) {
      const int src = ta.row_src ? ta.row_src[r] : r;
      cnt += (ta.targets[(size_t)(src / ta.L) * ta.Lt + src % ta.L] != -1.f) ? 1.f : 0.f;
    }
  } else {                                        // un-pad loop (CRF.py:348-350): rows i < len_b
    for (int b = threadIdx.x; b < ta.B; b += 64 * ROW_WAVES) cnt += (float)(ta.lengths ? min(max(ta.lengths[b], 0), ta.L) : ta.L);
  }
  cnt = wave_sum(cnt);
  if (lane == 0) cred[wave] = cnt;
  __syncthreads();
  cnt = ((cred[0] + cred[1]) + cred[2]) + cred[3];
  const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
  float hb[NHG];
#pragma unroll
  for (int c = 0; c < NHG; ++c) hb[c] = head_b[c];

  float acc[NHG][NV][4], dxs[NV][4], tsum[NHG];
#pragma unroll
  for (int i = 0; i < NV; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      dxs[i][j] = 0.f;
#pragma unroll
      for (int c = 0; c < NHG; ++c) acc[c][i][j] = 0.f;
    }
#pragma unroll
  for (int c = 0; c < NHG; ++c) tsum[c] = 0.f;
  float lacc = 0.f;
  const float invD = 1.0f / (float)D;
  const int stride = gridDim.x * ROW_WAVES;
  Pack<T, 4> px[NV];
  float y_n = 0.f;
  int i_n = 0, len_n = 0;
  auto fetch = [&](int r) {
    const int src = ta.row_src ? ta.row_src[r] : r;
    const int b = src / ta.L;
    i_n = src - b * ta.L;
    y_n = ta.targets[(size_t)b * ta.Lt + i_n];
    len_n = ta.lengths ? ta.lengths[b] : ta.L;
#pragma unroll
    for (int i = 0; i < NV; ++i) px[i].load(x + (size_t)r * D + 4 * (lane + 64 * i));
  };
  int row = blockIdx.x * ROW_WAVES + wave;
  if (row >= rows) row = -1;
  if (row >= 0) fetch(row);
  while (row >= 0) {
    const float yt = y_n;
    const int ipos = i_n, len = len_n;
    float xh[NV][4];
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) xh[i][j] = px[i].get(j);
    const int nrow = (row + stride < rows) ? row + stride : -1;
    if (nrow >= 0) fetch(nrow);
    // ---- LayerNorm forward + head (ln_fwd_kernel)
    float mu, rs;
    row_stats<NV>(xh, D, lane, mu, rs, eps);
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) xh[i][j] = (xh[i][j] - mu) * rs;          // xhat from here on (the forward's own first two operations)
    float hs[NHG];
#pragma unroll
    for (int c = 0; c < NHG; ++c) hs[c] = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = 4 * (lane + 64 * i);
      const float4 g4v = *reinterpret_cast<const float4*>(gam_s + e), b4v = *reinterpret_cast<const float4*>(bet_s + e);
      const float gv[4] = {g4v.x, g4v.y, g4v.z, g4v.w}, bv[4] = {b4v.x, b4v.y, b4v.z, b4v.w};
      float yv[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) yv[j] = xh[i][j] * gv[j] + bv[j];
#pragma unroll
      for (int c = 0; c < NHG; ++c) {
        const float4 w4v = *reinterpret_cast<const float4*>(hw_s[c] + e);
        const float wv[4] = {w4v.x, w4v.y, w4v.z, w4v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) hs[c] += to_f32(from_f32<T>(yv[j])) * wv[j];
      }
    }
    float sc[NHG];
#pragma unroll
    for (int c = 0; c < NHG; ++c) sc[c] = wave_sum(hs[c]) + hb[c];
    asm volatile("" ::: "memory");          // (the row-invariant LDS operands are read again in every phase: kept in registers across the
                                            // wave reductions they are 56-84 registers the kernel does not have)
    // ---- loss + its gradient wrt the scores (tagger_loss_kernel; the same value in every lane)
    float dl[NHG];
#pragma unroll
    for (int c = 0; c < NHG; ++c) dl[c] = 0.f;
    if constexpr (NHG == 2) {
      if (yt != -1.f) {
        float g0, g1;
        lacc += ce2_elem(sc[0], sc[1], (int)yt, g0, g1);
        dl[0] = g0 * inv;
        dl[1] = g1 * inv;
      }
    } else {
      if (ipos < len) {
        float gr;
        lacc += (ta.kind == MTS_LOSS_FOCAL) ? focal_elem(sc[0], yt, ta.alpha, ta.gamma_f, gr) : bce_elem(sc[0], yt, gr);
        dl[0] = gr * inv;
      }
    }
    if (ta.grad_scale != 1.0f) {
#pragma unroll
      for (int c = 0; c < NHG; ++c) dl[c] *= ta.grad_scale;
    }
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < NHG; ++c) ta.scores[(size_t)row * NHG + c] = sc[c];
    }
    // ---- head data gradient + LayerNorm backward (ln_bwd_kernel, NHG branch)
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = 4 * (lane + 64 * i);
      float dv[4] = {0.f, 0.f, 0.f, 0.f};
      const float4 g4v = *reinterpret_cast<const float4*>(gam_s + e);
      const float gv[4] = {g4v.x, g4v.y, g4v.z, g4v.w};
#pragma unroll
      for (int c = 0; c < NHG; ++c) {
        const float4 w4v = *reinterpret_cast<const float4*>(hw_s[c] + e);
        const float wv[4] = {w4v.x, w4v.y, w4v.z, w4v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) dv[j] += dl[c] * wv[j];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float gyv = dv[j] * gv[j];
        s1 += gyv;
        s2 += gyv * xh[i][j];
#pragma unroll
        for (int c = 0; c < NHG; ++c) acc[c][i][j] += dl[c] * xh[i][j];
      }
    }
#pragma unroll
    for (int c = 0; c < NHG; ++c) tsum[c] += dl[c];
    s1 = wave_sum(s1) * invD;
    s2 = wave_sum(s2) * invD;
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = 4 * (lane + 64 * i);
      // (dv * gamma is formed a second time, from LDS operands, instead of being kept: 28 registers at D = 1792, the same bits)
      float dv[4] = {0.f, 0.f, 0.f, 0.f};
      const float4 g4v = *reinterpret_cast<const float4*>(gam_s + e);
      const float gv[4] = {g4v.x, g4v.y, g4v.z, g4v.w};
#pragma unroll
      for (int c = 0; c < NHG; ++c) {
        const float4 w4v = *reinterpret_cast<const float4*>(hw_s[c] + e);
        const float wv[4] = {w4v.x, w4v.y, w4v.z, w4v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) dv[j] += dl[c] * wv[j];
      }
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        o[j] = rs * (dv[j] * gv[j] - s1 - xh[i][j] * s2);
        if constexpr (sizeof(T) == 2) o[j] = to_f32(from_f32<T>(o[j]));
        dxs[i][j] += o[j];
      }
      store4<T>(dx + (size_t)row * D + e, o);
    }
    row = nrow;
  }
  // ---- combine the waves of the workgroup: slabs (as ln_bwd_kernel NHG) and the loss partial
  float* slab = partial + (size_t)blockIdx.x * NS * D;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int c = 0; c < NHG; ++c) red[wave][c][lane * 4 + j] = acc[c][i][j];
      red[wave][NHG][lane * 4 + j] = dxs[i][j];
    }
    __syncthreads();
    for (int t = threadIdx.x; t < (NHG + 1) * 256; t += 64 * ROW_WAVES) {
      const int slot = t / 256, col = t % 256;
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < ROW_WAVES; ++w) s += red[w][slot][col];
      slab[(size_t)slot * D + 256 * i + col] = s;
    }
  }
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < NHG; ++c) red[wave][c][0] = tsum[c];
    red[wave][NHG][0] = lacc;
  }
  __syncthreads();
  if (threadIdx.x <= NHG) {
    float s = 0.f;
    for (int w = 0; w < ROW_WAVES; ++w) s += red[w][threadIdx.x][0];
    if (threadIdx.x < NHG) slab[(size_t)(NHG + 1) * D + threadIdx.x] = s;
    else { ta.loss_part[blockIdx.x] = s; if (blockIdx.x == 0) ta.loss_out[1] = cnt; }
  }
}

// Final step of the EMB form: dpos[i][e] = sum_chunks part[chunk][i][e] for every position, in a fixed order, and the column sums
// of the workgroup's positions -> tpart[blockIdx.y][e] (their sum over y is the token-type row = the pre-LN gradient summed over all
// rows; slab_reduce_kernel adds them).  Workgroup = 32 columns x 32 row-groups over L / gridDim.y positions.
#define POS_SPLIT 8
__global__ __launch_bounds__(256) void pos_sum_kernel(const float* __restrict__ part, int nchunks, int L, int D, float* __restrict__ dpos_rows,
                                                      float* __restrict__ tpart) {
  __shared__ float4 red[32][8];
  const int c4 = threadIdx.x & 7, grp = threadIdx.x >> 3;
  const int e = blockIdx.x * 32 + 4 * c4;
  const int per = (L + gridDim.y - 1) / gridDim.y, i0 = blockIdx.y * per, i1 = min(L, i0 + per);
  float4 tot = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e < D) {
    for (int i = i0 + grp; i < i1; i += 32) {
      float4 s = *reinterpret_cast<const float4*>(part + (size_t)i * D + e);
      int c = 1;
      for (; c + 6 < nchunks; c += 7) {                   // seven loads in flight, the adds in chunk order
        float4 v[7];
#pragma unroll
        for (int u = 0; u < 7; ++u) v[u] = *reinterpret_cast<const float4*>(part + ((size_t)(c + u) * L + i) * D + e);
#pragma unroll
        for (int u = 0; u < 7; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
      }
      for (; c < nchunks; ++c) {
        const float4 v = *reinterpret_cast<const float4*>(part + ((size_t)c * L + i) * D + e);
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
      }
      *reinterpret_cast<float4*>(dpos_rows + (size_t)i * D + e) = s;
      tot.x += s.x; tot.y += s.y; tot.z += s.z; tot.w += s.w;
    }
  }
  red[grp][c4] = tot;
  __syncthreads();
  if (grp == 0 && e < D) {
    float4 t = red[0][c4];
    for (int gi = 1; gi < 32; ++gi) { const float4 v = red[gi][c4]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
    *reinterpret_cast<float4*>(tpart + (size_t)blockIdx.y * D + e) = t;
  }
}

// head parameter gradient: dw[c,:] = sum_r ds[r,c] x[r,:]   (slab slots 0..3 = c); db handled by the final kernel
// FULL (D == NV * 256, one column chunk): no bounds checks, hence no divergent branches around the loads (see ln_bwd_kernel);
// the next row (x and its head gradients) is fetched before the current one is accumulated.
template <typename T, int NV, bool FULL = false>
__global__ __launch_bounds__(64 * ROW_WAVES) void head_bwd_params_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ ds,
                                                                       int n_out, int rows, int D, float* __restrict__ partial) {
  __shared__ float red[ROW_WAVES][4][64 * 4];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int col0 = (int)blockIdx.y * (NV * 256);                  // D > NV*256: blockIdx.y picks the column chunk
  float dw[4][NV][4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) dw[c][i][j] = 0.f;
  float dbs[4] = {0.f, 0.f, 0.f, 0.f};
  Pack<T, 4> px[NV];
  float dl_n[4] = {0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](int r) {
#pragma unroll
    for (int c = 0; c < 4; ++c) dl_n[c] = ds[(size_t)r * n_out + min(c, n_out - 1)];   // unconditional; columns >= n_out are zeroed at use
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = col0 + 4 * (lane + 64 * i);
      if (FULL || e < D) px[i].load(x + (size_t)r * ldx + e);
    }
  };
  const int stride = gridDim.x * ROW_WAVES;
  int row = blockIdx.x * ROW_WAVES + wave;
  if (row < rows) fetch(row);
  for (; row < rows; row += stride) {
    float dl[4], xv[NV][4];
#pragma unroll
    for (int c = 0; c < 4; ++c) dl[c] = (c < n_out) ? dl_n[c] : 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int e = col0 + 4 * (lane + 64 * i);
#pragma unroll
      for (int j = 0; j < 4; ++j) xv[i][j] = (FULL || e < D) ? px[i].get(j) : 0.f;
    }
    if (row + stride < rows) fetch(row + stride);
#pragma unroll
    for (int c = 0; c < 4; ++c) dbs[c] += dl[c];
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) dw[c][i][j] += dl[c] * xv[i][j];
  }
  // slab: [block][5][D] -> slots 0..3 dw rows, slot 4: first 4 entries = db partial of this workgroup
  float* slab = partial + (size_t)blockIdx.x * 5 * D;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int j = 0; j < 4; ++j) red[wave][c][lane * 4 + j] = dw[c][i][j];
    __syncthreads();
    for (int t = threadIdx.x; t < 4 * 256; t += 64 * ROW_WAVES) {
      const int slot = t / 256, col = t % 256;
      const int e = col0 + 256 * i + col;
      if (e < D) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < ROW_WAVES; ++w) s += red[w][slot][col];
        slab[(size_t)slot * D + e] = s;
      }
    }
  }
  __syncthreads();
  if (lane == 0)
    for (int c = 0; c < 4; ++c) red[wave][c][0] = dbs[c];
  __syncthreads();
  if (threadIdx.x < 4) {
    float s = 0.f;
    for (int w = 0; w < ROW_WAVES; ++w) s += red[w][threadIdx.x][0];
    slab[(size_t)4 * D + threadIdx.x] = s;
  }
}

// out[slot][e] = sum_blocks partial[block][slot][e].  Workgroup = 32 columns (8 lanes x float4) x 32 block-groups: every
// lane streams 16-byte loads with up to 8 in flight, then the 32 groups are combined through LDS in a fixed order
// (bitwise reproducible).  D must be a multiple of 4 (the row kernels above require it already).
#define SR_COLS 32
// one output row per slot (NULL: skipped); slot `short_slot` has only short_len entries; slot `alt_slot` (if >= 0) is reduced from
// another buffer, alt[alt_nb][D] (one row per partial)
struct SlabOuts { float* o[6]; int short_slot, short_len; const float* alt; int alt_slot, alt_nb; };
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ partial, int nblocks, int nslots, int D, const SlabOuts so) {
  // (partial / nblocks / nslots are re-pointed below for the slot that lives in another buffer)
  __shared__ float4 red[32][8];
  const int c4 = threadIdx.x & 7, grp = threadIdx.x >> 3;
  const int e = blockIdx.x * SR_COLS + 4 * c4;
  const int slot = blockIdx.y;
  float* o = so.o[slot];
  if (!o) return;
  if (slot == so.alt_slot) { partial = so.alt - (size_t)slot * D; nblocks = so.alt_nb; nslots = 1; }
  const int len = (slot == so.short_slot) ? so.short_len : D;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e < len) {
    const float* p = partial + (size_t)slot * D + e;
    const size_t bs = (size_t)nslots * D;
    int b = grp;
    for (; b + 32 * 7 < nblocks; b += 32 * 8) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(p + (size_t)(b + 32 * u) * bs);
#pragma unroll
      for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; b < nblocks; b += 32) {
      const float4 v = *reinterpret_cast<const float4*>(p + (size_t)b * bs);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  red[grp][c4] = s;
  __syncthreads();
  if (grp == 0 && e < len) {
    float4 t = red[0][c4];
#pragma unroll
    for (int gidx = 1; gidx < 32; ++gidx) { const float4 v = red[gidx][c4]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
    if (e + 3 < len) *reinterpret_cast<float4*>(o + e) = t;
    else { const float tv[4] = {t.x, t.y, t.z, t.w}; for (int j = 0; j < 4 && e + j < len; ++j) o[e + j] = tv[j]; }
  }
}

// used by band_attn.hip for the fused q/k/v bias gradient
int mts_slab_reduce_rows(hipStream_t st, const float* partial, int nblocks, int D, float* out) {
  SlabOuts so = {{out, nullptr, nullptr, nullptr, nullptr, nullptr}, -1, 0, nullptr, -1, 0};
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(ceil_div(D, SR_COLS), 1), dim3(256), 0, st, partial, nblocks, 1, D, so);
  MTS_LAUNCH_CHECK("slab_reduce");
  return MTS_OK;
}

// dpos[pos_offset+i,:] += sum_b dpre[b,i,:]
template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const T* __restrict__ dpre, int B, int L, int D, float* __restrict__ dpos, int pos_offset,
                                                        const int32_t* __restrict__ row0, const int32_t* __restrict__ lengths) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int per_row = D / 4;
  if (idx >= (size_t)L * per_row) return;
  const int i = (int)(idx / per_row), e = 4 * (int)(idx % per_row);
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  int b = 0;
  if (row0) {                                // packed rows: document b owns rows row0[b] .. row0[b] + lengths[b] - 1
    for (; b < B; ++b) {
      if (i < lengths[b]) {
        float v[4];
        load4<T>(dpre + ((size_t)row0[b] + i) * D + e, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] += v[j];
      }
    }
  }
  for (; b + 7 < B; b += 8) {            // 8 documents' loads in flight per lane
    float v[8][4];
#pragma unroll
    for (int u = 0; u < 8; ++u) load4<T>(dpre + ((size_t)(b + u) * L + i) * D + e, v[u]);
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) s[j] += v[u][j];
  }
  for (; b < B; ++b) {
    float v[4];
    load4<T>(dpre + ((size_t)b * L + i) * D + e, v);
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] += v[j];
  }
  float* o = dpos + (size_t)(pos_offset + i) * D + e;
  float cur[4];
  load4<float>(o, cur);
#pragma unroll
  for (int j = 0; j < 4; ++j) cur[j] += s[j];
  store4<float>(o, cur);
}
// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static inline int pick_nv(int D) {
  const int need = ceil_div(D, 256);
  if (need <= 1) return 1;
  if (need <= 2) return 2;
  if (need <= 4) return 4;
  if (need == 7) return 7;            // D = 1792, the embedding width of the reference's configs: no idle vector slot
  if (need <= 8) return 8;
  if (need <= 16) return 16;
  return 0;
}

template <typename F> static inline void dispatch_nv(int nv, F&& f) {
  switch (nv) {
    case 1: f(std::integral_constant<int, 1>{}); break;
    case 2: f(std::integral_constant<int, 2>{}); break;
    case 4: f(std::integral_constant<int, 4>{}); break;
    case 7: f(std::integral_constant<int, 7>{}); break;
    case 8: f(std::integral_constant<int, 8>{}); break;
    default: f(std::integral_constant<int, 16>{}); break;
  }
}
template <typename F> static inline void dispatch_nv8(int nv, F&& f) {   // kernels that keep several rows of state: D <= 2048
  switch (nv) {
    case 1: f(std::integral_constant<int, 1>{}); break;
    case 2: f(std::integral_constant<int, 2>{}); break;
    case 4: f(std::integral_constant<int, 4>{}); break;
    case 7: f(std::integral_constant<int, 7>{}); break;
    default: f(std::integral_constant<int, 8>{}); break;
  }
}

template <typename T, int EMBED>
static int ln_fwd_launch(hipStream_t st, const void* x, const float* pos, int pos_offset, int L, const float* type0, const float* gamma,
                         const float* beta, float eps, int rows, int D, void* y, void* pre, float* mean, float* rstd,
                         const float* head_w, const float* head_b, int n_out, float* scores, const int32_t* row_src = nullptr,
                         const float* x2 = nullptr, int D1 = 0) {
  const int nv = pick_nv(D);
  MTS_UNSUPPORTED(nv > 0 && D % 4 == 0, "layernorm: D=%d must be a multiple of 4 and <= 4096", D);
  dim3 grid(ceil_div(rows, ROW_WAVES)), block(64 * ROW_WAVES);
  dispatch_nv(nv, [&](auto nvc) {
    constexpr int NV = decltype(nvc)::value;
    if (D == NV * 256)
      hipLaunchKernelGGL((ln_fwd_kernel<T, NV, EMBED, true>), grid, block, 0, st, x, pos, pos_offset, L, type0, gamma, beta, eps, rows, D, (T*)y,
                         (T*)pre, mean, rstd, head_w, head_b, n_out, scores, row_src, x2, D1);
    else
      hipLaunchKernelGGL((ln_fwd_kernel<T, NV, EMBED>), grid, block, 0, st, x, pos, pos_offset, L, type0, gamma, beta, eps, rows, D, (T*)y,
                         (T*)pre, mean, rstd, head_w, head_b, n_out, scores, row_src, x2, D1);
  });
  MTS_LAUNCH_CHECK("layernorm_fwd");
  return MTS_OK;
}

static int embed_ln_fwd(void* stream, int dtype, int B, int L, int D, const float* x, const float* x2, int D1, const float* pos, int pos_offset,
                        const float* type0, const float* gamma, const float* beta, float eps, void* y, void* pre, float* mean, float* rstd,
                        const int32_t* row_src, int n_rows, const char* who) {
  MTS_CHECK_ARG(B > 0 && L > 0 && D > 0 && x && pos && type0 && gamma && beta && y, "%s: bad arguments", who);
  MTS_CHECK_ARG(!row_src || (n_rows > 0 && n_rows <= B * L), "%s: packed form needs 0 < n_rows <= B*L", who);
  MTS_CHECK_ARG(!x2 || (D1 > 0 && D1 < D && D1 % 4 == 0), "%s: the first part's width must be a multiple of 4 inside (0, D)", who);
  const int rows = row_src ? n_rows : B * L;
  if (dtype == MTS_F32)
    return ln_fwd_launch<float, 1>((hipStream_t)stream, x, pos, pos_offset, L, type0, gamma, beta, eps, rows, D, y, pre, mean, rstd,
                                      nullptr, nullptr, 0, nullptr, row_src, x2, D1);
  if (dtype == MTS_BF16)
    return ln_fwd_launch<bf16_t, 1>((hipStream_t)stream, x, pos, pos_offset, L, type0, gamma, beta, eps, rows, D, y, pre, mean, rstd,
                                       nullptr, nullptr, 0, nullptr, row_src, x2, D1);
  mts_set_error("%s: bad dtype %d", who, dtype);
  return MTS_ERR_INVALID;
}

extern "C" int mts_embed_layernorm_fwd(void* stream, int dtype, int B, int L, int D, const float* x, const float* pos, int pos_offset,
                                       const float* type0, const float* gamma, const float* beta, float eps, void* y, void* pre,
                                       float* mean, float* rstd, const int32_t* row_src, int n_rows) {
  return embed_ln_fwd(stream, dtype, B, L, D, x, nullptr, 0, pos, pos_offset, type0, gamma, beta, eps, y, pre, mean, rstd, row_src, n_rows,
                      "mts_embed_layernorm_fwd");
}

// the embedding block on a bf16 batch (one source, bf16 activations): bit for bit what mts_embed_layernorm_fwd gives on the fp32 values of the same numbers
extern "C" int mts_embed_layernorm_fwd_x16(void* stream, int B, int L, int D, const void* x_bf16, const float* pos, int pos_offset, const float* type0,
                                           const float* gamma, const float* beta, float eps, void* y, void* pre, float* mean, float* rstd,
                                           const int32_t* row_src, int n_rows) {
  MTS_CHECK_ARG(B > 0 && L > 0 && D > 0 && x_bf16 && pos && type0 && gamma && beta && y, "mts_embed_layernorm_fwd_x16: bad arguments");
  MTS_CHECK_ARG(!row_src || (n_rows > 0 && n_rows <= B * L), "mts_embed_layernorm_fwd_x16: packed form needs 0 < n_rows <= B*L");
  MTS_CHECK_ARG(((uintptr_t)x_bf16 & 7) == 0, "mts_embed_layernorm_fwd_x16: x must be 8-byte aligned");
  const int rows = row_src ? n_rows : B * L;
  return ln_fwd_launch<bf16_t, 2>((hipStream_t)stream, x_bf16, pos, pos_offset, L, type0, gamma, beta, eps, rows, D, y, pre, mean, rstd, nullptr, nullptr, 0,
                                  nullptr, row_src, nullptr, 0);
}

extern "C" int mts_embed_layernorm_fwd2(void* stream, int dtype, int B, int L, int D1, int D2, const float* x1, const float* x2, const float* pos,
                                        int pos_offset, const float* type0, const float* gamma, const float* beta, float eps, void* y, void* pre,
                                        float* mean, float* rstd, const int32_t* row_src, int n_rows) {
  MTS_CHECK_ARG(x2 && D2 > 0 && D2 % 4 == 0, "mts_embed_layernorm_fwd2: second part missing or its width not a multiple of 4");
  return embed_ln_fwd(stream, dtype, B, L, D1 + D2, x1, x2, D1, pos, pos_offset, type0, gamma, beta, eps, y, pre, mean, rstd, row_src, n_rows,
                      "mts_embed_layernorm_fwd2");
}

extern "C" int mts_layernorm_fwd(void* stream, int dtype, int rows, int D, const void* x, const float* gamma, const float* beta, float eps,
                                 void* y, float* mean, float* rstd, const float* head_w, const float* head_b, int n_out, float* scores) {
  MTS_CHECK_ARG(rows > 0 && D > 0 && x && gamma && beta && (y || (head_w && mean && rstd)), "mts_layernorm_fwd: bad arguments (y may be NULL only with a fused head and saved statistics)");
  MTS_CHECK_ARG(!head_w || (head_b && scores && n_out >= 1 && n_out <= 4), "mts_layernorm_fwd: fused head needs head_b, scores, n_out<=4");
  if (dtype == MTS_F32)
    return ln_fwd_launch<float, 0>((hipStream_t)stream, x, nullptr, 0, 1, nullptr, gamma, beta, eps, rows, D, y, nullptr, mean, rstd,
                                       head_w, head_b, n_out, scores);
  if (dtype == MTS_BF16)
    return ln_fwd_launch<bf16_t, 0>((hipStream_t)stream, x, nullptr, 0, 1, nullptr, gamma, beta, eps, rows, D, y, nullptr, mean, rstd,
                                        head_w, head_b, n_out, scores);
  mts_set_error("mts_layernorm_fwd: bad dtype %d", dtype);
  return MTS_ERR_INVALID;
}

extern "C" size_t mts_layernorm_bwd_workspace(int D) { return (size_t)BWD_MAX_BLOCKS * 6 * (size_t)D * sizeof(float); }

template <typename T>
static int ln_bwd_launch(hipStream_t st, int rows, int D, const void* x, const void* dy, const float* dlogit, const float* head_w, int n_out,
                         const float* gamma, const float* beta, const float* mean, const float* rstd, void* dx, float* dgamma, float* dbeta,
                         float* dxsum, float* dhead_w, float* dhead_b, void* partial) {
  const int nv = pick_nv(D);
  MTS_UNSUPPORTED(nv > 0 && D % 4 == 0, "layernorm_bwd: D=%d must be a multiple of 4 and <= 4096", D);
  const int blocks = std::min(BWD_MAX_BLOCKS, ceil_div(rows, ROW_WAVES));
  const int nhg = dhead_w ? n_out : 0;
  const EmbArgs ea = {0, 0, 0, 0, nullptr, nullptr, nullptr};
  auto go = [&](auto k, dim3 grid) {
    hipLaunchKernelGGL(k, grid, dim3(64 * ROW_WAVES), 0, st, (const T*)x, (const T*)dy, dlogit, head_w, n_out, gamma, mean, rstd, rows, D,
                       (T*)dx, (float*)partial, ea);
  };
  if (nv <= 8) {
    dispatch_nv8(nv, [&](auto nvc) {
      constexpr int NV = decltype(nvc)::value;
      if (D == NV * 256) {
        if (nhg == 1) go(ln_bwd_kernel<T, NV, false, true, 1>, dim3(blocks));
        else if (nhg == 2) go(ln_bwd_kernel<T, NV, false, true, 2>, dim3(blocks));
        else go(ln_bwd_kernel<T, NV, false, true>, dim3(blocks));
      } else {
        if (nhg == 1) go(ln_bwd_kernel<T, NV, false, false, 1>, dim3(blocks));
        else if (nhg == 2) go(ln_bwd_kernel<T, NV, false, false, 2>, dim3(blocks));
        else go(ln_bwd_kernel<T, NV, false>, dim3(blocks));
      }
    });
  } else {   // 2048 < D <= 4096 (e.g. 768 + 1536 = 2304): two column chunks of 2048
    if (nhg == 1) go(ln_bwd_kernel<T, 8, true, false, 1>, dim3(blocks, ceil_div(D, 2048)));
    else if (nhg == 2) go(ln_bwd_kernel<T, 8, true, false, 2>, dim3(blocks, ceil_div(D, 2048)));
    else go(ln_bwd_kernel<T, 8, true>, dim3(blocks, ceil_div(D, 2048)));
  }
  if (nhg == 1)
    hipLaunchKernelGGL(ln_head_final_kernel<1>, dim3(ceil_div(D, 32)), dim3(256), 0, st, (const float*)partial, blocks, D, head_w, gamma, beta, dgamma,
                       dbeta, dxsum, dhead_w, dhead_b, (const float*)nullptr, (float*)nullptr);
  else if (nhg == 2)
    hipLaunchKernelGGL(ln_head_final_kernel<2>, dim3(ceil_div(D, 32)), dim3(256), 0, st, (const float*)partial, blocks, D, head_w, gamma, beta, dgamma,
                       dbeta, dxsum, dhead_w, dhead_b, (const float*)nullptr, (float*)nullptr);
  else {
    SlabOuts so = {{dgamma, dbeta, dxsum, nullptr, nullptr, nullptr}, -1, 0, nullptr, -1, 0};
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(ceil_div(D, SR_COLS), 3), dim3(256), 0, st, (const float*)partial, blocks, 3, D, so);
  }
  MTS_LAUNCH_CHECK("layernorm_bwd");
  return MTS_OK;
}

extern "C" int mts_layernorm_bwd(void* stream, int dtype, int rows, int D, const void* x, const void* dy, const float* dlogit,
                                 const float* head_w, int n_out, const float* gamma, const float* mean, const float* rstd, void* dx,
                                 float* dgamma, float* dbeta, float* dxsum, void* partial, const float* beta, float* dhead_w, float* dhead_b) {
  MTS_CHECK_ARG(rows > 0 && D > 0 && x && gamma && mean && rstd && dx && partial, "mts_layernorm_bwd: bad arguments");
  MTS_CHECK_ARG(dy || head_w, "mts_layernorm_bwd: needs dy and/or a fused head gradient");
  MTS_CHECK_ARG(!head_w || (dlogit && n_out >= 1 && n_out <= 4), "mts_layernorm_bwd: fused head needs dlogit, n_out<=4");
  MTS_CHECK_ARG(!dhead_w || (head_w && beta && dhead_b && n_out <= 2 && !dy),
                "mts_layernorm_bwd: head parameter gradients need head_w, beta, dhead_b, n_out <= 2 and dy == NULL (the last layer)");
  if (dtype == MTS_F32)
    return ln_bwd_launch<float>((hipStream_t)stream, rows, D, x, dy, dlogit, head_w, n_out, gamma, beta, mean, rstd, dx, dgamma, dbeta, dxsum, dhead_w, dhead_b, partial);
  if (dtype == MTS_BF16)
    return ln_bwd_launch<bf16_t>((hipStream_t)stream, rows, D, x, dy, dlogit, head_w, n_out, gamma, beta, mean, rstd, dx, dgamma, dbeta, dxsum, dhead_w, dhead_b, partial);
  mts_set_error("mts_layernorm_bwd: bad dtype %d", dtype);
  return MTS_ERR_INVALID;
}

// ---- the last layer's tail in one pass (ln_tail_kernel) ---------------------------------------------------------------------------------
extern "C" int mts_layernorm_loss_tail_supported(int dtype, int D, int n_out) {
  return (dtype == MTS_F32 || dtype == MTS_BF16) && (n_out == 1 || n_out == 2) && D % 256 == 0 &&
         (D == 256 || D == 512 || D == 1024 || D == 1792 || D == 2048);
}

template <typename T>
static int ln_tail_launch(hipStream_t st, int rows, int D, const void* x, const float* gamma, const float* beta, float eps, const float* head_w,
                          const float* head_b, int n_out, void* dx, float* dgamma, float* dbeta, float* dxsum, float* dhead_w, float* dhead_b,
                          float* ws, const TailArgs& ta0) {
  const int nv = D / 256;
  const int blocks = std::min(BWD_MAX_BLOCKS, ceil_div(rows, ROW_WAVES));
  TailArgs ta = ta0;
  ta.loss_part = ws + (size_t)BWD_MAX_BLOCKS * 5 * D;               // behind the slabs (at most 4 full-width slots per workgroup)
  dispatch_nv8(nv, [&](auto nvc) {
    constexpr int NV = decltype(nvc)::value;
    if (n_out == 1)
      hipLaunchKernelGGL((ln_tail_kernel<T, NV, 1>), dim3(blocks), dim3(64 * ROW_WAVES), 0, st, (const T*)x, gamma, beta, eps, head_w, head_b, rows, D, (T*)dx, ws, ta);
    else
      hipLaunchKernelGGL((ln_tail_kernel<T, NV, 2>), dim3(blocks), dim3(64 * ROW_WAVES), 0, st, (const T*)x, gamma, beta, eps, head_w, head_b, rows, D, (T*)dx, ws, ta);
  });
  const dim3 fgrid(ceil_div(D, 32));
  if (n_out == 1)
    hipLaunchKernelGGL(ln_head_final_kernel<1>, fgrid, dim3(256), 0, st, (const float*)ws, blocks, D, head_w, gamma, beta, dgamma, dbeta, dxsum, dhead_w, dhead_b,
                       (const float*)ta.loss_part, ta.loss_out);
  else
    hipLaunchKernelGGL(ln_head_final_kernel<2>, fgrid, dim3(256), 0, st, (const float*)ws, blocks, D, head_w, gamma, beta, dgamma, dbeta, dxsum, dhead_w, dhead_b,
                       (const float*)ta.loss_part, ta.loss_out);
  MTS_LAUNCH_CHECK("mts_layernorm_loss_tail");
  return MTS_OK;
}

extern "C" int mts_layernorm_loss_tail(void* stream, int dtype, int rows, int D, const void* x, const float* gamma, const float* beta, float eps,
                                       const float* head_w, const float* head_b, int n_out, int loss_kind, int B, int L, int Lt, const float* targets,
                                       const int32_t* lengths, float alpha, float gamma_focal, float grad_scale, const int32_t* row_src, int n_rows,
                                       float* scores, float* loss_out, void* dx, float* dgamma, float* dbeta, float* dxsum, float* dhead_w,
                                       float* dhead_b, void* workspace) {
  MTS_CHECK_ARG(rows > 0 && D > 0 && x && gamma && beta && head_w && head_b && targets && scores && loss_out && dx && dgamma && dbeta && dhead_w &&
                dhead_b && workspace, "mts_layernorm_loss_tail: bad arguments");
  MTS_CHECK_ARG(B > 0 && L > 0 && Lt >= L, "mts_layernorm_loss_tail: bad batch shape");
  MTS_CHECK_ARG(loss_kind == MTS_LOSS_CE || loss_kind == MTS_LOSS_BCE || loss_kind == MTS_LOSS_FOCAL, "Choose one of CrossEntropy or BinaryCrossEntropy as loss function");
  MTS_CHECK_ARG((loss_kind == MTS_LOSS_CE) ? n_out == 2 : n_out == 1, "mts_layernorm_loss_tail: n_out=%d does not match the loss kind", n_out);
  MTS_CHECK_ARG(row_src ? (n_rows == rows && rows <= B * L) : rows == B * L, "mts_layernorm_loss_tail: rows must be B*L, or n_rows of a packed batch");
  MTS_UNSUPPORTED(mts_layernorm_loss_tail_supported(dtype, D, n_out), "mts_layernorm_loss_tail: D=%d / n_out=%d not covered (see mts_layernorm_loss_tail_supported)", D, n_out);
  const TailArgs ta = {loss_kind, B, L, Lt, n_rows, targets, lengths, row_src, alpha, gamma_focal, grad_scale, scores, nullptr, loss_out};
  if (dtype == MTS_F32)
    return ln_tail_launch<float>((hipStream_t)stream, rows, D, x, gamma, beta, eps, head_w, head_b, n_out, dx, dgamma, dbeta, dxsum, dhead_w, dhead_b, (float*)workspace, ta);
  return ln_tail_launch<bf16_t>((hipStream_t)stream, rows, D, x, gamma, beta, eps, head_w, head_b, n_out, dx, dgamma, dbeta, dxsum, dhead_w, dhead_b, (float*)workspace, ta);
}

// ---- backward of the embedding block in one pass: LayerNorm backward + position / token-type gradient ------------------------------
static inline int emb_chunks(int B, int L) { return std::max(1, std::min(B, ceil_div(4 * BWD_MAX_BLOCKS, L))); }
extern "C" size_t mts_embed_layernorm_bwd_workspace(int B, int L, int D) {
  return ((size_t)BWD_MAX_BLOCKS * 2 + POS_SPLIT + (size_t)emb_chunks(B, L) * L) * (size_t)D * sizeof(float);
}

template <typename T>
static int emb_bwd_launch(hipStream_t st, int B, int L, int D, const void* pre, const void* dh, const float* gamma, const float* mean, const float* rstd,
                          float* dgamma, float* dbeta, float* dtype0, float* dpos, int pos_offset, const int32_t* row0, const int32_t* lengths,
                          int n_rows, float* ws) {
  const int nv = pick_nv(D);
  MTS_UNSUPPORTED(nv > 0 && nv <= 8 && D % 4 == 0, "embed_layernorm_bwd: D=%d must be a multiple of 4 and <= 2048", D);
  const int C = emb_chunks(B, L), Bc = ceil_div(B, C);
  const int nchunks = ceil_div(B, Bc);
  const int ntasks = L * nchunks;
  const int blocks = std::min(BWD_MAX_BLOCKS, ceil_div(ntasks, ROW_WAVES));
  float* slabs = ws;
  float* tpart = ws + (size_t)BWD_MAX_BLOCKS * 2 * D;               // [POS_SPLIT][D] partial token-type sums
  float* part = tpart + (size_t)POS_SPLIT * D;
  float* out_rows = dpos + (size_t)pos_offset * D;
  EmbArgs ea = {B, L, Bc, nchunks, row0, lengths, part};
  if (row0) {                        // ragged documents: a (position, chunk) no document reaches is not written by the kernel
    hipError_t e = hipMemsetAsync(part, 0, (size_t)nchunks * L * D * sizeof(float), st);
    if (e != hipSuccess) { mts_set_error("embed_layernorm_bwd: hipMemsetAsync: %s", hipGetErrorString(e)); return MTS_ERR_LAUNCH; }
  }
  dispatch_nv8(nv, [&](auto nvc) {
    constexpr int NV = decltype(nvc)::value;
    if (D == NV * 256)
      hipLaunchKernelGGL((ln_bwd_kernel<T, NV, false, true, 0, true>), dim3(blocks), dim3(64 * ROW_WAVES), 0, st, (const T*)pre, (const T*)dh,
                         (const float*)nullptr, (const float*)nullptr, 0, gamma, mean, rstd, n_rows, D, (T*)nullptr, slabs, ea);
    else
      hipLaunchKernelGGL((ln_bwd_kernel<T, NV, false, false, 0, true>), dim3(blocks), dim3(64 * ROW_WAVES), 0, st, (const T*)pre, (const T*)dh,
                         (const float*)nullptr, (const float*)nullptr, 0, gamma, mean, rstd, n_rows, D, (T*)nullptr, slabs, ea);
  });
  const int ysplit = std::max(1, std::min(POS_SPLIT, L / 32));
  hipLaunchKernelGGL(pos_sum_kernel, dim3(ceil_div(D, 32), ysplit), dim3(256), 0, st, (const float*)part, nchunks, L, D, out_rows, tpart);
  SlabOuts so = {{dgamma, dbeta, dtype0, nullptr, nullptr, nullptr}, -1, 0, tpart, 2, ysplit};
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(ceil_div(D, SR_COLS), 3), dim3(256), 0, st, (const float*)slabs, blocks, 2, D, so);
  MTS_LAUNCH_CHECK("embed_layernorm_bwd");
  return MTS_OK;
}

extern "C" int mts_embed_layernorm_bwd(void* stream, int dtype, int B, int L, int D, const void* pre, const void* dh, const float* gamma,
                                       const float* mean, const float* rstd, float* dgamma, float* dbeta, float* dtype0, float* dpos,
                                       int pos_offset, const int32_t* row0, const int32_t* lengths, int n_rows, void* workspace, size_t workspace_bytes) {
  MTS_CHECK_ARG(B > 0 && L > 0 && D > 0 && pre && dh && gamma && mean && rstd && dgamma && dbeta && dtype0 && dpos && workspace,
                "mts_embed_layernorm_bwd: bad arguments");
  MTS_CHECK_ARG(!row0 || (lengths && n_rows > 0 && n_rows <= B * L), "mts_embed_layernorm_bwd: packed form needs lengths and 0 < n_rows <= B*L");
  if (workspace_bytes < mts_embed_layernorm_bwd_workspace(B, L, D)) { mts_set_error("mts_embed_layernorm_bwd: workspace too small"); return MTS_ERR_WORKSPACE; }
  const int rows = row0 ? n_rows : B * L;
  if (dtype == MTS_F32)
    return emb_bwd_launch<float>((hipStream_t)stream, B, L, D, pre, dh, gamma, mean, rstd, dgamma, dbeta, dtype0, dpos, pos_offset, row0, lengths, rows, (float*)workspace);
  if (dtype == MTS_BF16)
    return emb_bwd_launch<bf16_t>((hipStream_t)stream, B, L, D, pre, dh, gamma, mean, rstd, dgamma, dbeta, dtype0, dpos, pos_offset, row0, lengths, rows, (float*)workspace);
  mts_set_error("mts_embed_layernorm_bwd: bad dtype %d", dtype);
  return MTS_ERR_INVALID;
}

extern "C" int mts_embed_bwd(void* stream, int dtype, int B, int L, int D, const void* dpre, float* dpos, int pos_offset, const int32_t* row0,
                             const int32_t* lengths) {
  MTS_CHECK_ARG(B > 0 && L > 0 && D > 0 && D % 4 == 0 && dpre && dpos, "mts_embed_bwd: bad arguments");
  MTS_CHECK_ARG(!row0 || lengths, "mts_embed_bwd: packed form needs lengths");
  hipStream_t st = (hipStream_t)stream;
  const int blocks = (int)(((size_t)L * (D / 4) + 255) / 256);
  if (dtype == MTS_F32) hipLaunchKernelGGL(embed_bwd_kernel<float>, dim3(blocks), dim3(256), 0, st, (const float*)dpre, B, L, D, dpos, pos_offset, row0, lengths);
  else if (dtype == MTS_BF16) hipLaunchKernelGGL(embed_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, (const bf16_t*)dpre, B, L, D, dpos, pos_offset, row0, lengths);
  else { mts_set_error("mts_embed_bwd: bad dtype %d", dtype); return MTS_ERR_INVALID; }
  MTS_LAUNCH_CHECK("mts_embed_bwd");
  return MTS_OK;
}

extern "C" int mts_head_fwd(void* stream, int dtype, int rows, int D, int n_out, const void* x, int ldx, const float* w, const float* b,
                            float* scores) {
  MTS_CHECK_ARG(rows > 0 && D > 0 && D % 4 == 0 && ldx % 4 == 0 && n_out >= 1 && n_out <= 4 && x && w && b && scores, "mts_head_fwd: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(ceil_div(rows, ROW_WAVES)), block(64 * ROW_WAVES);
  if (dtype == MTS_F32) hipLaunchKernelGGL(head_fwd_kernel<float>, grid, block, 0, st, (const float*)x, ldx, rows, D, n_out, w, b, scores);
  else if (dtype == MTS_BF16) hipLaunchKernelGGL(head_fwd_kernel<bf16_t>, grid, block, 0, st, (const bf16_t*)x, ldx, rows, D, n_out, w, b, scores);
  else { mts_set_error("mts_head_fwd: bad dtype %d", dtype); return MTS_ERR_INVALID; }
  MTS_LAUNCH_CHECK("mts_head_fwd");
  return MTS_OK;
}

extern "C" int mts_head_bwd_params(void* stream, int dtype, int rows, int D, int n_out, const void* x, int ldx, const float* dscores,
                                   float* dw, float* db, void* partial) {
  MTS_CHECK_ARG(rows > 0 && D > 0 && D % 4 == 0 && ldx % 4 == 0 && n_out >= 1 && n_out <= 4 && x && dscores && dw && db && partial,
                "mts_head_bwd_params: bad arguments");
  const int nv = pick_nv(D);
  MTS_UNSUPPORTED(nv > 0, "mts_head_bwd_params: D=%d must be <= 4096", D);
  hipStream_t st = (hipStream_t)stream;
  const int blocks = std::min(BWD_MAX_BLOCKS, ceil_div(rows, ROW_WAVES));
  const int chunks = nv <= 8 ? 1 : ceil_div(D, 2048);             // 2048 < D <= 4096: two column chunks
  if (dtype == MTS_F32) {
    dispatch_nv8(nv, [&](auto nvc) {
      constexpr int NV = decltype(nvc)::value;
      if (D == NV * 256)
        hipLaunchKernelGGL((head_bwd_params_kernel<float, NV, true>), dim3(blocks, chunks), dim3(64 * ROW_WAVES), 0, st, (const float*)x, ldx, dscores,
                           n_out, rows, D, (float*)partial);
      else
        hipLaunchKernelGGL((head_bwd_params_kernel<float, NV>), dim3(blocks, chunks), dim3(64 * ROW_WAVES), 0, st, (const float*)x, ldx, dscores, n_out,
                           rows, D, (float*)partial);
    });
  } else if (dtype == MTS_BF16) {
    dispatch_nv8(nv, [&](auto nvc) {
      constexpr int NV = decltype(nvc)::value;
      if (D == NV * 256)
        hipLaunchKernelGGL((head_bwd_params_kernel<bf16_t, NV, true>), dim3(blocks, chunks), dim3(64 * ROW_WAVES), 0, st, (const bf16_t*)x, ldx, dscores,
                           n_out, rows, D, (float*)partial);
      else
        hipLaunchKernelGGL((head_bwd_params_kernel<bf16_t, NV>), dim3(blocks, chunks), dim3(64 * ROW_WAVES), 0, st, (const bf16_t*)x, ldx, dscores, n_out,
                           rows, D, (float*)partial);
    });
  } else { mts_set_error("mts_head_bwd_params: bad dtype %d", dtype); return MTS_ERR_INVALID; }
  float* rowp[4] = {nullptr, nullptr, nullptr, nullptr};
  for (int c = 0; c < n_out; ++c) rowp[c] = dw + (size_t)c * D;
  SlabOuts so = {{rowp[0], rowp[1], rowp[2], rowp[3], db, nullptr}, 4, n_out, nullptr, -1, 0};
  hipLaunchKernelGGL(slab_reduce_kernel, dim3(ceil_div(D, SR_COLS), 5), dim3(256), 0, st, (const float*)partial, blocks, 5, D, so);
  MTS_LAUNCH_CHECK("mts_head_bwd_params");
  return MTS_OK;
}

extern "C" int mts_head_bwd_data(void* stream, int dtype, int rows, int D, int n_out, const float* dscores, const float* w, void* dx, int lddx,
                                 int accumulate) {
  MTS_CHECK_ARG(rows > 0 && D > 0 && D % 4 == 0 && lddx % 4 == 0 && n_out >= 1 && n_out <= 4 && dscores && w && dx, "mts_head_bwd_data: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  const int blocks = (int)(((size_t)rows * (D / 4) + 255) / 256);
  if (dtype == MTS_F32) hipLaunchKernelGGL(head_bwd_data_kernel<float>, dim3(blocks), dim3(256), 0, st, dscores, w, rows, D, n_out, (float*)dx, lddx, accumulate);
  else if (dtype == MTS_BF16) hipLaunchKernelGGL(head_bwd_data_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, dscores, w, rows, D, n_out, (bf16_t*)dx, lddx, accumulate);
  else { mts_set_error("mts_head_bwd_data: bad dtype %d", dtype); return MTS_ERR_INVALID; }
  MTS_LAUNCH_CHECK("mts_head_bwd_data");
  return MTS_OK;
}

// dx = dy * gelu'(u), in place on dy (FFN backward; modeling_longformer.py:1113-1116 uses erf-GELU)
template <typename T>
__global__ __launch_bounds__(256) void gelu_bwd_kernel(size_t n, const T* __restrict__ u, T* __restrict__ dy) {
  const size_t stride = (size_t)gridDim.x * 256 * 4;
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i + 3 < n; i += stride) {
    float uv[4], dv[4];
    load4<T>(u + i, uv); load4<T>(dy + i, dv);
#pragma unroll
    for (int j = 0; j < 4; ++j) dv[j] *= gelu_erf_grad_f(uv[j]);
    store4<T>(dy + i, dv);
  }
}
extern "C" int mts_gelu_bwd(void* stream, int dtype, size_t n, const void* u, void* dy) {
  MTS_CHECK_ARG(u && dy && n % 4 == 0, "mts_gelu_bwd: bad arguments (n must be a multiple of 4)");
  if (n == 0) return MTS_OK;
  const int blocks = (int)std::min<size_t>(2048, (n / 4 + 255) / 256);
  if (dtype == MTS_F32) hipLaunchKernelGGL(gelu_bwd_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, (const float*)u, (float*)dy);
  else if (dtype == MTS_BF16) hipLaunchKernelGGL(gelu_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, (const bf16_t*)u, (bf16_t*)dy);
  else { mts_set_error("mts_gelu_bwd: bad dtype %d", dtype); return MTS_ERR_INVALID; }
  MTS_LAUNCH_CHECK("mts_gelu_bwd");
  return MTS_OK;
}

// dy *= (u > 0), in place on dy (backward of the legacy layer's ReLU FFN, models/RestrictedTransformerLayer.py:308)
template <typename T>
__global__ __launch_bounds__(256) void relu_bwd_kernel(size_t n, const T* __restrict__ u, T* __restrict__ dy) {
  const size_t stride = (size_t)gridDim.x * 256 * 4;
  for (size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4; i + 3 < n; i += stride) {
    float uv[4], dv[4];
    load4<T>(u + i, uv); load4<T>(dy + i, dv);
#pragma unroll
    for (int j = 0; j < 4; ++j) dv[j] = uv[j] > 0.0f ? dv[j] : 0.0f;
    store4<T>(dy + i, dv);
  }
}
extern "C" int mts_relu_bwd(void* stream, int dtype, size_t n, const void* u, void* dy) {
  MTS_CHECK_ARG(u && dy && n % 4 == 0, "mts_relu_bwd: bad arguments (n must be a multiple of 4)");
  if (n == 0) return MTS_OK;
  const int blocks = (int)std::min<size_t>(2048, (n / 4 + 255) / 256);
  if (dtype == MTS_F32) hipLaunchKernelGGL(relu_bwd_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, (const float*)u, (float*)dy);
  else if (dtype == MTS_BF16) hipLaunchKernelGGL(relu_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, n, (const bf16_t*)u, (bf16_t*)dy);
  else { mts_set_error("mts_relu_bwd: bad dtype %d", dtype); return MTS_ERR_INVALID; }
  MTS_LAUNCH_CHECK("mts_relu_bwd");
  return MTS_OK;
}
