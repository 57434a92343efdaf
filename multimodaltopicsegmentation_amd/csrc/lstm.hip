// LSTM recurrence with packed-sequence semantics (forward and backward through time).
//
// The input projection x W_ih^T + b_ih + b_hh for every position and both directions is one MFMA GEMM done
// by the caller (mts_gemm); what is left is the dependent chain h_t = f(xproj_t + h_{t-1} W_hh^T): DG
// documents x one direction per workgroup, one hidden unit per thread, the cell state in registers for the
// whole sequence, h exchanged through a double-buffered LDS tile (one barrier per time step), W_hh read
// through L2 in the orientation that makes the 64 lanes of a wave read 256 contiguous bytes
// (forward: W_hh^T [H][4H], backward: W_hh [4H][H]).  The weight gradient is NOT accumulated in the
// time loop: dW_hh = dA^T . H_prev is one TN GEMM over all positions after the loop.
//
// Packed semantics (models/NeuralArchitectures.py:98-115): a document of length n runs n steps; the reverse
// direction starts at its own last sentence; rows >= n of `out` are exactly 0.
#include <algorithm>
#include "common.h"

#define LSTM_DG 8   // documents per workgroup

template <typename T>
__global__ void lstm_fwd_kernel(int B, int L, int H, int ndir, const T* __restrict__ xproj, const float* __restrict__ whhT /*[ndir][H][4H]*/,
                                const float* __restrict__ bhh /*[ndir][4H] or null*/, const int32_t* __restrict__ lengths, T* __restrict__ out, T* __restrict__ gates, float* __restrict__ cells) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* hbuf = reinterpret_cast<float*>(smem);          // [2][DG][H]
  const int d = blockIdx.y;
  const int b0 = blockIdx.x * LSTM_DG;
  const int j = threadIdx.x;
  const bool unit = j < H;
  const float* W = whhT + (size_t)d * H * 4 * H;
  const int ldx = ndir * 4 * H, ldo = ndir * H;

  int len[LSTM_DG];
  int maxlen = 0;
#pragma unroll
  for (int q = 0; q < LSTM_DG; ++q) {
    const int b = b0 + q;
    len[q] = (b < B) ? (lengths ? min(lengths[b], L) : L) : 0;
    maxlen = max(maxlen, len[q]);
  }
  float c[LSTM_DG];
#pragma unroll
  for (int q = 0; q < LSTM_DG; ++q) c[q] = 0.f;
  float bh[4] = {0.f, 0.f, 0.f, 0.f};
  if (bhh && unit) {
#pragma unroll
    for (int gi_ = 0; gi_ < 4; ++gi_) bh[gi_] = bhh[(size_t)d * 4 * H + gi_ * H + j];
  }
  for (int idx = threadIdx.x; idx < 2 * LSTM_DG * H; idx += blockDim.x) hbuf[idx] = 0.f;
  __syncthreads();

  for (int s = 0; s < maxlen; ++s) {
    const float* hcur = hbuf + (s & 1) * LSTM_DG * H;
    float* hnext = hbuf + ((s + 1) & 1) * LSTM_DG * H;
    float acc[LSTM_DG][4];
#pragma unroll
    for (int q = 0; q < LSTM_DG; ++q) acc[q][0] = acc[q][1] = acc[q][2] = acc[q][3] = 0.f;
    if (unit) {
      for (int k = 0; k < H; ++k) {
        const float* wr = W + (size_t)k * 4 * H + j;
        const float w0 = wr[0], w1 = wr[H], w2 = wr[2 * H], w3 = wr[3 * H];
#pragma unroll
        for (int q = 0; q < LSTM_DG; ++q) {
          const float hv = hcur[q * H + k];
          acc[q][0] = fmaf(hv, w0, acc[q][0]);
          acc[q][1] = fmaf(hv, w1, acc[q][1]);
          acc[q][2] = fmaf(hv, w2, acc[q][2]);
          acc[q][3] = fmaf(hv, w3, acc[q][3]);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < LSTM_DG; ++q) {
      float hn = hcur[q * H + (unit ? j : 0)];
      if (unit && s < len[q]) {
        const int t = (d == 0) ? s : (len[q] - 1 - s);
        const size_t row = (size_t)(b0 + q) * L + t;
        const T* xp = xproj + row * ldx + (size_t)d * 4 * H + j;
        const float gi = sigmoid_f((to_f32(xp[0]) + bh[0]) + acc[q][0]);
        const float gf = sigmoid_f((to_f32(xp[H]) + bh[1]) + acc[q][1]);
        const float gg = tanhf((to_f32(xp[2 * H]) + bh[2]) + acc[q][2]);
        const float go = sigmoid_f((to_f32(xp[3 * H]) + bh[3]) + acc[q][3]);
        c[q] = gf * c[q] + gi * gg;
        hn = go * tanhf(c[q]);
        T* gp = gates + row * ldx + (size_t)d * 4 * H + j;
        gp[0] = from_f32<T>(gi); gp[H] = from_f32<T>(gf); gp[2 * H] = from_f32<T>(gg); gp[3 * H] = from_f32<T>(go);
        cells[row * ldo + (size_t)d * H + j] = c[q];
        const T hq = from_f32<T>(hn);
        out[row * ldo + (size_t)d * H + j] = hq;
        hn = to_f32(hq);            // the next step (and the next layer) see the stored value
      }
      if (unit) hnext[q * H + j] = hn;
    }
    __syncthreads();
  }
  // padded rows are exactly zero
  for (int q = 0; q < LSTM_DG; ++q) {
    const int b = b0 + q;
    if (b >= B) break;
    for (int idx = threadIdx.x; idx < (L - len[q]) * H; idx += blockDim.x) {
      const int t = len[q] + idx / H, jj = idx % H;
      out[((size_t)b * L + t) * ldo + (size_t)d * H + jj] = from_f32<T>(0.f);
    }
  }
}

// backward through time: writes dxproj (pre-activation gate gradients) and hprev (the h each step consumed)
template <typename T>
__global__ void lstm_bwd_kernel(int B, int L, int H, int ndir, const float* __restrict__ whh /*[ndir][4H][H]*/,
                                const int32_t* __restrict__ lengths, const T* __restrict__ out, const T* __restrict__ gates,
                                const float* __restrict__ cells, const T* __restrict__ dout, T* __restrict__ dxproj, T* __restrict__ hprev) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* da = reinterpret_cast<float*>(smem);            // [DG][4H] pre-activation gradients of this step
  const int d = blockIdx.y;
  const int b0 = blockIdx.x * LSTM_DG;
  const int j = threadIdx.x;
  const bool unit = j < H;
  const float* W = whh + (size_t)d * 4 * H * H;
  const int ldx = ndir * 4 * H, ldo = ndir * H;

  int len[LSTM_DG];
  int maxlen = 0;
#pragma unroll
  for (int q = 0; q < LSTM_DG; ++q) {
    const int b = b0 + q;
    len[q] = (b < B) ? (lengths ? min(lengths[b], L) : L) : 0;
    maxlen = max(maxlen, len[q]);
  }
  float dh[LSTM_DG], dc[LSTM_DG];
#pragma unroll
  for (int q = 0; q < LSTM_DG; ++q) dh[q] = dc[q] = 0.f;

  // walk the steps in reverse: step s of the forward pass processed position t(s)
  for (int s = maxlen - 1; s >= 0; --s) {
#pragma unroll
    for (int q = 0; q < LSTM_DG; ++q) {
      float ai = 0.f, af = 0.f, ag = 0.f, ao = 0.f;
      if (unit && s < len[q]) {
        const int t = (d == 0) ? s : (len[q] - 1 - s);
        const int tp = (d == 0) ? t - 1 : t + 1;          // position processed one step earlier
        const bool has_prev = (s > 0);
        const size_t row = (size_t)(b0 + q) * L + t;
        const size_t prow = (size_t)(b0 + q) * L + tp;
        const T* gp = gates + row * ldx + (size_t)d * 4 * H + j;
        const float gi = to_f32(gp[0]), gf = to_f32(gp[H]), gg = to_f32(gp[2 * H]), go = to_f32(gp[3 * H]);
        const float ct = cells[row * ldo + (size_t)d * H + j];
        const float cp = has_prev ? cells[prow * ldo + (size_t)d * H + j] : 0.f;
        const float tc = tanhf(ct);
        const float dht = to_f32(dout[row * ldo + (size_t)d * H + j]) + dh[q];
        const float dct = dc[q] + dht * go * (1.f - tc * tc);
        ai = dct * gg * gi * (1.f - gi);
        af = dct * cp * gf * (1.f - gf);
        ag = dct * gi * (1.f - gg * gg);
        ao = dht * tc * go * (1.f - go);
        dc[q] = dct * gf;
        T* dx = dxproj + row * ldx + (size_t)d * 4 * H + j;
        const T qi = from_f32<T>(ai), qf = from_f32<T>(af), qg = from_f32<T>(ag), qo = from_f32<T>(ao);
        dx[0] = qi; dx[H] = qf; dx[2 * H] = qg; dx[3 * H] = qo;
        ai = to_f32(qi); af = to_f32(qf); ag = to_f32(qg); ao = to_f32(qo);   // propagate what the GEMMs will see
        hprev[row * ldo + (size_t)d * H + j] = has_prev ? out[prow * ldo + (size_t)d * H + j] : from_f32<T>(0.f);
      }
      if (unit) {
        float* dq = da + q * 4 * H;
        dq[j] = ai; dq[H + j] = af; dq[2 * H + j] = ag; dq[3 * H + j] = ao;
      }
    }
    __syncthreads();
    // dh_{prev}[k] = sum_n da[n] W[n][k]   (thread k; inactive documents have da = 0 so their dh stays untouched below)
    if (unit) {
      float acc[LSTM_DG];
#pragma unroll
      for (int q = 0; q < LSTM_DG; ++q) acc[q] = 0.f;
      for (int n = 0; n < 4 * H; ++n) {
        const float w = W[(size_t)n * H + j];
#pragma unroll
        for (int q = 0; q < LSTM_DG; ++q) acc[q] = fmaf(da[q * 4 * H + n], w, acc[q]);
      }
#pragma unroll
      for (int q = 0; q < LSTM_DG; ++q)
        if (s < len[q]) dh[q] = acc[q];
    }
    __syncthreads();
  }
  // rows >= len: zero gradients (the GEMMs that follow read every row)
  for (int q = 0; q < LSTM_DG; ++q) {
    const int b = b0 + q;
    if (b >= B) break;
    for (int idx = threadIdx.x; idx < (L - len[q]) * 4 * H; idx += blockDim.x) {
      const int t = len[q] + idx / (4 * H), n = idx % (4 * H);
      dxproj[((size_t)b * L + t) * ldx + (size_t)d * 4 * H + n] = from_f32<T>(0.f);
    }
    for (int idx = threadIdx.x; idx < (L - len[q]) * H; idx += blockDim.x) {
      const int t = len[q] + idx / H, jj = idx % H;
      hprev[((size_t)b * L + t) * ldo + (size_t)d * H + jj] = from_f32<T>(0.f);
    }
  }
}

__global__ void transpose_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols) {
  __shared__ float tile[32][33];
  const float* s = src + (size_t)blockIdx.z * rows * cols;
  float* dd = dst + (size_t)blockIdx.z * rows * cols;
  int x = blockIdx.x * 32 + threadIdx.x, y = blockIdx.y * 32 + threadIdx.y;
  for (int i = 0; i < 32; i += 8)
    if (x < cols && y + i < rows) tile[threadIdx.y + i][threadIdx.x] = s[(size_t)(y + i) * cols + x];
  __syncthreads();
  x = blockIdx.y * 32 + threadIdx.x; y = blockIdx.x * 32 + threadIdx.y;
  for (int i = 0; i < 32; i += 8)
    if (x < rows && y + i < cols) dd[(size_t)(y + i) * rows + x] = tile[threadIdx.x][threadIdx.y + i];
}

// MFMA fast path (lstm_mfma.hip): bf16, H = 256
bool mts_lstm_mfma_supported(int dtype, int H);
size_t mts_lstm_mfma_workspace(int H, int ndir);
int mts_lstm_mfma_fwd(hipStream_t st, int B, int L, int H, int ndir, const void* xproj, const float* w_hh, const float* b_hh,
                      const int32_t* lengths, void* out, void* gates, float* cells, void* ws);
int mts_lstm_mfma_bwd(hipStream_t st, int B, int L, int H, int ndir, const float* w_hh, const int32_t* lengths, const void* out, const void* gates,
                      const float* cells, const void* dout, void* dxproj, void* hprev, void* ws);
// CU-pair form (lstm_pair.hip): weights fully register-resident, forward only so far
bool mts_lstm_pair_supported(int dtype, int H);
size_t mts_lstm_pair_workspace(int B, int H, int ndir);
int mts_lstm_pair_fwd(hipStream_t st, int B, int L, int H, int ndir, const void* xproj, const float* w_hh, const float* b_hh,
                      const int32_t* lengths, void* out, void* gates, float* cells, void* ws);
int mts_lstm_pair_bwd(hipStream_t st, int B, int L, int H, int ndir, const float* w_hh, const int32_t* lengths, const void* out, const void* gates,
                      const float* cells, const void* dout, void* dxproj, void* hprev, void* ws);
// CU-quad recurrences in fp32 (lstm_pair.hip), H = 256: parity mode
int mts_lstm_quad_f32_fwd(hipStream_t st, int B, int L, int H, int ndir, const void* xproj, const float* w_hh, const float* b_hh,
                          const int32_t* lengths, void* out, void* gates, float* cells, void* ws);
int mts_lstm_pair_hprev(hipStream_t st, int dtype, int B, int L, int H, int ndir, const int32_t* lengths, const void* out, void* hprev);
int mts_lstm_quad_f32_bwd(hipStream_t st, int B, int L, int H, int ndir, const float* w_hh, const int32_t* lengths, const void* out, const void* gates,
                          const float* cells, const void* dout, void* dxproj, void* hprev, void* ws);
unsigned mts_lstm_pair_take_error();                  // lstm_pair.hip: sticky timeout word (pinned host memory), reading clears
static thread_local int g_lstm_mfma = 1;
void mts_lstm_pair_force_parts(int n);      // lstm_pair.hip
int mts_lstm_pair_parts();
// Which recurrence form wrote a saved state, keyed by its `out` buffer.  The form fixes the layout of the opaque gates / cells and who produces
// h_{t-1}; it follows from options that are per host thread ("lstm_mfma", "lstm_parts"), and torch's autograd runs backward nodes on a thread of
// its own: the backward FOLLOWS what the forward recorded instead of re-reading its own thread's options.  0: the generic / MFMA
// kernels ([B*L, ndir*4H] gates), 2: CU pair, 4: CU quad (step-major blocks).  A few hundred entries at most (one per live `out` buffer).
#include <mutex>
#include <unordered_map>
static std::mutex g_form_mu;
static std::unordered_map<const void*, int> g_form;
static void lstm_form_note(const void* out, int form) {
  std::lock_guard<std::mutex> lk(g_form_mu);
  if (g_form.size() > 4096) g_form.clear();
  g_form[out] = form;
}
static int lstm_form_of(const void* out) {
  std::lock_guard<std::mutex> lk(g_form_mu);
  auto it = g_form.find(out);
  return it == g_form.end() ? -1 : it->second;
}

// Asynchronous device-side errors, reported without synchronising: today the only source is a CU-pair LSTM launch whose partner
// poll gave up.  Every mts_lstm_* entry calls this first, so the error surfaces at the next step at the latest.
static int lstm_report_async(const char* where) {
  const unsigned v = mts_lstm_pair_take_error();
  if (!v) return MTS_OK;
  mts_set_error("%s: an earlier CU-pair LSTM launch (%s%s) timed out waiting for its partner workgroup; the results of that "
                "launch are invalid", where, (v & 1u) ? "forward" : "", (v & 2u) ? ((v & 1u) ? "+backward" : "backward") : "");
  return MTS_ERR_TIMEOUT;
}
extern "C" int mts_async_status(void) { return lstm_report_async("mts_async_status"); }

extern "C" void mts_lstm_set_mfma(int on) { g_lstm_mfma = on; }

static int lstm_threads(int H) { return ((H + 63) / 64) * 64; }

// workspace: fwd needs W_hh^T (fp32 [ndir,H,4H]); bwd needs hprev (act dtype [B*L, ndir*H])
// first part of the workspace: scratch of whichever recurrence kernel runs (transposed / packed W_hh, the CU-pair exchange buffer);
// once the recurrence has finished it is the split-K slab area of the dW_hh GEMMs (4H x H outputs, K = all tokens: 16 tiles at
// H = 256 -- unsplit they would run on 16 of the 256 CUs), hence room for 16 partial planes.
static size_t lstm_scratch_bytes(int B, int H, int ndir) {
  size_t a = std::max(align_up((size_t)ndir * 4 * H * H * sizeof(float), 256), mts_lstm_mfma_workspace(H, ndir));
  a = std::max(a, align_up(mts_lstm_pair_workspace(B, H, ndir), 256));
  return std::max(a, align_up((size_t)16 * 4 * H * H * sizeof(float), 256));
}

extern "C" size_t mts_lstm_workspace(int dtype, int B, int L, int H, int ndir) {
  const size_t esz = dtype == MTS_F32 ? 4 : 2;
  const size_t a = lstm_scratch_bytes(B, H, ndir);
  const size_t b = align_up((size_t)B * L * ndir * H * esz, 256);
  return a + b;
}

extern "C" int mts_lstm_fwd(void* stream, int dtype, int B, int L, int H, int ndir, const void* xproj, const float* w_hh,
                            const float* b_hh, const int32_t* lengths, void* out, void* gates, float* cells, void* workspace) {
  MTS_CHECK_ARG(B > 0 && L > 0 && H > 0 && (ndir == 1 || ndir == 2), "mts_lstm_fwd: bad shape");
  MTS_CHECK_ARG(xproj && w_hh && out && gates && cells && workspace, "mts_lstm_fwd: null pointer");
  MTS_CHECK_ARG(dtype == MTS_F32 || dtype == MTS_BF16, "mts_lstm_fwd: bad dtype %d", dtype);
  MTS_UNSUPPORTED(H <= 1024, "mts_lstm_fwd: hidden size %d > 1024", H);
  if (int rc = lstm_report_async("mts_lstm_fwd")) return rc;
  hipStream_t st = (hipStream_t)stream;
  if (g_lstm_mfma && mts_lstm_pair_supported(dtype, H)) {
    lstm_form_note(out, mts_lstm_pair_parts());
    return dtype == MTS_F32 ? mts_lstm_quad_f32_fwd(st, B, L, H, ndir, xproj, w_hh, b_hh, lengths, out, gates, cells, workspace)
                            : mts_lstm_pair_fwd(st, B, L, H, ndir, xproj, w_hh, b_hh, lengths, out, gates, cells, workspace);
  }
  lstm_form_note(out, 0);
  if (g_lstm_mfma && mts_lstm_mfma_supported(dtype, H))
    return mts_lstm_mfma_fwd(st, B, L, H, ndir, xproj, w_hh, b_hh, lengths, out, gates, cells, workspace);
  float* whhT = (float*)workspace;
  hipLaunchKernelGGL(transpose_f32_kernel, dim3(ceil_div(H, 32), ceil_div(4 * H, 32), ndir), dim3(32, 8), 0, st, w_hh, whhT, 4 * H, H);
  const size_t lds = (size_t)2 * LSTM_DG * H * sizeof(float);
  dim3 grid(ceil_div(B, LSTM_DG), ndir), block(lstm_threads(H));
  if (dtype == MTS_F32)
    hipLaunchKernelGGL(lstm_fwd_kernel<float>, grid, block, lds, st, B, L, H, ndir, (const float*)xproj, whhT, b_hh, lengths, (float*)out, (float*)gates, cells);
  else
    hipLaunchKernelGGL(lstm_fwd_kernel<bf16_t>, grid, block, lds, st, B, L, H, ndir, (const bf16_t*)xproj, whhT, b_hh, lengths, (bf16_t*)out, (bf16_t*)gates, cells);
  MTS_LAUNCH_CHECK("mts_lstm_fwd");
  return MTS_OK;
}

// part 1: the recurrence (dxproj; + h_{t-1} where the kernel produces it itself), 2: dW_hh (+ h_{t-1} for the CU-pair / CU-quad paths), 3: both
static int lstm_bwd_impl(int part, void* stream, int dtype, int B, int L, int H, int ndir, const float* w_hh, const int32_t* lengths, const void* out,
                         const void* gates, const float* cells, const void* dout, void* dxproj, float* dw_hh, void* workspace) {
  const char* who = part == 1 ? "mts_lstm_bwd_recurrence" : part == 2 ? "mts_lstm_bwd_whh" : "mts_lstm_bwd";
  MTS_CHECK_ARG(B > 0 && L > 0 && H > 0 && (ndir == 1 || ndir == 2), "%s: bad shape", who);
  MTS_CHECK_ARG(out && dxproj && workspace, "%s: null pointer", who);
  MTS_CHECK_ARG(!(part & 1) || (w_hh && gates && cells && dout), "%s: null pointer", who);
  MTS_CHECK_ARG(!(part & 2) || dw_hh, "%s: null pointer", who);
  MTS_CHECK_ARG(dtype == MTS_F32 || dtype == MTS_BF16, "%s: bad dtype %d", who, dtype);
  MTS_UNSUPPORTED(H <= 1024, "%s: hidden size %d > 1024", who, H);
  if (int rc = lstm_report_async(who)) return rc;
  hipStream_t st = (hipStream_t)stream;
  const size_t hoff = lstm_scratch_bytes(B, H, ndir);
  char* hprev = (char*)workspace + hoff;
  // the form the forward recorded for this saved state wins over the calling thread's options (see lstm_form_note)
  const int form = lstm_form_of(out);
  struct ForceParts { ForceParts(int f) { mts_lstm_pair_force_parts(f); } ~ForceParts() { mts_lstm_pair_force_parts(0); } } force_parts(form > 0 ? form : 0);
  const bool pairq = form >= 0 ? form > 0 : (g_lstm_mfma && mts_lstm_pair_supported(dtype, H));
  const bool fast = pairq || (g_lstm_mfma && mts_lstm_mfma_supported(dtype, H));
  if (part & 1) {
    if (fast) {
      // the CU-pair / CU-quad recurrences take h_{t-1} from a kernel of its own: with the recurrence when both parts run here, otherwise with part 2
      void* hp = (part & 2) ? (void*)hprev : nullptr;
      int rc = pairq ? (dtype == MTS_F32 ? mts_lstm_quad_f32_bwd(st, B, L, H, ndir, w_hh, lengths, out, gates, cells, dout, dxproj, hp, workspace)
                                         : mts_lstm_pair_bwd(st, B, L, H, ndir, w_hh, lengths, out, gates, cells, dout, dxproj, hp, workspace))
                     : mts_lstm_mfma_bwd(st, B, L, H, ndir, w_hh, lengths, out, gates, cells, dout, dxproj, hprev, workspace);
      if (rc) return rc;
    } else {
      const size_t lds = (size_t)LSTM_DG * 4 * H * sizeof(float);
      MTS_UNSUPPORTED(lds <= 160 * 1024, "%s: hidden size %d needs too much LDS", who, H);
      dim3 grid(ceil_div(B, LSTM_DG), ndir), block(lstm_threads(H));
      if (dtype == MTS_F32) {
        auto k = lstm_bwd_kernel<float>;
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k, grid, block, lds, st, B, L, H, ndir, w_hh, lengths, (const float*)out, (const float*)gates, cells, (const float*)dout,
                           (float*)dxproj, (float*)hprev);
      } else {
        auto k = lstm_bwd_kernel<bf16_t>;
        if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k, grid, block, lds, st, B, L, H, ndir, w_hh, lengths, (const bf16_t*)out, (const bf16_t*)gates, cells, (const bf16_t*)dout,
                           (bf16_t*)dxproj, (bf16_t*)hprev);
      }
    }
    MTS_LAUNCH_CHECK(who);
  }
  if (!(part & 2)) return MTS_OK;
  if (part == 2 && pairq) {
    if (int rc = mts_lstm_pair_hprev(st, dtype, B, L, H, ndir, lengths, out, hprev)) return rc;
  }
  // dW_hh[d] = dA_d^T [4H, B*L] . Hprev_d [B*L, H]
  const size_t esz = dtype == MTS_F32 ? 4 : 2;
  const bool mfma_ok = (dtype == MTS_F32) || ((4 * H) % 8 == 0 && H % 8 == 0 && (ndir * 4 * H) % 8 == 0 && (ndir * H) % 8 == 0);
  MTS_UNSUPPORTED(mfma_ok, "%s(bf16): hidden size %d must be a multiple of 8", who, H);
  for (int d = 0; d < ndir; ++d) {
    int rc = mts_gemm(stream, dtype, MTS_F32, MTS_TN, 4 * H, H, B * L, (const char*)dxproj + (size_t)d * 4 * H * esz, ndir * 4 * H,
                      hprev + (size_t)d * H * esz, ndir * H, dw_hh + (size_t)d * 4 * H * H, H, nullptr, nullptr, 0, nullptr, 0, 0u, 1.f, 0,
                      workspace, hoff);                  // the recurrence kernels are done with their scratch: split-K slabs
    if (rc) return rc;
  }
  return MTS_OK;
}

extern "C" int mts_lstm_bwd(void* stream, int dtype, int B, int L, int H, int ndir, const float* w_hh, const int32_t* lengths, const void* out,
                            const void* gates, const float* cells, const void* dout, void* dxproj, float* dw_hh, void* workspace) {
  return lstm_bwd_impl(3, stream, dtype, B, L, H, ndir, w_hh, lengths, out, gates, cells, dout, dxproj, dw_hh, workspace);
}
// The two halves of mts_lstm_bwd as calls of their own, so that a caller can put the weight-gradient half on another stream: the recurrent
// taggers' next dependent launch (the data gradient of the layer below) then starts right behind the recurrence instead of behind
// h_{t-1} + two split-K GEMMs + their reduces (~80 us per layer at 64 x 256, H = 256).  `workspace`: the SAME buffer for both calls of a layer, not
// touched by anything else in between (mts_lstm_bwd_whh finds h_{t-1} there or builds it there, and uses the recurrence's scratch for its slabs).
extern "C" int mts_lstm_bwd_recurrence(void* stream, int dtype, int B, int L, int H, int ndir, const float* w_hh, const int32_t* lengths, const void* out,
                                       const void* gates, const float* cells, const void* dout, void* dxproj, void* workspace) {
  return lstm_bwd_impl(1, stream, dtype, B, L, H, ndir, w_hh, lengths, out, gates, cells, dout, dxproj, nullptr, workspace);
}
extern "C" int mts_lstm_bwd_whh(void* stream, int dtype, int B, int L, int H, int ndir, const int32_t* lengths, const void* out, const void* dxproj,
                                float* dw_hh, void* workspace) {
  return lstm_bwd_impl(2, stream, dtype, B, L, H, ndir, nullptr, lengths, out, nullptr, nullptr, nullptr, const_cast<void*>(dxproj), dw_hh, workspace);
}
