// LSTM recurrence, "CU pair" form (bf16, H = 256): W_hh fully resident in registers.
//
// Measured on MI355X (tools/lstm_bench.py, MTS_LSTM_EXP ablations): with one workgroup per (16 documents, direction)
// the recurrent matrix (512 KiB bf16) does not fit a CU and re-streaming even half of it from L2 costs ~5.7 us of a
// 10.6 us step (a CU pulls only ~45 GB/s through fragment-shaped loads).  So the hidden units of one direction are
// split over TWO workgroups (half p = 0/1, 128 units each, 8 waves = two per SIMD so that one wave's gate math
// overlaps the other's MFMAs): wave w keeps the 4 gate tiles x 8 k-steps of its 16 units (128 VGPRs) for the whole
// sequence and there is NO weight traffic in the time loop.  What the pair exchanges per step is the new h of its
// half (16 docs x 128 units bf16 = 4 KiB) as 8-byte {tag = step+1, 2 x bf16} granules written with agent-scope
// atomic stores and polled with agent-scope atomic loads (the data is its own flag: no fence, no separate flag
// word; cdna guide G16/R2).  A pair can serve LP_GROUPS groups of 16 documents alternately so that one group's exchange
// latency is covered by the other's compute; at B = 64 that measured SLOWER (9.0 us per step for two groups vs 4.9 us for
// one: it halves the number of pairs while 97 % of the CUs idle), hence LP_GROUPS = 1.
//
// Safety: a workgroup needs ~84 KiB (forward) / ~97 KiB (backward) of LDS, i.e. one per CU, and a pair can only make progress
// with BOTH halves resident.  A launch therefore covers at most LP_MAX_PAIRS = 64 pairs (128 workgroups, 512 documents x 2
// directions): larger batches run as several launches over consecutive document ranges, so that even two such grids on two HIP
// streams (late fusion) fit the 256 CUs together and no resident workgroup can wait for one that cannot be scheduled.
// Every spin is bounded and a timeout is REPORTED (lp_report_timeout -> MTS_ERR_TIMEOUT at the next call); the exchange buffer
// is zeroed by a memset node before every launch; tags count steps within the call.  Placement (pair members 8 block ids apart =
// same XCD under round-robin dispatch) is a speed hint only.
#include <stdlib.h>
#include <algorithm>
#include <type_traits>
#include "common.h"

#define LP_DOCS 16
#define LP_GROUPS 1
#define LP_SPIN_LIMIT (1u << 22)

typedef unsigned long long u64;

// A partner poll that gives up (LP_SPIN_LIMIT re-polls of ~64 cycles each, about 0.1 s) marks the launch in its own status word
// and in a pinned HOST word (system-scope store): the host reads that word without synchronising at the next mts_lstm_* call /
// mts_async_status() and reports MTS_ERR_TIMEOUT, so a launch that computed on stale h never passes silently.
__device__ __forceinline__ void lp_report_timeout(unsigned* status, unsigned* sticky, unsigned bit) {
  atomicOr(status, bit);
  __hip_atomic_store(sticky, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// v_rcp_f32 (1 ulp) instead of __frcp_rn: the correctly rounded reciprocal is a ten-instruction division sequence, and the
// 20 of them per step were a third of the time loop's instruction stream; the results are rounded to bf16 anyway
__device__ __forceinline__ float fsig2(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float ftanh2(float x) { return 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(-2.0f * x)) - 1.0f; }
__device__ __forceinline__ uint2 pk4(const float (&v)[4]) { return make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])); }
__device__ __forceinline__ void upk4(const uint2& u, float (&v)[4]) { v[0] = bf16_lo(u.x); v[1] = bf16_hi(u.x); v[2] = bf16_lo(u.y); v[3] = bf16_hi(u.y); }

// packed weights: wpk[d][p][w][gate][ks][lane] = 8 bf16 = the MFMA A fragment (row = gate column, k) wave w of half p needs
__global__ void lstm_pack_weights_kernel(const float* __restrict__ w_hh, bf16_t* __restrict__ wpk, int H, int ndir) {
  const int KS = H / 32, NW = H / 32;                                  // waves per half = (H/2)/16
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;          // over d, p, w, gate, ks, lane
  const size_t total = (size_t)ndir * 2 * NW * 4 * KS * 64;
  if (idx >= total) return;
  const int lane = idx % 64;
  size_t r = idx / 64;
  const int ks = r % KS; r /= KS;
  const int gt = r % 4; r /= 4;
  const int w = r % NW; r /= NW;
  const int p = r % 2; r /= 2;
  const int d = (int)r;
  const int col = gt * H + p * (H / 2) + w * 16 + (lane & 15);
  const int k = ks * 32 + 8 * (lane >> 4);
  const float* src = w_hh + ((size_t)d * 4 * H + col) * H + k;
  bf16_t* dst = wpk + idx * 8;
#pragma unroll
  for (int j = 0; j < 8; ++j) dst[j] = (bf16_t)src[j];
}

template <int KS>
__global__ __launch_bounds__(KS * 64, 2) void lstm_fwd_pair_kernel(int B, int L, int ndir, int npairs, const bf16_t* __restrict__ xproj,
                                                                   const bf16_t* __restrict__ wpk, const float* __restrict__ bhh,
                                                                   const int32_t* __restrict__ lengths, bf16_t* __restrict__ out,
                                                                   bf16_t* __restrict__ gates, float* __restrict__ cells, u64* __restrict__ xch,
                                                                   unsigned* __restrict__ status, char* __restrict__ dump, int xflags,
                                                                   unsigned spin_limit, unsigned* __restrict__ sticky) {
  constexpr int H = KS * 32;
  constexpr int HH = H / 2;                              // units per workgroup
  constexpr int NT = KS * 64;                            // threads
  constexpr int GPT = 1024 / NT;                         // granules each thread fetches
  constexpr int HROW = (H + 8) * 2;
  // xflags (MTS_LSTM_EXP, timing diagnostics, only in a -DMTS_LSTM_ABLATE build: the run-time switches cost scalar branches
  // and code size in the time loop): 1 no gate/cell/out stores, 2 do not wait for the partner, 4 no posts, 8 no fetch,
  // 16 no transcendentals, 32 no MFMA, 64 no barrier, 128 no x loads
#ifdef MTS_LSTM_ABLATE
  const bool x_nostore = xflags & 1, x_nowait = xflags & 2, x_nopost = xflags & 4;
  const bool x_nofetch = xflags & 8, x_nomath = xflags & 16, x_nomfma = xflags & 32, x_nobar = xflags & 64, x_nox = xflags & 128;
#else
  constexpr bool x_nostore = false, x_nowait = false, x_nopost = false, x_nofetch = false, x_nomath = false, x_nomfma = false, x_nobar = false,
                 x_nox = false;
  (void)xflags;
#endif
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* hbuf = smem;                                     // [group][parity][16][HROW]
  float* blds = reinterpret_cast<float*>(smem + LP_GROUPS * 2 * LP_DOCS * HROW);   // [4][HH] this half's recurrent bias
  // pair index and half from a 1-D grid: the two halves of a pair are 8 block ids apart
  const int chunk = blockIdx.x / 16, within = blockIdx.x % 16;
  const int p = within / 8, pair = chunk * 8 + within % 8;
  if (pair >= npairs) return;
  const int gx = pair / ndir, d = pair % ndir;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int doc = lane & 15, g4 = lane >> 4;
  const int ul = w * 16 + 4 * g4;                        // this lane's 4 units, within the half
  const int u = p * HH + ul;                             // ... within H
  const int ldx = ndir * 4 * H, ldo = ndir * H;

  // resident weights: gate tiles i, f, g in registers (96 VGPRs), the o tile in a wave-private LDS area (lane-linear 1 KiB
  // per k-step: conflict-free) -- all four in registers leaves too few VGPRs for the step's working set and spills
  bf16x8 wreg[3][KS];
  char* wl = smem + LP_GROUPS * 2 * LP_DOCS * HROW + 4 * HH * sizeof(float) + (size_t)w * KS * 1024;
  {
    const bf16_t* base = wpk + ((((size_t)d * 2 + p) * KS + w) * 4 * KS * 64) * 8;
#pragma unroll
    for (int gt = 0; gt < 3; ++gt)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) wreg[gt][ks] = *reinterpret_cast<const bf16x8*>(base + ((size_t)(gt * KS + ks) * 64 + lane) * 8);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      *reinterpret_cast<bf16x8*>(wl + ks * 1024 + lane * 16) = *reinterpret_cast<const bf16x8*>(base + ((size_t)(3 * KS + ks) * 64 + lane) * 8);
  }
  for (int i = tid; i < 4 * HH; i += NT) blds[i] = bhh ? bhh[(size_t)d * 4 * H + (i / HH) * H + p * HH + (i % HH)] : 0.f;

  int bdoc[LP_GROUPS], len[LP_GROUPS];
  int maxlen = 0;
#pragma unroll
  for (int g = 0; g < LP_GROUPS; ++g) {
    bdoc[g] = (gx * LP_GROUPS + g) * LP_DOCS + doc;
    len[g] = (bdoc[g] < B) ? (lengths ? min(lengths[bdoc[g]], L) : L) : 0;
    maxlen = max(maxlen, len[g]);
  }
#pragma unroll
  for (int off = 1; off < 16; off <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off, 64));

  // exchange areas: xch[pair][group][half][parity][1024 granules]
  auto xarea = [&](int g, int half, int par) { return xch + ((((size_t)pair * LP_GROUPS + g) * 2 + half) * 2 + par) * 1024; };

  float c[LP_GROUPS][4];
  uint2 hq[LP_GROUPS];
  uint2 xb[2][LP_GROUPS][4];                             // x rows of the next TWO steps of each group, by step parity (refilled right after use)
#pragma unroll
  for (int g = 0; g < LP_GROUPS; ++g) {
    hq[g] = make_uint2(0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r) c[g][r] = 0.f;
  }
  for (int i = tid; i < LP_GROUPS * 2 * LP_DOCS * HROW / 16; i += NT) reinterpret_cast<uint4*>(hbuf)[i] = make_uint4(0, 0, 0, 0);

  // x rows are fetched UNCONDITIONALLY from a clamped (always valid) address: a load inside a divergent branch makes the
  // compiler's wait-count bookkeeping give up and turn every later counted wait of the step (the poll check first of all)
  // into "wait for everything".  Lanes past their document's end read some row of it and never use the value.
  const bf16_t* xbase[LP_GROUPS];
  size_t grow0[LP_GROUPS], orow0[LP_GROUPS];             // element offsets of this lane's columns in row 0 of its document
#pragma unroll
  for (int g = 0; g < LP_GROUPS; ++g) {
    const size_t b0 = (size_t)min(bdoc[g], B - 1) * L;
    grow0[g] = b0 * ldx + (size_t)d * 4 * H + u;
    orow0[g] = b0 * ldo + (size_t)d * H + u;
    xbase[g] = xproj + grow0[g];
  }
  auto load_x = [&](auto parc, int g, int s) {
    constexpr int PAR = decltype(parc)::value;
    const int t = (d == 0) ? s : (len[g] - 1 - s);
    const bf16_t* src = xbase[g] + (size_t)min(max(t, 0), L - 1) * ldx;
#pragma unroll
    for (int gt = 0; gt < 4; ++gt) xb[PAR][g][gt] = *reinterpret_cast<const uint2*>(src + gt * H);
  };
  bool dead = x_nowait;
  // partner's half of h for (group g, after step s) -> LDS hbuf[g][(s+1)&1].  The first poll is ISSUED early (before the
  // caller's stores: vmcnt retires in order, a load queued behind stores waits for their write acks) and only checked
  // here; bounded re-polling if a tag is not there yet.
  u64 v[GPT];
  auto fetch_issue = [&](int g, int s) {
    const u64* src = xarea(g, 1 - p, (s + 1) & 1);
#pragma unroll
    for (int k = 0; k < GPT; ++k) v[k] = __hip_atomic_load(src + tid + NT * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto fetch = [&](int g, int s) {
    const u64* src = xarea(g, 1 - p, (s + 1) & 1);
    const unsigned epoch = (unsigned)(s + 1);
    auto tags_ok = [&]() {
      bool ok = true;
#pragma unroll
      for (int k = 0; k < GPT; ++k) ok &= ((unsigned)(v[k] >> 32) == epoch);
      return __all(ok);
    };
    // the check of the poll issued earlier stands OUTSIDE the retry loop: inside it, it shares the loop header with the
    // re-polls and the compiler can only wait for everything (x rows still in flight included) instead of a counted wait
    if (!tags_ok() && !dead) {
      unsigned spins = 0;
#pragma clang loop unroll(disable)
      for (;;) {
        asm volatile("" : "+s"(spins));                 // opaque counter: keeps the compiler from replicating the poll hundreds of times
        if (++spins > spin_limit) { dead = true; if (lane == 0) lp_report_timeout(status, sticky, 1u); break; }
        __builtin_amdgcn_s_sleep(1);
#pragma unroll
        for (int k = 0; k < GPT; ++k) v[k] = __hip_atomic_load(src + tid + NT * k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tags_ok()) break;
      }
    }
    char* dst = hbuf + ((g * 2 + ((s + 1) & 1)) * LP_DOCS) * HROW;
#pragma unroll
    for (int k = 0; k < GPT; ++k) {
      const int gi = tid + NT * k;                       // granule: half-quad k2 = gi / 512, quad = (gi % 512) / 16, doc = gi % 16
      const int k2 = gi >> 9, quad = (gi & 511) >> 4, dd = gi & 15;
      *reinterpret_cast<unsigned*>(dst + dd * HROW + ((1 - p) * HH + quad * 4 + k2 * 2) * 2) = (unsigned)v[k];
    }
  };

#pragma unroll
  for (int g = 0; g < LP_GROUPS; ++g) {
    load_x(std::integral_constant<int, 0>{}, g, 0);
    load_x(std::integral_constant<int, 1>{}, g, 1);
  }
  __syncthreads();
  // this lane's 4 x 4 recurrent biases stay in registers for the whole sequence
  float bia[4][4];
#pragma unroll
  for (int gt = 0; gt < 4; ++gt) {
    const float4 b4 = *reinterpret_cast<const float4*>(blds + gt * HH + ul);
    bia[gt][0] = b4.x; bia[gt][1] = b4.y; bia[gt][2] = b4.z; bia[gt][3] = b4.w;
  }

  // one time step of both groups; PAR = s & 1 is a compile-time constant so the x buffers are statically indexed
  auto step = [&](auto parc, int s) {
    constexpr int PAR = decltype(parc)::value;
#pragma unroll
    for (int g = 0; g < LP_GROUPS; ++g) {
      const char* hcur = hbuf + ((g * 2 + PAR) * LP_DOCS) * HROW;
      char* hnext = hbuf + ((g * 2 + (PAR ^ 1)) * LP_DOCS) * HROW;
      bf16x8 hf[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) hf[ks] = *reinterpret_cast<const bf16x8*>(hcur + doc * HROW + (ks * 32 + 8 * g4) * 2);
      f32x4 acc[4];
#pragma unroll
      for (int gt = 0; gt < 4; ++gt) acc[gt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (!x_nomfma) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
          for (int gt = 0; gt < 3; ++gt) acc[gt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[gt][ks], hf[ks], acc[gt], 0, 0, 0);
          acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(wl + ks * 1024 + lane * 16), hf[ks], acc[3], 0, 0, 0);
        }
      }

      const int og = (g + 1) % LP_GROUPS;
      const int os = (g == LP_GROUPS - 1) ? s : s - 1;   // group og last finished step os
      // Gate math runs on EVERY lane (documents that have ended included) and `active` only selects what is kept: no
      // divergent branch in the step, so the compiler schedules it as one block and counts its memory waits exactly.
      const bool active = s < len[g];
      float gi[4], gf[4], gg[4], go[4];
      {
        float xi[4], xf[4], xg[4], xo[4], hn[4];
        upk4(xb[PAR][g][0], xi); upk4(xb[PAR][g][1], xf); upk4(xb[PAR][g][2], xg); upk4(xb[PAR][g][3], xo);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float cn;
          if (x_nomath) {
            gi[r] = (xi[r] + bia[0][r]) + acc[0][r]; gf[r] = (xf[r] + bia[1][r]) + acc[1][r];
            gg[r] = (xg[r] + bia[2][r]) + acc[2][r]; go[r] = (xo[r] + bia[3][r]) + acc[3][r];
            cn = gf[r] * c[g][r] + gi[r] * gg[r];
            hn[r] = go[r] * cn;
          } else {
            gi[r] = fsig2((xi[r] + bia[0][r]) + acc[0][r]);
            gf[r] = fsig2((xf[r] + bia[1][r]) + acc[1][r]);
            gg[r] = ftanh2((xg[r] + bia[2][r]) + acc[2][r]);
            go[r] = fsig2((xo[r] + bia[3][r]) + acc[3][r]);
            cn = gf[r] * c[g][r] + gi[r] * gg[r];
            hn[r] = go[r] * ftanh2(cn);
          }
          c[g][r] = active ? cn : c[g][r];
        }
        const uint2 hnew = pk4(hn);
        hq[g].x = active ? hnew.x : hq[g].x;
        hq[g].y = active ? hnew.y : hq[g].y;
      }
      // 1) post the own half of the new h first (shortest exchange path): LDS for this workgroup, tagged granules for
      //    the partner (inactive documents carry h)
      *reinterpret_cast<uint2*>(hnext + doc * HROW + u * 2) = hq[g];
      if (!x_nopost) {
        u64* mine = xarea(g, p, PAR ^ 1);
        const u64 tag = (u64)(unsigned)(s + 1) << 32;
        // granule k of (document, unit quad) sits at k * 512 + quad * 16 + document: each of the two stores of a wave is one
        // contiguous 512-byte access (quad = ul / 4 = 4 * wave + lane / 16), and so is each poll of the partner
        __hip_atomic_store(mine + (ul >> 2) * 16 + doc, tag | hq[g].x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(mine + 512 + (ul >> 2) * 16 + doc, tag | hq[g].y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // 2) first poll of the OTHER group's partner half (posted about one phase ago when LP_GROUPS = 2)
      if (os >= 0 && !x_nofetch) fetch_issue(og, os);
      // 3) x rows two steps ahead (same parity buffer, just consumed): loads are queued BEFORE the bulk stores -- the
      //    vector-memory counter retires in order, so anything waited on behind a store also waits for its write ack
      if (!x_nox) load_x(parc, g, s + 2);
      // the OTHER group's partner half was posted one compute phase ago: finish its fetch, then both halves are in LDS
      if (os >= 0 && !x_nofetch) fetch(og, os);
      // 4) saved state for the backward pass, last: nothing in this step waits behind these stores, their acks arrive
      //    under the next step's LDS reads and MFMAs
      //    Lanes whose document has ended store to a dump area instead of branching around the stores.
      if (!x_nostore) {
        const int t = (d == 0) ? s : (len[g] - 1 - s);
        bf16_t* gp = active ? gates + grow0[g] + (size_t)t * ldx : reinterpret_cast<bf16_t*>(dump + tid * 16);
        float* cptr = active ? cells + orow0[g] + (size_t)t * ldo : reinterpret_cast<float*>(dump + tid * 16);
        bf16_t* optr = active ? out + orow0[g] + (size_t)t * ldo : reinterpret_cast<bf16_t*>(dump + tid * 16);
        *reinterpret_cast<uint2*>(gp) = pk4(gi);
        *reinterpret_cast<uint2*>(gp + H) = pk4(gf);
        *reinterpret_cast<uint2*>(gp + 2 * H) = pk4(gg);
        *reinterpret_cast<uint2*>(gp + 3 * H) = pk4(go);
        *reinterpret_cast<float4*>(cptr) = make_float4(c[g][0], c[g][1], c[g][2], c[g][3]);
        *reinterpret_cast<uint2*>(optr) = hq[g];
      }
      if (!x_nobar) __syncthreads();
    }
  };
  __builtin_amdgcn_s_waitcnt(0);                         // clean scoreboard at the loop header (weights, biases, first x rows have landed)
  for (int s = 0; s < maxlen; s += 2) {
    step(std::integral_constant<int, 0>{}, s);
    if (s + 1 < maxlen) step(std::integral_constant<int, 1>{}, s + 1);
  }
  // rows >= len are exactly zero
#pragma unroll
  for (int g = 0; g < LP_GROUPS; ++g)
    if (bdoc[g] < B)
      for (int t = len[g] + g4; t < L; t += 4)
        for (int e = 0; e < 16; e += 4) *reinterpret_cast<uint2*>(out + ((size_t)bdoc[g] * L + t) * ldo + (size_t)d * H + p * HH + w * 16 + e) = make_uint2(0, 0);
}

// ---- host side ---------------------------------------------------------------------------------------
static thread_local int g_pair_mode = -1;   // MTS_LSTM_PAIR=0 disables
#define LP_MAX_PAIRS 64
static thread_local unsigned g_spin_limit = LP_SPIN_LIMIT;      // mts_set_option("lstm_pair_spin_limit", n): tests force a timeout with n = 0
static thread_local int g_max_pairs = LP_MAX_PAIRS;             // mts_set_option("lstm_pair_max_pairs", n): tests exercise the multi-launch path
static unsigned* g_sticky_host = nullptr;          // pinned, device-visible: non-zero = some pair launch timed out
static unsigned* g_sticky_dev = nullptr;

void mts_lstm_pair_set_spin_limit(int n) { g_spin_limit = n < 0 ? LP_SPIN_LIMIT : (unsigned)n; }
void mts_lstm_pair_set_max_pairs(int n) { g_max_pairs = (n <= 0 || n > LP_MAX_PAIRS) ? LP_MAX_PAIRS : n; }

static int lp_ensure_sticky() {
  if (g_sticky_dev) return MTS_OK;
  void* h = nullptr;
  if (hipHostMalloc(&h, 256, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) { mts_set_error("lstm_pair: hipHostMalloc failed"); return MTS_ERR_LAUNCH; }
  *(volatile unsigned*)h = 0u;
  void* d = nullptr;
  if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { mts_set_error("lstm_pair: hipHostGetDevicePointer failed"); return MTS_ERR_LAUNCH; }
  g_sticky_host = (unsigned*)h;
  g_sticky_dev = (unsigned*)d;
  return MTS_OK;
}

// bit 0: a forward pair launch timed out, bit 1: a backward one; reading clears
unsigned mts_lstm_pair_take_error() {
  if (!g_sticky_host) return 0u;
  const unsigned v = *(volatile unsigned*)g_sticky_host;
  if (v) *(volatile unsigned*)g_sticky_host = 0u;
  return v;
}

static thread_local int g_lstm_parts_opt = 4;      // mts_set_option("lstm_parts", 2 | 4): CU pair or CU quad form of the recurrences
static thread_local int g_lstm_parts_forced = 0;   // 2 | 4 while a backward call follows the form its forward recorded (lstm.hip), else 0
#define g_lstm_parts (g_lstm_parts_forced ? g_lstm_parts_forced : g_lstm_parts_opt)
void mts_lstm_pair_set_parts(int n) { g_lstm_parts_opt = (n == 2) ? 2 : 4; }
void mts_lstm_pair_force_parts(int n) { g_lstm_parts_forced = (n == 2 || n == 4) ? n : 0; }
int mts_lstm_pair_parts() { return g_lstm_parts; }
bool mts_lstm_pair_supported(int dtype, int H) {
  if (g_pair_mode < 0) { const char* e = getenv("MTS_LSTM_PAIR"); g_pair_mode = (e && e[0] == '0') ? 0 : 1; }
  // fp32 (parity mode): only the CU-quad form exists (W_hh in fp32 is 256 registers per lane of a quad's waves)
  return g_pair_mode && H == 256 && (dtype == MTS_BF16 || (dtype == MTS_F32 && g_lstm_parts == 4));
}

// workspace: packed weights (bf16) | exchange granules | status word
static size_t pair_wbytes(int H, int ndir) { return align_up((size_t)ndir * 4 * H * H * 4, 256); }     // packed W_hh: bf16 forms use half of it
// exchange granules per (16 documents, direction): pair form 2 halves x 2 parities x 1024; quad forward 4 x 2 x 512 (the same);
// quad backward 4 sources x 4 destinations x 2 parities x 512 (fp32 form: 1024, one value per granule) -- sized for the largest
static size_t pair_xbytes(int B, int ndir) { return align_up((size_t)ceil_div(B, LP_DOCS * LP_GROUPS) * ndir * LP_GROUPS * 4 * 4 * 2 * 1024 * sizeof(u64), 256); }
// ... | status word (256 B) | dump area (stores of lanes whose document has ended)
#define LP_DUMP_BYTES 16384
size_t mts_lstm_pair_workspace(int B, int H, int ndir) { return pair_wbytes(H, ndir) + pair_xbytes(B, ndir) + 256 + LP_DUMP_BYTES; }

static int lstm_quad_fwd_launch(hipStream_t st, int B, int L, int H, int ndir, const void* xproj, const bf16_t* wpk, const float* b_hh, const int32_t* lengths,
                                void* out, void* gates, float* cells, u64* xch, unsigned* status);

int mts_lstm_pair_fwd(hipStream_t st, int B, int L, int H, int ndir, const void* xproj, const float* w_hh, const float* b_hh,
                      const int32_t* lengths, void* out, void* gates, float* cells, void* ws) {
  constexpr int KS = 8;
  bf16_t* wpk = (bf16_t*)ws;
  u64* xch = (u64*)((char*)ws + pair_wbytes(H, ndir));
  unsigned* status = (unsigned*)((char*)xch + pair_xbytes(B, ndir));
  const size_t total = (size_t)ndir * 2 * KS * 4 * KS * 64;
  hipLaunchKernelGGL(lstm_pack_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w_hh, wpk, H, ndir);
  if (hipMemsetAsync(xch, 0, pair_xbytes(B, ndir) + 256, st) != hipSuccess) { mts_set_error("lstm_pair: memset failed"); return MTS_ERR_LAUNCH; }
  const size_t lds = (size_t)LP_GROUPS * 2 * LP_DOCS * (H + 8) * 2 + (size_t)4 * (H / 2) * sizeof(float) + (size_t)KS * KS * 1024;
  auto k = lstm_fwd_pair_kernel<KS>;
  static std::atomic<bool> attr{false};
  if (!attr) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      mts_set_error("lstm_pair_fwd: cannot reserve %zu bytes of LDS", lds);
      return MTS_ERR_LAUNCH;
    }
    attr = true;
  }
  static int xflags = -1;
  if (xflags < 0) { const char* e = getenv("MTS_LSTM_EXP"); xflags = e ? atoi(e) : 0; }
  if (int rc = lp_ensure_sticky()) return rc;
  if (g_lstm_parts == 4) return lstm_quad_fwd_launch(st, B, L, H, ndir, xproj, wpk, b_hh, lengths, out, gates, cells, xch, status);
  // at most g_max_pairs pairs per launch (see "Safety" at the top): documents [b0, b0 + bc) per launch, rows are b * L + i
  const int docs_per_launch = std::max(1, g_max_pairs / ndir) * LP_DOCS * LP_GROUPS;
  for (int b0 = 0; b0 < B; b0 += docs_per_launch) {
    const int bc = std::min(docs_per_launch, B - b0);
    const size_t r0 = (size_t)b0 * L;
    if (b0 > 0 && hipMemsetAsync(xch, 0, pair_xbytes(bc, ndir) + 256, st) != hipSuccess) { mts_set_error("lstm_pair: memset failed"); return MTS_ERR_LAUNCH; }
    const int npairs = ceil_div(bc, LP_DOCS * LP_GROUPS) * ndir;
    hipLaunchKernelGGL(k, dim3(16 * ceil_div(npairs, 8)), dim3(KS * 64), lds, st, bc, L, ndir, npairs, (const bf16_t*)xproj + r0 * ndir * 4 * H,
                       (const bf16_t*)wpk, b_hh, lengths ? lengths + b0 : nullptr, (bf16_t*)out + r0 * ndir * H, (bf16_t*)gates + r0 * ndir * 4 * H,
                       cells + r0 * ndir * H, xch, status, (char*)status + 256, xflags, g_spin_limit, g_sticky_dev);
  }
  MTS_LAUNCH_CHECK("mts_lstm_fwd(pair)");
  return MTS_OK;
}

// =====================================================================================================
// backward, CU-pair form.  Half p owns the gate columns of its own 128 units (da_own, 16 x 512 per group) and keeps
// W_hh[own columns][all 256 units] resident; per step it computes the partial  dh[:, all units] = da_own . W_own,
// keeps the 128 columns of its own units (LDS, fp32) and sends the other 128 to the partner as bf16 granules.
// Elementwise gate-gradient work is spread over all 512 threads (thread = document x 4 own units).
// =====================================================================================================
// packed: wpkT[d][p][w][t][ks][lane] = 8 bf16: A fragment row = output unit j, tile t = 0: unit w*16 + (lane&15) of THIS half,
// t = 1: the same unit of the PARTNER's half; k = own column index kk = ks*32 + 8*(lane>>4) .. +7, i.e.
// W_hh[gate*H + p*H/2 + kk%128][j] with gate = kk/128
__global__ void lstm_pack_weights_T_kernel(const float* __restrict__ w_hh, bf16_t* __restrict__ wpk, int H, int ndir) {
  const int HH = H / 2, KT = 4 * HH / 32, NW = H / 32;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;          // over d, p, w, t, ks, lane
  const size_t total = (size_t)ndir * 2 * NW * 2 * KT * 64;
  if (idx >= total) return;
  const int lane = idx % 64;
  size_t r = idx / 64;
  const int ks = r % KT; r /= KT;
  const int t = r % 2; r /= 2;
  const int w = r % NW; r /= NW;
  const int p = r % 2; r /= 2;
  const int d = (int)r;
  const int j = (t == 0 ? p : 1 - p) * HH + w * 16 + (lane & 15);
  bf16_t* dst = wpk + idx * 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int kk = ks * 32 + 8 * (lane >> 4) + e;
    const int n = (kk / HH) * H + p * HH + (kk % HH);
    dst[e] = (bf16_t)w_hh[((size_t)d * 4 * H + n) * H + j];
  }
}

struct PairBwdIn { uint2 gi, gf, gg, go, dov; float4 ct, cp; };

// Step structure (the forward kernel's, mirrored: one workgroup barrier and one LDS round trip per time step).  Wave w
// owns output tile 0 = units w*16..+15 of THIS half and tile 1 = the same units of the PARTNER's half; both are reduced over
// this half's 512 gate columns.  The MFMA result layout (lane = document x 4 units) IS the elementwise layout: a lane
// keeps its own partial of dh in registers, posts the partner-tile partial, polls the partner's partial for exactly its
// (document, unit quad) into registers, does the gate-gradient math and writes da to LDS (double-buffered by step parity);
// after the barrier every wave reads da and runs its MFMAs for the previous time step.
template <int KS>
__global__ __launch_bounds__(KS * 64, 2) void lstm_bwd_pair_kernel(int B, int L, int ndir, int npairs, const bf16_t* __restrict__ wpkT,
                                                                   const int32_t* __restrict__ lengths, const bf16_t* __restrict__ gates,
                                                                   const float* __restrict__ cells, const bf16_t* __restrict__ dout,
                                                                   bf16_t* __restrict__ dxproj, u64* __restrict__ xch, unsigned* __restrict__ status,
                                                                   char* __restrict__ dump, unsigned spin_limit, unsigned* __restrict__ sticky) {
  static_assert(LP_GROUPS == 1 && KS == 8, "the backward pair kernel serves one group of documents per pair, H = 256");
  constexpr int H = KS * 32, HH = H / 2, NT = KS * 64;
  constexpr int KT = 4 * HH / 32;                        // k-steps over the 512 own gate columns
  constexpr int RK = KT + KT / 2;                        // fragments kept in registers: tile 0 all, tile 1 first half
  constexpr int DAROW = (4 * HH + 8) * 2;                // bytes per da row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* dabuf = smem;                                    // [2 (step parity)][16][DAROW] bf16
  char* wlds = dabuf + 2 * LP_DOCS * DAROW;              // [wave][KT/2][1024]
  const int chunk = blockIdx.x / 16, within = blockIdx.x % 16;
  const int p = within / 8, pair = chunk * 8 + within % 8;
  if (pair >= npairs) return;
  const int gx = pair / ndir, d = pair % ndir;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int doc = lane & 15, g4 = lane >> 4;
  const int ul = w * 16 + 4 * g4;                        // this lane's 4 units, within the half
  const int u = p * HH + ul;                             // ... within H
  const int ldx = ndir * 4 * H, ldo = ndir * H;

  bf16x8 wreg[RK];
  char* wl = wlds + (size_t)w * (KT / 2) * 1024;
  {
    const bf16_t* base = wpkT + ((((size_t)d * 2 + p) * KS + w) * 2 * KT * 64) * 8;
#pragma unroll
    for (int f = 0; f < RK; ++f) wreg[f] = *reinterpret_cast<const bf16x8*>(base + ((size_t)f * 64 + lane) * 8);
#pragma unroll
    for (int f = 0; f < KT / 2; ++f)
      *reinterpret_cast<bf16x8*>(wl + f * 1024 + lane * 16) = *reinterpret_cast<const bf16x8*>(base + ((size_t)(RK + f) * 64 + lane) * 8);
  }

  const int bdoc = gx * LP_DOCS + doc;
  const int len = (bdoc < B) ? (lengths ? min(lengths[bdoc], L) : L) : 0;
  int maxlen = len;
#pragma unroll
  for (int off = 1; off < 16; off <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off, 64));

  // exchange areas: xch[pair][half][parity][1024 granules]; granule k of (document, unit quad) sits at k * 512 + quad * 16 +
  // document = k * 512 + tid for the lane that owns the position on EITHER side: posts and polls are contiguous 512-byte
  // wave accesses and a poll lands in the consuming lane's registers
  auto xarea = [&](int half, int par) { return xch + (((size_t)pair * 2 + half) * 2 + par) * 1024; };
  for (int i = tid; i < 2 * LP_DOCS * DAROW / 4; i += NT) reinterpret_cast<unsigned*>(smem)[i] = 0u;

  // saved state of step s, fetched UNCONDITIONALLY from clamped (always valid) addresses -- see load_x in the forward
  // kernel; steps past a document's end read some row of it and the values are never used
  const size_t b0 = (size_t)min(bdoc, B - 1) * L;
  const size_t grow0 = b0 * ldx + (size_t)d * 4 * H + u;
  const size_t orow0 = b0 * ldo + (size_t)d * H + u;
  auto load_in = [&](int s, PairBwdIn& in) {
    const int sc = max(s, 0);
    const int t = min(max((d == 0) ? sc : (len - 1 - sc), 0), L - 1);
    const int tp = min(max((d == 0) ? t - 1 : t + 1, 0), L - 1);
    const bf16_t* gp = gates + grow0 + (size_t)t * ldx;
    in.gi = *reinterpret_cast<const uint2*>(gp);
    in.gf = *reinterpret_cast<const uint2*>(gp + H);
    in.gg = *reinterpret_cast<const uint2*>(gp + 2 * H);
    in.go = *reinterpret_cast<const uint2*>(gp + 3 * H);
    in.dov = *reinterpret_cast<const uint2*>(dout + orow0 + (size_t)t * ldo);
    in.ct = *reinterpret_cast<const float4*>(cells + orow0 + (size_t)t * ldo);
    in.cp = *reinterpret_cast<const float4*>(cells + orow0 + (size_t)tp * ldo);       // zeroed at use when s == 0
  };
  bool dead = false;
  u64 v[2];
  // partner's partial dh for my quad, produced in its MFMA of step s; tag = maxlen - s (>= 1)
  auto fetch_issue = [&](int s) {
    const u64* src = xarea(1 - p, s & 1) + tid;
    v[0] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v[1] = __hip_atomic_load(src + NT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto fetch = [&](int s) {
    const unsigned epoch = (unsigned)(maxlen - s);
    auto tags_ok = [&]() { return __all(((unsigned)(v[0] >> 32) == epoch) & ((unsigned)(v[1] >> 32) == epoch)); };
    // the check of the poll issued earlier stands OUTSIDE the retry loop (counted wait, see the forward kernel)
    if (!tags_ok() && !dead) {
      unsigned spins = 0;
#pragma clang loop unroll(disable)
      for (;;) {
        asm volatile("" : "+s"(spins));                 // opaque counter: keeps the compiler from replicating the poll hundreds of times
        if (++spins > spin_limit) { dead = true; if (lane == 0) lp_report_timeout(status, sticky, 2u); break; }
        __builtin_amdgcn_s_sleep(1);
        fetch_issue(s);
        if (tags_ok()) break;
      }
    }
  };

  float dc[4] = {0.f, 0.f, 0.f, 0.f};
  f32x4 own = (f32x4){0.f, 0.f, 0.f, 0.f};               // own partial of dh for this lane's quad, from the previous MFMA phase
  PairBwdIn in[2];                                       // saved state of the next TWO steps, by iteration parity
  load_in(maxlen - 1, in[0]);
  load_in(maxlen - 2, in[1]);
  __syncthreads();

  // one time step; PAR = (maxlen - 1 - s) & 1 is a compile-time constant so that `in` and the da buffer are statically indexed
  auto step = [&](auto parc, int s) {
    constexpr int PAR = decltype(parc)::value;
    char* da = dabuf + PAR * LP_DOCS * DAROW;
    const bool active = s < len;
    const PairBwdIn& cur = in[PAR];
    // ---- dh = own partial + partner partial (+ dOut) -> gate pre-activation gradients; math on every lane ----------
    uint2 oth = make_uint2(0, 0);
    if (s + 1 <= maxlen - 1) {
      fetch(s + 1);
      oth = make_uint2((unsigned)v[0], (unsigned)v[1]);
    }
    uint2 q[4];
    {
      const float dhv[4] = {own[0] + bf16_lo(oth.x), own[1] + bf16_hi(oth.x), own[2] + bf16_lo(oth.y), own[3] + bf16_hi(oth.y)};
      float gi[4], gf[4], gg[4], go[4], dov[4], ai[4], af[4], ag[4], ao[4];
      upk4(cur.gi, gi); upk4(cur.gf, gf); upk4(cur.gg, gg); upk4(cur.go, go); upk4(cur.dov, dov);
      const float ct[4] = {cur.ct.x, cur.ct.y, cur.ct.z, cur.ct.w};
      const bool has_prev = s > 0;                       // the first processed position has c_prev = 0
      const float cp[4] = {has_prev ? cur.cp.x : 0.f, has_prev ? cur.cp.y : 0.f, has_prev ? cur.cp.z : 0.f, has_prev ? cur.cp.w : 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float tc = ftanh2(ct[r]);
        const float dht = dov[r] + dhv[r];
        const float dct = dc[r] + dht * go[r] * (1.f - tc * tc);
        ai[r] = active ? dct * gg[r] * gi[r] * (1.f - gi[r]) : 0.f;
        af[r] = active ? dct * cp[r] * gf[r] * (1.f - gf[r]) : 0.f;
        ag[r] = active ? dct * gi[r] * (1.f - gg[r] * gg[r]) : 0.f;
        ao[r] = active ? dht * tc * go[r] * (1.f - go[r]) : 0.f;
        dc[r] = active ? dct * gf[r] : dc[r];
      }
      q[0] = pk4(ai); q[1] = pk4(af); q[2] = pk4(ag); q[3] = pk4(ao);
    }
    char* dr = da + doc * DAROW + ul * 2;
#pragma unroll
    for (int gt = 0; gt < 4; ++gt) *reinterpret_cast<uint2*>(dr + gt * HH * 2) = q[gt];
    // inputs two steps ahead (same parity buffer, just consumed), then the input-projection gradients; lanes whose document
    // has ended store to the dump area instead of branching
    load_in(s - 2, in[PAR]);
    {
      const int t = (d == 0) ? s : (len - 1 - s);
      bf16_t* dx = active ? dxproj + grow0 + (size_t)t * ldx : reinterpret_cast<bf16_t*>(dump + tid * 16);
#pragma unroll
      for (int gt = 0; gt < 4; ++gt) *reinterpret_cast<uint2*>(dx + gt * H) = q[gt];
    }
    __syncthreads();
    // ---- partial dh from the own gate columns: tile 0 = my units (kept), tile 1 = the partner's units (posted) ------
    f32x4 acc[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ks = 0; ks < KT; ++ks) {
      const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(da + doc * DAROW + (ks * 32 + 8 * g4) * 2);
      const bf16x8 w1 = (ks < KT / 2) ? wreg[KT + ks] : *reinterpret_cast<const bf16x8*>(wl + (ks - KT / 2) * 1024 + lane * 16);
      acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1, bfr, acc[1], 0, 0, 0);
      acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[ks], bfr, acc[0], 0, 0, 0);
    }
    {
      const u64 tag = (u64)(unsigned)(maxlen - s) << 32;
      u64* dst = xarea(p, s & 1) + tid;
      const float vv[4] = {acc[1][0], acc[1][1], acc[1][2], acc[1][3]};
      const uint2 pk = pk4(vv);
      __hip_atomic_store(dst, tag | pk.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(dst + NT, tag | pk.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    own = acc[0];
    if (s - 1 >= 0) fetch_issue(s);                      // the partner computes its partial for my units in this same phase
  };
  __builtin_amdgcn_s_waitcnt(0);                         // clean scoreboard at the loop header
  for (int s = maxlen - 1; s >= 0; s -= 2) {
    step(std::integral_constant<int, 0>{}, s);
    if (s - 1 >= 0) step(std::integral_constant<int, 1>{}, s - 1);
  }
  // rows >= len: zero gradients (the GEMMs that follow read every row)
  if (bdoc < B)
    for (int t = len; t < L; ++t)
#pragma unroll
      for (int gt = 0; gt < 4; ++gt) *reinterpret_cast<uint2*>(dxproj + ((size_t)bdoc * L + t) * ldx + (size_t)d * 4 * H + gt * H + u) = make_uint2(0, 0);
}

// =====================================================================================================
// "CU quad" form of the forward recurrence: the hidden units of one (16 documents, direction) are split over FOUR workgroups
// (64 units each, 4 waves = ONE per SIMD, all four gate tiles of a wave's 16 units in registers: 128 VGPRs).  Against the pair
// form a step's matrix-core time and gate math halve (32 MFMAs + one wave's elementwise work per SIMD instead of two waves' worth)
// and nothing of W_hh lives in LDS; what grows is the exchange: every workgroup posts its quarter of the new h (16 x 64 bf16 = 512
// tagged granules) and polls the three other quarters (6 granules per thread instead of 2).  The exchange latency itself (store ->
// L2 -> poll, ~1 us) is unchanged, so the dependent step is  ~0.25 (MFMA) + ~0.3 (gates) + ~1.0 (hand-off) + LDS / barrier.
// Same packed weights, same exchange-buffer size and the same safety rules as the pair form (bounded polls, reported timeouts, a
// launch never holds more workgroups than stay co-resident: LQ_MAX_QUADS x 4 = 128).
// =====================================================================================================
#define LQ_MAX_QUADS 32
template <int KS>
__global__ __launch_bounds__(256, 1) void lstm_fwd_quad_kernel(int B, int L, int ndir, int nquads, const bf16_t* __restrict__ xproj,
                                                               const bf16_t* __restrict__ wpk, const float* __restrict__ bhh,
                                                               const int32_t* __restrict__ lengths, bf16_t* __restrict__ out,
                                                               bf16_t* __restrict__ gates, float* __restrict__ cells, u64* __restrict__ xch,
                                                               unsigned* __restrict__ status, char* __restrict__ dump, unsigned spin_limit,
                                                               unsigned* __restrict__ sticky) {
  constexpr int H = KS * 32, HQ = H / 4, NT = 256, HROW = (H + 8) * 2;
  static_assert(KS == 8, "H = 256");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* hbuf = smem;                                     // [parity][16][HROW]
  float* blds = reinterpret_cast<float*>(smem + 2 * LP_DOCS * HROW);   // [4][HQ] this quarter's recurrent bias
  // quad index and part from a 1-D grid: the four parts of a quad are 8 block ids apart (same XCD under round-robin dispatch)
  const int chunk = blockIdx.x / 32, within = blockIdx.x % 32;
  const int p = within / 8, quad = chunk * 8 + within % 8;
  if (quad >= nquads) return;
  const int gx = quad / ndir, d = quad % ndir;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int doc = lane & 15, g4 = lane >> 4;
  const int ul = w * 16 + 4 * g4;                        // this lane's 4 units, within the quarter
  const int u = p * HQ + ul;                             // ... within H
  const int ldx = ndir * 4 * H, ldo = ndir * H;

  // resident weights: all four gate tiles x 8 k-steps of the wave's 16 units.  Packed as for the pair form: units p*64 + w*16 .. are
  // "half p >> 1, wave (p & 1) * 4 + w" there.
  bf16x8 wreg[4][KS];
  {
    const bf16_t* base = wpk + ((((size_t)d * 2 + (p >> 1)) * KS + ((p & 1) * 4 + w)) * 4 * KS * 64) * 8;
#pragma unroll
    for (int gt = 0; gt < 4; ++gt)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) wreg[gt][ks] = *reinterpret_cast<const bf16x8*>(base + ((size_t)(gt * KS + ks) * 64 + lane) * 8);
  }
  for (int i = tid; i < 4 * HQ; i += NT) blds[i] = bhh ? bhh[(size_t)d * 4 * H + (i / HQ) * H + p * HQ + (i % HQ)] : 0.f;

  const int bdoc = gx * LP_DOCS + doc;
  const int len = (bdoc < B) ? (lengths ? min(lengths[bdoc], L) : L) : 0;
  int maxlen = len;
#pragma unroll
  for (int off = 1; off < 16; off <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off, 64));

  // exchange areas: xch[quad][part][parity][512 granules]; granule k2 of (document, unit quad q = ul / 4) sits at k2 * 256 + q * 16 + doc
  auto xarea = [&](int part, int par) { return xch + (((size_t)quad * 4 + part) * 2 + par) * 512; };

  float c[4] = {0.f, 0.f, 0.f, 0.f};
  uint2 hq = make_uint2(0, 0);
  uint2 xb[2][4];                                        // x rows of the next TWO steps, by step parity
  for (int i = tid; i < 2 * LP_DOCS * HROW / 16; i += NT) reinterpret_cast<uint4*>(hbuf)[i] = make_uint4(0, 0, 0, 0);

  const size_t b0 = (size_t)min(bdoc, B - 1) * L;
  const size_t grow0 = b0 * ldx + (size_t)d * 4 * H + u, orow0 = b0 * ldo + (size_t)d * H + u;
  const bf16_t* xbase = xproj + grow0;
  // Saved state (gates, cells) in STEP-MAJOR blocks, private to this (document group, direction, part): what step s writes is one
  // contiguous block [doc][gate][64 units] (8 KB of gates, 4 KB of cells), and the backward pass reads the same block at the same s.
  // In the [document][time] layout of xproj / out a step touches 16 rows that lie L rows apart (16 DRAM pages and TLB entries per tensor
  // and part); from 32 768 tokens on that cost 0.3 / 0.6 us per dependent step (64 x 512: 2.53 / 3.25 us against 2.27 / 2.68 at 64 x 256).
  // The group's region is the rows of its documents in the caller's buffer: [direction][part][step][doc < ndg][...], ndg = documents
  // present in the group, exactly ndg * L * ndir * 4H elements.
  const int ndg = min(LP_DOCS, B - gx * LP_DOCS);
  const size_t gstep = (size_t)ndg * 4 * HQ, cstep = (size_t)ndg * HQ;
  bf16_t* gsave = gates + (size_t)gx * LP_DOCS * L * ldx + ((size_t)(d * 4 + p) * L) * gstep + (size_t)min(doc, ndg - 1) * 4 * HQ + ul;
  float* csave = cells + (size_t)gx * LP_DOCS * L * ldo + ((size_t)(d * 4 + p) * L) * cstep + (size_t)min(doc, ndg - 1) * HQ + ul;
  auto load_x = [&](auto parc, int s) {                  // unconditional, clamped (see the pair kernel)
    constexpr int PAR = decltype(parc)::value;
    const int t = (d == 0) ? s : (len - 1 - s);
    const bf16_t* src = xbase + (size_t)min(max(t, 0), L - 1) * ldx;
#pragma unroll
    for (int gt = 0; gt < 4; ++gt) xb[PAR][gt] = *reinterpret_cast<const uint2*>(src + gt * H);
  };
  bool dead = false;
  // the three other quarters of h after step s -> LDS hbuf[(s+1)&1]; q-th partner = part (p + 1 + q) & 3
  u64 v[3][2];
  auto fetch_issue = [&](int s) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const u64* src = xarea((p + 1 + q) & 3, (s + 1) & 1) + tid;
      v[q][0] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v[q][1] = __hip_atomic_load(src + 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  auto fetch = [&](int s) {
    const unsigned epoch = (unsigned)(s + 1);
    auto tags_ok = [&]() {
      bool ok = true;
#pragma unroll
      for (int q = 0; q < 3; ++q) ok &= ((unsigned)(v[q][0] >> 32) == epoch) & ((unsigned)(v[q][1] >> 32) == epoch);
      return __all(ok);
    };
    if (!tags_ok() && !dead) {
      unsigned spins = 0;
#pragma clang loop unroll(disable)
      for (;;) {
        asm volatile("" : "+s"(spins));
        if (++spins > spin_limit) { dead = true; if (lane == 0) lp_report_timeout(status, sticky, 1u); break; }
        __builtin_amdgcn_s_sleep(1);
        fetch_issue(s);
        if (tags_ok()) break;
      }
    }
    char* dst = hbuf + (((s + 1) & 1) * LP_DOCS) * HROW;
    const int qd = tid >> 4, dd = tid & 15;              // granule tid (+ 256 k2): unit quad qd, document dd
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int part = (p + 1 + q) & 3;
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2)
        *reinterpret_cast<unsigned*>(dst + dd * HROW + (part * HQ + qd * 4 + k2 * 2) * 2) = (unsigned)v[q][k2];
    }
  };

  load_x(std::integral_constant<int, 0>{}, 0);
  load_x(std::integral_constant<int, 1>{}, 1);
  __syncthreads();
  float bia[4][4];
#pragma unroll
  for (int gt = 0; gt < 4; ++gt) {
    const float4 b4 = *reinterpret_cast<const float4*>(blds + gt * HQ + ul);
    bia[gt][0] = b4.x; bia[gt][1] = b4.y; bia[gt][2] = b4.z; bia[gt][3] = b4.w;
  }

  auto step = [&](auto parc, int s) {
    constexpr int PAR = decltype(parc)::value;
    const char* hcur = hbuf + (PAR * LP_DOCS) * HROW;
    char* hnext = hbuf + ((PAR ^ 1) * LP_DOCS) * HROW;
    bf16x8 hf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) hf[ks] = *reinterpret_cast<const bf16x8*>(hcur + doc * HROW + (ks * 32 + 8 * g4) * 2);
    f32x4 acc[4];
#pragma unroll
    for (int gt = 0; gt < 4; ++gt) acc[gt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int gt = 0; gt < 4; ++gt) acc[gt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[gt][ks], hf[ks], acc[gt], 0, 0, 0);
    const bool active = s < len;
    float gi[4], gf[4], gg[4], go[4];
    {
      float xi[4], xf[4], xg[4], xo[4], hn[4];
      upk4(xb[PAR][0], xi); upk4(xb[PAR][1], xf); upk4(xb[PAR][2], xg); upk4(xb[PAR][3], xo);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        gi[r] = fsig2((xi[r] + bia[0][r]) + acc[0][r]);
        gf[r] = fsig2((xf[r] + bia[1][r]) + acc[1][r]);
        gg[r] = ftanh2((xg[r] + bia[2][r]) + acc[2][r]);
        go[r] = fsig2((xo[r] + bia[3][r]) + acc[3][r]);
        const float cn = gf[r] * c[r] + gi[r] * gg[r];
        hn[r] = go[r] * ftanh2(cn);
        c[r] = active ? cn : c[r];
      }
      const uint2 hnew = pk4(hn);
      hq.x = active ? hnew.x : hq.x;
      hq.y = active ? hnew.y : hq.y;
    }
    // 1) post the own quarter of the new h: LDS for this workgroup, tagged granules for the three others
    *reinterpret_cast<uint2*>(hnext + doc * HROW + u * 2) = hq;
    {
      u64* mine = xarea(p, PAR ^ 1);
      const u64 tag = (u64)(unsigned)(s + 1) << 32;
      __hip_atomic_store(mine + (ul >> 2) * 16 + doc, tag | hq.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(mine + 256 + (ul >> 2) * 16 + doc, tag | hq.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // 2) poll the partners' quarters, 3) THEN request the x rows two steps ahead: the vector-memory counter retires in order, so a re-poll
    //    (younger loads) also waits for every older load -- requested before the poll, the x rows had to land within the poll itself
    //    (fine out of the Infinity Cache, not from HBM: 64 x 512); requested after it they have a whole step
    fetch_issue(s);
    fetch(s);
    load_x(parc, s + 2);
    // 4) saved state for the backward pass, last (ended documents store to the dump area)
    {
      const int t = (d == 0) ? s : (len - 1 - s);
      bf16_t* gp = active ? gsave + (size_t)s * gstep : reinterpret_cast<bf16_t*>(dump + tid * 16);
      float* cptr = active ? csave + (size_t)s * cstep : reinterpret_cast<float*>(dump + tid * 16);
      bf16_t* optr = active ? out + orow0 + (size_t)t * ldo : reinterpret_cast<bf16_t*>(dump + tid * 16);
      const int gs = active ? HQ : 0;
      *reinterpret_cast<uint2*>(gp) = pk4(gi);
      *reinterpret_cast<uint2*>(gp + gs) = pk4(gf);
      *reinterpret_cast<uint2*>(gp + 2 * gs) = pk4(gg);
      *reinterpret_cast<uint2*>(gp + 3 * gs) = pk4(go);
      *reinterpret_cast<float4*>(cptr) = make_float4(c[0], c[1], c[2], c[3]);
      *reinterpret_cast<uint2*>(optr) = hq;
    }
    __syncthreads();
  };
  __builtin_amdgcn_s_waitcnt(0);
  for (int s = 0; s < maxlen; s += 2) {
    step(std::integral_constant<int, 0>{}, s);
    if (s + 1 < maxlen) step(std::integral_constant<int, 1>{}, s + 1);
  }
  // rows >= len are exactly zero
  if (bdoc < B)
    for (int t = len + g4; t < L; t += 4)
      for (int e = 0; e < 16; e += 4) *reinterpret_cast<uint2*>(out + ((size_t)bdoc * L + t) * ldo + (size_t)d * H + p * HQ + w * 16 + e) = make_uint2(0, 0);
}

static int lstm_quad_fwd_launch(hipStream_t st, int B, int L, int H, int ndir, const void* xproj, const bf16_t* wpk, const float* b_hh, const int32_t* lengths,
                                void* out, void* gates, float* cells, u64* xch, unsigned* status) {
  constexpr int KS = 8;
  const size_t lds = (size_t)2 * LP_DOCS * (H + 8) * 2 + (size_t)4 * (H / 4) * sizeof(float);
  auto k = lstm_fwd_quad_kernel<KS>;
  const int max_quads = std::max(1, std::min(LQ_MAX_QUADS, g_max_pairs / 2));
  const int docs_per_launch = std::max(1, max_quads / ndir) * LP_DOCS;
  for (int b0 = 0; b0 < B; b0 += docs_per_launch) {
    const int bc = std::min(docs_per_launch, B - b0);
    const size_t r0 = (size_t)b0 * L;
    if (b0 > 0 && hipMemsetAsync(xch, 0, pair_xbytes(bc, ndir) + 256, st) != hipSuccess) { mts_set_error("lstm_quad: memset failed"); return MTS_ERR_LAUNCH; }
    const int nquads = ceil_div(bc, LP_DOCS) * ndir;
    hipLaunchKernelGGL(k, dim3(32 * ceil_div(nquads, 8)), dim3(256), lds, st, bc, L, ndir, nquads, (const bf16_t*)xproj + r0 * ndir * 4 * H, wpk, b_hh,
                       lengths ? lengths + b0 : nullptr, (bf16_t*)out + r0 * ndir * H, (bf16_t*)gates + r0 * ndir * 4 * H, cells + r0 * ndir * H, xch, status,
                       (char*)status + 256, g_spin_limit, g_sticky_dev);
  }
  MTS_LAUNCH_CHECK("mts_lstm_fwd(quad)");
  return MTS_OK;
}

// =====================================================================================================
// CU-quad form of the backward recurrence (mirror of lstm_bwd_pair_kernel): part p owns the 256 gate columns of its 64 units and
// reduces dh over them for ALL 256 units -- four 16-unit tiles per wave: its own part's (kept) and one for each partner (posted as
// bf16 granules); a lane's dh = own partial + the three partners' partials for its (document, unit quad), polled into registers.
// 32 MFMAs per wave and step as in the pair form, but ONE wave per SIMD and all 32 weight fragments in registers.
// Packed weights wpkQ[d][p][w][tt][ks][lane] = 8 bf16: tile tt = 0 -> units of part p, tt = q + 1 -> part (p + 1 + q) & 3; A-fragment
// row = unit tt-part * 64 + w * 16 + (lane & 15), k = own gate column kk = ks * 32 + 8 * (lane >> 4) + e = gate (kk / 64), unit p * 64 + kk % 64.
// =====================================================================================================
__global__ void lstm_pack_weights_TQ_kernel(const float* __restrict__ w_hh, bf16_t* __restrict__ wpk, int H, int ndir) {
  const int HQ = H / 4, KT = 4 * HQ / 32;                             // 8 k-steps over the 256 own gate columns
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;          // over d, p, w, tt, ks, lane
  const size_t total = (size_t)ndir * 4 * 4 * 4 * KT * 64;
  if (idx >= total) return;
  const int lane = idx % 64;
  size_t r = idx / 64;
  const int ks = r % KT; r /= KT;
  const int tt = r % 4; r /= 4;
  const int w = r % 4; r /= 4;
  const int p = r % 4; r /= 4;
  const int d = (int)r;
  const int part = (tt == 0) ? p : ((p + tt) & 3);
  const int j = part * HQ + w * 16 + (lane & 15);
  bf16_t* dst = wpk + idx * 8;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int kk = ks * 32 + 8 * (lane >> 4) + e;
    const int n = (kk / HQ) * H + p * HQ + (kk % HQ);
    dst[e] = (bf16_t)w_hh[((size_t)d * 4 * H + n) * H + j];
  }
}

template <int KS>
__global__ __launch_bounds__(256, 1) void lstm_bwd_quad_kernel(int B, int L, int ndir, int nquads, const bf16_t* __restrict__ wpkQ,
                                                               const int32_t* __restrict__ lengths, const bf16_t* __restrict__ gates,
                                                               const float* __restrict__ cells, const bf16_t* __restrict__ dout,
                                                               bf16_t* __restrict__ dxproj, u64* __restrict__ xch, unsigned* __restrict__ status,
                                                               char* __restrict__ dump, unsigned spin_limit, unsigned* __restrict__ sticky) {
  static_assert(KS == 8, "H = 256");
  constexpr int H = KS * 32, HQ = H / 4, NT = 256;
  constexpr int KT = 4 * HQ / 32;                        // 8 k-steps over the 256 own gate columns
  constexpr int DAROW = (4 * HQ + 8) * 2;                // bytes per da row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* dabuf = smem;                                    // [2 (step parity)][16][DAROW] bf16
  const int chunk = blockIdx.x / 32, within = blockIdx.x % 32;
  const int p = within / 8, quad = chunk * 8 + within % 8;
  if (quad >= nquads) return;
  const int gx = quad / ndir, d = quad % ndir;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int doc = lane & 15, g4 = lane >> 4;
  const int ul = w * 16 + 4 * g4;                        // this lane's 4 units, within the quarter
  const int u = p * HQ + ul;                             // ... within H
  const int ldx = ndir * 4 * H, ldo = ndir * H;

  bf16x8 wreg[4][KT];
  {
    const bf16_t* base = wpkQ + ((((size_t)d * 4 + p) * 4 + w) * 4 * KT * 64) * 8;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int ks = 0; ks < KT; ++ks) wreg[tt][ks] = *reinterpret_cast<const bf16x8*>(base + ((size_t)(tt * KT + ks) * 64 + lane) * 8);
  }

  const int bdoc = gx * LP_DOCS + doc;
  const int len = (bdoc < B) ? (lengths ? min(lengths[bdoc], L) : L) : 0;
  int maxlen = len;
#pragma unroll
  for (int off = 1; off < 16; off <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off, 64));

  // exchange areas: xch[quad][src][dst][parity][512 granules]; granule k2 of the position lane `tid` owns on EITHER side sits at
  // k2 * 256 + tid: posts and polls are contiguous 512-byte wave accesses and a poll lands in the consuming lane's registers
  auto xarea = [&](int src, int dst, int par) { return xch + ((((size_t)quad * 4 + src) * 4 + dst) * 2 + par) * 512; };
  for (int i = tid; i < 2 * LP_DOCS * DAROW / 4; i += NT) reinterpret_cast<unsigned*>(smem)[i] = 0u;

  const size_t b0 = (size_t)min(bdoc, B - 1) * L;
  const size_t grow0 = b0 * ldx + (size_t)d * 4 * H + u;
  const size_t orow0 = b0 * ldo + (size_t)d * H + u;
  // saved state in the forward kernel's step-major blocks (lstm_fwd_quad_kernel): step s of this part is one contiguous block, and the
  // previous cell state of step s is the cell block of step s - 1
  const int ndg = min(LP_DOCS, B - gx * LP_DOCS);
  const size_t gstep = (size_t)ndg * 4 * HQ, cstep = (size_t)ndg * HQ;
  const bf16_t* gsave = gates + (size_t)gx * LP_DOCS * L * ldx + ((size_t)(d * 4 + p) * L) * gstep + (size_t)min(doc, ndg - 1) * 4 * HQ + ul;
  const float* csave = cells + (size_t)gx * LP_DOCS * L * ldo + ((size_t)(d * 4 + p) * L) * cstep + (size_t)min(doc, ndg - 1) * HQ + ul;
  auto load_in = [&](int s, PairBwdIn& in) {             // unconditional, clamped (see the pair kernels)
    const int sc = min(max(s, 0), L - 1);
    const int t = min(max((d == 0) ? sc : (len - 1 - sc), 0), L - 1);
    const bf16_t* gp = gsave + (size_t)sc * gstep;
    in.gi = *reinterpret_cast<const uint2*>(gp);
    in.gf = *reinterpret_cast<const uint2*>(gp + HQ);
    in.gg = *reinterpret_cast<const uint2*>(gp + 2 * HQ);
    in.go = *reinterpret_cast<const uint2*>(gp + 3 * HQ);
    in.dov = *reinterpret_cast<const uint2*>(dout + orow0 + (size_t)t * ldo);
    in.ct = *reinterpret_cast<const float4*>(csave + (size_t)sc * cstep);
    in.cp = *reinterpret_cast<const float4*>(csave + (size_t)max(sc - 1, 0) * cstep);
  };
  bool dead = false;
  u64 v[3][2];
  // the partners' partial dh for my quad, produced in their MFMAs of step s; tag = maxlen - s (>= 1)
  auto fetch_issue = [&](int s) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const u64* src = xarea((p + 1 + q) & 3, p, s & 1) + tid;
      v[q][0] = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      v[q][1] = __hip_atomic_load(src + NT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  auto fetch = [&](int s) {
    const unsigned epoch = (unsigned)(maxlen - s);
    auto tags_ok = [&]() {
      bool ok = true;
#pragma unroll
      for (int q = 0; q < 3; ++q) ok &= ((unsigned)(v[q][0] >> 32) == epoch) & ((unsigned)(v[q][1] >> 32) == epoch);
      return __all(ok);
    };
    if (!tags_ok() && !dead) {
      unsigned spins = 0;
#pragma clang loop unroll(disable)
      for (;;) {
        asm volatile("" : "+s"(spins));
        if (++spins > spin_limit) { dead = true; if (lane == 0) lp_report_timeout(status, sticky, 2u); break; }
        __builtin_amdgcn_s_sleep(1);
        fetch_issue(s);
        if (tags_ok()) break;
      }
    }
  };

  float dc[4] = {0.f, 0.f, 0.f, 0.f};
  f32x4 own = (f32x4){0.f, 0.f, 0.f, 0.f};
  PairBwdIn in[2];
  load_in(maxlen - 1, in[0]);
  load_in(maxlen - 2, in[1]);
  __syncthreads();

  auto step = [&](auto parc, int s) {
    constexpr int PAR = decltype(parc)::value;
    char* da = dabuf + PAR * LP_DOCS * DAROW;
    const bool active = s < len;
    const PairBwdIn& cur = in[PAR];
    // ---- dh = own partial + the partners' partials (fixed order) (+ dOut) -> gate pre-activation gradients ---------------------
    float dhv[4] = {own[0], own[1], own[2], own[3]};
    if (s + 1 <= maxlen - 1) {
      fetch(s + 1);
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const unsigned lo = (unsigned)v[q][0], hi = (unsigned)v[q][1];
        dhv[0] += bf16_lo(lo); dhv[1] += bf16_hi(lo); dhv[2] += bf16_lo(hi); dhv[3] += bf16_hi(hi);
      }
    }
    uint2 qv[4];
    {
      float gi[4], gf[4], gg[4], go[4], dov[4], ai[4], af[4], ag[4], ao[4];
      upk4(cur.gi, gi); upk4(cur.gf, gf); upk4(cur.gg, gg); upk4(cur.go, go); upk4(cur.dov, dov);
      const float ct[4] = {cur.ct.x, cur.ct.y, cur.ct.z, cur.ct.w};
      const bool has_prev = s > 0;
      const float cp[4] = {has_prev ? cur.cp.x : 0.f, has_prev ? cur.cp.y : 0.f, has_prev ? cur.cp.z : 0.f, has_prev ? cur.cp.w : 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float tc = ftanh2(ct[r]);
        const float dht = dov[r] + dhv[r];
        const float dct = dc[r] + dht * go[r] * (1.f - tc * tc);
        ai[r] = active ? dct * gg[r] * gi[r] * (1.f - gi[r]) : 0.f;
        af[r] = active ? dct * cp[r] * gf[r] * (1.f - gf[r]) : 0.f;
        ag[r] = active ? dct * gi[r] * (1.f - gg[r] * gg[r]) : 0.f;
        ao[r] = active ? dht * tc * go[r] * (1.f - go[r]) : 0.f;
        dc[r] = active ? dct * gf[r] : dc[r];
      }
      qv[0] = pk4(ai); qv[1] = pk4(af); qv[2] = pk4(ag); qv[3] = pk4(ao);
    }
    char* dr = da + doc * DAROW + ul * 2;
#pragma unroll
    for (int gt = 0; gt < 4; ++gt) *reinterpret_cast<uint2*>(dr + gt * HQ * 2) = qv[gt];
    load_in(s - 2, in[PAR]);
    {
      const int t = (d == 0) ? s : (len - 1 - s);
      bf16_t* dx = active ? dxproj + grow0 + (size_t)t * ldx : reinterpret_cast<bf16_t*>(dump + tid * 16);
#pragma unroll
      for (int gt = 0; gt < 4; ++gt) *reinterpret_cast<uint2*>(dx + gt * H) = qv[gt];
    }
    __syncthreads();
    // ---- partial dh from the own gate columns: tile 0 = my units (kept), tiles 1..3 = the partners' units (posted) ------------
    f32x4 acc[4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) acc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KT; ++ks) {
      const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(da + doc * DAROW + (ks * 32 + 8 * g4) * 2);
#pragma unroll
      for (int tt = 3; tt >= 0; --tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[tt][ks], bfr, acc[tt], 0, 0, 0);
    }
    {
      const u64 tag = (u64)(unsigned)(maxlen - s) << 32;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        u64* dst = xarea(p, (p + 1 + q) & 3, s & 1) + tid;
        const float vv[4] = {acc[q + 1][0], acc[q + 1][1], acc[q + 1][2], acc[q + 1][3]};
        const uint2 pk = pk4(vv);
        __hip_atomic_store(dst, tag | pk.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + NT, tag | pk.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    own = acc[0];
    if (s - 1 >= 0) fetch_issue(s);
  };
  __builtin_amdgcn_s_waitcnt(0);
  for (int s = maxlen - 1; s >= 0; s -= 2) {
    step(std::integral_constant<int, 0>{}, s);
    if (s - 1 >= 0) step(std::integral_constant<int, 1>{}, s - 1);
  }
  if (bdoc < B)
    for (int t = len; t < L; ++t)
#pragma unroll
      for (int gt = 0; gt < 4; ++gt) *reinterpret_cast<uint2*>(dxproj + ((size_t)bdoc * L + t) * ldx + (size_t)d * 4 * H + gt * H + u) = make_uint2(0, 0);
}

// hprev[b, t] = out[b, t_prev] (zero at a document's first processed position and on padded rows), per direction
template <typename T>
__global__ void lstm_hprev_kernel(int B, int L, int H, int ndir, const int32_t* __restrict__ lengths, const T* __restrict__ out, T* __restrict__ hprev) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;          // over rows x ndir x H/4
  const int per_row = ndir * H / 4;
  if (idx >= (size_t)B * L * per_row) return;
  const size_t row = idx / per_row;
  const int c = (int)(idx % per_row) * 4, d = c / H;
  const int b = (int)(row / L), t = (int)(row % L);
  const int len = lengths ? min(lengths[b], L) : L;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  const int tp = (d == 0) ? t - 1 : t + 1;
  if (t < len && tp >= 0 && tp < len) load4<T>(out + ((size_t)b * L + tp) * ndir * H + c, v);
  store4<T>(hprev + row * ndir * H + c, v);
}

// h_{t-1} of every (document, step) as a matrix [B*L, ndir*H] -- the second operand of the dW_hh GEMMs (the CU-pair / CU-quad recurrences do
// not produce it themselves)
int mts_lstm_pair_hprev(hipStream_t st, int dtype, int B, int L, int H, int ndir, const int32_t* lengths, const void* out, void* hprev) {
  const size_t rows4 = (size_t)B * L * ndir * H / 4;
  if (dtype == MTS_F32)
    hipLaunchKernelGGL(lstm_hprev_kernel<float>, dim3((unsigned)((rows4 + 255) / 256)), dim3(256), 0, st, B, L, H, ndir, lengths, (const float*)out, (float*)hprev);
  else
    hipLaunchKernelGGL(lstm_hprev_kernel<bf16_t>, dim3((unsigned)((rows4 + 255) / 256)), dim3(256), 0, st, B, L, H, ndir, lengths, (const bf16_t*)out,
                       (bf16_t*)hprev);
  MTS_LAUNCH_CHECK("mts_lstm_bwd(hprev)");
  return MTS_OK;
}

int mts_lstm_pair_bwd(hipStream_t st, int B, int L, int H, int ndir, const float* w_hh, const int32_t* lengths, const void* out, const void* gates,
                      const float* cells, const void* dout, void* dxproj, void* hprev, void* ws) {
  constexpr int KS = 8;
  bf16_t* wpk = (bf16_t*)ws;
  u64* xch = (u64*)((char*)ws + pair_wbytes(H, ndir));
  unsigned* status = (unsigned*)((char*)xch + pair_xbytes(B, ndir));
  const int HH = H / 2, KT = 4 * HH / 32;
  const bool quad = g_lstm_parts == 4;
  if (quad) {
    const size_t total = (size_t)ndir * 4 * 4 * 4 * 8 * 64;
    hipLaunchKernelGGL(lstm_pack_weights_TQ_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w_hh, wpk, H, ndir);
  } else {
    const size_t total = (size_t)ndir * 2 * KS * 2 * KT * 64;
    hipLaunchKernelGGL(lstm_pack_weights_T_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w_hh, wpk, H, ndir);
  }
  if (hipMemsetAsync(xch, 0, pair_xbytes(B, ndir) + 256, st) != hipSuccess) { mts_set_error("lstm_pair: memset failed"); return MTS_ERR_LAUNCH; }
  const size_t rows4 = (size_t)B * L * ndir * H / 4;
  if (hprev)      // (NULL: the caller launches mts_lstm_pair_hprev itself, off the recurrence's stream: mts_lstm_bwd_recurrence / _whh)
    hipLaunchKernelGGL(lstm_hprev_kernel<bf16_t>, dim3((unsigned)((rows4 + 255) / 256)), dim3(256), 0, st, B, L, H, ndir, lengths, (const bf16_t*)out,
                       (bf16_t*)hprev);
  if (quad) {
    if (int rc = lp_ensure_sticky()) return rc;
    const size_t ldsq = (size_t)2 * LP_DOCS * ((4 * (H / 4) + 8) * 2);
    const int max_quads = std::max(1, std::min(LQ_MAX_QUADS, g_max_pairs / 2));
    const int docs_per_launch = std::max(1, max_quads / ndir) * LP_DOCS;
    for (int b0 = 0; b0 < B; b0 += docs_per_launch) {
      const int bc = std::min(docs_per_launch, B - b0);
      const size_t r0 = (size_t)b0 * L;
      if (b0 > 0 && hipMemsetAsync(xch, 0, pair_xbytes(bc, ndir) + 256, st) != hipSuccess) { mts_set_error("lstm_quad: memset failed"); return MTS_ERR_LAUNCH; }
      const int nquads = ceil_div(bc, LP_DOCS) * ndir;
      hipLaunchKernelGGL(lstm_bwd_quad_kernel<KS>, dim3(32 * ceil_div(nquads, 8)), dim3(256), ldsq, st, bc, L, ndir, nquads, (const bf16_t*)wpk,
                         lengths ? lengths + b0 : nullptr, (const bf16_t*)gates + r0 * ndir * 4 * H, cells + r0 * ndir * H, (const bf16_t*)dout + r0 * ndir * H,
                         (bf16_t*)dxproj + r0 * ndir * 4 * H, xch, status, (char*)status + 256, g_spin_limit, g_sticky_dev);
    }
    MTS_LAUNCH_CHECK("mts_lstm_bwd(quad)");
    return MTS_OK;
  }
  const size_t lds = (size_t)2 * LP_DOCS * ((4 * HH + 8) * 2) + (size_t)KS * (KT / 2) * 1024;
  auto k = lstm_bwd_pair_kernel<KS>;
  static std::atomic<bool> attr{false};
  if (!attr) {
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
      mts_set_error("lstm_pair_bwd: cannot reserve %zu bytes of LDS", lds);
      return MTS_ERR_LAUNCH;
    }
    attr = true;
  }
  if (int rc = lp_ensure_sticky()) return rc;
  const int docs_per_launch = std::max(1, g_max_pairs / ndir) * LP_DOCS * LP_GROUPS;
  for (int b0 = 0; b0 < B; b0 += docs_per_launch) {
    const int bc = std::min(docs_per_launch, B - b0);
    const size_t r0 = (size_t)b0 * L;
    if (b0 > 0 && hipMemsetAsync(xch, 0, pair_xbytes(bc, ndir) + 256, st) != hipSuccess) { mts_set_error("lstm_pair: memset failed"); return MTS_ERR_LAUNCH; }
    const int npairs = ceil_div(bc, LP_DOCS * LP_GROUPS) * ndir;
    hipLaunchKernelGGL(k, dim3(16 * ceil_div(npairs, 8)), dim3(KS * 64), lds, st, bc, L, ndir, npairs, (const bf16_t*)wpk, lengths ? lengths + b0 : nullptr,
                       (const bf16_t*)gates + r0 * ndir * 4 * H, cells + r0 * ndir * H, (const bf16_t*)dout + r0 * ndir * H,
                       (bf16_t*)dxproj + r0 * ndir * 4 * H, xch, status, (char*)status + 256, g_spin_limit, g_sticky_dev);
  }
  MTS_LAUNCH_CHECK("mts_lstm_bwd(pair)");
  return MTS_OK;
}

// =====================================================================================================
// CU-quad recurrences in FP32 (parity mode, H = 256).  The drop-in classes default to fp32; the generic fp32 recurrence
// (lstm.hip: one workgroup per 8 documents streaming the whole W_hh through L2 in every time step) took 54 us per dependent
// step -- 58 ms per BiLSTM training step at 64 x 256 x 1792, the paper's own configuration.  Same decomposition as the bf16
// quad kernels above: four workgroups of 64 units per (16 documents, direction), one wave per SIMD, and W_hh in fp32 is exactly
// 256 registers per lane of those waves (4 gate tiles x 64 k-steps of v_mfma_f32_16x16x4_f32).  h / da live in LDS in fp32 and are
// read 16 bytes per lane: the contraction index is PERMUTED so that one ds_read_b128 feeds four k-steps (k-step 4m + j takes
// k = 16 m + 4 g + j from lane group g; the packed weights use the same map), i.e. the sum over k is taken in a different --
// fixed -- order than the oracle's: fp32 reassociation, ~1e-7 relative.  Exchange granules carry one fp32 each.  Transcendentals are
// the generic kernel's (expf / tanhf), not the bf16 kernels' fast forms.
// =====================================================================================================
__global__ void lstm_pack_weights_f32_kernel(const float* __restrict__ w_hh, float* __restrict__ wpk, int H, int ndir, int transposed) {
  // forward  (transposed = 0): wpk[d][p][w][gt][ks][lane] = W[gt*H + p*64 + w*16 + (lane&15)][k],  k = 16 (ks>>2) + 4 (lane>>4) + (ks&3)
  // backward (transposed = 1): wpk[d][p][w][tt][ks][lane] = W[n][part(tt)*64 + w*16 + (lane&15)],  n = (kk/64)*H + p*64 + kk%64,
  //                            kk = 16 (ks>>2) + 4 (lane>>4) + (ks&3),  part(0) = p, part(q+1) = (p+1+q) & 3
  const int HQ = H / 4;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;          // over d, p, w, t, ks, lane
  const size_t total = (size_t)ndir * 4 * 4 * 4 * 64 * 64;
  if (idx >= total) return;
  const int lane = idx % 64;
  size_t r = idx / 64;
  const int ks = r % 64; r /= 64;
  const int t = r % 4; r /= 4;
  const int w = r % 4; r /= 4;
  const int p = r % 4; r /= 4;
  const int d = (int)r;
  const int kk = 16 * (ks >> 2) + 4 * (lane >> 4) + (ks & 3);
  float v;
  if (!transposed) {
    const int col = t * H + p * HQ + w * 16 + (lane & 15);
    v = w_hh[((size_t)d * 4 * H + col) * H + kk];
  } else {
    const int part = (t == 0) ? p : ((p + t) & 3);
    const int j = part * HQ + w * 16 + (lane & 15);
    const int n = (kk / HQ) * H + p * HQ + (kk % HQ);
    v = w_hh[((size_t)d * 4 * H + n) * H + j];
  }
  wpk[idx] = v;
}

__global__ __launch_bounds__(256, 1) void lstm_fwd_quad_f32_kernel(int B, int L, int ndir, int nquads, const float* __restrict__ xproj,
                                                                   const float* __restrict__ wpk, const float* __restrict__ bhh,
                                                                   const int32_t* __restrict__ lengths, float* __restrict__ out,
                                                                   float* __restrict__ gates, float* __restrict__ cells, u64* __restrict__ xch,
                                                                   unsigned* __restrict__ status, char* __restrict__ dump, unsigned spin_limit,
                                                                   unsigned* __restrict__ sticky) {
  constexpr int H = 256, HQ = 64, NT = 256, HROW = (H + 4) * 4;      // 1040-byte rows: a 16-lane group reads 64 distinct banks
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* hbuf = smem;                                     // [parity][16][HROW] fp32
  const int chunk = blockIdx.x / 32, within = blockIdx.x % 32;
  const int p = within / 8, quad = chunk * 8 + within % 8;
  if (quad >= nquads) return;
  const int gx = quad / ndir, d = quad % ndir;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int doc = lane & 15, g4 = lane >> 4;
  const int ul = w * 16 + 4 * g4, u = p * HQ + ul;
  const int ldx = ndir * 4 * H, ldo = ndir * H;

  float wreg[4][64];
  {
    const float* base = wpk + ((((size_t)d * 4 + p) * 4 + w) * 4 * 64 * 64);
#pragma unroll
    for (int gt = 0; gt < 4; ++gt)
#pragma unroll
      for (int ks = 0; ks < 64; ++ks) wreg[gt][ks] = base[(size_t)(gt * 64 + ks) * 64 + lane];
  }
  float bia[4][4];
#pragma unroll
  for (int gt = 0; gt < 4; ++gt)
#pragma unroll
    for (int r = 0; r < 4; ++r) bia[gt][r] = bhh ? bhh[(size_t)d * 4 * H + gt * H + u + r] : 0.f;

  const int bdoc = gx * LP_DOCS + doc;
  const int len = (bdoc < B) ? (lengths ? min(lengths[bdoc], L) : L) : 0;
  int maxlen = len;
#pragma unroll
  for (int off = 1; off < 16; off <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off, 64));

  // exchange areas: xch[quad][part][parity][1024 granules]; granule k2 (unit ul + k2) of (document, unit quad q = ul / 4) at k2 * 256 + q * 16 + doc
  auto xarea = [&](int part, int par) { return xch + (((size_t)quad * 4 + part) * 2 + par) * 1024; };

  float c[4] = {0.f, 0.f, 0.f, 0.f};
  float hq[4] = {0.f, 0.f, 0.f, 0.f};
  float4 xb[2][4];
  for (int i = tid; i < 2 * LP_DOCS * HROW / 16; i += NT) reinterpret_cast<uint4*>(hbuf)[i] = make_uint4(0, 0, 0, 0);

  const size_t b0 = (size_t)min(bdoc, B - 1) * L;
  const size_t grow0 = b0 * ldx + (size_t)d * 4 * H + u, orow0 = b0 * ldo + (size_t)d * H + u;
  const float* xbase = xproj + grow0;
  auto load_x = [&](auto parc, int s) {                  // unconditional, clamped (see the bf16 kernels)
    constexpr int PAR = decltype(parc)::value;
    const int t = (d == 0) ? s : (len - 1 - s);
    const float* src = xbase + (size_t)min(max(t, 0), L - 1) * ldx;
#pragma unroll
    for (int gt = 0; gt < 4; ++gt) xb[PAR][gt] = *reinterpret_cast<const float4*>(src + gt * H);
  };
  bool dead = false;
  u64 v[3][4];
  auto fetch_issue = [&](int s) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const u64* src = xarea((p + 1 + q) & 3, (s + 1) & 1) + tid;
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) v[q][k2] = __hip_atomic_load(src + 256 * k2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  auto fetch = [&](int s) {
    const unsigned epoch = (unsigned)(s + 1);
    auto tags_ok = [&]() {
      bool ok = true;
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) ok &= ((unsigned)(v[q][k2] >> 32) == epoch);
      return __all(ok);
    };
    if (!tags_ok() && !dead) {
      unsigned spins = 0;
#pragma clang loop unroll(disable)
      for (;;) {
        asm volatile("" : "+s"(spins));
        if (++spins > spin_limit) { dead = true; if (lane == 0) lp_report_timeout(status, sticky, 1u); break; }
        __builtin_amdgcn_s_sleep(1);
        fetch_issue(s);
        if (tags_ok()) break;
      }
    }
    char* dst = hbuf + (((s + 1) & 1) * LP_DOCS) * HROW;
    const int qd = tid >> 4, dd = tid & 15;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int part = (p + 1 + q) & 3;
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) *reinterpret_cast<unsigned*>(dst + dd * HROW + (part * HQ + qd * 4 + k2) * 4) = (unsigned)v[q][k2];
    }
  };

  load_x(std::integral_constant<int, 0>{}, 0);
  load_x(std::integral_constant<int, 1>{}, 1);
  __syncthreads();

  auto step = [&](auto parc, int s) {
    constexpr int PAR = decltype(parc)::value;
    const char* hcur = hbuf + (PAR * LP_DOCS) * HROW;
    char* hnext = hbuf + ((PAR ^ 1) * LP_DOCS) * HROW;
    f32x4 acc[4];
#pragma unroll
    for (int gt = 0; gt < 4; ++gt) acc[gt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const float4 hf = *reinterpret_cast<const float4*>(hcur + doc * HROW + (16 * m + 4 * g4) * 4);
      const float hv[4] = {hf.x, hf.y, hf.z, hf.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int gt = 0; gt < 4; ++gt) acc[gt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[gt][4 * m + j], hv[j], acc[gt], 0, 0, 0);
    }
    const bool active = s < len;
    float gi[4], gf[4], gg[4], go[4];
    {
      const float xi[4] = {xb[PAR][0].x, xb[PAR][0].y, xb[PAR][0].z, xb[PAR][0].w}, xf[4] = {xb[PAR][1].x, xb[PAR][1].y, xb[PAR][1].z, xb[PAR][1].w};
      const float xg[4] = {xb[PAR][2].x, xb[PAR][2].y, xb[PAR][2].z, xb[PAR][2].w}, xo[4] = {xb[PAR][3].x, xb[PAR][3].y, xb[PAR][3].z, xb[PAR][3].w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        gi[r] = sigmoid_f((xi[r] + bia[0][r]) + acc[0][r]);
        gf[r] = sigmoid_f((xf[r] + bia[1][r]) + acc[1][r]);
        gg[r] = tanhf((xg[r] + bia[2][r]) + acc[2][r]);
        go[r] = sigmoid_f((xo[r] + bia[3][r]) + acc[3][r]);
        const float cn = gf[r] * c[r] + gi[r] * gg[r];
        const float hn = go[r] * tanhf(cn);
        c[r] = active ? cn : c[r];
        hq[r] = active ? hn : hq[r];
      }
    }
    *reinterpret_cast<float4*>(hnext + doc * HROW + u * 4) = make_float4(hq[0], hq[1], hq[2], hq[3]);
    {
      u64* mine = xarea(p, PAR ^ 1) + (ul >> 2) * 16 + doc;
      const u64 tag = (u64)(unsigned)(s + 1) << 32;
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) __hip_atomic_store(mine + 256 * k2, tag | (u64)__float_as_uint(hq[k2]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    fetch_issue(s);
    fetch(s);
    load_x(parc, s + 2);                                   // after the poll (see the bf16 kernel)
    {
      const int t = (d == 0) ? s : (len - 1 - s);
      float* gp = active ? gates + grow0 + (size_t)t * ldx : reinterpret_cast<float*>(dump + tid * 16);
      float* cptr = active ? cells + orow0 + (size_t)t * ldo : reinterpret_cast<float*>(dump + tid * 16);
      float* optr = active ? out + orow0 + (size_t)t * ldo : reinterpret_cast<float*>(dump + tid * 16);
      const int gstep = active ? H : 0;
      *reinterpret_cast<float4*>(gp) = make_float4(gi[0], gi[1], gi[2], gi[3]);
      *reinterpret_cast<float4*>(gp + gstep) = make_float4(gf[0], gf[1], gf[2], gf[3]);
      *reinterpret_cast<float4*>(gp + 2 * gstep) = make_float4(gg[0], gg[1], gg[2], gg[3]);
      *reinterpret_cast<float4*>(gp + 3 * gstep) = make_float4(go[0], go[1], go[2], go[3]);
      *reinterpret_cast<float4*>(cptr) = make_float4(c[0], c[1], c[2], c[3]);
      *reinterpret_cast<float4*>(optr) = make_float4(hq[0], hq[1], hq[2], hq[3]);
    }
    __syncthreads();
  };
  __builtin_amdgcn_s_waitcnt(0);
  for (int s = 0; s < maxlen; s += 2) {
    step(std::integral_constant<int, 0>{}, s);
    if (s + 1 < maxlen) step(std::integral_constant<int, 1>{}, s + 1);
  }
  if (bdoc < B)
    for (int t = len + g4; t < L; t += 4)
      for (int e = 0; e < 16; e += 4) *reinterpret_cast<float4*>(out + ((size_t)bdoc * L + t) * ldo + (size_t)d * H + p * HQ + w * 16 + e) = make_float4(0.f, 0.f, 0.f, 0.f);
}

struct QuadBwdInF { float4 gi, gf, gg, go, dov, ct, cp; };

__global__ __launch_bounds__(256, 1) void lstm_bwd_quad_f32_kernel(int B, int L, int ndir, int nquads, const float* __restrict__ wpkT,
                                                                   const int32_t* __restrict__ lengths, const float* __restrict__ gates,
                                                                   const float* __restrict__ cells, const float* __restrict__ dout,
                                                                   float* __restrict__ dxproj, u64* __restrict__ xch, unsigned* __restrict__ status,
                                                                   char* __restrict__ dump, unsigned spin_limit, unsigned* __restrict__ sticky) {
  constexpr int H = 256, HQ = 64, NT = 256;
  constexpr int DAROW = (4 * HQ + 4) * 4;                // bytes per da row (fp32)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* dabuf = smem;                                    // [2][16][DAROW]
  const int chunk = blockIdx.x / 32, within = blockIdx.x % 32;
  const int p = within / 8, quad = chunk * 8 + within % 8;
  if (quad >= nquads) return;
  const int gx = quad / ndir, d = quad % ndir;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int doc = lane & 15, g4 = lane >> 4;
  const int ul = w * 16 + 4 * g4, u = p * HQ + ul;
  const int ldx = ndir * 4 * H, ldo = ndir * H;

  float wreg[4][64];
  {
    const float* base = wpkT + ((((size_t)d * 4 + p) * 4 + w) * 4 * 64 * 64);
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
      for (int ks = 0; ks < 64; ++ks) wreg[tt][ks] = base[(size_t)(tt * 64 + ks) * 64 + lane];
  }
  const int bdoc = gx * LP_DOCS + doc;
  const int len = (bdoc < B) ? (lengths ? min(lengths[bdoc], L) : L) : 0;
  int maxlen = len;
#pragma unroll
  for (int off = 1; off < 16; off <<= 1) maxlen = max(maxlen, __shfl_xor(maxlen, off, 64));

  // exchange areas: xch[quad][src][dst][parity][1024 granules]; granule k2 of the position lane `tid` owns on either side at k2 * 256 + tid
  auto xarea = [&](int src, int dst, int par) { return xch + ((((size_t)quad * 4 + src) * 4 + dst) * 2 + par) * 1024; };
  for (int i = tid; i < 2 * LP_DOCS * DAROW / 4; i += NT) reinterpret_cast<unsigned*>(smem)[i] = 0u;

  const size_t b0 = (size_t)min(bdoc, B - 1) * L;
  const size_t grow0 = b0 * ldx + (size_t)d * 4 * H + u, orow0 = b0 * ldo + (size_t)d * H + u;
  auto load_in = [&](int s, QuadBwdInF& in) {
    const int sc = max(s, 0);
    const int t = min(max((d == 0) ? sc : (len - 1 - sc), 0), L - 1);
    const int tp = min(max((d == 0) ? t - 1 : t + 1, 0), L - 1);
    const float* gp = gates + grow0 + (size_t)t * ldx;
    in.gi = *reinterpret_cast<const float4*>(gp);
    in.gf = *reinterpret_cast<const float4*>(gp + H);
    in.gg = *reinterpret_cast<const float4*>(gp + 2 * H);
    in.go = *reinterpret_cast<const float4*>(gp + 3 * H);
    in.dov = *reinterpret_cast<const float4*>(dout + orow0 + (size_t)t * ldo);
    in.ct = *reinterpret_cast<const float4*>(cells + orow0 + (size_t)t * ldo);
    in.cp = *reinterpret_cast<const float4*>(cells + orow0 + (size_t)tp * ldo);
  };
  bool dead = false;
  u64 v[3][4];
  auto fetch_issue = [&](int s) {
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const u64* src = xarea((p + 1 + q) & 3, p, s & 1) + tid;
#pragma unroll
      for (int k2 = 0; k2 < 4; ++k2) v[q][k2] = __hip_atomic_load(src + NT * k2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  auto fetch = [&](int s) {
    const unsigned epoch = (unsigned)(maxlen - s);
    auto tags_ok = [&]() {
      bool ok = true;
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) ok &= ((unsigned)(v[q][k2] >> 32) == epoch);
      return __all(ok);
    };
    if (!tags_ok() && !dead) {
      unsigned spins = 0;
#pragma clang loop unroll(disable)
      for (;;) {
        asm volatile("" : "+s"(spins));
        if (++spins > spin_limit) { dead = true; if (lane == 0) lp_report_timeout(status, sticky, 2u); break; }
        __builtin_amdgcn_s_sleep(1);
        fetch_issue(s);
        if (tags_ok()) break;
      }
    }
  };

  float dc[4] = {0.f, 0.f, 0.f, 0.f};
  f32x4 own = (f32x4){0.f, 0.f, 0.f, 0.f};
  QuadBwdInF in[2];
  load_in(maxlen - 1, in[0]);
  load_in(maxlen - 2, in[1]);
  __syncthreads();

  auto step = [&](auto parc, int s) {
    constexpr int PAR = decltype(parc)::value;
    char* da = dabuf + PAR * LP_DOCS * DAROW;
    const bool active = s < len;
    const QuadBwdInF& cur = in[PAR];
    float dhv[4] = {own[0], own[1], own[2], own[3]};
    if (s + 1 <= maxlen - 1) {
      fetch(s + 1);
#pragma unroll
      for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) dhv[k2] += __uint_as_float((unsigned)v[q][k2]);
    }
    float ai[4], af[4], ag[4], ao[4];
    {
      const float gi[4] = {cur.gi.x, cur.gi.y, cur.gi.z, cur.gi.w}, gf[4] = {cur.gf.x, cur.gf.y, cur.gf.z, cur.gf.w};
      const float gg[4] = {cur.gg.x, cur.gg.y, cur.gg.z, cur.gg.w}, go[4] = {cur.go.x, cur.go.y, cur.go.z, cur.go.w};
      const float dov[4] = {cur.dov.x, cur.dov.y, cur.dov.z, cur.dov.w}, ct[4] = {cur.ct.x, cur.ct.y, cur.ct.z, cur.ct.w};
      const bool has_prev = s > 0;
      const float cp[4] = {has_prev ? cur.cp.x : 0.f, has_prev ? cur.cp.y : 0.f, has_prev ? cur.cp.z : 0.f, has_prev ? cur.cp.w : 0.f};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float tc = tanhf(ct[r]);
        const float dht = dov[r] + dhv[r];
        const float dct = dc[r] + dht * go[r] * (1.f - tc * tc);
        ai[r] = active ? dct * gg[r] * gi[r] * (1.f - gi[r]) : 0.f;
        af[r] = active ? dct * cp[r] * gf[r] * (1.f - gf[r]) : 0.f;
        ag[r] = active ? dct * gi[r] * (1.f - gg[r] * gg[r]) : 0.f;
        ao[r] = active ? dht * tc * go[r] * (1.f - go[r]) : 0.f;
        dc[r] = active ? dct * gf[r] : dc[r];
      }
    }
    char* dr = da + doc * DAROW + ul * 4;
    *reinterpret_cast<float4*>(dr) = make_float4(ai[0], ai[1], ai[2], ai[3]);
    *reinterpret_cast<float4*>(dr + HQ * 4) = make_float4(af[0], af[1], af[2], af[3]);
    *reinterpret_cast<float4*>(dr + 2 * HQ * 4) = make_float4(ag[0], ag[1], ag[2], ag[3]);
    *reinterpret_cast<float4*>(dr + 3 * HQ * 4) = make_float4(ao[0], ao[1], ao[2], ao[3]);
    load_in(s - 2, in[PAR]);
    {
      const int t = (d == 0) ? s : (len - 1 - s);
      float* dx = active ? dxproj + grow0 + (size_t)t * ldx : reinterpret_cast<float*>(dump + tid * 16);
      const int gstep = active ? H : 0;
      *reinterpret_cast<float4*>(dx) = make_float4(ai[0], ai[1], ai[2], ai[3]);
      *reinterpret_cast<float4*>(dx + gstep) = make_float4(af[0], af[1], af[2], af[3]);
      *reinterpret_cast<float4*>(dx + 2 * gstep) = make_float4(ag[0], ag[1], ag[2], ag[3]);
      *reinterpret_cast<float4*>(dx + 3 * gstep) = make_float4(ao[0], ao[1], ao[2], ao[3]);
    }
    __syncthreads();
    f32x4 acc[4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) acc[tt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      const float4 bf = *reinterpret_cast<const float4*>(da + doc * DAROW + (16 * m + 4 * g4) * 4);
      const float bv[4] = {bf.x, bf.y, bf.z, bf.w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int tt = 3; tt >= 0; --tt) acc[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[tt][4 * m + j], bv[j], acc[tt], 0, 0, 0);
    }
    {
      const u64 tag = (u64)(unsigned)(maxlen - s) << 32;
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        u64* dst = xarea(p, (p + 1 + q) & 3, s & 1) + tid;
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) __hip_atomic_store(dst + NT * k2, tag | (u64)__float_as_uint(acc[q + 1][k2]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    own = acc[0];
    if (s - 1 >= 0) fetch_issue(s);
  };
  __builtin_amdgcn_s_waitcnt(0);
  for (int s = maxlen - 1; s >= 0; s -= 2) {
    step(std::integral_constant<int, 0>{}, s);
    if (s - 1 >= 0) step(std::integral_constant<int, 1>{}, s - 1);
  }
  if (bdoc < B)
    for (int t = len; t < L; ++t)
#pragma unroll
      for (int gt = 0; gt < 4; ++gt) *reinterpret_cast<float4*>(dxproj + ((size_t)bdoc * L + t) * ldx + (size_t)d * 4 * H + gt * H + u) = make_float4(0.f, 0.f, 0.f, 0.f);
}

static int quad_f32_common(hipStream_t st, int B, int H, int ndir, const float* w_hh, void* ws, int transposed, float** wpk, u64** xch, unsigned** status) {
  *wpk = (float*)ws;
  *xch = (u64*)((char*)ws + pair_wbytes(H, ndir));
  *status = (unsigned*)((char*)*xch + pair_xbytes(B, ndir));
  const size_t total = (size_t)ndir * 4 * 4 * 4 * 64 * 64;
  hipLaunchKernelGGL(lstm_pack_weights_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w_hh, *wpk, H, ndir, transposed);
  if (hipMemsetAsync(*xch, 0, pair_xbytes(B, ndir) + 256, st) != hipSuccess) { mts_set_error("lstm_quad(f32): memset failed"); return MTS_ERR_LAUNCH; }
  return lp_ensure_sticky();
}

int mts_lstm_quad_f32_fwd(hipStream_t st, int B, int L, int H, int ndir, const void* xproj, const float* w_hh, const float* b_hh,
                          const int32_t* lengths, void* out, void* gates, float* cells, void* ws) {
  float* wpk; u64* xch; unsigned* status;
  if (int rc = quad_f32_common(st, B, H, ndir, w_hh, ws, 0, &wpk, &xch, &status)) return rc;
  const size_t lds = (size_t)2 * LP_DOCS * (H + 4) * 4;
  const int max_quads = std::max(1, std::min(LQ_MAX_QUADS, g_max_pairs / 2));
  const int docs_per_launch = std::max(1, max_quads / ndir) * LP_DOCS;
  for (int b0 = 0; b0 < B; b0 += docs_per_launch) {
    const int bc = std::min(docs_per_launch, B - b0);
    const size_t r0 = (size_t)b0 * L;
    if (b0 > 0 && hipMemsetAsync(xch, 0, pair_xbytes(bc, ndir) + 256, st) != hipSuccess) { mts_set_error("lstm_quad(f32): memset failed"); return MTS_ERR_LAUNCH; }
    const int nquads = ceil_div(bc, LP_DOCS) * ndir;
    hipLaunchKernelGGL(lstm_fwd_quad_f32_kernel, dim3(32 * ceil_div(nquads, 8)), dim3(256), lds, st, bc, L, ndir, nquads, (const float*)xproj + r0 * ndir * 4 * H,
                       (const float*)wpk, b_hh, lengths ? lengths + b0 : nullptr, (float*)out + r0 * ndir * H, (float*)gates + r0 * ndir * 4 * H,
                       cells + r0 * ndir * H, xch, status, (char*)status + 256, g_spin_limit, g_sticky_dev);
  }
  MTS_LAUNCH_CHECK("mts_lstm_fwd(quad f32)");
  return MTS_OK;
}

int mts_lstm_quad_f32_bwd(hipStream_t st, int B, int L, int H, int ndir, const float* w_hh, const int32_t* lengths, const void* out, const void* gates,
                          const float* cells, const void* dout, void* dxproj, void* hprev, void* ws) {
  float* wpk; u64* xch; unsigned* status;
  if (int rc = quad_f32_common(st, B, H, ndir, w_hh, ws, 1, &wpk, &xch, &status)) return rc;
  const size_t rows4 = (size_t)B * L * ndir * H / 4;
  if (hprev) hipLaunchKernelGGL(lstm_hprev_kernel<float>, dim3((unsigned)((rows4 + 255) / 256)), dim3(256), 0, st, B, L, H, ndir, lengths, (const float*)out, (float*)hprev);
  const size_t lds = (size_t)2 * LP_DOCS * (4 * (H / 4) + 4) * 4;
  const int max_quads = std::max(1, std::min(LQ_MAX_QUADS, g_max_pairs / 2));
  const int docs_per_launch = std::max(1, max_quads / ndir) * LP_DOCS;
  for (int b0 = 0; b0 < B; b0 += docs_per_launch) {
    const int bc = std::min(docs_per_launch, B - b0);
    const size_t r0 = (size_t)b0 * L;
    if (b0 > 0 && hipMemsetAsync(xch, 0, pair_xbytes(bc, ndir) + 256, st) != hipSuccess) { mts_set_error("lstm_quad(f32): memset failed"); return MTS_ERR_LAUNCH; }
    const int nquads = ceil_div(bc, LP_DOCS) * ndir;
    hipLaunchKernelGGL(lstm_bwd_quad_f32_kernel, dim3(32 * ceil_div(nquads, 8)), dim3(256), lds, st, bc, L, ndir, nquads, (const float*)wpk,
                       lengths ? lengths + b0 : nullptr, (const float*)gates + r0 * ndir * 4 * H, cells + r0 * ndir * H, (const float*)dout + r0 * ndir * H,
                       (float*)dxproj + r0 * ndir * 4 * H, xch, status, (char*)status + 256, g_spin_limit, g_sticky_dev);
  }
  MTS_LAUNCH_CHECK("mts_lstm_bwd(quad f32)");
  return MTS_OK;
}
