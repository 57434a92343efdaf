// Host-side collation (no device code): ragged per-document embedding matrices -> ONE zero-padded batch, written straight into the caller's
// buffer -- which the product's dataset makes a pinned ring slot, so that the host-to-device copy can start from it as it is.
//
// Replaces the `merge` closure of AudioPortionDataset.collater (EncoderDataset.py:20-27, :103-109: a fresh pageable fp32 torch.zeros batch
// per step, filled document by document on one thread: 117 MB per step at BASELINE configs[1], 12-18 ms against a 2 ms step).  The copy that
// pads is the only pass over the data; it is split over threads by destination rows and can narrow fp32 -> bf16 (round to nearest even, NaN
// kept quiet: the bits torch's .to(torch.bfloat16) and the device cast kernel produce) on the way.
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>
#include <algorithm>
#include <mutex>
#include <condition_variable>
#include <functional>
#include <emmintrin.h>
#include "common.h"

// Same-dtype rows go out with non-temporal 16-byte stores: the destination (a pinned ring slot the GPU's DMA engine reads next) is written
// once and not read by the host, and an ordinary store first READS the line it is about to overwrite -- a third more memory traffic for a
// pass that is memory-bound.  (glibc's memcpy switches to such stores only above several MB per call; a document's piece is < 1 MB.)
static inline void copy_stream(char* d, const char* s, size_t n) {
  if ((((uintptr_t)d | n) & 15) != 0) { memcpy(d, s, n); return; }
  const __m128i* sp = (const __m128i*)s;
  __m128i* dp = (__m128i*)d;
  const size_t v = n >> 4;
  size_t k = 0;
  for (; k + 4 <= v; k += 4) {
    const __m128i a = _mm_loadu_si128(sp + k), b = _mm_loadu_si128(sp + k + 1), c = _mm_loadu_si128(sp + k + 2), e = _mm_loadu_si128(sp + k + 3);
    _mm_stream_si128(dp + k, a); _mm_stream_si128(dp + k + 1, b); _mm_stream_si128(dp + k + 2, c); _mm_stream_si128(dp + k + 3, e);
  }
  for (; k < v; ++k) _mm_stream_si128(dp + k, _mm_loadu_si128(sp + k));
}

// A small persistent pool: starting a thread costs 20-30 us, sixteen of them per call were a fifth of a 58-MB collation.  One job at a time
// (calls are serialised by a mutex: the collater runs on one producer thread); workers sleep on a condition variable between jobs.
namespace {
struct Pool {
  std::mutex call_mu;                       // one mts_collate_pad at a time
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  std::vector<std::thread> workers;
  std::function<void(int)> job;             // job(part), part = 1 .. parts - 1 (part 0 runs on the calling thread)
  int parts = 0, next = 0, left = 0;
  unsigned long long epoch = 0;
  bool stop = false;
  void ensure(int n) {
    while ((int)workers.size() < n) workers.emplace_back([this] { loop(); });
  }
  void loop() {
    unsigned long long seen = 0;
    std::unique_lock<std::mutex> lk(mu);
    for (;;) {
      cv_work.wait(lk, [&] { return stop || (epoch != seen && next < parts); });
      if (stop) return;
      while (next < parts) {
        const int part = next++;
        lk.unlock();
        job(part);
        lk.lock();
        if (--left == 0) cv_done.notify_all();
      }
      seen = epoch;
    }
  }
  void run(int nparts, std::function<void(int)> f) {
    {
      std::lock_guard<std::mutex> lk(mu);
      job = f; parts = nparts; next = 1; left = nparts - 1; ++epoch;
    }
    cv_work.notify_all();
    f(0);
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [&] { return left == 0; });
    parts = 0;
  }
  ~Pool() {
    { std::lock_guard<std::mutex> lk(mu); stop = true; }
    cv_work.notify_all();
    for (auto& t : workers) t.join();
  }
};
Pool* pool() { static Pool* p = new Pool(); return p; }      // leaked on purpose: no destructor order problems at process exit
}  // namespace

static inline uint16_t f32_to_bf16_rne(uint32_t u) {
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);      // NaN stays NaN
  return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// rows [r0, r1) of the padded batch (row = b * Lmax + i)
static void collate_rows(int src_dtype, int dst_dtype, int Lmax, int D, const void* const* docs, const int64_t* doc_rows, void* dst, int64_t r0,
                         int64_t r1, float pad) {
  const size_t dsz = dst_dtype == MTS_F32 ? 4 : 2, ssz = src_dtype == MTS_F32 ? 4 : 2;
  int64_t r = r0;
  while (r < r1) {
    const int64_t b = r / Lmax, i = r - b * Lmax;
    const int64_t have = std::min<int64_t>(std::max<int64_t>(doc_rows[b], 0), Lmax);
    const int64_t end = std::min<int64_t>(r1, (b + 1) * (int64_t)Lmax);                // rows of this document inside the range
    const int64_t ncopy = std::max<int64_t>(0, std::min<int64_t>(end - r, have - i));
    char* d = (char*)dst + (size_t)r * D * dsz;
    if (ncopy > 0) {
      const char* s = (const char*)docs[b] + (size_t)i * D * ssz;
      const size_t n = (size_t)ncopy * D;
      if (src_dtype == dst_dtype) {
        copy_stream(d, s, n * dsz);
      } else if (src_dtype == MTS_F32) {                                               // fp32 -> bf16
        const uint32_t* sp = (const uint32_t*)s;
        uint16_t* dp = (uint16_t*)d;
        for (size_t k = 0; k < n; ++k) dp[k] = f32_to_bf16_rne(sp[k]);
      } else {                                                                         // bf16 -> fp32 (exact)
        const uint16_t* sp = (const uint16_t*)s;
        uint32_t* dp = (uint32_t*)d;
        for (size_t k = 0; k < n; ++k) dp[k] = (uint32_t)sp[k] << 16;
      }
    }
    const int64_t nzero = (end - r) - ncopy;
    if (nzero > 0) {
      char* z = d + (size_t)ncopy * D * dsz;
      if (pad == 0.0f) memset(z, 0, (size_t)nzero * D * dsz);
      else if (dst_dtype == MTS_F32) std::fill((float*)z, (float*)z + (size_t)nzero * D, pad);
      else { uint32_t u; memcpy(&u, &pad, 4); std::fill((uint16_t*)z, (uint16_t*)z + (size_t)nzero * D, f32_to_bf16_rne(u)); }
    }
    r = end;
  }
  _mm_sfence();                                            // the streamed rows are globally visible before the caller enqueues the DMA
}

extern "C" int mts_collate_pad(int src_dtype, int dst_dtype, int B, int Lmax, int D, const void* const* docs, const int64_t* doc_rows, void* dst,
                               float pad_value, int nthreads) {
  MTS_CHECK_ARG(B > 0 && Lmax > 0 && D > 0 && docs && doc_rows && dst, "mts_collate_pad: bad arguments");
  MTS_CHECK_ARG((src_dtype == MTS_F32 || src_dtype == MTS_BF16) && (dst_dtype == MTS_F32 || dst_dtype == MTS_BF16), "mts_collate_pad: dtypes must be fp32 or bf16");
  for (int b = 0; b < B; ++b) MTS_CHECK_ARG(docs[b] || doc_rows[b] <= 0, "mts_collate_pad: document %d is NULL", b);
  const int64_t rows = (int64_t)B * Lmax;
  const size_t bytes = (size_t)rows * D * (dst_dtype == MTS_F32 ? 4 : 2);
  int nt = std::max(1, std::min(nthreads, 64));
  if (bytes < (size_t)(1 << 20)) nt = 1;                                               // small batches: waking the pool costs more than the copy
  nt = (int)std::min<int64_t>(nt, rows);
  if (nt == 1) {
    collate_rows(src_dtype, dst_dtype, Lmax, D, docs, doc_rows, dst, 0, rows, pad_value);
    return MTS_OK;
  }
  Pool* p = pool();
  std::lock_guard<std::mutex> call(p->call_mu);
  p->ensure(nt - 1);
  const int64_t per = (rows + nt - 1) / nt;
  p->run(nt, [&](int part) {
    const int64_t r0 = std::min<int64_t>(rows, part * per), r1 = std::min<int64_t>(rows, r0 + per);
    if (r0 < r1) collate_rows(src_dtype, dst_dtype, Lmax, D, docs, doc_rows, dst, r0, r1, pad_value);
  });
  return MTS_OK;
}
