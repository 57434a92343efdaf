// Shared by the two band-attention translation units (band_attn.hip: generic VALU kernels, fp32/bf16, any head dim;
// band_attn_mfma.hip: bf16 matrix-core kernels for head dims that are multiples of 32).
#pragma once
#include "common.h"

struct BandArgs {
  const void* qkv; const int32_t* lengths; void* ctx; float* probs;
  const void* dctx; void* dqkv; float* dscores;
  int B, L, D, heads, hd, radius, slots;
  int rs;      // LDS row stride (bytes) of staged q/k/v rows
  int ps;      // LDS row stride (floats) of the probability tile
  float q_scale;
  const int32_t* row0;  // packed batches: first row of document b (rows of a document are contiguous, only its `lengths[b]` valid rows exist); NULL = padded [B, L]
  float drop_scale; uint32_t drop_thr; uint64_t drop_seed;   // dropout on the attention probabilities (drop_thr = 0: off)
  float* bias_slab;   // MFMA backward: per-(document, tile) column sums of dqkv, [<= B*ceil(L/128)][3D], or NULL
  int img_bytes;      // MFMA kernels: size of the staged-row LDS image
};

__host__ __device__ inline int band_slots(int radius) { return ((2 * radius + 1 + 31) / 32) * 32; }

// decode the XCD-remapped linear block id into (tile, head, doc): consecutive tiles of one (doc, head) stay on one XCD
__device__ __forceinline__ void decode_block(int ntiles, int heads, int nblocks, int& tile, int& h, int& b) {
  int bid = blockIdx.x;
  const int q = nblocks >> 3, rr = nblocks & 7, xcd = bid & 7, idx = bid >> 3;
  bid = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + idx;
  tile = bid % ntiles;
  h = (bid / ntiles) % heads;
  b = bid / (ntiles * heads);
}


// where document b lives: first row and number of rows that exist for it
struct DocView { int base; int Lb; };
__device__ __forceinline__ DocView doc_view(const BandArgs& a, int b) {
  if (a.row0) return DocView{a.row0[b], min(a.lengths[b], a.L)};
  return DocView{b * a.L, a.L};
}

// attention-probability dropout (modeling_longformer.py:590): keep decision of probability (packed row, head, slot)
__device__ __forceinline__ bool band_keep(const BandArgs& a, int grow, int h, int c) {
  return mts_hash32(a.drop_seed, ((uint64_t)grow * a.heads + h) * a.slots + c) >= a.drop_thr;
}

// band_attn_mfma.hip: returns MTS_OK after launching, or -1 when the shape is outside what the MFMA kernels cover
// (the caller then takes the generic kernels).
int mts_band_mfma_fwd(const BandArgs& a, hipStream_t st);
int mts_band_mfma_bwd(const BandArgs& a, hipStream_t st, int* slab_rows);   // *slab_rows: rows of bias_slab the kernels filled
