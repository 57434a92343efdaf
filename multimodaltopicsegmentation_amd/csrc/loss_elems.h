// Per-element loss terms of the tagger heads (models/focal_loss.py:38-57, models/CRF.py:303,345-356), shared by tagger_loss_kernel (loss.hip) and the
// fused LayerNorm + head + loss + LayerNorm-backward tail of the last encoder layer (norm.hip): the same functions, so the same bits.
#pragma once
#include "common.h"

// models/focal_loss.py:38-57 for one element; returns loss, writes dloss/dx
__device__ __forceinline__ float focal_elem(float x, float y, float alpha, float gamma, float& grad) {
  const float p = sigmoid_f(x);
  // BCE with logits, stable: max(x,0) - x*y + log1p(exp(-|x|))
  const float ce = fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
  const float pt = p * y + (1.f - p) * (1.f - y);
  const float om = 1.f - pt;
  float mod, dmod;   // (1-pt)^gamma and its derivative wrt pt
  if (gamma == 2.f) { mod = om * om; dmod = -2.f * om; }
  else if (gamma == 0.f) { mod = 1.f; dmod = 0.f; }
  else { mod = powf(om, gamma); dmod = (om > 0.f) ? -gamma * powf(om, gamma - 1.f) : 0.f; }
  const float at = (alpha >= 0.f) ? alpha * y + (1.f - alpha) * (1.f - y) : 1.f;
  // d ce/dx = p - y ; d pt/dx = (2y-1) p (1-p)
  const float dpt = (2.f * y - 1.f) * p * (1.f - p);
  grad = at * ((p - y) * mod + ce * dmod * dpt);
  return at * ce * mod;
}

// nn.BCELoss(sigmoid(x), y) with the log clamp at -100 (models/CRF.py:303, :345-352)
__device__ __forceinline__ float bce_elem(float x, float y, float& grad) {
  const float p = sigmoid_f(x);
  const float lp = fmaxf(logf(p), -100.f), l1p = fmaxf(logf(1.f - p), -100.f);
  // gradient through the clamps as autograd sees them: d/dp [-y log p] = -y/p unless clamped
  const float dlp = (logf(p) > -100.f) ? 1.f / p : 0.f;
  const float dl1p = (logf(1.f - p) > -100.f) ? -1.f / (1.f - p) : 0.f;
  grad = -(y * dlp + (1.f - y) * dl1p) * p * (1.f - p);
  return -(y * lp + (1.f - y) * l1p);
}

// nn.CrossEntropyLoss over two classes for one row (ignore_index handled by the caller): loss and d loss / d (x0, x1)
__device__ __forceinline__ float ce2_elem(float x0, float x1, int t, float& g0, float& g1) {
  const float m = fmaxf(x0, x1);
  const float lse = m + logf(expf(x0 - m) + expf(x1 - m));
  const float p0 = expf(x0 - lse), p1 = expf(x1 - lse);
  g0 = p0 - (t == 0 ? 1.f : 0.f);
  g1 = p1 - (t == 1 ? 1.f : 0.f);
  return lse - (t == 0 ? x0 : x1);
}
