// select_dev.hpp -- device code shared by select.hip (the general per-window filter kernels) and binned.hip (the same stage fused per
// reference bin): what the filter looks at of a raw alignment, and the order in which the greedy of SequentialGuideAligner.scala:315-320
// takes the alignments of a window.  Private to those translation units.
#pragma once
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace calitas {

namespace {

struct Derived {
  int32_t start, end, score;
  uint16_t gaps, edits;
  uint32_t ekey;      // enumeration order inside the window: strand list << 19 | end column << 6 | start matrix << 4 | PAM + 1
  uint32_t widx;      // global window index
};

// The order in which the per-window greedy takes records (score desc, gap bases asc (GA:125-129), then the enumeration order: a
// stable sort), as a number: the larger, the earlier.  Bit 63 = strand list (the two lists are filtered one after the other), bit 0
// set so that no record has key 0.  ekey is unique inside a window, so the order is total.
__device__ __forceinline__ unsigned long long order_key(const Derived& d) {
  return ((unsigned long long)(d.ekey >> 19) << 63) | ((unsigned long long)(uint32_t)(d.score + (1 << 21)) << 40) |
         ((unsigned long long)(0xFFFFu - d.gaps) << 24) | ((unsigned long long)(0x7FFFFu - (d.ekey & 0x7FFFFu)) << 1) | 1ull;
}

// What the filter looks at, from one raw alignment.
__device__ __forceinline__ Derived derive(const RawAln* rp, const GuideDev* guides, const uint64_t* win_base, const int2* win, uint32_t window_lo,
                                          uint32_t windows_per_guide) {
  const uint32_t contig = rp->contig, window_k = rp->window_k, guide = rp->guide;
  const int pam = rp->pam, offset = rp->offset, n_ops = rp->n_ops, dir = rp->dir;
  const OpCounts oc = count_ops(load_ops_words(rp->ops), n_ops);
  int diffs = oc.non_eq, gaps = oc.gaps, pam_len = 0;
  if (pam >= 0) { pam_len = guides[guide].pam_len[pam]; diffs += offset + __popc((unsigned)rp->pam_x); gaps += offset; }
  const uint64_t wi = win_base[contig] + window_k;
  const int2 w = win[wi];
  const int start_s = (int)rp->t_start - 1, end_s = (int)rp->t_end_guide + offset + pam_len;
  Derived d;
  if (dir == 0) { d.start = w.x + start_s; d.end = w.x + end_s; }
  else          { d.start = w.y - end_s;   d.end = w.y - start_s; }
  d.score = rp->score; d.gaps = (uint16_t)gaps; d.edits = (uint16_t)diffs;
  const uint32_t pam5 = guides[guide].pam5;
  const uint32_t list = pam5 ? (dir == 1 ? 0u : 1u) : (dir == 0 ? 0u : 1u);   // 0 = forward-strand list (SGA:316)
  d.ekey = (list << 19) | ((uint32_t)rp->t_end_guide << 6) | ((uint32_t)rp->pad << 4) | (uint32_t)(pam + 1);
  d.widx = guide * windows_per_guide + ((uint32_t)wi - window_lo);
  return d;
}

// 64-bit wave maximum (all lanes get it)
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_xor(v, off);
    v = o > v ? o : v;
  }
  return v;
}

}  // namespace

}  // namespace calitas
