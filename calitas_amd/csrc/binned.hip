// binned.hip -- everything behind trace_kernel for one guide, one wave per reference bin (see binned.hpp for the why).
//
// A bin is a power-of-two stretch of ONE contig (32 kb for the default window).  trace_kernel drops every raw alignment into the bin
// its window STARTS in (kernels.hip); bin_hits_kernel's wave for bin b owns the hits whose coordinate_start -- the first key of
// ReferenceHit.sort (ReferenceHit.scala:284) -- lies in [lo, hi) = the bin's stretch, so the rows of consecutive bins are consecutive
// pieces of hits.txt.  What the wave needs to decide those hits exactly:
//   * every window that can hold such a hit: windows starting in (lo - W, hi) -- its own bin and the tail of bin b-1;
//   * removeOverlaps (SearchReference.scala:653-675) walks a (chromosome, strand) group left to right and restarts wherever a hit
//     overlaps nothing before it (hits.hip, prep_kernel).  A hit's fate hangs on its cluster: from the restart point at or left of it
//     to the first hit right of it that it does not swallow.  Hits are at most HIT_MAX_LEN (128) bases long, so the right side needs
//     the hits starting within 128 bases behind hi (windows starting below hi + 128: the head of bin b+1), and the left side needs a
//     restart point whose own look-back (128 bases) is inside the stretch where all hits are known.  The wave takes the windows
//     starting in [lo - 2W, hi + 128): all hits with coordinate_start >= lo - W are known, restart points from lo - W + 128 on are
//     certain, and a hit of the bin that no walk from a certain restart point reaches raises BIN_FLAG_HALO (a tandem repeat with
//     chained hits over more than W - 128 bases): the call then finishes on the general kernels.
// Per wave: (1) the context's raw alignments (<= 128, two per lane's worth of LDS), (2) the greedy of SGA:315-320 window by window --
// a wave-wide maximum of the order keys per round, as filter_wave_kernel does --, (3) GuideAlignment coordinates of the accepted ones
// (<= 64, one per lane), (4) their order by counting, restart points, the cluster walks, (5) the row lengths of the kept hits of the
// bin (the row builder's own arithmetic without the text: build_middle<false>) and the bin's (rows, bytes) with a two-level sum.
// bin_rows_kernel then places every bin from those sums and builds its rows at their final offsets -- no staging copy of the text.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "binned.hpp"
#include "tuning.hpp"
#include "hits_dev.hpp"
#include "select_dev.hpp"

namespace calitas {

namespace {

constexpr int BIN_WAVES = 1;            // waves (= bins) per workgroup: waves never wait for each other, and a one-wave workgroup gives its slot back when its bin is done
constexpr uint32_t ACC_MAX = 64;        // accepted alignments of a bin's context: one per lane
constexpr uint32_t CHUNK_SHIFT = 8;     // bins per chunk / chunks per super-chunk of the three-level sum that places the bins' text
constexpr uint32_t SUPER_SHIFT = 2 * CHUNK_SHIFT;
static_assert(BIN_CAP == 64, "a lane holds one alignment of each of the three bins of a context");

struct BinRow {            // a kept hit of a bin, in final order
  uint32_t raw;            // its index in the lane's list of raw alignments
  uint32_t len;            // length of the row's middle part (what build_middle returns)
};

struct BinArgs {
  const RawAln* raw;                 // the lane's raw alignments (trace_kernel's list)
  const uint32_t* bin_idx;           // n_bins x BIN_CAP indices into raw[]
  const uint32_t* bin_count;
  const uint32_t* bin_base;          // per contig (n_contigs + 1), absolute bin indices
  const uint32_t* bin_contig;        // per bin (absolute index): its contig
  int n_contigs;
  uint32_t bin_first, n_bins, bin_shift;
  const GuideDev* guides;
  const uint64_t* win_base;
  const int2* win;
  int W, step, max_total_diffs, max_overlap;
  BinRow* rows;                      // n_bins x BIN_ROWS
  uint32_t* bin_rows;                // per bin: kept rows
  uint32_t* bin_bytes;               // per bin: their text bytes
  unsigned long long* chunk_bytes;   // per 256 bins (zero at launch)
  unsigned long long* super_bytes;   // per 65536 bins (zero at launch)
  uint32_t* chunk_rows;
  uint32_t* chunk_acc;               // accepted alignments (after the per-window filter) of the windows that start in the chunk's bins
  uint32_t* flags;
  // the hits this call owns, as keys (contig << 32 | coordinate_start): [own_lo, own_hi); everything by default.  A process of a
  // multi-GPU job owns a stretch of the genome that may begin and end anywhere (binned.hpp, BinnedParams)
  unsigned long long own_lo, own_hi;
  uint32_t* rows_list;               // bins with rows, in no particular order (one append per wave)
  uint32_t* rows_count;              // (zero at launch)
  unsigned long long* stamps;        // [0] align_kernel's start (written there), [1] bin_hits_small_kernel's: the device's wall clock
  uint32_t dbg;                      // timing experiments (CALITAS_BINNED_SKIP; always 0 unless built with -DCALITAS_EXPERIMENTS)
};

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const uint32_t o = (uint32_t)__shfl_xor((int)v, off);
    v = o < v ? o : v;
  }
  return v;
}

__device__ __forceinline__ unsigned long long wave_sum_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// What the filter looks at of a raw alignment of a contig whose windows start at wbase in the window table (select_dev.hpp derive()
// without the dependent load of win_base[contig]: the contig is the bin's).
__device__ __forceinline__ Derived derive_in_contig(const RawAln* rp, const GuideDev* guides, uint32_t wbase, const int2* win) {
  const uint32_t window_k = rp->window_k, guide = rp->guide;
  const int pam = rp->pam, offset = rp->offset, n_ops = rp->n_ops, dir = rp->dir;
  const OpCounts oc = count_ops(load_ops_words(rp->ops), n_ops);
  int diffs = oc.non_eq, gaps = oc.gaps, pam_len = 0;
  if (pam >= 0) { pam_len = guides[guide].pam_len[pam]; diffs += offset + __popc((unsigned)rp->pam_x); gaps += offset; }
  const uint32_t wi = wbase + window_k;
  const int2 w = win[wi];
  const int start_s = (int)rp->t_start - 1, end_s = (int)rp->t_end_guide + offset + pam_len;
  Derived d;
  if (dir == 0) { d.start = w.x + start_s; d.end = w.x + end_s; }
  else          { d.start = w.y - end_s;   d.end = w.y - start_s; }
  d.score = rp->score; d.gaps = (uint16_t)gaps; d.edits = (uint16_t)diffs;
  const uint32_t pam5 = guides[guide].pam5;
  const uint32_t list = pam5 ? (dir == 1 ? 0u : 1u) : (dir == 0 ? 0u : 1u);   // 0 = forward-strand list (SGA:316)
  d.ekey = (list << 19) | ((uint32_t)rp->t_end_guide << 6) | ((uint32_t)rp->pad << 4) | (uint32_t)(pam + 1);
  d.widx = wi;
  return d;
}

// One bin by a whole wave: the bins whose context holds more than SMALL_MAX alignments (bin_hits_kernel's waves stride over the
// list bin_hits_small_kernel made of them).
__device__ __forceinline__ void bin_hits_wave(const BinArgs& a, const MidArgs& m, const uint32_t rel, const int lane) {
  __shared__ int32_t k_s[BIN_WAVES][ACC_MAX], k_e[BIN_WAVES][ACC_MAX];  // kept intervals of the window being filtered
  __shared__ uint32_t acc[BIN_WAVES][ACC_MAX];                          // accepted alignments in arrival order: index into raw[]
  // accepted alignments by arrival (t_*) and in ReferenceHit.sort order (s_*)
  __shared__ int32_t t_start[BIN_WAVES][ACC_MAX], t_score[BIN_WAVES][ACC_MAX];
  __shared__ int32_t s_start[BIN_WAVES][ACC_MAX], s_end[BIN_WAVES][ACC_MAX], s_score[BIN_WAVES][ACC_MAX];
  __shared__ uint8_t t_minus[BIN_WAVES][ACC_MAX], s_cs[BIN_WAVES][ACC_MAX], s_idx[BIN_WAVES][ACC_MAX], s_head[BIN_WAVES][ACC_MAX],
      s_keep[BIN_WAVES][ACC_MAX], s_done[BIN_WAVES][ACC_MAX];
  constexpr uint32_t wv = 0;                                            // one wave per workgroup
  wave_lds_sync();                                                      // the previous bin of this wave is done with the arrays
  if (rel >= a.n_bins) return;
  const uint32_t b = a.bin_first + rel;
  auto finish = [&](uint32_t n_rows, uint32_t bytes, uint32_t n_acc) {
    if (lane == 0) {
      a.bin_rows[rel] = n_rows; a.bin_bytes[rel] = bytes;
      if (n_rows) {
        atomicAdd(a.chunk_rows + (rel >> CHUNK_SHIFT), n_rows); atomicAdd(a.chunk_bytes + (rel >> CHUNK_SHIFT), (unsigned long long)bytes);
        atomicAdd(a.super_bytes + (rel >> SUPER_SHIFT), (unsigned long long)bytes);
        a.rows_list[atomicAdd(a.rows_count, 1u)] = rel;
      }
      if (n_acc) atomicAdd(a.chunk_acc + (rel >> CHUNK_SHIFT), n_acc);
    }
  };
  auto decline = [&](uint32_t why) { if (lane == 0) atomicOr(a.flags, why); finish(0, 0, 0); };
  const uint32_t n_own = a.bin_count[rel];
  const uint32_t n_prev_raw = rel > 0 ? a.bin_count[rel - 1] : 0u;
  if (n_prev_raw == 0 && n_own == 0) { finish(0, 0, 0); return; }       // no window that could hold a hit of this bin has alignments
  const uint32_t c = a.bin_contig[b];
  const uint32_t wbase = (uint32_t)a.win_base[c];                       // the contig's first entry of the window table
  const uint32_t bb = b - a.bin_base[c];                                // bin inside its contig
  const bool has_prev = bb > 0 && rel > 0, has_next = b + 1 < a.bin_base[c + 1] && rel + 1 < a.n_bins;
  const uint32_t n_prev = has_prev ? n_prev_raw : 0u, n_next = has_next ? a.bin_count[rel + 1] : 0u;
  if (n_prev == 0 && n_own == 0) { finish(0, 0, 0); return; }
  if (n_prev > BIN_CAP || n_own > BIN_CAP || n_next > BIN_CAP) { decline(BIN_FLAG_CROWDED); return; }
  const int64_t lo = (int64_t)bb << a.bin_shift, hi = lo + ((int64_t)1 << a.bin_shift);
  const int64_t ctx_lo = lo - 2 * (int64_t)a.W, ctx_hi = hi + HIT_MAX_LEN;
  // hits are all known from here on; restart points are certain HIT_MAX_LEN further right.  At the start of a contig nothing is missing.
  const int64_t known_from = ctx_lo <= 0 ? -((int64_t)1 << 40) : lo - (int64_t)a.W;

  // ---- 1. the context's raw alignments: lane l holds alignment l of the previous, the own and the next bin (those whose window
  //         starts in [lo - 2W, hi + 128)) in registers ----
  unsigned long long key[3];        // order_key(), 0 = none / taken
  int st[3], en[3];
  uint32_t ed[3], widx[3], src[3], wk[3];
#pragma unroll
  for (int q = 0; q < 3; q++) {
    const uint32_t nq = q == 0 ? n_prev : q == 1 ? n_own : n_next;
    key[q] = 0; st[q] = 0; en[q] = 0; ed[q] = 0; widx[q] = 0xFFFFFFFFu; src[q] = 0; wk[q] = 0;
    if ((uint32_t)lane < nq) {
      const uint32_t idx = a.bin_idx[(size_t)((int64_t)rel + q - 1) * BIN_CAP + (uint32_t)lane];
      const RawAln* rp = a.raw + idx;
      const uint32_t window_k = rp->window_k;
      const int64_t ws = (int64_t)window_k * (int64_t)a.step;            // where the window starts on the contig
      if (ws >= ctx_lo && ws < ctx_hi) {
        const Derived d = derive_in_contig(rp, a.guides, wbase, a.win);      // widx = the window's index in the table
        key[q] = order_key(d); st[q] = d.start; en[q] = d.end; ed[q] = d.edits; widx[q] = d.widx; src[q] = idx; wk[q] = window_k;
      }
    }
  }

  // ---- 2. the greedy of SGA:315-320, window by window in ascending order: acc[] = the accepted alignments in arrival order ----
  uint32_t nA = 0, nA_own = 0;
  bool over = false;
  for (uint32_t cur = 0;;) {
    uint32_t mw = 0xFFFFFFFFu;
#pragma unroll
    for (int q = 0; q < 3; q++) if (widx[q] != 0xFFFFFFFFu && widx[q] >= cur && widx[q] < mw) mw = widx[q];
    const uint32_t w = wave_min_u32(mw);
    if (w == 0xFFFFFFFFu) break;
    cur = w + 1u;
    uint32_t nk = 0;
    for (uint32_t list = 0; list < 2; list++) {
      const uint32_t first_kept = nk;                                   // overlaps are only tested against the same strand (SGA:317)
      for (;;) {
        unsigned long long mine = 0;
#pragma unroll
        for (int q = 0; q < 3; q++) if (widx[q] == w && (uint32_t)(key[q] >> 63) == list && key[q] > mine) mine = key[q];
        const unsigned long long bk = wave_max_u64(mine);
        if (bk == 0) break;
        const int owner = __ffsll((long long)__ballot(mine == bk)) - 1;   // keys are unique inside a window
        int b_start = 0, b_end = 0;
        uint32_t b_edits = 0, b_src = 0, b_wk = 0;
#pragma unroll
        for (int q = 0; q < 3; q++)
          if (widx[q] == w && key[q] == bk) { b_start = st[q]; b_end = en[q]; b_edits = ed[q]; b_src = src[q]; b_wk = wk[q]; key[q] = 0; }   // the owner takes it
        b_start = __shfl(b_start, owner); b_end = __shfl(b_end, owner); b_edits = (uint32_t)__shfl((int)b_edits, owner);
        b_src = (uint32_t)__shfl((int)b_src, owner); b_wk = (uint32_t)__shfl((int)b_wk, owner);
        if ((int)b_edits > a.max_total_diffs) continue;
        bool clash = false;
        for (uint32_t k = first_kept + (uint32_t)lane; k < nk && k < ACC_MAX; k += 64) {
          const int o = min(b_end, k_e[wv][k]) - max(b_start, k_s[wv][k]);   // GA:119-122
          clash = clash || o > a.max_overlap;
        }
        if (__ballot(clash) == 0) {
          if (nA < ACC_MAX && nk < ACC_MAX) { if (lane == 0) { k_s[wv][nk] = b_start; k_e[wv][nk] = b_end; acc[wv][nA] = b_src; } }
          else over = true;
          nk++; nA++;
          const int64_t ws = (int64_t)b_wk * (int64_t)a.step;            // (statistics) counted by the bin the window starts in
          if (ws >= lo && ws < hi) nA_own++;
          wave_lds_sync();
        }
      }
    }
  }
  if (over) { decline(BIN_FLAG_CROWDED); return; }
  if (nA == 0) { finish(0, 0, 0); return; }
  wave_lds_sync();

  // ---- 3. GuideAlignment coordinates of the accepted alignments (lane i = arrival i) ----
  const bool isA = (uint32_t)lane < nA;
  HitRec h{};
  uint32_t hsrc = 0;
  if (isA) {
    hsrc = acc[wv][lane];
    h = hit_record(a.raw + hsrc, a.guides, a.win_base, a.win);
    t_start[wv][lane] = h.gstart; t_score[wv][lane] = h.score; t_minus[wv][lane] = (uint8_t)h.minus;
  }
  if (__ballot(isA && h.gstart < 0) != 0) { decline(BIN_FLAG_RANGE); return; }
  wave_lds_sync();
  // ---- 4. ReferenceHit.sort among them by counting (equal keys keep their arrival order: a stable sort), restart points, walks ----
  uint32_t my_rank = 0;                                                 // lane i (arrival) -> its sorted position
  if (isA) {
    uint32_t rank = 0;
    for (uint32_t j = 0; j < nA; j++) {
      const int gs = t_start[wv][j], sc = t_score[wv][j];
      const uint32_t mi = t_minus[wv][j];
      const bool less = gs < h.gstart || (gs == h.gstart && (mi < h.minus || (mi == h.minus && sc > h.score)));
      const bool same = gs == h.gstart && mi == h.minus && sc == h.score;
      rank += (less || (same && j < (uint32_t)lane)) ? 1u : 0u;
    }
    s_start[wv][rank] = h.gstart; s_end[wv][rank] = h.rh_end; s_score[wv][rank] = h.score; s_cs[wv][rank] = (uint8_t)h.minus;
    s_idx[wv][rank] = (uint8_t)lane;
    my_rank = rank;
  }
  wave_lds_sync();
  const uint32_t r = (uint32_t)lane;                                    // from here on a lane is (also) a sorted position
  bool head = false;
  if (r < nA) {
    head = true;                                                        // hits.hip prep_body on the wave's arrays
    const int hs = s_start[wv][r];
    const uint32_t cs = s_cs[wv][r];
    for (uint32_t j = r; j-- > 0;) {
      if (s_start[wv][j] + HIT_MAX_LEN - 1 - hs < a.max_overlap) break;
      if (s_cs[wv][j] == cs && s_end[wv][j] - hs >= a.max_overlap) { head = false; break; }
    }
    s_head[wv][r] = head ? 1 : 0; s_keep[wv][r] = 0; s_done[wv][r] = 0;
  }
  wave_lds_sync();
  if (r < nA && head && (int64_t)s_start[wv][r] >= known_from + HIT_MAX_LEN) {   // hits.hip cluster_body: the reference's loop, SR:661-671
    const uint32_t cs = s_cs[wv][r];
    auto next = [&](uint32_t j) { for (j++; j < nA; j++) if (s_cs[wv][j] == cs) return j; return nA; };
    uint32_t j = r;
    for (;;) {
      const uint32_t hit = j;
      s_done[wv][hit] = 1;
      j = next(j);
      const int hs = s_start[wv][hit], he = s_end[wv][hit], hsc = s_score[wv][hit];
      bool more = false;
      int ov = 0;
      for (;;) {
        more = j < nA && !s_head[wv][j];
        if (!more) break;
        ov = max(0, min(s_end[wv][j], he) - max(s_start[wv][j], hs));   // RH:141-144
        if (!(ov >= a.max_overlap && s_score[wv][j] <= hsc)) break;
        s_done[wv][j] = 1;                                               // swallowed
        j = next(j);
      }
      if (!more || ov < a.max_overlap) s_keep[wv][hit] = 1;
      if (!more) break;
    }
  }
  wave_lds_sync();
  const unsigned long long okey = r < nA ? (((unsigned long long)c << 32) | (unsigned long long)(uint32_t)s_start[wv][r]) : 0ull;
  const bool mine = r < nA && (int64_t)s_start[wv][r] >= lo && (int64_t)s_start[wv][r] < hi && okey >= a.own_lo && okey < a.own_hi;
  if (__ballot(mine && !s_done[wv][r]) != 0) { decline(BIN_FLAG_HALO); return; }
  const unsigned long long kept = __ballot(mine && s_keep[wv][r] != 0);
  const uint32_t n_rows = (uint32_t)__popcll(kept);
  if (n_rows > BIN_ROWS) { decline(BIN_FLAG_CROWDED); return; }

  // ---- 5. the kept hits of the bin in final order: which alignment, and the length of the row's middle part -- every kept hit in
  //         the lane that holds its record (arrival lane), all of them at once ----
  const bool kept_here = isA && ((kept >> my_rank) & 1ull);
  int len = 0;
  uint32_t row_bytes = 0;
  if (kept_here) {
    const RawAln* rp = a.raw + hsrc;
    const GuideDev* gp = a.guides + rp->guide;
    const int pam = rp->pam;
    len = middle_length(rp, h, gp->L, pam >= 0 ? gp->pam_len[pam] : 0, (int)m.rc.pu_len[pam + 1], (int)m.n_max, (int)m.mid_bound);
    if (len >= 0) {
      const uint32_t name_len = m.name_off[h.contig + 1] - m.name_off[h.contig];
      row_bytes = m.rc.head_len + name_len + 1u + (uint32_t)len + m.rc.tail_len;
      a.rows[(size_t)rel * BIN_ROWS + (uint32_t)__popcll(kept & bits_below((int)my_rank))] = BinRow{hsrc, (uint32_t)len};
    }
  }
  const bool bad_row = __ballot(kept_here && len < 0) != 0;
  const uint32_t bytes = (uint32_t)wave_sum_u64(row_bytes);
  if (bad_row) { decline(BIN_FLAG_ROW); return; }
  finish(n_rows, bytes, nA_own);
}

// The listed bins: a fixed grid strides over the list (its length is on the device).
__global__ __launch_bounds__(64) void bin_hits_kernel(BinArgs a, MidArgs m, const uint32_t* list, const uint32_t* n_list) {
  CALITAS_TAIL_PRIO();
  const int lane = (int)(threadIdx.x & 63);
  const uint32_t n_todo = *n_list;
  for (uint32_t it = blockIdx.x; it < n_todo; it += gridDim.x) bin_hits_wave(a, m, list[it], lane);
}


// ---- the common case, one LANE per bin ------------------------------------------------------------------------------------------
// At max-guide-diffs 5 on a genome-sized reference a bin's context holds 0-4 raw alignments (0.4 on average), and a wave per bin spends
// its time in dependent loads and wave-wide reductions over two or three records.  bin_hits_small_kernel gives every bin one lane:
// a context of at most SMALL_MAX alignments runs the same five steps in registers (fixed-size arrays, constant indices only);
// everything else goes to a list that bin_hits_kernel works off with a wave per bin.

constexpr int SMALL_MAX = 4;

struct SmallArgs {
  uint32_t* complex_list;      // bins left to bin_hits_kernel
  uint32_t* complex_count;
  uint32_t force_complex;      // tests: every bin with alignments goes to the list
};

__global__ __launch_bounds__(64) void bin_hits_small_kernel(BinArgs a, MidArgs m, SmallArgs sa) {
  CALITAS_TAIL_PRIO();
  const int lane = (int)(threadIdx.x & 63);
  const uint32_t rel = blockIdx.x * 64 + threadIdx.x;
  const bool in_range = rel < a.n_bins;
  if (rel == 0) a.stamps[1] = (unsigned long long)wall_clock64();
  bool is_complex = false;
  uint32_t out_rows = 0, out_bytes = 0, out_acc = 0, out_flags = 0;
  BinRow out_row[SMALL_MAX];
#pragma unroll
  for (int k = 0; k < SMALL_MAX; k++) out_row[k] = BinRow{0u, 0u};
  if (in_range) {
    const uint32_t b = a.bin_first + rel;
    const uint32_t n_own = a.bin_count[rel];
    const uint32_t n_prev_raw = rel > 0 ? a.bin_count[rel - 1] : 0u;
    if (n_prev_raw != 0 || n_own != 0) {
      const uint32_t c = a.bin_contig[b];
      const uint32_t wbase = (uint32_t)a.win_base[c];
      const uint32_t bb = b - a.bin_base[c];
      const bool has_prev = bb > 0 && rel > 0, has_next = b + 1 < a.bin_base[c + 1] && rel + 1 < a.n_bins;
      const uint32_t n_prev = has_prev ? n_prev_raw : 0u, n_next = has_next ? a.bin_count[rel + 1] : 0u;
      if (n_prev != 0 || n_own != 0) {
        if (n_prev + n_own + n_next > (uint32_t)SMALL_MAX || sa.force_complex) is_complex = true;
        else {
          const int64_t lo = (int64_t)bb << a.bin_shift, hi = lo + ((int64_t)1 << a.bin_shift);
          const int64_t ctx_lo = lo - 2 * (int64_t)a.W, ctx_hi = hi + HIT_MAX_LEN;
          const int64_t known_from = ctx_lo <= 0 ? -((int64_t)1 << 40) : lo - (int64_t)a.W;
          // ---- 1. context: candidate j of (previous | own | next) bin ----
          unsigned long long key[SMALL_MAX];
          int st[SMALL_MAX], en[SMALL_MAX];
          uint32_t ed[SMALL_MAX], widx[SMALL_MAX], src[SMALL_MAX], wk[SMALL_MAX];
#pragma unroll
          for (int j = 0; j < SMALL_MAX; j++) {
            key[j] = 0; st[j] = 0; en[j] = 0; ed[j] = 0; widx[j] = 0xFFFFFFFFu; src[j] = 0; wk[j] = 0;
            const uint32_t uj = (uint32_t)j;
            if (uj < n_prev + n_own + n_next) {
              const int q = uj < n_prev ? 0 : uj < n_prev + n_own ? 1 : 2;
              const uint32_t e = q == 0 ? uj : q == 1 ? uj - n_prev : uj - n_prev - n_own;
              const uint32_t idx = a.bin_idx[(size_t)((int64_t)rel + q - 1) * BIN_CAP + e];
              const RawAln* rp = a.raw + idx;
              const uint32_t window_k = rp->window_k;
              const int64_t ws = (int64_t)window_k * (int64_t)a.step;
              if (ws >= ctx_lo && ws < ctx_hi) {
                const Derived d = derive_in_contig(rp, a.guides, wbase, a.win);
                key[j] = order_key(d); st[j] = d.start; en[j] = d.end; ed[j] = d.edits; widx[j] = d.widx; src[j] = idx; wk[j] = window_k;
              }
            }
          }
          // ---- 2. the greedy of SGA:315-320 window by window; acc_*[] = accepted alignments in arrival order ----
          uint32_t acc_src[SMALL_MAX];
#pragma unroll
          for (int k = 0; k < SMALL_MAX; k++) acc_src[k] = 0;
          uint32_t nA = 0;
          uint32_t cur = 0;
#pragma unroll
          for (int wround = 0; wround < SMALL_MAX; wround++) {             // at most SMALL_MAX distinct windows
            uint32_t w = 0xFFFFFFFFu;
#pragma unroll
            for (int j = 0; j < SMALL_MAX; j++) if (widx[j] != 0xFFFFFFFFu && widx[j] >= cur && widx[j] < w) w = widx[j];
            if (__ballot(w != 0xFFFFFFFFu) == 0) break;                     // (wave-uniform: no lane of the wave has another window)
            if (w != 0xFFFFFFFFu) {
              cur = w + 1u;
              int ks[SMALL_MAX], ke[SMALL_MAX];
#pragma unroll
              for (int k = 0; k < SMALL_MAX; k++) { ks[k] = 0; ke[k] = 0; }
              uint32_t nk = 0;
#pragma unroll
              for (int list = 0; list < 2; list++) {
                const uint32_t first_kept = nk;
#pragma unroll
                for (int round = 0; round < SMALL_MAX; round++) {
                  unsigned long long bk = 0;
#pragma unroll
                  for (int j = 0; j < SMALL_MAX; j++) if (widx[j] == w && (uint32_t)(key[j] >> 63) == (uint32_t)list && key[j] > bk) bk = key[j];
                  if (__ballot(bk != 0) == 0) break;                        // (no lane has another alignment in this list)
                  if (bk != 0) {
                    int b_start = 0, b_end = 0;
                    uint32_t b_edits = 0, b_src = 0, b_wk = 0;
#pragma unroll
                    for (int j = 0; j < SMALL_MAX; j++)
                      if (widx[j] == w && key[j] == bk) { b_start = st[j]; b_end = en[j]; b_edits = ed[j]; b_src = src[j]; b_wk = wk[j]; key[j] = 0; }
                    if ((int)b_edits <= a.max_total_diffs) {
                      bool clash = false;
#pragma unroll
                      for (int k = 0; k < SMALL_MAX; k++)
                        if ((uint32_t)k >= first_kept && (uint32_t)k < nk) clash = clash || (min(b_end, ke[k]) - max(b_start, ks[k]) > a.max_overlap);   // GA:119-122
                      if (!clash) {
#pragma unroll
                        for (int k = 0; k < SMALL_MAX; k++) {
                          if ((uint32_t)k == nk) { ks[k] = b_start; ke[k] = b_end; }
                          if ((uint32_t)k == nA) acc_src[k] = b_src;
                        }
                        nk++; nA++;
                        const int64_t ws = (int64_t)b_wk * (int64_t)a.step;
                        if (ws >= lo && ws < hi) out_acc++;
                      }
                    }
                  }
                }
              }
            }
          }
          // ---- 3. coordinates; 4. order by counting, restart points, the walk of SR:661-671 over the (at most four) sorted hits ----
          HitRec hr[SMALL_MAX];
#pragma unroll
          for (int i = 0; i < SMALL_MAX; i++) hr[i] = HitRec{};
#pragma unroll
          for (int i = 0; i < SMALL_MAX; i++) {
            if (__ballot((uint32_t)i < nA) == 0) break;                   // (no lane accepted that many)
            if ((uint32_t)i < nA) { hr[i] = hit_record(a.raw + acc_src[i], a.guides, a.win_base, a.win); if (hr[i].gstart < 0) out_flags |= BIN_FLAG_RANGE; }
          }
          int s_start[SMALL_MAX], s_end[SMALL_MAX], s_score[SMALL_MAX], s_arr[SMALL_MAX];
          uint32_t s_cs[SMALL_MAX];
#pragma unroll
          for (int p = 0; p < SMALL_MAX; p++) { s_start[p] = 0; s_end[p] = 0; s_score[p] = 0; s_cs[p] = 0; s_arr[p] = 0; }
#pragma unroll
          for (int i = 0; i < SMALL_MAX; i++) {
            if (__ballot((uint32_t)i < nA) == 0) break;
            if ((uint32_t)i < nA) {
              uint32_t rank = 0;
#pragma unroll
              for (int j = 0; j < SMALL_MAX; j++) {
                if ((uint32_t)j < nA) {
                  const bool less = hr[j].gstart < hr[i].gstart ||
                                    (hr[j].gstart == hr[i].gstart && (hr[j].minus < hr[i].minus || (hr[j].minus == hr[i].minus && hr[j].score > hr[i].score)));
                  const bool same = hr[j].gstart == hr[i].gstart && hr[j].minus == hr[i].minus && hr[j].score == hr[i].score;
                  rank += (less || (same && j < i)) ? 1u : 0u;
                }
              }
#pragma unroll
              for (int p = 0; p < SMALL_MAX; p++)
                if ((uint32_t)p == rank) { s_start[p] = hr[i].gstart; s_end[p] = hr[i].rh_end; s_score[p] = hr[i].score; s_cs[p] = hr[i].minus; s_arr[p] = i; }
            }
          }
          bool head[SMALL_MAX], keep[SMALL_MAX], done[SMALL_MAX];
#pragma unroll
          for (int p = 0; p < SMALL_MAX; p++) {
            head[p] = (uint32_t)p < nA; keep[p] = false; done[p] = false;
            bool stop = false;                                            // hits.hip prep_body: look back over the hits that can reach into this one
#pragma unroll
            for (int j = SMALL_MAX - 1; j >= 0; j--) {
              if (j < p && (uint32_t)p < nA && !stop) {
                if (s_start[j] + HIT_MAX_LEN - 1 - s_start[p] < a.max_overlap) stop = true;
                else if (s_cs[j] == s_cs[p] && s_end[j] - s_start[p] >= a.max_overlap) { head[p] = false; stop = true; }
              }
            }
          }
#pragma unroll
          for (uint32_t cs = 0; cs < 2; cs++) {                            // one (chromosome, strand) group after the other, left to right
            int curp = -1;                                                // position of the walk's current hit (cluster_body's `hit`)
            bool active = false;                                          // the walk started at a certain restart point
            int c_s = 0, c_e = 0, c_sc = 0;
#pragma unroll
            for (int p = 0; p < SMALL_MAX; p++) {
              if ((uint32_t)p < nA && s_cs[p] == cs) {
                if (head[p]) {
                  if (curp >= 0 && active) {                              // the cluster before ends: !more
#pragma unroll
                    for (int k = 0; k < SMALL_MAX; k++) if (k == curp) keep[k] = true;
                  }
                  active = (int64_t)s_start[p] >= known_from + HIT_MAX_LEN;
                  curp = p; c_s = s_start[p]; c_e = s_end[p]; c_sc = s_score[p];
                  if (active) done[p] = true;
                } else if (active) {
                  const int ov = max(0, min(s_end[p], c_e) - max(s_start[p], c_s));   // RH:141-144
                  done[p] = true;
                  if (!(ov >= a.max_overlap && s_score[p] <= c_sc)) {      // not swallowed: the walk goes on from here
                    if (ov < a.max_overlap) {
#pragma unroll
                      for (int k = 0; k < SMALL_MAX; k++) if (k == curp) keep[k] = true;
                    }
                    curp = p; c_s = s_start[p]; c_e = s_end[p]; c_sc = s_score[p];
                  }
                }
              }
            }
            if (curp >= 0 && active) {
#pragma unroll
              for (int k = 0; k < SMALL_MAX; k++) if (k == curp) keep[k] = true;
            }
          }
          // ---- 5. the bin's kept hits in final order ----
#pragma unroll
          for (int p = 0; p < SMALL_MAX; p++) {
            const unsigned long long okey = ((unsigned long long)c << 32) | (unsigned long long)(uint32_t)s_start[p];
            const bool mine = (uint32_t)p < nA && (int64_t)s_start[p] >= lo && (int64_t)s_start[p] < hi && okey >= a.own_lo && okey < a.own_hi;
            if (mine && !done[p]) out_flags |= BIN_FLAG_HALO;
            if (mine && keep[p]) {
              uint32_t rsrc = 0;
              HitRec hh{};
#pragma unroll
              for (int i = 0; i < SMALL_MAX; i++) if (s_arr[p] == i) { rsrc = acc_src[i]; hh = hr[i]; }
              const RawAln* rp = a.raw + rsrc;
              const GuideDev* gp = a.guides + rp->guide;
              const int pam = rp->pam;
              const int len = middle_length(rp, hh, gp->L, pam >= 0 ? gp->pam_len[pam] : 0, (int)m.rc.pu_len[pam + 1], (int)m.n_max, (int)m.mid_bound);
              if (len < 0) out_flags |= BIN_FLAG_ROW;
              else {
                const uint32_t name_len = m.name_off[hh.contig + 1] - m.name_off[hh.contig];
                out_bytes += m.rc.head_len + name_len + 1u + (uint32_t)len + m.rc.tail_len;
#pragma unroll
                for (int k = 0; k < SMALL_MAX; k++) if ((uint32_t)k == out_rows) out_row[k] = BinRow{rsrc, (uint32_t)len};
                out_rows++;
              }
            }
          }
          if (out_flags) { out_rows = 0; out_bytes = 0; out_acc = 0; }
        }
      }
    }
  }
  // ---- results: per bin (the list's bins are written by bin_hits_kernel), per chunk (one atomic per wave and sum: the 64 bins of a
  //      wave lie in one chunk), flags, the list ----
  if (in_range && !is_complex) {
    a.bin_rows[rel] = out_rows; a.bin_bytes[rel] = out_bytes;
#pragma unroll
    for (int k = 0; k < SMALL_MAX; k++) if ((uint32_t)k < out_rows) a.rows[(size_t)rel * BIN_ROWS + k] = out_row[k];
  }
  const unsigned long long wbytes = wave_sum_u64(out_bytes);
  const uint32_t wrows = (uint32_t)wave_sum_u64(out_rows), wacc = (uint32_t)wave_sum_u64(out_acc);
  const unsigned long long fl = __ballot(out_flags != 0);
  if (fl) { uint32_t f = out_flags; for (int off = 32; off > 0; off >>= 1) f |= (uint32_t)__shfl_xor((int)f, off); if (lane == 0) atomicOr(a.flags, f); }
  if (lane == 0) {
    const uint32_t rel0 = blockIdx.x * 64;
    if (wrows) {
      atomicAdd(a.chunk_rows + (rel0 >> CHUNK_SHIFT), wrows); atomicAdd(a.chunk_bytes + (rel0 >> CHUNK_SHIFT), wbytes);
      atomicAdd(a.super_bytes + (rel0 >> SUPER_SHIFT), wbytes);
    }
    if (wacc) atomicAdd(a.chunk_acc + (rel0 >> CHUNK_SHIFT), wacc);
  }
  const unsigned long long rm = __ballot(out_rows != 0);
  if (rm) {
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(a.rows_count, (uint32_t)__popcll(rm));
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (out_rows != 0) a.rows_list[base + (uint32_t)__popcll(rm & bits_below(lane))] = rel;
  }
  const unsigned long long cm = __ballot(is_complex);
  if (cm) {
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(sa.complex_count, (uint32_t)__popcll(cm));
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (is_complex) sa.complex_list[base + (uint32_t)__popcll(cm & bits_below(lane))] = rel;
  }
}

struct RowsArgs {
  RowConstDev rc;
  const char* names;
  char* text;
  unsigned long long text_cap;
  char* host_text;              // page-locked host memory: a text of up to host_cap bytes is written there, over the bus, and needs no copy
  unsigned long long host_cap;
  const uint32_t* counters;     // the lane's eight counters, posted with the result
  uint32_t* box;                // mailbox (device view)
  uint32_t seq;
  uint32_t n_chunks, n_supers;
  const uint32_t* complex_count;   // (statistics) bins that took a whole wave
};

// One wave per bin: where the bin's text starts = bytes of the chunks before its chunk + bytes of the bins before it in its chunk;
// then row by row: the middle part in the wave's LDS line (build_middle), head | chromosome | middle | tail at the row's final place.
// The first wave of the grid also posts the totals (final since bin_hits_kernel ended) to the host when it starts.
__global__ __launch_bounds__(64 * BIN_WAVES) void bin_rows_kernel(BinArgs a, MidArgs m, RowsArgs o) {
  CALITAS_TAIL_PRIO();
  __shared__ __attribute__((aligned(16))) uint8_t lds[BIN_WAVES * (MID_LINE + MID_FWD)];   // per wave: line | fwd
  const int lane = (int)(threadIdx.x & 63);
  const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  uint8_t* line = lds + wv * (MID_LINE + MID_FWD);
  uint8_t* fwd = line + MID_LINE;
  const uint8_t* blob = reinterpret_cast<const uint8_t*>(m.blob);     // constant strings straight from global memory (cache-resident)
  const uint32_t n_todo = *a.rows_count;                              // bins with rows (bin_hits_small_kernel / bin_hits_kernel listed them)
  const bool posts = blockIdx.x == 0;                                 // the first wave of the grid also posts the totals
  for (uint32_t it = blockIdx.x; it < n_todo || (posts && it == 0); it += gridDim.x) {
  const uint32_t rel = it < n_todo ? a.rows_list[it] : 0u;
  const uint32_t n = it < n_todo ? a.bin_rows[rel] : 0u;
  // where the bin's text starts: super-chunks before its super-chunk + chunks before its chunk + bins before it -- all loads independent
  const uint32_t my_chunk = rel >> CHUNK_SHIFT, my_super = rel >> SUPER_SHIFT;
  unsigned long long tot = 0, before = 0;
  for (uint32_t sc = (uint32_t)lane; sc < o.n_supers; sc += 64) {
    const unsigned long long v = a.super_bytes[sc];
    tot += v;
    if (sc < my_super) before += v;
  }
  for (uint32_t ch = (my_super << CHUNK_SHIFT) + (uint32_t)lane; ch < my_chunk; ch += 64) before += a.chunk_bytes[ch];
  for (uint32_t x = (my_chunk << CHUNK_SHIFT) + (uint32_t)lane; x < rel; x += 64) before += a.bin_bytes[x];
  tot = wave_sum_u64(tot); before = wave_sum_u64(before);
  const bool to_host = tot <= o.host_cap;                             // (the same for every wave of the grid; the host decides by the posted bytes)
  const uint32_t flags = *a.flags | (!to_host && tot > o.text_cap ? BIN_FLAG_TEXT : 0u);
  if (posts && it == 0) {
    uint32_t rows_tot = 0, acc_tot = 0;
    for (uint32_t ch = (uint32_t)lane; ch < o.n_chunks; ch += 64) { rows_tot += a.chunk_rows[ch]; acc_tot += a.chunk_acc[ch]; }
    rows_tot = (uint32_t)wave_sum_u64(rows_tot); acc_tot = (uint32_t)wave_sum_u64(acc_tot);
    if (lane == 0) {
      for (int i = 0; i < 8; i++) o.box[BIN_BOX_COUNTERS + i] = __hip_atomic_load(o.counters + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      o.box[BIN_BOX_ROWS] = rows_tot; o.box[BIN_BOX_BYTES] = (uint32_t)tot; o.box[BIN_BOX_BYTES + 1] = (uint32_t)(tot >> 32);
      o.box[BIN_BOX_FLAGS] = flags; o.box[BIN_BOX_ACCEPTED] = acc_tot; o.box[BIN_BOX_COMPLEX] = *o.complex_count;
      const unsigned long long now = (unsigned long long)wall_clock64();
      for (int i = 0; i < 3; i++) {
        const unsigned long long t = i < 2 ? a.stamps[i] : now;
        o.box[BIN_BOX_STAMPS + 2 * i] = (uint32_t)t; o.box[BIN_BOX_STAMPS + 2 * i + 1] = (uint32_t)(t >> 32);
      }
      __threadfence_system();
      __hip_atomic_store(o.box, o.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  if (flags != 0 || (a.dbg & 2u)) return;
  if (n == 0) continue;
  const uint8_t* head = reinterpret_cast<const uint8_t*>(m.blob) + o.rc.head_off;     // constant pieces straight from global memory (L2-resident)
  const uint8_t* tail = reinterpret_cast<const uint8_t*>(m.blob) + o.rc.tail_off;
  unsigned long long at = before;
  for (uint32_t k = 0; k < n; k++) {
    const auto* row = uniform_ptr(a.rows) + ((size_t)rel * BIN_ROWS + k);
    const uint32_t ridx = row->raw, want = row->len;
    const auto* rp = uniform_ptr(a.raw) + ridx;
    RowIn rin;
    const auto* ow = (const __attribute__((address_space(4))) uint32_t*)rp->ops;
    rin.w0 = ow[0]; rin.w1 = ow[1]; rin.w2 = ow[2]; rin.w3 = ow[3]; rin.w4 = ow[4];
    rin.n_ops = rp->n_ops; rin.pam = rp->pam; rin.offset = rp->offset; rin.pam_x = rp->pam_x;
    HitRec h;                                             // the same coordinates bin_hits_kernel derived (every lane computes them: uniform)
    {
      const HitRec hv = hit_record(a.raw + ridx, a.guides, a.win_base, a.win);
      h.contig = __builtin_amdgcn_readfirstlane(hv.contig); h.start = __builtin_amdgcn_readfirstlane(hv.start);
      h.end = __builtin_amdgcn_readfirstlane(hv.end); h.gstart = __builtin_amdgcn_readfirstlane(hv.gstart);
      h.gend = __builtin_amdgcn_readfirstlane(hv.gend); h.score = __builtin_amdgcn_readfirstlane(hv.score);
      h.rh_end = __builtin_amdgcn_readfirstlane(hv.rh_end); h.minus = (uint32_t)__builtin_amdgcn_readfirstlane((int)hv.minus);
    }
    const auto* gp = uniform_ptr(a.guides) + rp->guide;
    RowGuide g;
    g.L = gp->L; g.pam5 = gp->pam5; g.pam_len = rin.pam >= 0 ? gp->pam_len[rin.pam] : 0;
    wave_lds_sync();                                      // the copy-out of the previous row is done with line[]
    const int len = build_middle<true>(line, fwd, m, blob, rin, h, g, lane);
    if (len < 0 || (uint32_t)len != want) {               // cannot happen: both kernels run the same arithmetic
      if (lane == 0) __hip_atomic_fetch_or(o.box + BIN_BOX_LATE, BIN_FLAG_INTERNAL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return;
    }
    wave_lds_sync();
    const uint32_t nb = uniform_ptr(m.name_off)[h.contig], nl = uniform_ptr(m.name_off)[h.contig + 1] - nb;
    const uint32_t s0 = o.rc.head_len, s1 = s0 + nl + 1, s2 = s1 + (uint32_t)len, total = s2 + o.rc.tail_len;
    char* dst = (to_host ? o.host_text : o.text) + at;
    for (uint32_t x = (uint32_t)lane; x < total; x += 64) {
      uint8_t ch;
      if (x < s0) ch = head[x];
      else if (x < s1) ch = (x - s0 < nl) ? (uint8_t)o.names[nb + x - s0] : (uint8_t)'\t';
      else if (x < s2) ch = line[x - s1];
      else ch = tail[x - s2];
      dst[x] = (char)ch;
    }
    at += total;
  }
  }
}

template <typename T>
hipError_t grow_to(T** p, size_t& cap, size_t need) {
  if (need <= cap) return hipSuccess;
  (void)hipFree(*p); *p = nullptr; cap = 0;
  hipError_t e = hipMalloc((void**)p, need * sizeof(T));
  if (e == hipSuccess) cap = need;
  return e;
}

}  // namespace

struct BinnedWork {
  uint32_t* bin_idx = nullptr; size_t bin_idx_cap = 0;
  BinRow* rows = nullptr; size_t rows_cap = 0;
  // one allocation for everything that is cleared per call: bin_count[n_bins] | chunk_bytes | chunk_rows | chunk_acc | flags
  uint8_t* clear = nullptr; size_t clear_cap = 0;
  uint32_t *bin_count = nullptr, *chunk_rows = nullptr, *chunk_acc = nullptr, *flags = nullptr;
  unsigned long long *chunk_bytes = nullptr, *super_bytes = nullptr;
  uint32_t *bin_rows = nullptr, *bin_bytes = nullptr; size_t bin_rows_cap = 0, bin_bytes_cap = 0;
  uint32_t* rows_list = nullptr; size_t rows_list_cap = 0;    // bins with rows
  uint32_t* rows_count = nullptr;                             // (inside `clear`)
  uint32_t* complex_list = nullptr; size_t complex_cap = 0;   // bins bin_hits_small_kernel leaves to bin_hits_kernel
  uint32_t* complex_count = nullptr;                          // (inside `clear`)
  uint32_t n_bins = 0, n_chunks = 0, n_supers = 0;
  // A short text (most calls on a bacterial genome, a rare guide on a slice of a large one) is written straight into page-locked host
  // memory by the rows kernel: the device-to-host copy of such a call -- waiting for the kernel, starting the copy engine, waiting
  // for it -- cost 25 of its 165 us.
  char* host_text = nullptr; unsigned long long host_cap = 0, host_alloc = 0;
  unsigned long long* stamps = nullptr;                       // (BinArgs::stamps)
  double stamp_khz = 0;                                       // the wall clock's rate
};

double binned_stamp_ms(const BinnedWork* w, const Mailbox& box, int from, int to) {
  if (!w || !box.host || w->stamp_khz <= 0) return 0;
  auto at = [&](int i) { return (unsigned long long)box.host[BIN_BOX_STAMPS + 2 * i] | ((unsigned long long)box.host[BIN_BOX_STAMPS + 2 * i + 1] << 32); };
  const unsigned long long a = at(from), b = at(to);
  return b > a ? (double)(b - a) / w->stamp_khz : 0.0;
}

const char* binned_host_text(const BinnedWork* w) { return w ? w->host_text : nullptr; }
unsigned long long binned_host_cap(const BinnedWork* w) { return w && w->host_text ? w->host_cap : 0; }

void binned_destroy(BinnedWork* w) {
  if (!w) return;
  if (w->host_text) (void)hipHostFree(w->host_text);
  (void)hipFree(w->stamps);
  (void)hipFree(w->bin_idx); (void)hipFree(w->rows); (void)hipFree(w->clear); (void)hipFree(w->bin_rows); (void)hipFree(w->bin_bytes); (void)hipFree(w->complex_list); (void)hipFree(w->rows_list);
  delete w;
}

int binned_shift(int window_size) {
  if (window_size < 2 * HIT_MAX_LEN) return 0;
  int shift = 13;                                                     // 8 kb: room for two default windows of context on the left
  while (shift < 20 && ((int64_t)1 << shift) < 2 * (int64_t)window_size + 4 * HIT_MAX_LEN) shift++;
  return ((int64_t)1 << shift) >= 2 * (int64_t)window_size + 4 * HIT_MAX_LEN ? shift : 0;
}

#define TRY(x) do { e = (x); if (e != hipSuccess) return e; } while (0)

hipError_t binned_prepare(BinnedWork** pw, uint32_t n_bins, hipStream_t stream) {
  void* clear = nullptr;
  size_t bytes = 0;
  hipError_t e = binned_prepare_host(pw, n_bins, &clear, &bytes);
  return e == hipSuccess ? hipMemsetAsync(clear, 0, bytes, stream) : e;
}

hipError_t binned_prepare_host(BinnedWork** pw, uint32_t n_bins, void** clear, size_t* clear_bytes) {
  if (!*pw) *pw = new BinnedWork();
  BinnedWork& w = **pw;
  hipError_t e;
  const uint32_t n_chunks = (n_bins >> CHUNK_SHIFT) + 1;
  TRY(grow_to(&w.bin_idx, w.bin_idx_cap, (size_t)n_bins * BIN_CAP));
  TRY(grow_to(&w.rows, w.rows_cap, (size_t)n_bins * BIN_ROWS));
  TRY(grow_to(&w.bin_rows, w.bin_rows_cap, (size_t)n_bins));
  TRY(grow_to(&w.bin_bytes, w.bin_bytes_cap, (size_t)n_bins));
  TRY(grow_to(&w.complex_list, w.complex_cap, (size_t)n_bins));
  TRY(grow_to(&w.rows_list, w.rows_list_cap, (size_t)n_bins));
  // the 64-bit sums first, then the 32-bit arrays; the whole block is a multiple of 16 bytes (MI355X_MICROARCH: memset sizes)
  const uint32_t n_supers = (n_bins >> SUPER_SHIFT) + 1;
  const size_t bytes = (((size_t)(n_chunks + n_supers) * 8 + (size_t)n_chunks * 4 * 2 + (size_t)n_bins * 4 + 12) + 15) & ~(size_t)15;
  TRY(grow_to(&w.clear, w.clear_cap, bytes));
  w.chunk_bytes = reinterpret_cast<unsigned long long*>(w.clear);
  w.super_bytes = w.chunk_bytes + n_chunks;
  w.chunk_rows = reinterpret_cast<uint32_t*>(w.super_bytes + n_supers);
  w.chunk_acc = w.chunk_rows + n_chunks;
  w.flags = w.chunk_acc + n_chunks;
  w.complex_count = w.flags + 1;
  w.rows_count = w.flags + 2;
  w.bin_count = w.flags + 3;
  w.n_bins = n_bins; w.n_chunks = n_chunks; w.n_supers = n_supers;
  unsigned long long host_want = BIN_HOST_TEXT;
  if (const char* env = TUNE_GET("CALITAS_BINNED_HOST_TEXT_KB")) host_want = (unsigned long long)std::max(0, std::atoi(env)) << 10;   // (experiments)
  if (w.host_text && w.host_alloc < host_want) { (void)hipHostFree(w.host_text); w.host_text = nullptr; }
  if (!w.host_text && host_want) {
    if (hipHostMalloc((void**)&w.host_text, host_want, hipHostMallocDefault) != hipSuccess) {   // (coherent: the device writes through)
      w.host_text = nullptr;                                 // not an error: every text takes the copy then
      (void)hipGetLastError();
    } else w.host_alloc = host_want;
  }
  w.host_cap = host_want;
  if (!w.stamps) {
    TRY(hipMalloc((void**)&w.stamps, 4 * sizeof(unsigned long long)));
    int dev = 0, khz = 0;
    TRY(hipGetDevice(&dev));
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess || khz <= 0) { khz = 100000; (void)hipGetLastError(); }   // 100 MHz on CDNA
    w.stamp_khz = khz;
  }
  // (tests: CALITAS_BINNED_TEXT_KB forces the regrow path of the device buffer, CALITAS_BINNED_HOST_TEXT=0 the copy for every text)
  if (TUNE_GET("CALITAS_BINNED_TEXT_KB")) w.host_cap = 0;
  if (const char* env = TUNE_GET("CALITAS_BINNED_HOST_TEXT")) w.host_cap = std::atoi(env) != 0 ? host_want : 0;
  *clear = w.clear; *clear_bytes = bytes;
  return hipSuccess;
}

void binned_fill_align_args(const BinnedWork* w, const BinnedGeometry& geo, AlignArgs& aa) {
  aa.bin_idx = w->bin_idx; aa.bin_count = w->bin_count; aa.bin_base = geo.d_bin_base; aa.bin_first = geo.bin_first; aa.bin_n = geo.n_bins;
  aa.bin_shift = geo.bin_shift;
  aa.bin_cap = BIN_CAP;
  aa.stamps = w->stamps;
}

const char* binned_text(const HitsWork* hits) { return hits ? hits->text : nullptr; }

static hipError_t launch_rows(BinnedWork& w, HitsWork& hw, const BinArgs& ba, const MidArgs& ma, const uint32_t* d_counters, hipStream_t stream,
                              Mailbox* post, hipEvent_t ev_start, hipEvent_t ev_done, char* host_dst = nullptr, unsigned long long host_dst_cap = 0) {
  hipError_t e;
  TRY(mailbox_open(*post));
  RowsArgs ro{};
  ro.rc = hw.rc; ro.names = hw.names; ro.text = hw.text; ro.text_cap = hw.text_cap; ro.host_text = w.host_text; ro.host_cap = w.host_text ? w.host_cap : 0;
  if (host_dst) { ro.host_text = host_dst; ro.host_cap = host_dst_cap; }   // the caller's page-locked destination: the text's final place
  ro.counters = d_counters; ro.box = post->dev; ro.seq = ++post->seq;
  ro.n_chunks = w.n_chunks; ro.n_supers = w.n_supers; ro.complex_count = w.complex_count;
  post->host[BIN_BOX_LATE] = 0;                              // raised by any wave while rows are written; read when the stream is done
  const unsigned grid = std::min<uint32_t>(std::max<uint32_t>(w.n_bins, 1u), 16384u);      // strides over the list of bins with rows
  hipExtLaunchKernelGGL(bin_rows_kernel, dim3(grid), dim3(64), 0, stream, ev_start, ev_done, 0, ba, ma, ro);
  return hipGetLastError();
}

static void fill_args(BinnedWork& w, HitsWork& hw, const BinnedGeometry& geo, const HitsRef& ref, const RawAln* d_raw, const GuideDev* d_guides,
                      const uint64_t* d_win_base, const int2* d_win, const BinnedParams& p, BinArgs& ba, MidArgs& ma) {
  ba = BinArgs{};
  ba.raw = d_raw; ba.bin_idx = w.bin_idx; ba.bin_count = w.bin_count; ba.bin_base = geo.d_bin_base; ba.bin_contig = geo.d_bin_contig; ba.n_contigs = geo.n_contigs; ba.bin_first = geo.bin_first;
  ba.n_bins = geo.n_bins; ba.bin_shift = geo.bin_shift; ba.guides = d_guides; ba.win_base = d_win_base; ba.win = d_win;
  ba.W = p.window_size; ba.step = p.step; ba.max_total_diffs = p.max_total_diffs; ba.max_overlap = p.max_overlap;
  ba.own_lo = p.own_lo; ba.own_hi = p.own_hi;
  ba.rows = w.rows; ba.bin_rows = w.bin_rows; ba.bin_bytes = w.bin_bytes; ba.chunk_bytes = w.chunk_bytes; ba.super_bytes = w.super_bytes; ba.chunk_rows = w.chunk_rows;
  ba.chunk_acc = w.chunk_acc; ba.flags = w.flags; ba.rows_list = w.rows_list; ba.rows_count = w.rows_count; ba.stamps = w.stamps;
  const uint32_t n_max = (uint32_t)std::min<int>(MID_COLS, std::max(1, p.max_ops));
  ma = MidArgs{};
  ma.ref = ref; ma.rc = hw.rc; ma.blob = hw.blob; ma.name_off = hw.name_off; ma.guides = d_guides;
  ma.mid_bound = (6 * n_max + 128 + 3) & ~3u; ma.n_max = n_max; ma.blob_bytes = (uint32_t)hw.blob_bytes;
}

hipError_t binned_run(BinnedWork* pw, HitsWork** phw, const BinnedGeometry& geo, const HitsRef& ref, const RawAln* d_raw, const GuideDev* d_guides,
                      const uint64_t* d_win_base, const int2* d_win, const BinnedParams& p, const uint32_t* d_counters, hipStream_t stream,
                      Mailbox* post, hipEvent_t ev_hits_done, hipEvent_t ev_rows_start, hipEvent_t ev_rows_done, bool with_rows) {
  if (!pw || !*phw || !(*phw)->prepared) return hipErrorInvalidValue;
  BinnedWork& w = *pw;
  HitsWork& hw = **phw;
  hw.prepared = false;
  hipError_t e;
  {
    size_t first_guess = (size_t)32 << 20;                     // BIN_FLAG_TEXT asks for more
    if (const char* env = TUNE_GET("CALITAS_BINNED_TEXT_KB")) first_guess = (size_t)std::max(1, std::atoi(env)) << 10;   // tests: force the regrow path
    if (hw.text_cap < first_guess) TRY(grow(&hw.text, hw.text_cap, first_guess));
  }
  BinArgs ba; MidArgs ma;
  fill_args(w, hw, geo, ref, d_raw, d_guides, d_win_base, d_win, p, ba, ma);
  // (a null table here would be a wild read on the device, not an error code: refuse on the host)
  if (!ba.raw || !ba.bin_idx || !ba.bin_count || !ba.bin_base || !ba.bin_contig || !ba.guides || !ba.win_base || !ba.win || !ba.rows || !ba.bin_rows ||
      !ba.bin_bytes || !ba.chunk_bytes || !ba.super_bytes || !ba.chunk_rows || !ba.chunk_acc || !ba.flags || !ba.rows_list || !ba.rows_count || !ba.stamps || !ma.blob || !ma.name_off || !hw.names ||
      !hw.text || w.n_bins < geo.n_bins)
    return hipErrorInvalidValue;
  SmallArgs sa{w.complex_list, w.complex_count, 0u};
  if (const char* env = TUNE_GET("CALITAS_BINNED_COMPLEX")) sa.force_complex = std::atoi(env) != 0;   // tests: the wave-per-bin kernel for every bin
  if (!sa.complex_list || !sa.complex_count) return hipErrorInvalidValue;
  // (the listed bins inside the lane kernel -- its waves doing the bins their lanes left over, one launch less -- measured slower at
  // every size: 0.176 against 0.165 ms for an E. coli-sized call, 0.685 against 0.588 ms for an eighth of the hg38-sized genome: the
  // lanes of a wave that does a listed bin wait for it, and the listed bins of a wave run one after the other)
  const dim3 sgrid((std::max<uint32_t>(geo.n_bins, 1u) + 63) / 64);
  hipLaunchKernelGGL(bin_hits_small_kernel, sgrid, dim3(64), 0, stream, ba, ma, sa);
  TRY(hipGetLastError());
  // the listed bins: a fixed grid that strides over the list (its length is on the device)
  const unsigned grid = std::min<uint32_t>(std::max<uint32_t>(geo.n_bins, 1u), 1024u);
  int skip = 0;                                            // timing experiments only (the text is wrong): 1 = no wave-per-bin kernel, 2 = rows kernel posts and returns
#ifdef CALITAS_EXPERIMENTS
  if (const char* env = TUNE_GET("CALITAS_BINNED_SKIP")) skip = std::atoi(env);
#endif
  ba.dbg = (skip & 2) ? 2u : 0u;
  if (!(skip & 1))
    hipExtLaunchKernelGGL(bin_hits_kernel, dim3(grid), dim3(64), 0, stream, nullptr, ev_hits_done, 0, ba, ma, (const uint32_t*)w.complex_list,
                          (const uint32_t*)w.complex_count);
  TRY(hipGetLastError());
  if (!with_rows) return hipSuccess;                         // (the caller launches them itself: binned_rows)
  return launch_rows(w, hw, ba, ma, d_counters, stream, post, ev_rows_start, ev_rows_done);
}

hipError_t binned_rows(BinnedWork* pw, HitsWork** phw, const BinnedGeometry& geo, const HitsRef& ref, const RawAln* d_raw, const GuideDev* d_guides,
                       const uint64_t* d_win_base, const int2* d_win, const BinnedParams& p, const uint32_t* d_counters, hipStream_t stream,
                       Mailbox* post, hipEvent_t ev_rows_done, char* host_dst, unsigned long long host_dst_cap) {
  if (!pw || !*phw) return hipErrorInvalidValue;
  BinArgs ba; MidArgs ma;
  fill_args(*pw, **phw, geo, ref, d_raw, d_guides, d_win_base, d_win, p, ba, ma);
  int skip = 0;
#ifdef CALITAS_EXPERIMENTS
  if (const char* env = TUNE_GET("CALITAS_BINNED_SKIP")) skip = std::atoi(env);
#endif
  ba.dbg = (skip & 2) ? 2u : 0u;
  return launch_rows(*pw, **phw, ba, ma, d_counters, stream, post, nullptr, ev_rows_done, host_dst, host_dst_cap);
}

hipError_t binned_rerun_rows(BinnedWork* pw, HitsWork** phw, const BinnedGeometry& geo, const HitsRef& ref, const RawAln* d_raw, const GuideDev* d_guides,
                             const uint64_t* d_win_base, const int2* d_win, const BinnedParams& p, const uint32_t* d_counters,
                             uint64_t bytes, hipStream_t stream, Mailbox* post, hipEvent_t ev_rows_done) {
  if (!pw || !*phw) return hipErrorInvalidValue;
  BinnedWork& w = *pw;
  HitsWork& hw = **phw;
  hipError_t e;
  TRY(hipStreamSynchronize(stream));                           // nobody is writing the old buffer any more
  TRY(grow(&hw.text, hw.text_cap, (size_t)bytes + (size_t)(bytes / 4)));
  BinArgs ba; MidArgs ma;
  fill_args(w, hw, geo, ref, d_raw, d_guides, d_win_base, d_win, p, ba, ma);
  return launch_rows(w, hw, ba, ma, d_counters, stream, post, nullptr, ev_rows_done);
}

#undef TRY

}  // namespace calitas
