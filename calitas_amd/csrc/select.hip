// select.hip -- the per-window stage of SequentialGuideAligner.align (SequentialGuideAligner.scala:315-320) on the GPU.
//
// trace_kernel appends extended alignments in arbitrary order.  Every alignment belongs to exactly one window, and windows have a
// dense global index (guide x all windows of all contigs), so grouping is a counting sort on that index, not a comparison sort:
//   1. count_kernel    per alignment: what the filter looks at (score, gap bases, edits, contig start / end, GA:100-101,119-122),
//                      its place in the reference's enumeration order inside the window (strand list, end column, PAM: fgbio
//                      emits ascending end columns, extendAndFilterRight keeps PAM order), and cnt[window]++;
//   2. exclusive scan of cnt over all windows (rocPRIM) -> first slot of every window;
//   3. scatter_kernel  alignment indices into their window's slots (cnt counts back down to zero: it needs no clearing between
//                      calls);
//   4. filter_kernel   one lane per window (the lane of the window's first slot): per strand list, repeatedly the best remaining alignment -- score desc, gap bases asc,
//                      enumeration order on ties = the reference's stable sort (GA:125-129) --, kept if edits <= maxTotalDiffs and
//                      no kept alignment of the same list overlaps it by more than maxOverlap;
//   5. exclusive scan of the kept counts, gather_kernel: accepted alignments in (guide, contig, window, output) order.
// Windows with more than LANE_MAX alignments (planted or repeated sites, PAM-less dense searches) go to filter_wave_kernel, one wave
// per window (in registers up to GROUP_MAX of them); only beyond 16 384 records or 512 kept alignments per strand in one window
// does a flag send the caller to the host implementation of the same stage for this search.  (A first version radix-sorted 64-bit keys: eight more launches per call,
// which is what a lane's tail is made of -- DESIGN.md 4.5.)
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "common.hpp"
#include "select.hpp"
#include "select_dev.hpp"

namespace calitas {

namespace {

constexpr uint32_t LANE_MAX = 8;      // records of a window one lane filters on its own (filter_kernel); larger windows get a wave
constexpr uint32_t GROUP_MAX = 256;   // records of a window a wave holds in registers (filter_wave_kernel); beyond: its global-memory loop

__global__ void count_kernel(const RawAln* raw, uint32_t n, const GuideDev* guides, const uint64_t* win_base, const int2* win,
                             uint32_t window_lo, uint32_t windows_per_guide, Derived* der, uint32_t* cnt, uint32_t* counts) {
  CALITAS_TAIL_PRIO();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) { counts[0] = 0; counts[1] = 0; counts[2] = 0; counts[3] = 0; }   // survivors, flags, listed windows, crowded windows: nobody reads them before step 4
  if (i >= n) return;
  const Derived d = derive(raw + i, guides, win_base, win, window_lo, windows_per_guide);
  der[i] = d;
  atomicAdd(&cnt[d.widx], 1u);
}

// ders[] = der[] in slot order: the filter kernels read a window's records as one contiguous run
__global__ void scatter_kernel(const Derived* der, uint32_t n, const uint32_t* offs, uint32_t* cnt, uint32_t* slot, Derived* ders,
                               uint8_t* taken, uint32_t* counts) {
  CALITAS_TAIL_PRIO();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const Derived d = der[i];
  const uint32_t pos = offs[d.widx] + atomicSub(&cnt[d.widx], 1u) - 1u;
  // cnt[] must have been all zero when count_kernel started (new buffers are filled with 0xFF and then cleared on the using
  // stream, see grow()): a clear that is missing or ordered wrongly shows here as a slot outside [0, n) -- reported, not written
  if (pos >= n) { atomicOr(&counts[1], SELECT_FLAG_INTERNAL); return; }
  slot[pos] = i;
  ders[pos] = d;
  taken[i] = 0;                                        // indexed by slot position below; any permutation clears all n
}

// One lane per slot position; only the first slot of a window works (a lane per window would leave 97 % of the lanes of an
// hg38-sized table idle).  kept[s] = number of survivors of the window whose slots start at s (0 for the other positions);
// their alignment indices, in output order, go to out_idx[s .. s + kept[s]).
__global__ __launch_bounds__(256) void filter_kernel(const Derived* ders, const uint32_t* offs, uint32_t n, int max_total_diffs, int max_overlap, uint8_t* taken,
                                                     uint32_t* kept, uint32_t* out_pos, uint32_t* counts, uint32_t* big) {
  CALITAS_TAIL_PRIO();
  // The block's 256 records (a window's records are consecutive slots) and their "taken" flags are staged in LDS: the greedy below
  // visits every record of the window once per round, and as loads from global memory those visits were a chain of dependent round
  // trips (33-60 us for windows of two or three records).  Slots beyond the block (a window that starts here and ends in the next
  // block) are read from global memory as before; only this window's lane touches them.
  __shared__ Derived s_d[256];
  __shared__ unsigned long long s_key[256];         // the greedy's order as one number per record (0 = taken), see order_key()
  __shared__ uint32_t s_out[256];                   // out_pos[] of the block's slots: the clash test reads what this lane just wrote
  const uint32_t base = blockIdx.x * blockDim.x, s = base + threadIdx.x;
  if (s < n) { const Derived d = ders[s]; s_d[threadIdx.x] = d; s_key[threadIdx.x] = order_key(d); }
  __syncthreads();
  if (s >= n) return;
  const uint32_t w = s_d[threadIdx.x].widx;
  if (offs[w] != s) { kept[s] = 0; return; }
  const uint32_t e = offs[w + 1];
  // A window of more than LANE_MAX records is left to filter_wave_kernel (the greedy below is quadratic in the window's size, and
  // alone in its wave for that long: a PAM-less search at eight differences has ~170 alignments per window and spent 3.6 of its
  // 4.5 device-seconds here).  One append per wave.
  {
    const bool listed = e - s > LANE_MAX;
    const unsigned long long lm = __ballot(listed);
    if (lm) {
      const int lane = (int)(threadIdx.x & 63), leader = __ffsll((long long)lm) - 1;
      uint32_t at = 0;
      if (lane == leader) at = atomicAdd(counts + 2, (uint32_t)__popcll(lm));
      at = __shfl(at, leader);
      if (listed) {
        big[at + (uint32_t)__popcll(lm & ((1ull << lane) - 1ull))] = s;
        kept[s] = 0;
        if (e - s > GROUP_MAX) atomicAdd(counts + 3, 1u);
        return;
      }
    }
  }
  const uint32_t lim = base + 256;                  // slots [base, lim) are in LDS
  uint32_t nk = 0;
  for (uint32_t list = 0; list < 2; list++) {
    const uint32_t first_kept = nk;                   // overlaps are only tested against the same strand (SGA:317)
    for (uint32_t round = s; round < e; round++) {
      // best remaining record of this list: the largest key.  One LDS read per record -- the loop is a chain of LDS round trips
      // and the whole stage hangs on its length (it read the taken flag, then the fields: 30 us for a window of six records).
      unsigned long long bk = 0;
      uint32_t best = 0;
      for (uint32_t m = s; m < e; m++) {
        unsigned long long k;
        if (m < lim) k = s_key[m - base];               // two explicit branches: one generic pointer would make every access a flat load
        else k = taken[m] ? 0ull : order_key(ders[m]);
        if ((uint32_t)(k >> 63) != list || k == 0) continue;
        if (k > bk) { bk = k; best = m; }
      }
      if (bk == 0) break;
      int b_start, b_end, b_edits;
      if (best < lim) { s_key[best - base] = 0; b_start = s_d[best - base].start; b_end = s_d[best - base].end; b_edits = s_d[best - base].edits; }
      else { taken[best] = 1; b_start = ders[best].start; b_end = ders[best].end; b_edits = ders[best].edits; }
      if (b_edits > max_total_diffs) continue;
      bool clash = false;
      for (uint32_t k = first_kept; k < nk; k++) {
        const uint32_t kp = s + k < lim ? s_out[s + k - base] : out_pos[s + k];
        int ks, ke;
        if (kp < lim) { ks = s_d[kp - base].start; ke = s_d[kp - base].end; } else { ks = ders[kp].start; ke = ders[kp].end; }
        const int o = min(b_end, ke) - max(b_start, ks);   // GA:119-122
        if (o > max_overlap) { clash = true; break; }
      }
      if (!clash) {
        if (s + nk < lim) s_out[s + nk - base] = best;
        out_pos[s + nk++] = best;
      }
    }
  }
  kept[s] = nk;
}

// Windows with more than LANE_MAX records: one wave per window, same greedy.
//   up to GROUP_MAX records: every lane holds four of them in registers; a round is a wave-wide maximum of the order keys, the
//     winner's interval broadcast from its lane, and the clash test spread over the lanes (kept intervals in LDS);
//   beyond (satellite repeats): the records stay in global memory, "taken" bits and kept intervals in LDS; a window beyond those
//     capacities raises the flag.
// LDS is what a workgroup of this kernel has to wait for while the scan of the next lane fills the CUs: kept to 6 KB, because the
// kernel is launched on every call and usually has little to do.
constexpr uint32_t BIG_MAX = 1u << 14;     // records per window
constexpr uint32_t BIG_KEPT = 512;         // kept alignments per strand list
static_assert(GROUP_MAX == 4 * 64 && GROUP_MAX <= BIG_KEPT, "filter_wave_kernel holds four records per lane");

__global__ __launch_bounds__(64) void filter_wave_kernel(const Derived* ders, const uint32_t* offs, int max_total_diffs, int max_overlap,
                                                         uint32_t* kept, uint32_t* out_pos, uint32_t* counts, const uint32_t* big) {
  CALITAS_TAIL_PRIO();
  __shared__ uint32_t s_taken[BIG_MAX / 32];
  __shared__ int s_ks[BIG_KEPT], s_ke[BIG_KEPT];
  const uint32_t lane = threadIdx.x;
  const uint32_t n_big = counts[2];
  for (uint32_t g = blockIdx.x; g < n_big; g += gridDim.x) {
    const uint32_t s = big[g];
    const uint32_t e = offs[ders[s].widx + 1];
    if (e - s > BIG_MAX) { if (lane == 0) { atomicOr(counts + 1, 1u); kept[s] = 0; } continue; }
    if (e - s <= GROUP_MAX) {
      // ---- the window in registers: lane holds records s + lane + 64 q ----
      unsigned long long key[4];
      int st[4], en[4], ed[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t m = s + lane + 64u * (uint32_t)q;
        key[q] = 0; st[q] = 0; en[q] = 0; ed[q] = 0;
        if (m < e) { const Derived d = ders[m]; key[q] = order_key(d); st[q] = d.start; en[q] = d.end; ed[q] = d.edits; }
      }
      uint32_t nk = 0;
      for (uint32_t list = 0; list < 2; list++) {
        const uint32_t first_kept = nk;                 // overlaps are only tested against the same strand (SGA:317)
        for (;;) {
          unsigned long long mine = 0;                  // this lane's best remaining record of the list (keys are unique and not 0)
#pragma unroll
          for (int q = 0; q < 4; q++) if ((uint32_t)(key[q] >> 63) == list && key[q] > mine) mine = key[q];
          const unsigned long long bk = wave_max_u64(mine);
          if (bk == 0) break;
          const int owner = __ffsll((long long)__ballot(mine == bk)) - 1;
          int b_start = 0, b_end = 0, b_edits = 0, b_q = 0;
#pragma unroll
          for (int q = 0; q < 4; q++) if (key[q] == bk) { b_start = st[q]; b_end = en[q]; b_edits = ed[q]; b_q = q; key[q] = 0; }   // the owner takes it
          b_start = __shfl(b_start, owner); b_end = __shfl(b_end, owner); b_edits = __shfl(b_edits, owner); b_q = __shfl(b_q, owner);
          if (b_edits > max_total_diffs) continue;
          bool clash = false;
          for (uint32_t k = first_kept + lane; k < nk; k += 64) {
            const int o = min(b_end, s_ke[k]) - max(b_start, s_ks[k]);   // GA:119-122
            clash = clash || o > max_overlap;
          }
          if (__ballot(clash) == 0) {
            if (lane == 0) { out_pos[s + nk] = s + (uint32_t)owner + 64u * (uint32_t)b_q; s_ks[nk] = b_start; s_ke[nk] = b_end; }
            nk++;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // one wave: the LDS writes above before the next round's reads
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          }
        }
      }
      if (lane == 0) kept[s] = nk;
      continue;
    }
    for (uint32_t i = lane; i < (e - s + 31) / 32; i += 64) s_taken[i] = 0;
    __syncthreads();
    uint32_t nk = 0;
    bool overflow = false;
    for (uint32_t list = 0; list < 2 && !overflow; list++) {
      const uint32_t first_kept = nk;
      for (;;) {
        // best remaining record of this list; the order is total (ekey is unique inside a window), so a max-reduction of a
        // packed key finds it: score | inverted gap bases | inverted enumeration key | (the slot follows separately)
        unsigned long long bk = 0;
        uint32_t bm = 0;
        for (uint32_t m = s + lane; m < e; m += 64) {
          const uint32_t rel = m - s;
          if ((s_taken[rel >> 5] >> (rel & 31)) & 1u) continue;
          const Derived d = ders[m];
          if ((d.ekey >> 19) != list) continue;
          const unsigned long long k = ((unsigned long long)(uint32_t)(d.score + (1 << 22)) << 40) | ((unsigned long long)(0xFFFFu - d.gaps) << 24) |
                                       (unsigned long long)(0xFFFFFu - (d.ekey & 0x7FFFFu)) << 1 | 1ull;
          if (k > bk) { bk = k; bm = m; }
        }
        for (int off = 32; off > 0; off >>= 1) {
          const unsigned long long ok = __shfl_xor(bk, off);
          const uint32_t om = __shfl_xor(bm, off);
          if (ok > bk) { bk = ok; bm = om; }
        }
        if (bk == 0) break;
        if (lane == 0) { const uint32_t rel = bm - s; s_taken[rel >> 5] |= 1u << (rel & 31); }
        const uint32_t best = bm;
        const Derived b = ders[best];
        if ((int)b.edits <= max_total_diffs) {
          bool mine = false;
          for (uint32_t k = first_kept + lane; k < nk; k += 64) {
            const int o = min(b.end, s_ke[k - first_kept]) - max(b.start, s_ks[k - first_kept]);   // GA:119-122
            mine = mine || o > max_overlap;
          }
          if (__ballot(mine) == 0) {
            if (nk - first_kept >= BIG_KEPT) { overflow = true; break; }
            if (lane == 0) { out_pos[s + nk] = best; s_ks[nk - first_kept] = b.start; s_ke[nk - first_kept] = b.end; }
            nk++;
          }
        }
        __syncthreads();
      }
    }
    if (lane == 0) { if (overflow) { atomicOr(counts + 1, 1u); kept[s] = 0; } else kept[s] = nk; }
    __syncthreads();
  }
}

// The whole stage for at most SELECT_SMALL raw alignments in one workgroup (an E. coli-sized call has ~45): grouping by window is a
// rank by counting in LDS instead of the dense per-window counters and their scan over the window table, the greedy is filter_kernel's
// on LDS arrays, the compaction a scan in LDS -- one launch instead of nine, and between dependent launches lie 4-10 us each.  A window
// with more than SMALL_WINDOW_MAX records (the greedy is quadratic and one lane runs it) raises SELECT_FLAG_RETRY: the caller runs the
// general kernels for this call.
constexpr uint32_t SELECT_SMALL = 1024, SMALL_WINDOW_MAX = 32;

__global__ __launch_bounds__(SELECT_SMALL) void select_small_kernel(const RawAln* raw, uint32_t n, const GuideDev* guides, const uint64_t* win_base,
                                                                    const int2* win, uint32_t window_lo, uint32_t windows_per_guide,
                                                                    int max_total_diffs, int max_overlap, RawAln* final_out, uint32_t* counts,
                                                                    uint32_t* box, uint32_t seq, const uint32_t* ctr, uint32_t rec_cap,
                                                                    uint32_t raw_cap, uint32_t item_cap) {
  CALITAS_TAIL_PRIO();
  // ctr: launched right behind trace_kernel, before the host has seen the call's counters (records, raw alignments, anomalies, passing
  // candidates, ...: eight words).  The kernel takes the number of alignments from there, checks what the host would have checked
  // (any overflow or anomaly, or more alignments than it handles: SELECT_FLAG_RETRY and nothing else done) and posts the counters
  // together with its own three counts: one host round trip instead of two on the path of a small call.
  if (ctr) {
    n = ctr[1];
    if (ctr[0] > rec_cap || n > raw_cap || ctr[3] > item_cap || ctr[2] != 0 || n > SELECT_SMALL) {
      if (threadIdx.x == 0) {
        counts[0] = 0; counts[1] = SELECT_FLAG_RETRY; counts[2] = 0; counts[3] = 0;
        for (int k = 0; k < 8; k++) box[1 + k] = ctr[k];
        box[9] = 0; box[10] = SELECT_FLAG_RETRY; box[11] = 0;
        __threadfence_system();
        __hip_atomic_store(box, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      }
      return;
    }
  }
  __shared__ uint32_t s_widx[SELECT_SMALL];             // by arrival
  __shared__ Derived s_d[SELECT_SMALL];                 // from here on by (window, arrival)
  __shared__ unsigned long long s_key[SELECT_SMALL];    // order_key(), 0 = taken
  __shared__ uint32_t s_src[SELECT_SMALL];              // arrival index
  __shared__ uint32_t s_out[SELECT_SMALL];              // per window, from its first position: the kept positions in output order
  __shared__ uint32_t s_sum[SELECT_SMALL];              // kept counts at the windows' first positions, then their inclusive scan
  __shared__ uint32_t s_flags;
  const uint32_t i = threadIdx.x;
  const bool mine = i < n;
  Derived d{};
  if (mine) { d = derive(raw + i, guides, win_base, win, window_lo, windows_per_guide); s_widx[i] = d.widx; }
  if (i == 0) s_flags = 0;
  __syncthreads();
  if (mine) {
    uint32_t p = 0;
    for (uint32_t j = 0; j < n; j++) { const uint32_t wj = s_widx[j]; p += (wj < d.widx || (wj == d.widx && j < i)) ? 1u : 0u; }
    s_d[p] = d; s_key[p] = order_key(d); s_src[p] = i;
  }
  __syncthreads();
  uint32_t nk = 0;
  bool first = false;
  if (mine) {
    const uint32_t s = i, w = s_d[s].widx;
    first = s == 0 || s_d[s - 1].widx != w;
    if (first) {
      uint32_t e = s + 1;
      while (e < n && s_d[e].widx == w) e++;
      if (e - s > SMALL_WINDOW_MAX) {
        atomicOr(&s_flags, SELECT_FLAG_RETRY);
      } else {
        for (uint32_t list = 0; list < 2; list++) {
          const uint32_t first_kept = nk;                 // overlaps are only tested against the same strand (SGA:317)
          for (uint32_t round = s; round < e; round++) {
            unsigned long long bk = 0;
            uint32_t best = 0;
            for (uint32_t m = s; m < e; m++) {
              const unsigned long long k = s_key[m];
              if ((uint32_t)(k >> 63) != list || k == 0) continue;
              if (k > bk) { bk = k; best = m; }
            }
            if (bk == 0) break;
            s_key[best] = 0;
            if ((int)s_d[best].edits > max_total_diffs) continue;
            bool clash = false;
            for (uint32_t k = first_kept; k < nk; k++) {
              const uint32_t kp = s_out[s + k];
              const int o = min(s_d[best].end, s_d[kp].end) - max(s_d[best].start, s_d[kp].start);   // GA:119-122
              if (o > max_overlap) { clash = true; break; }
            }
            if (!clash) s_out[s + nk++] = best;
          }
        }
      }
    }
  }
  s_sum[i] = nk;
  __syncthreads();
  for (uint32_t dd = 1; dd < SELECT_SMALL; dd <<= 1) {    // inclusive scan, Hillis-Steele
    const uint32_t add = i >= dd ? s_sum[i - dd] : 0u;
    __syncthreads();
    s_sum[i] += add;
    __syncthreads();
  }
  if (first && nk) {
    const uint32_t at = s_sum[i] - nk;
    for (uint32_t r = 0; r < nk; r++) final_out[at + r] = raw[s_src[s_out[i + r]]];
  }
  if (i == 0) {
    const uint32_t total = s_sum[SELECT_SMALL - 1], flags = s_flags;
    counts[0] = total; counts[1] = flags; counts[2] = 0; counts[3] = 0;
    if (box) {
      if (ctr) { for (int k = 0; k < 8; k++) box[1 + k] = ctr[k]; box[9] = total; box[10] = flags; box[11] = 0; }
      else { box[1] = total; box[2] = flags; box[3] = 0; }
      __threadfence_system();
      __hip_atomic_store(box, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// box: the thread that knows the total also posts the stage's three counts to the host's mailbox (mailbox.hpp).  The flags and the
// number of big groups were final when the filter kernels ended; the host only uses the counts to queue the next kernels on this
// stream, behind this one.
__global__ void gather_kernel(const RawAln* raw, const uint32_t* slot, const uint32_t* kept, const uint32_t* koffs, const uint32_t* out_pos,
                              uint32_t n, RawAln* final_out, uint32_t* counts, uint32_t* box, uint32_t seq) {
  CALITAS_TAIL_PRIO();
  const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const uint32_t nk = kept[s], d = koffs[s];
  if (s == n - 1) {
    counts[0] = d + nk;                                // total survivors
    if (box) {
      box[1] = d + nk;
      box[2] = __hip_atomic_load(counts + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      box[3] = __hip_atomic_load(counts + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // windows beyond GROUP_MAX ("crowded")
      __threadfence_system();
      __hip_atomic_store(box, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  for (uint32_t r = 0; r < nk; r++) final_out[d + r] = raw[slot[out_pos[s + r]]];
}

// zero_on: clear the new buffer, ordered on the stream that will use it.  (hipMemset on the null stream is asynchronous to the
// host and is not ordered against a non-blocking stream: a lane's count_kernel ran ahead of it and scatter_kernel then
// indexed with negative slots -- DESIGN.md 4.7.)
template <typename T>
hipError_t grow(T** p, size_t& cap, size_t need, hipStream_t zero_on = nullptr, bool zero = false) {
  if (need <= cap) return hipSuccess;
  (void)hipFree(*p); *p = nullptr; cap = 0;
  need += need / 8;
  hipError_t e = hipMalloc((void**)p, need * sizeof(T));
  if (e != hipSuccess) return e;
  cap = need;
  if (!zero) return hipSuccess;
  // poison first, then the clear, both on the using stream: if the clear ever gets lost or reordered again, the first call on the
  // new buffer trips scatter_kernel's range check deterministically instead of depending on what the allocator handed out
  e = hipMemsetAsync(*p, 0xFF, need * sizeof(T), zero_on);
  if (e != hipSuccess) return e;
  return hipMemsetAsync(*p, 0, need * sizeof(T), zero_on);
}

}  // namespace

struct SelectWork {
  Derived *der = nullptr, *ders = nullptr; size_t der_cap = 0, ders_cap = 0;
  uint32_t *slot = nullptr, *out_idx = nullptr, *big = nullptr; size_t slot_cap = 0, out_idx_cap = 0, big_cap = 0;
  uint8_t* taken = nullptr; size_t taken_cap = 0;
  uint32_t *cnt = nullptr, *offs = nullptr;       // per window (+ 1)
  uint32_t *kept = nullptr, *koffs = nullptr;     // per slot position
  size_t cnt_cap = 0, offs_cap = 0, kept_cap = 0, koffs_cap = 0;
  bool cnt_dirty = false;      // a call was cut short between count_kernel and scatter_kernel: cnt must be cleared
  RawAln* final_out = nullptr; size_t final_cap = 0;
  void* temp = nullptr; size_t temp_cap = 0;
  uint32_t* counts = nullptr;   // [0] survivors, [1] flags, [2] windows left to filter_wave_kernel, [3] those of them beyond GROUP_MAX
};

void select_destroy(SelectWork* w) {
  if (!w) return;
  (void)hipFree(w->der); (void)hipFree(w->ders); (void)hipFree(w->slot); (void)hipFree(w->out_idx); (void)hipFree(w->big); (void)hipFree(w->taken);
  (void)hipFree(w->cnt); (void)hipFree(w->offs); (void)hipFree(w->kept); (void)hipFree(w->koffs); (void)hipFree(w->final_out);
  (void)hipFree(w->temp); (void)hipFree(w->counts);
  delete w;
}

bool select_supported(uint64_t windows_per_guide, int window_size, int n_guides) {
  return window_size < (1 << 13) &&      // end columns take 13 bits of the enumeration key
         windows_per_guide > 0 && windows_per_guide * (uint64_t)n_guides < (1ull << 31) && n_guides <= 64;
}

void select_done(SelectWork* w) { if (w) w->cnt_dirty = false; }

hipError_t select_run_speculative(SelectWork** pw, const RawAln* d_raw, const uint32_t* d_counters, uint32_t rec_cap, uint32_t raw_cap,
                                  uint32_t item_cap, const GuideDev* d_guides, const uint64_t* d_win_base, const int2* d_win, uint64_t window_lo,
                                  uint64_t windows_per_guide, int max_total_diffs, int max_overlap, hipStream_t stream,
                                  const RawAln** d_final, Mailbox* post) {
  if (!*pw) *pw = new SelectWork();
  SelectWork& w = **pw;
  hipError_t e;
#define TRY(x) do { e = (x); if (e != hipSuccess) return e; } while (0)
  if (!w.counts) { TRY(hipMalloc((void**)&w.counts, 4 * sizeof(uint32_t))); TRY(hipMemsetAsync(w.counts, 0, 4 * sizeof(uint32_t), stream)); }
  TRY(grow(&w.final_out, w.final_cap, SELECT_SMALL));
  TRY(mailbox_open(*post));
  const uint32_t seq = ++post->seq;
  hipLaunchKernelGGL(select_small_kernel, dim3(1), dim3(SELECT_SMALL), 0, stream, d_raw, 0u, d_guides, d_win_base, d_win, (uint32_t)window_lo,
                     (uint32_t)windows_per_guide, max_total_diffs, max_overlap, w.final_out, w.counts, post->dev, seq, d_counters, rec_cap, raw_cap,
                     item_cap);
  TRY(hipGetLastError());
#undef TRY
  *d_final = w.final_out;
  return hipSuccess;
}

hipError_t select_run(SelectWork** pw, const RawAln* d_raw, uint32_t n_raw, const GuideDev* d_guides, const uint64_t* d_win_base,
                      const int2* d_win, uint64_t window_lo, uint64_t windows_per_guide, int n_guides, int max_total_diffs, int max_overlap,
                      hipStream_t stream, const RawAln** d_final, const uint32_t** d_counts, Mailbox* post, bool general) {
  if (!*pw) *pw = new SelectWork();
  SelectWork& w = **pw;
  hipError_t e;
  const size_t n = n_raw;
  const size_t nw = (size_t)windows_per_guide * (size_t)n_guides;
#define TRY(x) do { e = (x); if (e != hipSuccess) return e; } while (0)
  if (!w.counts) { TRY(hipMalloc((void**)&w.counts, 4 * sizeof(uint32_t))); TRY(hipMemsetAsync(w.counts, 0, 4 * sizeof(uint32_t), stream)); }
  *d_final = nullptr; *d_counts = w.counts;
  if (n == 0) {
    TRY(hipMemsetAsync(w.counts, 0, 4 * sizeof(uint32_t), stream));
    if (post) TRY(mailbox_post(*post, w.counts, 3, stream));
    return hipSuccess;
  }
  if (n <= SELECT_SMALL && !general) {     // the one-workgroup version (select_small_kernel)
    TRY(grow(&w.final_out, w.final_cap, n));
    uint32_t* box = nullptr;
    uint32_t seq = 0;
    if (post) { TRY(mailbox_open(*post)); box = post->dev; seq = ++post->seq; }
    hipLaunchKernelGGL(select_small_kernel, dim3(1), dim3(SELECT_SMALL), 0, stream, d_raw, n_raw, d_guides, d_win_base, d_win, (uint32_t)window_lo,
                       (uint32_t)windows_per_guide, max_total_diffs, max_overlap, w.final_out, w.counts, box, seq, (const uint32_t*)nullptr, 0u, 0u, 0u);
    TRY(hipGetLastError());
    *d_final = w.final_out;
    return hipSuccess;
  }
  const size_t cnt_cap_before = w.cnt_cap;
  TRY(grow(&w.cnt, w.cnt_cap, nw + 1, stream, true));
  if (w.cnt_dirty && w.cnt_cap == cnt_cap_before) TRY(hipMemsetAsync(w.cnt, 0, w.cnt_cap * sizeof(uint32_t), stream));
  TRY(grow(&w.offs, w.offs_cap, nw + 1)); TRY(grow(&w.kept, w.kept_cap, n)); TRY(grow(&w.koffs, w.koffs_cap, n));
  TRY(grow(&w.der, w.der_cap, n)); TRY(grow(&w.ders, w.ders_cap, n)); TRY(grow(&w.slot, w.slot_cap, n)); TRY(grow(&w.out_idx, w.out_idx_cap, n));
  TRY(grow(&w.taken, w.taken_cap, n)); TRY(grow(&w.big, w.big_cap, n / (LANE_MAX + 1) + 1)); TRY(grow(&w.final_out, w.final_cap, n));
  size_t t1 = 0, t2 = 0;
  TRY(rocprim::exclusive_scan(nullptr, t1, w.cnt, w.offs, 0u, nw + 1, rocprim::plus<uint32_t>(), stream));
  TRY(rocprim::exclusive_scan(nullptr, t2, w.kept, w.koffs, 0u, n, rocprim::plus<uint32_t>(), stream));
  t1 = std::max(t1, t2);
  if (t1 > w.temp_cap) { (void)hipFree(w.temp); w.temp = nullptr; w.temp_cap = 0; TRY(hipMalloc(&w.temp, t1)); w.temp_cap = t1; }
  const dim3 block(256), grid_n((unsigned)((n + 255) / 256));
  w.cnt_dirty = true;
  hipLaunchKernelGGL(count_kernel, grid_n, block, 0, stream, d_raw, n_raw, d_guides, d_win_base, d_win, (uint32_t)window_lo, (uint32_t)windows_per_guide, w.der,
                     w.cnt, w.counts);
  size_t ts = w.temp_cap;
  TRY(rocprim::exclusive_scan(w.temp, ts, w.cnt, w.offs, 0u, nw + 1, rocprim::plus<uint32_t>(), stream));   // cnt[nw] = 0: offs[nw] = n
  hipLaunchKernelGGL(scatter_kernel, grid_n, block, 0, stream, (const Derived*)w.der, n_raw, (const uint32_t*)w.offs, w.cnt, w.slot, w.ders, w.taken, w.counts);
  hipLaunchKernelGGL(filter_kernel, grid_n, block, 0, stream, (const Derived*)w.ders, (const uint32_t*)w.offs, n_raw, max_total_diffs,
                     max_overlap, w.taken, w.kept, w.out_idx, w.counts, w.big);
  hipLaunchKernelGGL(filter_wave_kernel, dim3((unsigned)std::min<size_t>(n / (LANE_MAX + 1) + 1, 8192)), dim3(64), 0, stream, (const Derived*)w.ders,
                     (const uint32_t*)w.offs, max_total_diffs, max_overlap, w.kept, w.out_idx, w.counts, (const uint32_t*)w.big);
  ts = w.temp_cap;
  TRY(rocprim::exclusive_scan(w.temp, ts, w.kept, w.koffs, 0u, n, rocprim::plus<uint32_t>(), stream));
  uint32_t* box = nullptr;
  uint32_t seq = 0;
  if (post) { TRY(mailbox_open(*post)); box = post->dev; seq = ++post->seq; }
  hipLaunchKernelGGL(gather_kernel, grid_n, block, 0, stream, d_raw, (const uint32_t*)w.slot, (const uint32_t*)w.kept, (const uint32_t*)w.koffs,
                     (const uint32_t*)w.out_idx, n_raw, w.final_out, w.counts, box, seq);
  TRY(hipGetLastError());
#undef TRY
  *d_final = w.final_out;
  return hipSuccess;
}

}  // namespace calitas
