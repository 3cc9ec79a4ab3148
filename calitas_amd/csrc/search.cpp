// search.cpp -- the search pipeline behind calitas_search / calitas_search_hits / calitas_search_hits_batch: planning, lane
// (stream + buffers) management, the scan -> align -> trace -> filter -> rows chain, and the lanes that pipeline contig ranges
// or guides against each other.  The C entry points themselves are in api.cpp.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "ctx.hpp"
#include "tuning.hpp"

// CALITAS_TRACE=2: host-side time line of a call (microseconds since the first mark), one line at the end of calitas_search_hits
namespace {
struct HostMarks {
  bool on = false;
  std::chrono::steady_clock::time_point t0;
  std::string line;
  void start() { start_at(std::chrono::steady_clock::now()); }
  void start_at(std::chrono::steady_clock::time_point t) { const char* e = TUNE_GET("CALITAS_TRACE"); on = e && std::atoi(e) >= 2; line.clear(); t0 = t; }
  void mark(const char* what) {
    if (!on) return;
    char b[64];
    std::snprintf(b, sizeof b, " %s %.0f", what, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    line += b;
  }
  void dump(int lane = -1) { if (on) std::fprintf(stderr, "[calitas] host marks (us)%s%s:%s\n", lane >= 0 ? " lane " : "", lane >= 0 ? std::to_string(lane).c_str() : "", line.c_str()); }
};
thread_local HostMarks g_marks;
}  // namespace

static int fail(calitas_ctx* ctx, int code, const std::string& msg) { return calitas_fail(ctx, code, msg); }
static void* out_alloc(size_t size) { return calitas_out_alloc(size); }

std::string build_guide_dev(const GuideHost& gh, const calitas_params_t& p, const Scores& sc, int max_guide_diffs, int max_pam_mismatches,
                            GuideDev& gd) {
  std::memset(&gd, 0, sizeof(gd));
  const int L = (int)gh.q.size();
  gd.L = L;
  gd.n_pams = (int)gh.pams_q.size();
  gd.cli_length = gh.cli_length;
  gd.min_guide_score = sc.match * L + sc.worst_guide_diff * max_guide_diffs;         // SGA:239-243
  gd.max_guide_diffs = max_guide_diffs;
  gd.max_pam_mismatches = max_pam_mismatches;
  gd.max_diffs_filtering = max_guide_diffs + p.max_gaps_between_guide_and_pam + max_pam_mismatches;   // SGA:249
  gd.pam5 = gh.pam5 ? 1 : 0;
  const int budget = sc.match * L - gd.min_guide_score;                              // = |worst| * d
  // score(all matches) - score(path) = sum of per-edit costs: mismatch |m|, guide-only base |b|, genome-only base |B|
  const int c_mm = iabs(p.guide_mismatch_net_cost), c_ins = iabs(p.genome_gap_net_cost), c_del = iabs(p.guide_gap_net_cost);
  const int c_min = std::min(c_mm, std::min(c_ins, c_del));
  if (c_min <= 0 || c_del <= 0) return "net costs of 0 are not supported (the candidate filter needs every edit to cost something)";
  gd.scan_max_edits = budget / c_min;
  const int max_del = budget / c_del;
  gd.span = L + max_del;
  if (gd.span + 1 + 16 > STRIP_MAX_COLS || gd.span + 1 > RAW_MAX_OPS)
    return "max-guide-diffs too large for this protospacer (strip wider than the aligner kernel supports)";
  if (L + gd.scan_max_edits > 64) return "max-guide-diffs too large for the scan warm-up";
  for (int code = 0; code < 4; code++) {
    uint32_t v = 0;
    for (int i = 0; i < L; i++) if (iupac_mask((unsigned char)gh.q[i]) & (1 << code)) v |= 1u << (32 - L + i);
    gd.peq_a[code] = v;
  }
  uint32_t all = 0;
  for (int i = 0; i < L; i++) all |= 1u << (32 - L + i);
  gd.peq_a[4] = 0; gd.peq_a[5] = all; gd.peq_a[6] = 0; gd.peq_a[7] = 0;
  for (int code = 0; code < 4; code++) gd.peq_b[code] = gd.peq_a[3 - code];
  for (int k = 4; k < 8; k++) gd.peq_b[k] = gd.peq_a[k];
  for (int i = 0; i < L; i++) gd.qmask[i] = (uint8_t)iupac_mask((unsigned char)gh.q[i]);
  for (int i = 0; i < L; i++) gd.row_sets[i >> 4] |= (uint64_t)gd.qmask[i] << ((i & 15) * 4);
  for (int pi = 0; pi < gd.n_pams; pi++) {
    gd.pam_len[pi] = (uint8_t)gh.pams_q[pi].size();
    for (size_t k = 0; k < gh.pams_q[pi].size(); k++) gd.pam_mask[pi][k] = (uint8_t)iupac_mask((unsigned char)gh.pams_q[pi][k]);
  }
  return "";
}

int ensure_buffers(calitas_ctx* ctx, uint32_t rec_cap, uint32_t raw_cap, uint64_t slab_per_rec, uint32_t item_cap) {
  rec_cap = std::max(rec_cap, ctx->rec_cap);
  if (const char* e = TUNE_GET("CALITAS_DEVICE_BUDGET_MB")) {   // refuse instead of trying: what a caller sharing the card can set
    const uint64_t want = (uint64_t)rec_cap * slab_per_rec + (uint64_t)rec_cap * sizeof(ScanRecord) +
                          (uint64_t)std::max(raw_cap, ctx->raw_cap) * sizeof(RawAln) + (uint64_t)std::max(item_cap, ctx->item_cap) * sizeof(uint64_t);
    if (want > (uint64_t)std::atoll(e) << 20)
      return calitas_fail(ctx, CALITAS_ENOMEM, "search buffers of " + std::to_string(want >> 20) + " MB exceed CALITAS_DEVICE_BUDGET_MB");
  }
  if (item_cap > ctx->item_cap) {
    (void)hipFree(ctx->d_items); ctx->d_items = nullptr; ctx->item_cap = 0;
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_items, (size_t)item_cap * sizeof(uint64_t)));
    ctx->item_cap = item_cap;
  }
  if ((uint64_t)rec_cap * slab_per_rec > ctx->slab_cap) {
    (void)hipFree(ctx->d_slab); ctx->d_slab = nullptr; ctx->slab_cap = 0;
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_slab, (size_t)rec_cap * slab_per_rec));
    ctx->slab_cap = (uint64_t)rec_cap * slab_per_rec;
  }
  if (rec_cap > ctx->rec_cap) {
    (void)hipFree(ctx->d_recs); ctx->d_recs = nullptr; ctx->rec_cap = 0;
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_recs, (size_t)rec_cap * sizeof(ScanRecord)));
    ctx->rec_cap = rec_cap;
  }
  if (raw_cap > ctx->raw_cap) {
    (void)hipFree(ctx->d_raw); ctx->d_raw = nullptr; ctx->raw_cap = 0;
    HIP_TRY(ctx, hipMalloc((void**)&ctx->d_raw, (size_t)raw_cap * sizeof(RawAln)));
    ctx->raw_cap = raw_cap;
  }
  return CALITAS_OK;
}

// A lane of a chunked search (see calitas_search_hits) is a child context: its own stream, buffers and scratch, the
// parent's resident reference and window table.
static inline calitas_ctx* ref_owner(calitas_ctx* ctx) { return ctx->parent ? ctx->parent : ctx; }

// Everything about one search that does not depend on the lane running it.
struct SearchPlan {
  calitas_params_t p{};
  int n_guides = 0, step = 0, max_total = 0;
  Scores sc{};
  std::vector<GuideHost> gh;
  std::vector<GuideDev> gd;
  uint32_t slots_per_rec = 0, slab_bytes = 0;
  uint64_t slab_per_rec = 0;
  int warm_words = 1;                 // 32-base warm-up words of a scan lane: L + E - 1 <= 32 * warm_words
  uint64_t rec_hint = 0;              // expected scan records of this job (from estimate_scan_records), 0 = unknown
  uint64_t gw_lo = 0, gw_hi = ~0ull;  // global window range of the call (calitas_params_t::first_window / n_windows); all by default
  // the part of the packed reference this job covers
  uint32_t tile_lo = 0, n_tiles = 0;
  uint64_t bases = 0;
  uint64_t win_lo = 0, win_n = 0;     // its entries of the device window table
  // its reference bins (binned.hpp); bin_shift = 0: the binned tail does not take this window size
  int bin_shift = 0;
  uint32_t bin_first = 0, n_bins = 0;
  // calitas_search_hits on a window range (a process of a multi-GPU job): the rows it owns, as keys (contig << 32 | coordinate_start)
  bool owned = false;
  uint64_t own_lo = 0, own_hi = ~0ull;
  bool narrow_tail = false;           // a range of a chunked call that is not the last: its tail shares the chip with the next scan
  bool three_ranges = false;          // a range of a call cut into three or more
  bool last_range = false;            // ... and the last of them: no scan runs beside its tail
  int range_index = 0;                // which range of a chunked call this is
  bool general_tail = false;          // the caller brings hits of its own into the row stage (HitsExt): the general kernels take them, the bins do not
};

// The bins of contigs [c0, c1) of the plan's geometry (the owner's bin_base must be built: ensure_bin_base).
static void plan_bins(const calitas_ctx* owner, SearchPlan& q, int c0, int c1) {
  if (!q.bin_shift || owner->bin_base.empty()) { q.bin_first = 0; q.n_bins = 0; return; }
  q.bin_first = owner->bin_base[c0];
  q.n_bins = owner->bin_base[c1] - q.bin_first;
}

// Accepted alignments left on the device by search_impl for calitas_search_hits.
struct DeviceSel {
  bool valid = false;
  const RawAln* d_final = nullptr;
  uint32_t n_sel = 0;
  bool crowded = false;    // some window held more records than a wave filters in registers (select.hip GROUP_MAX)
  std::chrono::steady_clock::time_point t_call;
};

// Copies the device-selected alignments back and converts them to GuideAlignment records (GA:21-31, SGA:260-313).
static int convert_selected(calitas_ctx* ctx, const RawAln* d_final, uint32_t n_sel, const std::vector<GuideHost>& gh,
                            const calitas_params_t& p, int step, calitas_aln_t** out) {
  const PackedRef& ref = ref_owner(ctx)->ref;
  if (n_sel > ctx->h_raw_cap) {
    if (ctx->h_raw) (void)hipHostFree(ctx->h_raw);
    ctx->h_raw = nullptr; ctx->h_raw_cap = 0;
    HIP_TRY(ctx, hipHostMalloc((void**)&ctx->h_raw, (size_t)ctx->raw_cap * sizeof(RawAln), hipHostMallocDefault));
    ctx->h_raw_cap = ctx->raw_cap;
  }
  if (n_sel) HIP_TRY(ctx, hipMemcpyAsync(ctx->h_raw, d_final, (size_t)n_sel * sizeof(RawAln), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, calitas_spin_sync(ctx->stream));
  const RawAln* raw = ctx->h_raw;
  calitas_aln_t* result = (calitas_aln_t*)out_alloc(std::max<size_t>(1, n_sel) * sizeof(calitas_aln_t));
  if (!result) return fail(ctx, CALITAS_EINVAL, "out of memory");
  ref_owner(ctx)->pool->for_blocks(n_sel, [&](size_t b, size_t e, int) {
    for (size_t i = b; i < e; i++) {
      const RawAln& r = raw[i];
      int64_t wa = 0, wb = 0;
      window_bounds(ref.runs.data(), (int64_t)ref.runs.size(), ref.contigs[r.contig].gbase, ref.contigs[r.contig].len, p.window_size, step,
                    r.window_k, wa, wb);
      raw_to_aln(r, gh[r.guide], wa, wb, result[i]);
    }
  });
  *out = result;
  return CALITAS_OK;
}

// Validation and the host-side constants of a search.  Covers the whole reference (or the one contig of chrom_index);
// a chunked search narrows tile_lo / n_tiles / bases per lane afterwards.
static int plan_search(calitas_ctx* ctx, int32_t n_guides, const calitas_guide_t* guides, const calitas_params_t* params, SearchPlan& pl) {
  if (!guides || !params) return fail(ctx, CALITAS_EINVAL, "NULL argument");
  if (ctx->device < 0) return fail(ctx, CALITAS_ENODEV, "host-only context: calitas_search needs a GPU (there is no CPU fallback)");
  if (!ref_owner(ctx)->has_ref) return fail(ctx, CALITAS_ESTATE, "calitas_set_reference has not been called");
  if (n_guides <= 0 || n_guides > MAX_GUIDES) return fail(ctx, CALITAS_EINVAL, "n_guides must be 1..64");
  const calitas_params_t& p = *params;
  if (p.window_size <= 0 || p.window_size > 60000) return fail(ctx, CALITAS_EINVAL, "window-size must be 1..60000");
  if (p.max_guide_diffs < 0 || p.max_pam_mismatches < 0 || p.max_gaps_between_guide_and_pam < 0 || p.max_gaps_between_guide_and_pam > 16)
    return fail(ctx, CALITAS_EINVAL, "limits out of range (max-gaps-between-guide-and-pam must be 0..16)");
  const PackedRef& ref = ref_owner(ctx)->ref;
  if (p.chrom_index >= (int)ref.contigs.size()) return fail(ctx, CALITAS_EINVAL, "chrom_index out of range");
  pl.p = p; pl.n_guides = n_guides;
  pl.sc = derive_scores(p.guide_mismatch_net_cost, p.pam_mismatch_net_cost, p.genome_gap_net_cost, p.guide_gap_net_cost);
  pl.max_total = p.max_total_diffs >= 0 ? p.max_total_diffs : p.max_guide_diffs + p.max_gaps_between_guide_and_pam + p.max_pam_mismatches;
  pl.gh.assign(n_guides, GuideHost());
  pl.gd.assign(n_guides, GuideDev());
  for (int i = 0; i < n_guides; i++) {
    std::string e = make_guide_host(guides[i], pl.gh[i]);
    if (e.empty()) e = build_guide_dev(pl.gh[i], p, pl.sc, p.max_guide_diffs, p.max_pam_mismatches, pl.gd[i]);
    if (!e.empty()) return fail(ctx, CALITAS_EINVAL, "guide " + std::to_string(i) + ": " + e);
    // SR:529-530: the window step depends on the CLI guide length; one pass shares one tiling
    int overlap = pl.gh[i].cli_length + p.max_guide_diffs + p.max_gaps_between_guide_and_pam - 1;
    int s = p.window_size - overlap;
    if (s <= 0) return fail(ctx, CALITAS_EINVAL, "window-size is not larger than guide length + max-guide-diffs + max-gaps - 1");
    if (i == 0) pl.step = s;
    else if (s != pl.step) return fail(ctx, CALITAS_EINVAL, "all guides of one batch must have the same length (same window tiling, SearchReference.scala:529)");
    if ((pl.gd[i].L + pl.gd[i].scan_max_edits + 15) / 16 > ref.chunk / 16) return fail(ctx, CALITAS_EINVAL, "scan warm-up exceeds the lane chunk");
    pl.warm_words = std::max(pl.warm_words, (pl.gd[i].L + pl.gd[i].scan_max_edits - 1 + 31) / 32);
  }
  // Strip slabs (align_kernel -> trace_kernel): fixed size and fixed address per (record, window slot).
  pl.slots_per_rec = (uint32_t)((p.window_size + 14) / pl.step + 1);   // windows a 16-base word can fall into
  if (pl.slots_per_rec > 8) return fail(ctx, CALITAS_EINVAL, "window step is too small relative to the window size (more than 8 windows per position)");
  pl.slab_bytes = 0;
  for (int i = 0; i < n_guides; i++) {
    const uint32_t ncols_max = 16 + pl.gd[i].span + 1;
    const uint32_t stride_max = (ncols_max + 4) & ~3u;
    const uint32_t ntb_max = (ncols_max + p.max_gaps_between_guide_and_pam + MAX_PAM_LEN + 3) & ~3u;
    pl.slab_bytes = std::max<uint32_t>(pl.slab_bytes, (uint32_t)((sizeof(SlabHeader) + ntb_max + pl.gd[i].L * stride_max + 15) & ~15u));
  }
  pl.slab_per_rec = (uint64_t)pl.slab_bytes * pl.slots_per_rec;
  pl.tile_lo = 0; pl.n_tiles = (uint32_t)ref.tiles.size();
  pl.bin_shift = binned_shift(p.window_size);
  pl.bases = p.chrom_index >= 0 ? ref.contigs[p.chrom_index].len : ref.total_bases;
  pl.win_lo = 0; pl.win_n = 0;
  for (auto& c : ref.contigs) pl.win_n += window_count(c.len, pl.step);
  if (p.n_windows != 0 || p.first_window != 0) {
    // a window range of the job: scan the tiles its windows touch, align only inside those windows
    if (p.first_window < 0 || p.n_windows <= 0 || (uint64_t)p.first_window + (uint64_t)p.n_windows > pl.win_n)
      return fail(ctx, CALITAS_EINVAL, "first_window / n_windows outside the window table (" + std::to_string(pl.win_n) + " windows)");
    if (p.chrom_index >= 0) return fail(ctx, CALITAS_EINVAL, "a window range and chrom_index exclude each other");
    pl.gw_lo = (uint64_t)p.first_window; pl.gw_hi = pl.gw_lo + (uint64_t)p.n_windows;
    uint64_t base = 0, g_lo = 0, g_hi = 0, bases = 0;
    bool first = true;
    for (auto& c : ref.contigs) {
      const uint64_t nw = window_count(c.len, pl.step);
      const uint64_t a = std::max(pl.gw_lo, base), b = std::min(pl.gw_hi, base + nw);     // this contig's share of the range
      if (a < b) {
        const uint64_t lo = (a - base) * (uint64_t)pl.step, hi = std::min<uint64_t>(c.len, (b - 1 - base) * (uint64_t)pl.step + (uint64_t)p.window_size);
        if (first) { g_lo = c.gbase + lo; first = false; }
        g_hi = c.gbase + hi;
        bases += hi - lo;
      }
      base += nw;
    }
    pl.tile_lo = (uint32_t)(g_lo / ref.tile);
    pl.n_tiles = (uint32_t)((g_hi + ref.tile - 1) / ref.tile) - pl.tile_lo;
    pl.bases = bases;
    pl.win_lo = pl.gw_lo; pl.win_n = pl.gw_hi - pl.gw_lo;
  }
  return CALITAS_OK;
}

// The device window table for (window size, step) lives with the reference; (re)built on `stream` when the tiling changes.
static int ensure_window_table(calitas_ctx* ctx, const SearchPlan& pl, hipStream_t stream) {
  calitas_ctx* o = ref_owner(ctx);
  if (o->win_W == pl.p.window_size && o->win_step == pl.step) return CALITAS_OK;
  const PackedRef& ref = o->ref;
  std::vector<uint64_t> wb(ref.contigs.size() + 1, 0);
  for (size_t c = 0; c < ref.contigs.size(); c++) wb[c + 1] = wb[c] + window_count(ref.contigs[c].len, pl.step);
  const uint64_t nw = wb.back();
  if (!o->d_win_base) HIP_TRY(ctx, hipMalloc((void**)&o->d_win_base, wb.size() * sizeof(uint64_t)));
  if (nw > o->win_cap) {
    (void)hipFree(o->d_win); o->d_win = nullptr; o->win_cap = 0;
    HIP_TRY(ctx, hipMalloc((void**)&o->d_win, std::max<uint64_t>(1, nw) * sizeof(int2)));
    o->win_cap = nw;
  }
  HIP_TRY(ctx, hipMemcpyAsync(o->d_win_base, wb.data(), wb.size() * sizeof(uint64_t), hipMemcpyHostToDevice, stream));   // ordered before the kernel below
  HIP_TRY(ctx, hipStreamSynchronize(stream));                                                                              // wb is a local
  HIP_TRY(ctx, launch_window_table(o->d_runs, (int64_t)ref.runs.size(), o->d_contigs, o->d_win_base, (int)ref.contigs.size(), nw,
                                   pl.p.window_size, pl.step, o->d_win, stream));
  o->win_W = pl.p.window_size; o->win_step = pl.step;
  return CALITAS_OK;
}

// Per contig the index of its first bin (binned.hpp), for the plan's bin size; lives with the reference like the window table.
static int ensure_bin_base(calitas_ctx* ctx, SearchPlan& pl, hipStream_t stream) {
  calitas_ctx* o = ref_owner(ctx);
  if (!pl.bin_shift) return CALITAS_OK;
  const PackedRef& ref = o->ref;
  if (o->bin_shift != pl.bin_shift || o->bin_base.size() != ref.contigs.size() + 1) {
    std::vector<uint32_t> bb(ref.contigs.size() + 1, 0);
    uint64_t acc = 0;
    for (size_t c = 0; c < ref.contigs.size(); c++) { bb[c] = (uint32_t)acc; acc += (ref.contigs[c].len >> pl.bin_shift) + 1; }
    bb[ref.contigs.size()] = (uint32_t)acc;
    if (acc >= 0x7FFFFFFFull) { pl.bin_shift = 0; return CALITAS_OK; }
    (void)hipFree(o->d_bin_base); o->d_bin_base = nullptr; o->bin_shift = 0;
    (void)hipFree(o->d_bin_contig); o->d_bin_contig = nullptr;
    std::vector<uint32_t> bc((size_t)acc + 1, 0);
    for (size_t c = 0; c < ref.contigs.size(); c++) std::fill(bc.begin() + bb[c], bc.begin() + bb[c + 1], (uint32_t)c);
    HIP_TRY(ctx, hipMalloc((void**)&o->d_bin_base, bb.size() * sizeof(uint32_t)));
    HIP_TRY(ctx, hipMalloc((void**)&o->d_bin_contig, bc.size() * sizeof(uint32_t)));
    HIP_TRY(ctx, hipMemcpyAsync(o->d_bin_base, bb.data(), bb.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    HIP_TRY(ctx, hipMemcpyAsync(o->d_bin_contig, bc.data(), bc.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    HIP_TRY(ctx, hipStreamSynchronize(stream));                                                                              // bb / bc are locals
    o->bin_base.swap(bb); o->bin_shift = pl.bin_shift;
  }
  if (pl.n_bins == 0) plan_bins(o, pl, pl.p.chrom_index >= 0 ? pl.p.chrom_index : 0, pl.p.chrom_index >= 0 ? pl.p.chrom_index + 1 : (int)ref.contigs.size());
  return CALITAS_OK;
}

static void fill_kernel_args(calitas_ctx* ctx, const SearchPlan& pl, ScanArgs& sa, AlignArgs& aa) {
  const calitas_ctx* o = ref_owner(ctx);
  const PackedRef& ref = o->ref;
  const calitas_params_t& p = pl.p;
  sa = ScanArgs{};
  sa.codes = o->d_codes; sa.planes = o->d_planes; sa.mask = o->d_mask; sa.tiles = o->d_tiles; sa.guides = ctx->d_guides;
  sa.recs = ctx->d_recs; sa.rec_count = ctx->d_counters; sa.rec_capacity = ctx->rec_cap;
  sa.n_guides = pl.n_guides; sa.chrom_index = p.chrom_index; sa.tile_offset = pl.tile_lo; sa.tile_stride = 1;
  aa = AlignArgs{};
  aa.codes = o->d_codes; aa.mask = o->d_mask; aa.runs = o->d_runs; aa.n_runs = (int64_t)ref.runs.size();
  aa.contigs = o->d_contigs; aa.tiles = o->d_tiles; aa.win_base = o->d_win_base; aa.win = o->d_win; aa.guides = ctx->d_guides; aa.recs = ctx->d_recs;
  aa.rec_count = ctx->d_counters; aa.out = ctx->d_raw; aa.out_count = ctx->d_counters + 1; aa.anomalies = ctx->d_counters + 2;
  aa.trace_done = ctx->d_counters + 5; aa.job_count = ctx->d_counters + 6;
  for (int g = 0; g < pl.n_guides; g++) aa.max_guide_len = std::max<int32_t>(aa.max_guide_len, pl.gd[g].L);
  {
    // two jobs per lane group (align_pk_kernel): one protospacer length for all guides of the launch, and every cell a passing
    // alignment can go through -- within +-max|cost| x L of zero -- times four, with a step's cost on top, inside sixteen bits
    bool same_L = true;
    for (int g = 1; g < pl.n_guides; g++) same_L = same_L && pl.gd[g].L == pl.gd[0].L;
    const int64_t big = std::max<int64_t>(std::max<int64_t>(std::abs(pl.sc.match), std::abs(pl.sc.mismatch)), std::max<int64_t>(std::abs(pl.sc.target_gap), std::abs(pl.sc.query_gap)));
    aa.pack16 = (same_L && aa.max_guide_len <= 20 && 4 * big * ((int64_t)aa.max_guide_len + 2) < 30000) ? 1 : 0;
  }
  aa.rec_capacity = ctx->rec_cap; aa.out_capacity = ctx->raw_cap;
  aa.slab = ctx->d_slab; aa.cand_count = ctx->d_counters + 4; aa.items = ctx->d_items; aa.item_count = ctx->d_counters + 3; aa.item_capacity = ctx->item_cap;
  aa.slab_bytes = pl.slab_bytes; aa.slots_per_rec = pl.slots_per_rec; aa.tile_words = (uint32_t)(ref.tile / 16);
  aa.gw_lo = pl.gw_lo; aa.gw_hi = pl.gw_hi;
  if (const char* e = TUNE_GET("CALITAS_TAIL_PRIO_NARROW")) aa.low_prio = (pl.narrow_tail && std::atoi(e) == 0) ? 1 : 0;
  aa.sp.window_size = p.window_size; aa.sp.step = pl.step; aa.sp.n_guides = pl.n_guides;
  aa.sp.max_guide_diffs = p.max_guide_diffs; aa.sp.max_pam_mismatches = p.max_pam_mismatches;
  aa.sp.max_gaps = p.max_gaps_between_guide_and_pam;
  aa.sp.max_diffs_filtering = p.max_guide_diffs + p.max_gaps_between_guide_and_pam + p.max_pam_mismatches;   // SGA:249
  aa.sp.match = pl.sc.match; aa.sp.mismatch = pl.sc.mismatch; aa.sp.pam_match = pl.sc.pam_match; aa.sp.pam_mismatch = pl.sc.pam_mismatch;
  aa.sp.query_gap = pl.sc.query_gap; aa.sp.target_gap = pl.sc.target_gap; aa.sp.eqx_by_score = p.eqx_by_score & 1; aa.sp.per_matrix = (p.eqx_by_score >> 1) & 1; aa.sp.chrom_index = p.chrom_index;
}

// Device buffers of one lane for this plan (allocation only).
static int lane_prepare(calitas_ctx* ctx, const SearchPlan& pl) {
  uint64_t want = std::max<uint64_t>(1u << 16, std::min<uint64_t>(1u << 20, pl.bases / 8 + 1024));
  uint64_t want_raw = want, want_items = 2 * want;
  if (pl.rec_hint) {   // a dense search (estimate_scan_records): no retry round per contig -- such searches yield ~3 alignments and passing candidates per record
    want = std::max<uint64_t>(want, std::min<uint64_t>(0xFFFFFFF0ull, pl.rec_hint + pl.rec_hint / 4 + 4096));
    want_raw = std::min<uint64_t>(0xFFFFFFF0ull, want * 7 / 2);
    want_items = std::min<uint64_t>(0xFFFFFFF0ull, want * 4);
  }
  return ensure_buffers(ctx, std::max<uint32_t>(ctx->rec_cap, (uint32_t)want), std::max<uint32_t>(ctx->raw_cap, (uint32_t)want_raw), pl.slab_per_rec,
                        std::max<uint32_t>(ctx->item_cap, (uint32_t)want_items));
}

// Guides, cleared counters and the scan kernel of this lane, queued on `stream` (the lane's own, or the shared scan stream of a
// chunked search).  t_scan0 / t_scan1 bracket the kernel: ev[0], and the lane's scan_done (the event its stream waits for) or ev[1].
// Both ride on the dispatch itself (hipExtLaunchKernel): as marker packets of their own on the lowest-priority stream they delayed
// whatever waited for the end of the scan by 60-90 us.
// (Queuing the inputs of all ranges first and their scans back to back was tried as well: no gain, the pause between two scans is
// where the previous range's tail gets onto the CUs.)
// The inputs of a lane's scan: guide constants and cleared counters, queued on `stream`.
static int queue_scan_inputs(calitas_ctx* ctx, const SearchPlan& pl, hipStream_t stream) {
  // from the context's pinned copy (an async copy from pageable memory may wait for the stream to drain)
  std::memcpy(ctx->h_guides, pl.gd.data(), sizeof(GuideDev) * pl.n_guides);
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_guides, ctx->h_guides, sizeof(GuideDev) * pl.n_guides, hipMemcpyHostToDevice, stream));
  g_marks.mark("guides");
  HIP_TRY(ctx, hipMemsetAsync(ctx->d_counters, 0, 8 * sizeof(uint32_t), stream));
  g_marks.mark("counters");
  return CALITAS_OK;
}

// columnwise: round 1's column-wise scan_kernel (kernels.hip) instead of scan_rows_kernel.  Only calitas_scan_candidates_columnwise
// asks for it -- a test hook that holds the two kernels' record sets against each other; no search path does.
// Contigs that are absent here (refpack.hpp: a process of a multi-GPU job holds what its window range touches): a search must not need
// one.  Checked where a plan is about to run -- the callers of a ranged search plan the whole job first and narrow it afterwards.
static int check_resident(calitas_ctx* ctx, const SearchPlan& pl) {
  const PackedRef& ref = ref_owner(ctx)->ref;
  if (ref.absent.empty()) return CALITAS_OK;
  int bad = -1;
  if (pl.p.chrom_index >= 0) {
    if (ref.is_absent((size_t)pl.p.chrom_index)) bad = pl.p.chrom_index;
  } else {
    uint64_t base = 0;
    for (size_t c = 0; c < ref.contigs.size() && bad < 0; c++) {
      const uint64_t nw = window_count(ref.contigs[c].len, pl.step);
      const uint64_t t0 = ref.contigs[c].gbase / ref.tile;          // (a chunked call's lanes: contig ranges by tiles)
      if (ref.is_absent(c) && std::max(pl.gw_lo, base) < std::min(pl.gw_hi, base + nw) && t0 >= pl.tile_lo && t0 < (uint64_t)pl.tile_lo + pl.n_tiles) bad = (int)c;
      base += nw;
    }
  }
  if (bad < 0) return CALITAS_OK;
  return calitas_fail(ctx, CALITAS_EINVAL, "contig " + ref.names[(size_t)bad] + " is not resident in this context (it was given without bases): "
                                           "search a window range that leaves it out");
}

static int launch_scan_stage(calitas_ctx* ctx, const SearchPlan& pl, hipStream_t stream, bool inputs_queued = false, bool columnwise = false) {
  { int rc = check_resident(ctx, pl); if (rc) return rc; }
  if (!inputs_queued) { int rc = queue_scan_inputs(ctx, pl, stream); if (rc) return rc; }
  ScanArgs sa; AlignArgs aa;
  fill_kernel_args(ctx, pl, sa, aa);
  ctx->t_scan0 = ctx->ev[0];
  ctx->t_scan1 = ctx->scan_done ? ctx->scan_done : ctx->ev[1];
  // the events ride on the dispatch
  if (columnwise) HIP_TRY(ctx, launch_scan(sa, ref_owner(ctx)->ref.chunk, pl.n_tiles, stream, ctx->t_scan0, ctx->t_scan1));
  else HIP_TRY(ctx, launch_scan_rows(sa, ref_owner(ctx)->ref.chunk, pl.warm_words, pl.n_tiles, stream, ctx->t_scan0, ctx->t_scan1));
  return CALITAS_OK;
}

// Kernel durations of the last search on this context, from its events (all of them complete).
static void kernel_times(calitas_ctx* ctx, calitas_timing_t& tm) {
  float ms = 0;
  (void)hipEventElapsedTime(&ms, ctx->t_scan0, ctx->t_scan1); tm.scan_kernel_ms = ms;
  if (ctx->align_ms_by_stamps >= 0) tm.align_kernel_ms = ctx->align_ms_by_stamps;   // (a search the binned tail started and declined: no ev[2])
  else { (void)hipEventElapsedTime(&ms, ctx->t_scan1, ctx->ev[2]); tm.align_kernel_ms = ms; }
  (void)hipEventElapsedTime(&ms, ctx->t_scan0, ctx->ev[3]); tm.gpu_total_ms = ms;
}

// Grids of align_kernel (units of four one-wave workgroups) and trace_kernel.  align_kernel: 2048 workgroups = 8 per CU, each looping
// over its share of the records.  4096 (all the CUs can hold next to nothing else: 16 x ~11 KB of LDS) was round 2's choice; swept again
// with three jobs per wave (tools/sweep_align.sh, profiles/r03_sweep_align.txt): 384-683 units beat 1024 at every size -- 2.33-2.35
// against 2.42 ms for the hg38-sized call, 0.77 against 0.87 ms for a quarter, 0.51 against 0.53 ms for an eighth -- because the
// scan of the next range keeps more of each CU while the tail runs beside it, and below 256 units the aligner itself runs out of waves.
// trace_kernel's grid makes no difference between 512 and 2048 (256 costs 0.3 ms at full size).
// CALITAS_ALIGN_BLOCKS / CALITAS_TRACE_BLOCKS (and ..._NARROW for the ranges whose tail runs beside the next range's scan) override.
constexpr int kAlignBlocks = 512, kTraceBlocks = 2048;
// ... with two jobs per lane group (align_pk_kernel, round 5) a wave does the work of two: 256 units beside a scan (an eighth of the
// genome as a rank's window range: 0.446 against 0.494 ms at 512, tools/owned_sweep.py), 384 for a tail that runs alone.
constexpr int kAlignBlocksPackedNarrow = 256, kAlignBlocksPacked = 384;
static int narrow_blocks(const char* e, int fallback) {   // e = the switch's value (TUNE_GET), or null
  if (e) { const int v = std::atoi(e); if (v >= 1 && v <= 8192) return v; }
  return fallback;
}

// calitas_search; with dev != nullptr the accepted alignments stay on the device when the device filter handled them
// (dev->valid), and *out stays NULL.  prelaunched: the scan stage of this lane was queued by the caller on another stream
// and ctx->stream already waits for it; an overflow then fails the call instead of retrying.
// resume: the stages through trace_kernel have run and the lane's counters are in ctx->h_counters (the binned tail declined, see
// lane_rows_binned): the first round starts at the per-window filter.
static int search_run(calitas_ctx* ctx, const SearchPlan& pl, calitas_aln_t** out, uint64_t* n_out, DeviceSel* dev, bool prelaunched,
                      bool resume = false) {
  const auto t_call = std::chrono::steady_clock::now();
  *out = nullptr; *n_out = 0;
  if (!resume) ctx->align_ms_by_stamps = -1;
  const calitas_params_t& p = pl.p;
  const PackedRef& ref = ref_owner(ctx)->ref;
  const int n_guides = pl.n_guides, step = pl.step, max_total = pl.max_total;
  const std::vector<GuideHost>& gh = pl.gh;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  if (!prelaunched) {
    int rc = lane_prepare(ctx, pl);
    if (rc) return rc;
    rc = ensure_window_table(ctx, pl, ctx->stream);
    if (rc) return rc;
  }
  calitas_timing_t tm{};
  tm.bases_scanned = pl.bases;
  tm.packed_bytes = (tm.bases_scanned + 3) / 4;
  uint32_t n_rec = 0, n_raw = 0;
  const calitas_ctx* own = ref_owner(ctx);
  const bool device_filter = !TUNE_GET("CALITAS_HOST_FILTER") && select_supported(pl.win_n, p.window_size, n_guides);
  // A small reference usually yields few alignments: the one-workgroup filter is queued right behind trace_kernel and reads the counts
  // on the device, so the host hears about the counters and the filter's result in one round trip (select_run_speculative).
  const bool speculate = !prelaunched && !resume && device_filter && pl.bases <= (64ull << 20);
  const RawAln* d_spec = nullptr;
  uint32_t spec_counts[3] = {0, 0, 0};
  bool spec_done = false;
  for (bool first_round = true;; first_round = false) {
    if (!(resume && first_round)) {
    ctx->align_ms_by_stamps = -1;                     // (this round's trace_kernel carries ev[2])
    if (!prelaunched) {
      int rc = launch_scan_stage(ctx, pl, ctx->stream);
      if (rc) return rc;
    }
    ScanArgs sa; AlignArgs aa;
    fill_kernel_args(ctx, pl, sa, aa);
    HIP_TRY(ctx, launch_align(aa, narrow_blocks(pl.narrow_tail ? TUNE_GET("CALITAS_ALIGN_BLOCKS_NARROW") : TUNE_GET("CALITAS_ALIGN_BLOCKS"), aa.pack16 && !aa.sp.per_matrix ? (pl.narrow_tail ? kAlignBlocksPackedNarrow : kAlignBlocksPacked) : kAlignBlocks), ctx->stream));
    // (trace_kernel can post the counters itself from its last workgroup -- launch_trace's `post` -- but finding the last of 2048
    // workgroups is 2048 atomics on one word, ~8 ns each: 20-30 us against the ~10 us of this launch)
    HIP_TRY(ctx, launch_trace(aa, narrow_blocks(pl.narrow_tail ? TUNE_GET("CALITAS_TRACE_BLOCKS_NARROW") : TUNE_GET("CALITAS_TRACE_BLOCKS"), kTraceBlocks), ctx->stream, ctx->ev[2]));
    if (speculate) {
      HIP_TRY(ctx, select_run_speculative(&ctx->select, ctx->d_raw, ctx->d_counters, ctx->rec_cap, ctx->raw_cap, ctx->item_cap, ctx->d_guides,
                                          own->d_win_base, own->d_win, pl.win_lo, pl.win_n, max_total, p.max_overlap, ctx->stream, &d_spec, &ctx->mbox));
      HIP_TRY(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
    } else {
      HIP_TRY(ctx, mailbox_post(ctx->mbox, ctx->d_counters, 8, ctx->stream));
    }
    g_marks.mark("queued-scan-align-trace");
    HIP_TRY(ctx, mailbox_wait(ctx->mbox, ctx->stream));
    g_marks.mark("counts1");
    for (int k = 0; k < 8; k++) ctx->h_counters[k] = ctx->mbox.host[1 + k];
    if (speculate) { for (int k = 0; k < 3; k++) spec_counts[k] = ctx->mbox.host[9 + k]; spec_done = !(spec_counts[1] & SELECT_FLAG_RETRY); }
    }
    n_rec = ctx->h_counters[0]; n_raw = ctx->h_counters[1];
    const uint32_t n_items = ctx->h_counters[3];
    if (ctx->h_counters[2] != 0) return fail(ctx, CALITAS_EHIP, "aligner kernel reported an inconsistent traceback (internal error)");
    if (n_rec > ctx->rec_cap || n_raw > ctx->raw_cap || n_items > ctx->item_cap) {
      if (prelaunched) return fail(ctx, CALITAS_ESTATE, "lane buffers overflowed");   // the caller reruns unchunked
      tm.retries++;
      uint64_t nr = n_rec > ctx->rec_cap ? (uint64_t)n_rec + n_rec / 4 : ctx->rec_cap;
      uint64_t nw = n_raw > ctx->raw_cap ? (uint64_t)n_raw * 2 : ctx->raw_cap;
      if (n_rec > ctx->rec_cap)   // the raw count was cut short as well: scale it with the record count
        nw = std::max<uint64_t>(nw, (uint64_t)((double)n_raw * nr / std::max<uint32_t>(1, ctx->rec_cap)) + 1024);
      if (nr > 0xFFFFFFF0ull || nw > 0xFFFFFFF0ull) return fail(ctx, CALITAS_EINVAL, "result volume exceeds 2^32 records");
      // passing candidates: as counted, or scaled with the record count when that was cut short
      uint64_t ni = n_items > ctx->item_cap ? (uint64_t)n_items + n_items / 4 : ctx->item_cap;
      if (n_rec > ctx->rec_cap) ni = std::max<uint64_t>(ni, (uint64_t)((double)std::max<uint32_t>(n_items, 1024) * nr / std::max<uint32_t>(1, ctx->rec_cap)) * 2);
      if (ni > 0xFFFFFFF0ull) return fail(ctx, CALITAS_EINVAL, "result volume exceeds 2^32 records");
      int rc = ensure_buffers(ctx, (uint32_t)nr, (uint32_t)nw, pl.slab_per_rec, (uint32_t)ni);
      if (rc) return rc;
      continue;
    }
    break;
  }
  // ---- per-window filter (SGA:315-320): on the GPU (select.hip) unless the tiling does not fit its sort key, a window
  //      exceeds its group limit, or CALITAS_HOST_FILTER asks for the host implementation of the same stage ----
  bool gpu_select = n_raw > 0 && device_filter;
  uint32_t n_sel = 0;
  const RawAln* d_sel = nullptr;
  if (gpu_select && spec_done) {                    // the filter ran with the aligner kernels: its counts came with theirs
    ctx->h_counters[5] = spec_counts[0]; ctx->h_counters[6] = spec_counts[1]; ctx->h_counters[7] = spec_counts[2];
    n_sel = spec_counts[0];
    d_sel = d_spec;
  } else if (gpu_select) {
    const RawAln* d_final = nullptr;
    const uint32_t* d_cnt = nullptr;
    // (second round: the one-workgroup version met a window it leaves to the general kernels -- here, or already behind trace_kernel)
    for (bool general = speculate && n_raw <= 1024;; general = true) {
      HIP_TRY(ctx, select_run(&ctx->select, ctx->d_raw, n_raw, ctx->d_guides, own->d_win_base, own->d_win, pl.win_lo, pl.win_n, n_guides, max_total,
                              p.max_overlap, ctx->stream, &d_final, &d_cnt, &ctx->mbox, general));   // its last kernel posts the three counts
      HIP_TRY(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
      g_marks.mark("queued-filter");
      HIP_TRY(ctx, mailbox_wait(ctx->mbox, ctx->stream));
      g_marks.mark("counts2");
      if (general || !(ctx->mbox.host[2] & SELECT_FLAG_RETRY)) break;
    }
    ctx->h_counters[5] = ctx->mbox.host[1]; ctx->h_counters[6] = ctx->mbox.host[2]; ctx->h_counters[7] = ctx->mbox.host[3];
    select_done(ctx->select);
    if (ctx->h_counters[6] & SELECT_FLAG_INTERNAL)
      return fail(ctx, CALITAS_EHIP, "per-window filter: window counters were not clear at the start of the stage (internal error)");
    if (ctx->h_counters[6] != 0) gpu_select = false;   // a window beyond what the device filter handles
    else {
      n_sel = ctx->h_counters[5];
      d_sel = d_final;
    }
  }
  if (!gpu_select && n_raw) {
    if (n_raw > ctx->h_raw_cap) {   // pinned staging for the copy-back
      if (ctx->h_raw) (void)hipHostFree(ctx->h_raw);
      ctx->h_raw = nullptr; ctx->h_raw_cap = 0;
      HIP_TRY(ctx, hipHostMalloc((void**)&ctx->h_raw, (size_t)ctx->raw_cap * sizeof(RawAln), hipHostMallocDefault));
      ctx->h_raw_cap = ctx->raw_cap;
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_raw, ctx->d_raw, (size_t)n_raw * sizeof(RawAln), hipMemcpyDeviceToHost, ctx->stream));
  }
  if (!gpu_select) {              // (the device filter recorded ev[3] and waited above)
    HIP_TRY(ctx, hipEventRecord(ctx->ev[3], ctx->stream));
    HIP_TRY(ctx, calitas_spin_sync(ctx->stream));
  }
  if (!(gpu_select && dev)) kernel_times(ctx, tm);   // calitas_search_hits asks later, while its row kernels run
  tm.scan_records = n_rec;
  tm.raw_alignments = n_raw;

  if (gpu_select) {
    tm.accepted_alignments = n_sel;
    tm.candidate_columns = ctx->h_counters[4];
    if (dev) {   // calitas_search_hits goes on from the device copy
      dev->valid = true; dev->d_final = d_sel; dev->n_sel = n_sel; dev->t_call = t_call; dev->crowded = ctx->h_counters[7] != 0;
      ctx->timing = tm;
      return CALITAS_OK;
    }
    // accepted alignments arrive in final order; only the coordinate conversion (GA:21-31, SGA:260-313) is left
    auto t0 = std::chrono::steady_clock::now();
    calitas_aln_t* result = nullptr;
    int rc = convert_selected(ctx, d_sel, n_sel, gh, p, step, &result);
    if (rc) return rc;
    tm.host_post_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    ctx->timing = tm;
    if (TUNE_GET("CALITAS_TRACE"))
      std::fprintf(stderr, "[calitas] search: scan %.3f ms, align %.3f ms, gpu total %.3f ms (incl. sort+filter on the GPU), copy+convert %.3f ms, call %.3f ms (%u records, %u raw, %u accepted)\n",
                   tm.scan_kernel_ms, tm.align_kernel_ms, tm.gpu_total_ms, tm.host_post_ms,
                   std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count(), n_rec, n_raw, n_sel);
    *n_out = n_sel;
    *out = result;
    return CALITAS_OK;
  }
  if (dev) dev->t_call = t_call;

  // ---- host: restore the reference's enumeration order, then the per-window filter (SGA:315-320) ----
  const RawAln* raw = ctx->h_raw;
  WorkerPool* pool = ref_owner(ctx)->pool;                                  // lanes share the owner's pool, one at a time
  std::lock_guard<std::mutex> host_lock(ref_owner(ctx)->host_mu);
  // Raw records arrive in atomic-append order.  They are bucketed by (guide, contig, 4096-window chunk), each bucket is
  // sorted by (window, strand list, end column, PAM) = fgbio's enumeration order (ascending end column) followed
  // by the PAM order of extendAndFilterRight (SGA:455), filtered window by window, and the buckets are concatenated.
  auto t0 = std::chrono::steady_clock::now();
  constexpr int WCHUNK_SHIFT = 12;
  const size_t n_contigs = ref.contigs.size();
  std::vector<uint64_t> chunk_base(n_contigs + 1, 0);   // bucket index base per contig (within one guide)
  for (size_t c = 0; c < n_contigs; c++)
    chunk_base[c + 1] = chunk_base[c] + ((window_count(ref.contigs[c].len, step) >> WCHUNK_SHIFT) + 1);
  const uint64_t buckets_per_guide = chunk_base[n_contigs];
  const size_t n_buckets = (size_t)(buckets_per_guide * (uint64_t)n_guides);
  auto bucket_of = [&](const RawAln& r) { return (size_t)(r.guide * buckets_per_guide + chunk_base[r.contig] + (r.window_k >> WCHUNK_SHIFT)); };
  std::vector<uint32_t> bucket_off(n_buckets + 1, 0);
  for (uint32_t i = 0; i < n_raw; i++) bucket_off[bucket_of(raw[i]) + 1]++;
  for (size_t b = 0; b < n_buckets; b++) bucket_off[b + 1] += bucket_off[b];
  std::vector<uint32_t> perm(n_raw);
  {
    std::vector<uint32_t> cur(bucket_off.begin(), bucket_off.end() - 1);
    for (uint32_t i = 0; i < n_raw; i++) perm[cur[bucket_of(raw[i])]++] = i;
  }
  const auto t_bucketed = std::chrono::steady_clock::now();
  std::vector<std::vector<calitas_aln_t>> bucket_out(n_buckets);
  {
    std::atomic<size_t> next(0);
    pool->run([&](int) {
      std::vector<std::pair<uint64_t, uint32_t>> keyed;
      std::vector<calitas_aln_t> win;
      std::vector<int> kept;
      for (;;) {
        size_t b = next.fetch_add(1);
        if (b >= n_buckets) break;
        const uint32_t lo = bucket_off[b], hi = bucket_off[b + 1];
        if (lo == hi) continue;
        keyed.clear();
        for (uint32_t i = lo; i < hi; i++) {
          const RawAln& r = raw[perm[i]];
          const uint64_t list = gh[r.guide].pam5 ? (r.dir == 1 ? 0 : 1) : (r.dir == 0 ? 0 : 1);   // 0 = forward-strand list (SGA:316)
          const uint64_t key = ((uint64_t)r.window_k << 24) | (list << 23) | ((uint64_t)r.t_end_guide << 7) | ((uint64_t)r.pad << 5) | (uint64_t)(r.pam + 1);
          keyed.emplace_back(key, perm[i]);
        }
        std::sort(keyed.begin(), keyed.end());
        auto& outv = bucket_out[b];
        size_t i = 0;
        while (i < keyed.size()) {
          const RawAln& f = raw[keyed[i].second];
          size_t j = i;
          while (j < keyed.size() && raw[keyed[j].second].window_k == f.window_k) j++;
          int64_t wa = 0, wb = 0;
          window_bounds(ref.runs.data(), (int64_t)ref.runs.size(), ref.contigs[f.contig].gbase, ref.contigs[f.contig].len, p.window_size,
                        step, f.window_k, wa, wb);
          if (win.size() < j - i) win.resize(j - i);
          for (size_t k = i; k < j; k++) raw_to_aln(raw[keyed[k].second], gh[f.guide], wa, wb, win[k - i]);
          window_filter(win.data(), (int)(j - i), max_total, p.max_overlap, kept);
          for (int k : kept) outv.push_back(win[k]);
          i = j;
        }
      }
    });
  }
  const auto t_filtered = std::chrono::steady_clock::now();
  std::vector<size_t> out_off(n_buckets + 1, 0);
  for (size_t b = 0; b < n_buckets; b++) out_off[b + 1] = out_off[b] + bucket_out[b].size();
  const size_t n_result = out_off[n_buckets];
  calitas_aln_t* result = (calitas_aln_t*)out_alloc(std::max<size_t>(1, n_result) * sizeof(calitas_aln_t));
  if (!result) return fail(ctx, CALITAS_EINVAL, "out of memory");
  {
    std::atomic<size_t> next(0);
    pool->run([&](int) {
      for (;;) {
        size_t b = next.fetch_add(1);
        if (b >= n_buckets) break;
        if (!bucket_out[b].empty()) std::memcpy(result + out_off[b], bucket_out[b].data(), bucket_out[b].size() * sizeof(calitas_aln_t));
      }
    });
  }
  tm.host_post_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  if (TUNE_GET("CALITAS_TRACE")) {
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    std::fprintf(stderr, "[calitas] host filter: bucket %.2f ms, sort+convert+filter %.2f ms, concat %.2f ms (%zu buckets)\n",
                 ms(t0, t_bucketed), ms(t_bucketed, t_filtered), ms(t_filtered, std::chrono::steady_clock::now()), n_buckets);
  }
  if (TUNE_GET("CALITAS_TRACE"))
    std::fprintf(stderr, "[calitas] search: scan %.3f ms, align %.3f ms, gpu total %.3f ms, host filter %.3f ms, call %.3f ms (%u records, %u raw, %zu accepted)\n",
                 tm.scan_kernel_ms, tm.align_kernel_ms, tm.gpu_total_ms, tm.host_post_ms,
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count(), n_rec, n_raw, n_result);
  tm.accepted_alignments = n_result;
  tm.candidate_columns = ctx->h_counters[4];   // end columns whose best bottom-row score reached minGuideScore inside a window
  ctx->timing = tm;

  *n_out = n_result;
  *out = result;
  return CALITAS_OK;
}

int calitas_search_impl(calitas_ctx* ctx, int32_t n_guides, const calitas_guide_t* guides, const calitas_params_t* params,
                        calitas_aln_t** out, uint64_t* n_out) {
  if (!ctx) return CALITAS_EINVAL;
  if (!out || !n_out) return fail(ctx, CALITAS_EINVAL, "NULL argument");
  *out = nullptr; *n_out = 0;
  SearchPlan pl;
  int rc = plan_search(ctx, n_guides, guides, params, pl);
  if (rc) return rc;
  return search_run(ctx, pl, out, n_out, nullptr, false);
}

// calitas_scan_candidates: plan, scan stage, records back (sorted).  Test and profiling entry; no lanes.
int calitas_scan_candidates_impl(calitas_ctx* ctx, int32_t n_guides, const calitas_guide_t* guides, const calitas_params_t* params,
                                 uint32_t** records, uint64_t* n_records, bool columnwise) {
  if (!ctx) return CALITAS_EINVAL;
  if (!records || !n_records) return fail(ctx, CALITAS_EINVAL, "NULL argument");
  *records = nullptr; *n_records = 0;
  SearchPlan pl;
  int rc = plan_search(ctx, n_guides, guides, params, pl);
  if (rc) return rc;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  rc = lane_prepare(ctx, pl);
  if (rc) return rc;
  uint32_t n_rec = 0;
  for (;;) {
    rc = launch_scan_stage(ctx, pl, ctx->stream, false, columnwise);
    if (rc) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_counters, ctx->d_counters, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    n_rec = ctx->h_counters[0];
    if (n_rec <= ctx->rec_cap) break;
    const uint64_t grown = (uint64_t)n_rec + n_rec / 4;        // 64-bit: a dense PAM-less scan can pass 3.4e9 records
    if (grown > 0xFFFFFFF0ull) return fail(ctx, CALITAS_EINVAL, "result volume exceeds 2^32 records");
    rc = ensure_buffers(ctx, (uint32_t)grown, ctx->raw_cap, pl.slab_per_rec, ctx->item_cap);
    if (rc) return rc;
  }
  static_assert(sizeof(ScanRecord) == 8, "two words per record");
  uint64_t* recs = (uint64_t*)out_alloc(std::max<size_t>(1, n_rec) * sizeof(ScanRecord));
  if (!recs) return fail(ctx, CALITAS_EINVAL, "out of memory");
  if (n_rec) HIP_TRY(ctx, hipMemcpy(recs, ctx->d_recs, (size_t)n_rec * sizeof(ScanRecord), hipMemcpyDeviceToHost));
  // {gword, info} little-endian as one 64-bit key: sort by info then gword would interleave; sort by (gword, info) instead
  std::sort(recs, recs + n_rec, [](uint64_t a, uint64_t b) {
    const uint64_t ka = (a << 32) | (a >> 32), kb = (b << 32) | (b >> 32);
    return ka < kb;
  });
  float ms = 0;
  (void)hipEventElapsedTime(&ms, ctx->t_scan0, ctx->t_scan1);
  ctx->timing = calitas_timing_t{};
  ctx->timing.scan_kernel_ms = ms; ctx->timing.scan_records = n_rec; ctx->timing.bases_scanned = pl.bases; ctx->timing.packed_bytes = (pl.bases + 3) / 4;
  *records = (uint32_t*)recs;
  *n_records = n_rec;
  return CALITAS_OK;
}

void calitas_default_version_and_stamp(const char* aligner_version, const char* time_stamp, std::string& version, std::string& stamp) {
  version = aligner_version ? aligner_version : "";
  stamp = time_stamp ? time_stamp : "";
  if (version.empty()) {  // EditasMetric.Version without a jar manifest: unknown-YYYY-MM-DD
    char b[32]; std::time_t t = std::time(nullptr); std::tm tmv; gmtime_r(&t, &tmv);
    std::strftime(b, sizeof b, "unknown-%Y-%m-%d", &tmv); version = b;
  }
  if (stamp.empty()) {    // RH:169-173 "EEE MMM dd HH:mm:ss z yyyy" in UTC
    char b[64]; std::time_t t = std::time(nullptr); std::tm tmv; gmtime_r(&t, &tmv);
    std::strftime(b, sizeof b, "%a %b %d %H:%M:%S UTC %Y", &tmv); stamp = b;
  }
}

// The finished text of a lane, device -> page-locked host.  Preferred: an SDMA engine through the HSA runtime (dma.hpp), after
// waiting for the lane's row kernels -- the CUs stay with the search kernels.  Otherwise the runtime's copy (a blit kernel) on the
// owner's low-priority copy stream (chunked / batch calls: one stream for all lanes) or on the lane's own stream.
static void dma_open_once(calitas_ctx* owner) {
  if (owner->dma_tried) return;
  std::lock_guard<std::mutex> lk(owner->host_mu);
  if (!owner->dma_tried) {
    const char* e = TUNE_GET("CALITAS_SDMA");
    if (!(e && std::atoi(e) == 0)) owner->dma.open(owner->device);
    owner->dma_tried = true;
  }
}

static int text_to_host(calitas_ctx* owner, calitas_ctx* lane, char* dst, const char* src, size_t n, std::mutex* copy_mu, double* ms_out,
                        hipEvent_t rows_done = nullptr) {
  dma_open_once(owner);
  if (lane->binned_late_check && src && src == binned_host_text(lane->binned)) {   // the rows kernel wrote the text into host memory itself
    if (rows_done) HIP_TRY(lane, calitas_spin_sync(rows_done)); else HIP_TRY(lane, calitas_spin_sync(lane->stream));
    g_marks.mark("rows-done");
    if (lane->mbox.host && lane->mbox.host[BIN_BOX_LATE] != 0)
      return fail(lane, CALITAS_EHIP, "binned rows kernel: a row's length differs between the two kernels (internal error)");
    std::memcpy(dst, src, n);
    if (ms_out) *ms_out = 0;
    return CALITAS_OK;
  }
  if (owner->dma.usable()) {
    // (rows_done: the caller recorded it behind the row kernels and other work may already be queued behind it on the stream)
    if (rows_done) HIP_TRY(lane, calitas_spin_sync(rows_done)); else HIP_TRY(lane, calitas_spin_sync(lane->stream));
    g_marks.mark("rows-done");
    const auto t0 = std::chrono::steady_clock::now();
    if (lane->binned_late_check && lane->mbox.host && lane->mbox.host[BIN_BOX_LATE] != 0)
      return fail(lane, CALITAS_EHIP, "binned rows kernel: a row's length differs between the two kernels (internal error)");
    if (owner->dma.copy_to_host(dst, src, n)) {
      g_marks.mark("copied");
      if (ms_out) *ms_out = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      return CALITAS_OK;
    }
    if (TUNE_GET("CALITAS_TRACE")) std::fprintf(stderr, "[calitas] SDMA copy declined (%s), using hipMemcpyAsync\n", DmaCopier::last_reason());
  }
  // a stream of its own for the copy whenever other work may be queued behind the rows on the lane's stream: the lanes of a chunked /
  // batch call, and the per-contig passes (rows_done given: the helper thread queues the next contig's kernels on ctx->stream)
  hipStream_t cs = (owner->copy_stream && (lane->parent || rows_done)) ? owner->copy_stream : lane->stream;
  if (cs != lane->stream) {
    hipEvent_t ready = rows_done;
    if (!ready) { HIP_TRY(lane, hipEventRecord(lane->rows_ready, lane->stream)); ready = lane->rows_ready; }
    std::lock_guard<std::mutex> lk(*copy_mu);
    HIP_TRY(lane, hipStreamWaitEvent(cs, ready, 0));
    HIP_TRY(lane, hipEventRecord(lane->ev[6], cs));
    HIP_TRY(lane, hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, cs));
    HIP_TRY(lane, hipEventRecord(lane->ev[7], cs));
  } else {
    HIP_TRY(lane, hipEventRecord(lane->ev[6], cs));
    HIP_TRY(lane, hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, cs));
    HIP_TRY(lane, hipEventRecord(lane->ev[7], cs));
  }
  HIP_TRY(lane, calitas_spin_sync(lane->ev[7]));
  if (lane->binned_late_check && lane->mbox.host && lane->mbox.host[BIN_BOX_LATE] != 0)
    return fail(lane, CALITAS_EHIP, "binned rows kernel: a row's length differs between the two kernels (internal error)");
  float ms = 0;
  (void)hipEventElapsedTime(&ms, lane->ev[6], lane->ev[7]);
  if (ms_out) *ms_out = ms;
  return CALITAS_OK;
}

// ---- calitas_search_hits ------------------------------------------------------------------------------------------------

// What one lane contributes to a hits.txt: rows on the device, or rows built by the host stages when a device stage declined.
struct LaneText {
  int rc = CALITAS_OK;
  const char* d_text = nullptr;
  uint64_t bytes = 0, rows = 0;
  uint64_t compact_bytes = 0;          // != 0: the device holds compact rows (post.hpp) of that many bytes; `bytes` is what they expand to
  bool on_host = false;
  bool in_place = false;               // the rows kernel wrote the text to its final place in the caller's page-locked buffer (LaneDest)
  std::string host_rows;
  const HitsWork* rows_by = nullptr;   // the general row stage that wrote d_text (its late flags are looked at once the text has been copied)
  const HitsExt* ext = nullptr;        // the caller's hits whose rows the caller writes into the text itself (HitsExtRows::fill_on_host) ...
  const uint64_t* ext_place = nullptr; // ... and where (HitsResult::ext_place)
  calitas_timing_t tm{};
};

// After the text of a lane has been copied (so its rows kernel is done): did the rows kernel of the general stage object to anything?
static int rows_late_check(calitas_ctx* lane, const LaneText& lt) {
  if (lt.rows_by && hits_late(lt.rows_by) != 0)
    return calitas_fail(lane, CALITAS_EHIP, "rows kernel: a row's length differs between the two kernels (internal error)");
  return CALITAS_OK;
}

// The compact rows of a lane (nbytes of `chromosome \t middle \n` in lt.d_text) become full rows at dst: the text crosses PCIe in
// pieces queued back to back on the DMA engine, and the worker pool expands what has landed while the rest is on the bus
// (post.cpp RowExpansion) -- the call's last expansion ends ~one piece after its copy instead of a whole expansion after it.
// *wrote: bytes written at dst, (size_t)-1 when the text does not hold lt.rows rows.
static int compact_rows_to_host(calitas_ctx* owner, calitas_ctx* lane, LaneText& lt, size_t nbytes, char* staging, const std::string& head,
                                const std::string& tail, char* dst, std::mutex* copy_mu, size_t* wrote, hipEvent_t rows_done = nullptr) {
  *wrote = 0;
  dma_open_once(owner);
  // (pieces of a sixth of the text, 256 KB to 2 MB -- less left to do behind the last piece of a short text: 2.028 against 2.001 ms)
  size_t piece = 2u << 20;
  if (const char* e = TUNE_GET("CALITAS_COMPACT_PIECE_KB")) piece = (size_t)std::max(64, std::atoi(e)) << 10;
  const bool in_host_text = lane->binned_late_check && lt.d_text == binned_host_text(lane->binned);
  auto whole = [&]() -> int {
    int r = text_to_host(owner, lane, staging, lt.d_text, nbytes, copy_mu, &lt.tm.hits_copy_ms, rows_done);
    if (r) return r;
    *wrote = expand_rows(staging, nbytes, lt.rows, head, tail, dst, owner->pool);
    g_marks.mark("expanded");
    return CALITAS_OK;
  };
  if (!owner->dma.usable() || in_host_text || nbytes < std::min<size_t>(1u << 20, 2 * piece)) return whole();   // (a short text: one copy, then the rows)
  // (rows_done: the caller recorded it behind the row kernels and other work may already be queued behind it on the stream)
  if (rows_done) HIP_TRY(lane, calitas_spin_sync(rows_done)); else HIP_TRY(lane, calitas_spin_sync(lane->stream));
  g_marks.mark("rows-done");
  if (lane->binned_late_check && lane->mbox.host && lane->mbox.host[BIN_BOX_LATE] != 0)
    return fail(lane, CALITAS_EHIP, "binned rows kernel: a row's length differs between the two kernels (internal error)");
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<unsigned long long> tickets;
  if (!owner->dma.start_pieces(staging, lt.d_text, nbytes, piece, tickets)) {
    if (TUNE_GET("CALITAS_TRACE")) std::fprintf(stderr, "[calitas] SDMA copy declined (%s), using hipMemcpyAsync\n", DmaCopier::last_reason());
    return whole();
  }
  auto job = expand_rows_begin(staging, nbytes, lt.rows, head, tail, dst, owner->pool);   // the workers wake while the first piece is on the bus
  bool ok = true;
  for (size_t i = 0; i < tickets.size(); i++) {
    if (!owner->dma.finish(tickets[i])) ok = false;           // (every ticket is waited for: nothing may land in a freed block)
    if (ok) expand_rows_arrived(*job, std::min(nbytes, (i + 1) * piece));
  }
  g_marks.mark("copied");
  lt.tm.hits_copy_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  *wrote = expand_rows_end(*job, ok);
  g_marks.mark("expanded");
  if (!ok) return fail(lane, CALITAS_EHIP, "SDMA copy failed");
  return CALITAS_OK;
}

// Kernel time of a lane's row stage once its last kernel is done.  General kernels: ev[4] .. ev[5] around hits_run.  Binned tail: no event
// sits between its kernels, so: end of the scan .. end of the rows kernel, less align_kernel + trace_kernel (by the stamps) -- the two bin
// kernels, the rows kernel and the kernel boundaries of the chain.
static double rows_stage_ms(calitas_ctx* lane, const calitas_timing_t& tm) {
  float ms = 0;
  if (lane->rows_ev0 >= 0) { (void)hipEventElapsedTime(&ms, lane->ev[lane->rows_ev0], lane->ev[5]); return ms; }
  (void)hipEventElapsedTime(&ms, lane->t_scan1, lane->ev[5]);
  return std::max(0.0, (double)ms - tm.align_kernel_ms);
}

// Whether this lane's search takes the binned tail (binned.hpp): one guide on the device path, a window size the bins handle, not
// a search planned as dense (rec_hint: per-contig passes of a permissive PAM-less search would crowd every bin), and not one at least
// as permissive as the last the bins declined on this reference.
static uint64_t guide_hash(const GuideDev& g) {
  uint64_t h = 1469598103934665603ull;                       // FNV-1a over what the kernels see of the guide
  auto mix = [&](uint64_t v) { for (int k = 0; k < 8; k++) { h ^= (v >> (8 * k)) & 0xFF; h *= 1099511628211ull; } };
  mix((uint64_t)g.L); mix((uint64_t)g.n_pams); mix((uint64_t)g.pam5);
  for (int i = 0; i < g.L; i++) mix(g.qmask[i]);
  for (int p = 0; p < g.n_pams; p++) { mix(g.pam_len[p]); for (int k = 0; k < g.pam_len[p]; k++) mix(g.pam_mask[p][k]); }
  return h;
}

// binned_possible: the search is one the per-bin kernels take at all.  binned_remembered: ... but this very guide met a crowded bin on this
// context before (a property of guide x reference: it would again).  binned_wanted: both considered -- what decides a lane's tail.
static bool binned_remembered(calitas_ctx* lane, const SearchPlan& pl) {
  const calitas_ctx* own = ref_owner(lane);
  const GuideDev& g = pl.gd[0];
  return own->bin_decl_pams == g.n_pams && own->bin_decl_L == g.L && g.min_guide_score <= own->bin_decl_min_score && own->bin_decl_guide == guide_hash(g);
}

static bool binned_possible(calitas_ctx* lane, const SearchPlan& pl) {
  const calitas_ctx* own = ref_owner(lane);
  if (!pl.bin_shift || pl.n_bins == 0 || pl.n_guides != 1 || pl.rec_hint != 0 || pl.general_tail) return false;
  if (!pl.owned && (pl.gw_lo != 0 || pl.gw_hi != ~0ull)) return false;   // (a window range of calitas_search: alignment records, no rows)
  if (TUNE_GET("CALITAS_HOST_FILTER") || TUNE_GET("CALITAS_HOST_HITS")) return false;
  // Which tail by default: the per-bin kernels wherever a call is one pass or two ranges (references up to 2 Gb: a rank's share of a
  // genome on 2-8 GPUs, a bacterial genome) -- 0.58 against 0.62 ms for an eighth of the hg38-sized genome, 0.164 against 0.190 ms for
  // an E. coli-sized one.  A call cut into three ranges (the whole hg38-sized genome on one GPU) is bound by its scans, and those
  // lose more to the per-bin kernels running beside them (many short waves) than the last range's tail gains: 2.39 ms per pass on
  // the general kernels against 2.45-2.53 (tools/sweep_lanes.py, profiles/r03_*).  The last range's tail has the chip to itself, and
  // since the leading ranges' rows cross PCIe compact (round 4) it ends the call: per-bin there, 2.15 / 2.23 / 2.17 against
  // 2.18 / 2.26 / 2.29 ms (three boxes, tools/sweep_env.py CALITAS_BINNED - last).  CALITAS_BINNED=1 / 0 / last force a choice.
  bool want = !pl.three_ranges || pl.owned || pl.last_range;   // (a stretch that cuts a contig: only the bins can own it)
  if (const char* e = TUNE_GET("CALITAS_BINNED")) {
    if (std::strcmp(e, "last") == 0) want = !pl.narrow_tail;
    else if (std::strcmp(e, "from1") == 0) want = !pl.three_ranges || pl.owned || pl.range_index >= 1;   // (experiment: every range but the first)
    else want = std::atoi(e) != 0;
  }
  if (!want) return false;
  if (pl.p.max_overlap < 1 || own->ref.contigs.size() >= (1u << 18) - 1) return false;
  return true;
}

static bool binned_wanted(calitas_ctx* lane, const SearchPlan& pl) { return binned_possible(lane, pl) && !binned_remembered(lane, pl); }

// The per-call constants of a lane's row stage -- and the cleared scratch of the bins when the lane takes the binned tail --, queued on
// its stream ahead of its kernels (callers that queue a wait for a scan on that stream do this first).
static hipError_t queue_row_constants(calitas_ctx* lane, const SearchPlan& pl, const RowStrings& rs) {
  hipError_t e = hits_prepare(&lane->hits, rs, lane->stream);
  if (e == hipSuccess && binned_wanted(lane, pl)) e = binned_prepare(&lane->binned, pl.n_bins, lane->stream);
  return e;
}

// Everything small a lane's search needs on the device in one launch (kernels.hpp, LaneSetupArgs): guide constants, cleared counters,
// and -- rs given -- what queue_row_constants would queue (constant row strings, the row stage's counts, the bins' scratch).  Returns
// false when the search does not fit that form (several guides, very long parameter strings, CALITAS_LANE_SETUP=0): the caller queues
// the separate commands then (*done = false).
static int queue_lane_setup(calitas_ctx* lane, const SearchPlan& pl, const RowStrings* rs, hipStream_t stream, bool* done, bool with_scan_inputs = true) {
  *done = false;
  if (pl.n_guides != 1) return CALITAS_OK;
  if (const char* e = TUNE_GET("CALITAS_LANE_SETUP")) if (std::atoi(e) == 0) return CALITAS_OK;
  LaneSetupArgs a{};
  if (rs) {
    HitsSetup hs{};
    HIP_TRY(lane, hits_prepare_host(&lane->hits, *rs, &hs));
    if (hs.blob_bytes > LANE_SETUP_BLOB) {                      // (the strings are assembled: bring them over the usual way)
      HIP_TRY(lane, hits_prepare(&lane->hits, *rs, stream));
    } else {
      std::memcpy(a.blob, hs.blob, hs.blob_bytes);
      a.blob_bytes = hs.blob_bytes; a.d_blob = hs.d_blob; a.d_row_counts = hs.d_counts;
    }
    if (binned_wanted(lane, pl)) {
      void* clear = nullptr;
      size_t bytes = 0;
      HIP_TRY(lane, binned_prepare_host(&lane->binned, pl.n_bins, &clear, &bytes));
      if (bytes > 0xFFFFFFF0u) HIP_TRY(lane, hipMemsetAsync(clear, 0, bytes, stream));
      else { a.clear = static_cast<uint4*>(clear); a.clear_bytes = (uint32_t)bytes; }
    }
  }
  if (with_scan_inputs) { a.guide = pl.gd[0]; a.d_guides = lane->d_guides; a.d_counters = lane->d_counters; }
  HIP_TRY(lane, launch_lane_setup(a, stream));
  g_marks.mark("lane-setup");
  *done = true;
  return CALITAS_OK;
}

// Where a lane's text finally goes, asked for when its row kernel is about to be launched: page-locked memory the device can address
// and the room there.  false: not known / not addressable -- the text takes the device buffer and the copy.
struct LaneDest { std::function<bool(char** dst, uint64_t* cap)> get; };

constexpr int kOwnedDeclined = -1000;   // (internal) a lane of an owned range (SearchPlan::owned) met bins it leaves to the general kernels
static std::atomic<double> g_pass_ms[2]; // (trace only; written by the one thread that runs a sequential call's passes -- atomics: two contexts may run such calls at once, and the figures are then both calls')
constexpr int kExtDeclined = -1001;     // (internal) a pass that brings hits of the caller's (HitsExt) met a stage the device declines: the caller merges on the host

struct LaneText;
static int lane_rows_binned(calitas_ctx* lane, const SearchPlan& pl, bool prelaunched, const RowStrings& rs, LaneText& lt, bool prepared,
                            bool* declined, const LaneDest* dest = nullptr, uint32_t* decline_flags = nullptr);

// One lane from the scan stage (queued here, or already queued by the caller) to its finished rows.
// hits_prepared: the caller queued hits_prepare on the lane's stream already -- *before* the stream's wait for the scan, so that
// the constants are in place while the scan runs instead of sitting between the end of the scan and align_kernel.
static int lane_rows(calitas_ctx* lane, const SearchPlan& pl, bool prelaunched, const RowStrings& rs, const std::string& guide_id,
                     const std::string& version, const std::string& stamp, LaneText& lt, bool hits_prepared = false, const LaneDest* dest = nullptr,
                     const HitsExtSource* ext_source = nullptr, int ext_contig = 0) {
  calitas_ctx* own = ref_owner(lane);
  const PackedRef& ref = own->ref;
  const calitas_params_t& p = pl.p;
  const GuideHost& gh = pl.gh[0];
  DeviceSel dev;
  calitas_aln_t* alns = nullptr;
  uint64_t n_alns = 0;
  bool resume = false;
  lane->binned_late_check = false;
  // A stretch (SearchPlan::owned) is the bins' to decide.  Where one of its bins is crowded -- the guide meets a repeat: more alignments
  // than a wave holds -- the general kernels finish it from the same alignments and keep the rows the stretch owns (HitsOwn): exact
  // unless a chain of overlapping hits reaches from the edge of the aligned context into the stretch, which they detect (HITS_FLAG_HALO)
  // and which the bins' own halo flag says as well; then, and for anything else, the caller searches the touched contigs whole.
  bool own_general = false;
  if (pl.owned && !binned_possible(lane, pl)) return kOwnedDeclined;   // (no bins for this window size / forced off: the caller's whole-contig path)
  if (pl.owned && binned_remembered(lane, pl)) {                        // this guide crowded a bin here before: the general kernels at once
    if (TUNE_GET("CALITAS_OWN_GENERAL_OFF")) return kOwnedDeclined;
    own_general = true;
  } else if (binned_wanted(lane, pl)) {
    bool declined = false;
    uint32_t why = 0;
    int rc = lane_rows_binned(lane, pl, prelaunched, rs, lt, hits_prepared, &declined, dest, &why);
    if (rc || !declined) return rc;
    if (pl.owned) {
      if (why != BIN_FLAG_CROWDED || TUNE_GET("CALITAS_OWN_GENERAL_OFF")) return kOwnedDeclined;
      own_general = true;
    }
    // the bins declined: the raw alignments are where the general kernels expect them, the lane's counters in h_counters
    resume = true;
    hits_prepared = false;                                   // binned_run consumed the row constants
  }
  if (!hits_prepared && !TUNE_GET("CALITAS_HOST_HITS")) HIP_TRY(lane, hits_prepare(&lane->hits, rs, lane->stream));   // ahead of the lane's kernels
  const auto t_pass0 = std::chrono::steady_clock::now();
  int rc = search_run(lane, pl, &alns, &n_alns, &dev, prelaunched, resume);
  if (rc) return rc;
  lt.tm = lane->timing;
  if (own_general && !dev.valid) { calitas_free(alns); return kOwnedDeclined; }
  const HitsExt* ext = nullptr;             // the caller's own hits of this contig: asked for now, the search kernels of the pass are behind us
  const auto t_pass1 = std::chrono::steady_clock::now();
  if (ext_source && ext_source->get(ext_contig, &ext) != 0) { calitas_free(alns); return kExtDeclined; }
  if (ext_source) {                         // (CALITAS_TRACE of the per-contig passes: the search kernels' part of a pass, and its wait for the caller's hits)
    const auto t_pass2 = std::chrono::steady_clock::now();
    g_pass_ms[0].store(g_pass_ms[0].load(std::memory_order_relaxed) + std::chrono::duration<double, std::milli>(t_pass1 - t_pass0).count(), std::memory_order_relaxed);
    g_pass_ms[1].store(g_pass_ms[1].load(std::memory_order_relaxed) + std::chrono::duration<double, std::milli>(t_pass2 - t_pass1).count(), std::memory_order_relaxed);
  }
  if (ext && !dev.valid && n_alns == 0) {   // nothing of the reference's own on this contig: the row stage still places the caller's hits
    dev.valid = true; dev.d_final = nullptr; dev.n_sel = 0; dev.crowded = true;
  }
  if (ext && (!dev.valid || TUNE_GET("CALITAS_HOST_HITS"))) { calitas_free(alns); return kExtDeclined; }
  if (dev.valid && !TUNE_GET("CALITAS_HOST_HITS")) {
    // removeOverlaps, ReferenceHit.sort and the rows on the device (hits.hip); only text crosses PCIe
    int max_pam = 0;
    for (auto& q : gh.pams) max_pam = std::max<int>(max_pam, (int)q.size());
    const int score_hi = pl.sc.match * (int)gh.protospacer.size() + pl.sc.pam_match * max_pam;
    const int worst_gap = std::max(iabs(pl.sc.query_gap), std::max(iabs(pl.sc.target_gap), iabs(pl.sc.mismatch)));
    const int score_lo = pl.gd[0].min_guide_score - iabs(pl.sc.pam_mismatch) * max_pam - worst_gap * (p.max_gaps_between_guide_and_pam + 1);
    if (hits_supported(ref.contigs.size(), p.max_overlap, score_lo, score_hi)) {
      if (lane->hits_names_serial != own->ref_serial) {
        HIP_TRY(lane, hits_set_names(&lane->hits, ref.names, lane->stream));
        lane->hits_names_serial = own->ref_serial;
      }
      HitsRef hr{own->d_codes, own->d_mask, own->d_runs, (int64_t)ref.runs.size(), own->d_contigs, (int)ref.contigs.size()};
      HitsResult res{};
      HitsOwn ho;
      if (own_general) {
        // from where on every hit that could overlap is known: the first aligned window's start + a window (hits of the windows left of it
        // end before that) + the longest hit; a context that starts with its contig knows everything
        ho.lo = pl.own_lo; ho.hi = pl.own_hi;
        uint64_t w0 = 0;
        for (size_t c = 0; c < ref.contigs.size(); c++) {
          const uint64_t nw = window_count(ref.contigs[c].len, pl.step);
          if (pl.gw_lo < w0 + nw || c + 1 == ref.contigs.size()) {
            const uint64_t pos = (pl.gw_lo - w0) * (uint64_t)pl.step;
            ho.safe = pos == 0 ? ((uint64_t)c << 32) : (((uint64_t)c << 32) | (pos + (uint64_t)p.window_size + CALITAS_MAX_OPS));
            break;
          }
          w0 += nw;
        }
      }
      HIP_TRY(lane, hipEventRecord(lane->ev[4], lane->stream));
      lane->rows_ev0 = 4;
      HIP_TRY(lane, hits_run(&lane->hits, hr, dev.d_final, dev.n_sel, lane->d_guides, own->d_win_base, own->d_win, rs, p.max_overlap, score_hi,
                             pl.gd[0].span + 1 + p.max_gaps_between_guide_and_pam + max_pam, dev.crowded ? 0u : (uint32_t)((p.window_size + pl.step - 1) / pl.step),
                             lane->stream, &res, ext, own_general ? &ho : nullptr));
      HIP_TRY(lane, hipEventRecord(lane->ev[5], lane->stream));
      g_marks.mark("rows-queued");
      kernel_times(lane, lt.tm);          // while out_kernel runs
      if (res.flags == 0) {
        lt.d_text = res.d_text; lt.bytes = res.text_bytes; lt.rows = res.n_rows; lt.rows_by = lane->hits;
        if (res.ext_place) { lt.ext = ext; lt.ext_place = res.ext_place; }
        if (own_general) lt.tm.owned_general_lanes = 1;
        return CALITAS_OK;
      }
      if (own_general) return kOwnedDeclined;
      if (TUNE_GET("CALITAS_TRACE")) std::fprintf(stderr, "[calitas] search_hits: device rows declined (flags %u), finishing on the host\n", res.flags);
    }
    if (ext) return kExtDeclined;       // (a contig without hits of the caller's needs no merge: any tail writes its text)
  }
  // host tail: the same stages as calitas_hits_tsv (one lane at a time: they share the owner's worker pool)
  std::lock_guard<std::mutex> host_lock(own->host_mu);
  if (dev.valid) {
    kernel_times(lane, lt.tm);
    rc = convert_selected(lane, dev.d_final, dev.n_sel, pl.gh, p, pl.step, &alns);
    if (rc) return rc;
    n_alns = dev.n_sel;
  }
  uint64_t rows = 0;
  char* text = hits_tsv(ref, gh, guide_id, p, alns, n_alns, version, stamp, &rows, own->pool, out_alloc, nullptr, 0);
  calitas_free(alns);
  if (!text) return fail(lane, CALITAS_EINVAL, "out of memory");
  lt.on_host = true;
  lt.host_rows.assign(text + rs.header.size());
  calitas_free(text);
  lt.bytes = lt.host_rows.size(); lt.rows = rows;
  return CALITAS_OK;
}

// The binned tail of one lane: scan (unless queued by the caller) -> align_kernel -> trace_kernel (alignments into the bins as well)
// -> bin_hits_kernel -> bin_rows_kernel, ONE host round trip (the rows kernel posts counters, rows, bytes and flags as it starts).
// prepared: the caller queued hits_prepare and binned_prepare on the lane's stream already, ahead of its wait for the scan.
// *declined: a bin was crowded / a repeat outran the halo / a lane buffer overflowed: nothing is lost, the general kernels take over.
static int lane_rows_binned(calitas_ctx* lane, const SearchPlan& pl, bool prelaunched, const RowStrings& rs, LaneText& lt, bool prepared,
                            bool* declined, const LaneDest* dest, uint32_t* decline_flags) {
  if (decline_flags) *decline_flags = 0;
  calitas_ctx* own = ref_owner(lane);
  const PackedRef& ref = own->ref;
  const calitas_params_t& p = pl.p;
  const GuideHost& gh = pl.gh[0];
  *declined = false;
  HIP_TRY(lane, hipSetDevice(lane->device));
  if (!prelaunched) {
    int rc = lane_prepare(lane, pl);
    if (rc) return rc;
    rc = ensure_window_table(lane, pl, lane->stream);
    if (rc) return rc;
  }
  const auto t_call = std::chrono::steady_clock::now();
  if (!prelaunched) {
    // one launch for everything small the lane needs (guide constants, cleared counters, row constants, the bins' scratch), then the scan
    bool one = false;
    int rc = prepared ? CALITAS_OK : queue_lane_setup(lane, pl, &rs, lane->stream, &one);
    if (rc) return rc;
    if (one) prepared = true;
    rc = launch_scan_stage(lane, pl, lane->stream, one);
    if (rc) return rc;
  }
  if (!prepared) HIP_TRY(lane, queue_row_constants(lane, pl, rs));
  if (lane->hits_names_serial != own->ref_serial) {
    HIP_TRY(lane, hits_set_names(&lane->hits, ref.names, lane->stream));
    lane->hits_names_serial = own->ref_serial;
  }
  int max_pam = 0;
  for (auto& q : gh.pams) max_pam = std::max<int>(max_pam, (int)q.size());
  const BinnedGeometry geo{own->d_bin_base, own->d_bin_contig, (int)ref.contigs.size(), pl.bin_first, pl.n_bins, (uint32_t)pl.bin_shift};
  const BinnedParams bp{p.window_size, pl.step, pl.max_total, p.max_overlap, pl.gd[0].span + 1 + p.max_gaps_between_guide_and_pam + max_pam,
                        pl.own_lo, pl.own_hi};
  const HitsRef hr{own->d_codes, own->d_mask, own->d_runs, (int64_t)ref.runs.size(), own->d_contigs, (int)ref.contigs.size()};
  ScanArgs sa; AlignArgs aa;
  fill_kernel_args(lane, pl, sa, aa);
  binned_fill_align_args(lane->binned, geo, aa);
  HIP_TRY(lane, launch_align(aa, narrow_blocks(pl.narrow_tail ? TUNE_GET("CALITAS_ALIGN_BLOCKS_NARROW") : TUNE_GET("CALITAS_ALIGN_BLOCKS"), aa.pack16 && !aa.sp.per_matrix ? (pl.narrow_tail ? kAlignBlocksPackedNarrow : kAlignBlocksPacked) : kAlignBlocks), lane->stream));
  // (no events on these dispatches: each would hold back the kernel behind it by ~5 us; the kernels stamp the device's wall clock instead)
  HIP_TRY(lane, launch_trace(aa, narrow_blocks(pl.narrow_tail ? TUNE_GET("CALITAS_TRACE_BLOCKS_NARROW") : TUNE_GET("CALITAS_TRACE_BLOCKS"), kTraceBlocks), lane->stream, nullptr));
  HIP_TRY(lane, binned_run(lane->binned, &lane->hits, geo, hr, lane->d_raw, lane->d_guides, own->d_win_base, own->d_win, bp, lane->d_counters, lane->stream,
                           &lane->mbox, nullptr, nullptr, lane->ev[5], dest == nullptr));
  char* host_dst = nullptr;
  uint64_t host_dst_cap = 0;
  if (dest) {
    // the rows kernel goes out once the text's final place is known (the byte counts of the ranges before this one: their row kernels
    // have started by then) and writes there itself -- no copy behind it
    if (!dest->get(&host_dst, &host_dst_cap)) { host_dst = nullptr; host_dst_cap = 0; }
    HIP_TRY(lane, binned_rows(lane->binned, &lane->hits, geo, hr, lane->d_raw, lane->d_guides, own->d_win_base, own->d_win, bp, lane->d_counters, lane->stream,
                              &lane->mbox, lane->ev[5], host_dst, host_dst_cap));
  }
  lane->rows_ev0 = -1;                                       // (the row stage's time: binned_rows_ms)
  g_marks.mark("queued-binned");
  HIP_TRY(lane, mailbox_wait(lane->mbox, lane->stream));
  g_marks.mark("binned-counts");
  for (int k = 0; k < 8; k++) lane->h_counters[k] = lane->mbox.host[BIN_BOX_COUNTERS + k];
  lane->align_ms_by_stamps = binned_stamp_ms(lane->binned, lane->mbox, 0, 1);                    // align_kernel + trace_kernel
  const uint32_t n_rec = lane->h_counters[0], n_raw = lane->h_counters[1], n_items = lane->h_counters[3];
  if (lane->h_counters[2] != 0) return fail(lane, CALITAS_EHIP, "aligner kernel reported an inconsistent traceback (internal error)");
  uint32_t flags = lane->mbox.host[BIN_BOX_FLAGS];
  const bool overflow = n_rec > lane->rec_cap || n_raw > lane->raw_cap || n_items > lane->item_cap;
  if (overflow || (flags & ~BIN_FLAG_TEXT)) {
    if (!overflow) {   // a property of this search on this reference: remember it
      own->bin_decl_L = pl.gd[0].L; own->bin_decl_pams = pl.gd[0].n_pams; own->bin_decl_min_score = pl.gd[0].min_guide_score;
      own->bin_decl_guide = guide_hash(pl.gd[0]);
      if (TUNE_GET("CALITAS_TRACE")) std::fprintf(stderr, "[calitas] binned tail declined (flags %u): finishing on the general kernels\n", flags);
    }
    HIP_TRY(lane, calitas_spin_sync(lane->stream));          // the rows kernel returns at once; nothing of it may linger over the retry
    *declined = true;
    if (decline_flags) *decline_flags = overflow ? ~0u : (flags & ~BIN_FLAG_TEXT);
    return CALITAS_OK;
  }
  uint64_t bytes = (uint64_t)lane->mbox.host[BIN_BOX_BYTES] | ((uint64_t)lane->mbox.host[BIN_BOX_BYTES + 1] << 32);
  if (flags & BIN_FLAG_TEXT) {                               // the text buffer was a guess: grow it, the rows kernel once more
    HIP_TRY(lane, binned_rerun_rows(lane->binned, &lane->hits, geo, hr, lane->d_raw, lane->d_guides, own->d_win_base, own->d_win, bp, lane->d_counters, bytes,
                                    lane->stream, &lane->mbox, lane->ev[5]));
    HIP_TRY(lane, mailbox_wait(lane->mbox, lane->stream));
    flags = lane->mbox.host[BIN_BOX_FLAGS];
    if (flags) return fail(lane, CALITAS_EHIP, "binned rows kernel: flags " + std::to_string(flags) + " after the text buffer was grown (internal error)");
  }
  calitas_timing_t tm{};
  tm.bases_scanned = pl.bases; tm.packed_bytes = (pl.bases + 3) / 4;
  tm.scan_records = n_rec; tm.raw_alignments = n_raw; tm.candidate_columns = lane->h_counters[4];
  tm.accepted_alignments = lane->mbox.host[BIN_BOX_ACCEPTED];
  if (TUNE_GET("CALITAS_TRACE"))
    std::fprintf(stderr, "[calitas] binned tail: %u bins, %u of them by a whole wave, %u rows, %llu bytes\n", pl.n_bins, (unsigned)lane->mbox.host[BIN_BOX_COMPLEX],
                 (unsigned)lane->mbox.host[BIN_BOX_ROWS], (unsigned long long)bytes);
  {                                                          // scan: its events; the kernels behind it: their stamps
    float ms = 0;
    (void)hipEventElapsedTime(&ms, lane->t_scan0, lane->t_scan1); tm.scan_kernel_ms = ms;
    tm.align_kernel_ms = lane->align_ms_by_stamps;
    tm.gpu_total_ms = tm.scan_kernel_ms + binned_stamp_ms(lane->binned, lane->mbox, 0, 2);       // ... + the bin kernels, up to the start of the rows kernel
  }
  tm.binned_lanes = 1;
  lane->timing = tm;
  lt.tm = tm;
  // (a short text is already on its way into the lane's page-locked buffer: text_to_host only waits for the kernel)
  if (host_dst) { lt.in_place = bytes <= host_dst_cap; lt.d_text = lt.in_place ? host_dst : binned_text(lane->hits); }
  else lt.d_text = bytes <= binned_host_cap(lane->binned) ? binned_host_text(lane->binned) : binned_text(lane->hits);
  lt.bytes = bytes; lt.rows = lane->mbox.host[BIN_BOX_ROWS];
  lane->binned_late_check = true;
  (void)t_call;
  return CALITAS_OK;
}

// The host threads behind lanes 1..K-1 of a chunked search (the caller's thread drives lane 0).  They live as long as the lanes:
// starting two threads took ~110 us of every call and joining them ~40 us after the last copy had finished -- on the caller's clock.
struct LaneThreads {
  std::vector<std::thread> threads;
  std::mutex m;
  std::condition_variable cv;
  unsigned long gen = 0;
  bool stop = false;
  size_t k = 0;                                    // lanes of the current job (worker i runs job(i) when i < k)
  std::function<void(size_t)> job;
  std::atomic<size_t> remaining{0};
  std::atomic<bool> threw{false};                  // a job ended with an exception (std::bad_alloc, ...): the call fails with CALITAS_EHIP
  std::string what;                                // ... and says which (the first one's text; under m)
  void note(const char* text) {
    std::lock_guard<std::mutex> lk(m);
    if (!threw.load(std::memory_order_relaxed)) what = text ? text : "";
    threw.store(true, std::memory_order_relaxed);
  }
  // Runs fn: an exception must not take the process down (a lane thread has no caller to unwind to), and must not be lost either.
  template <typename F>
  void guarded(F&& fn) {
    try { fn(); }
    catch (const std::exception& e) { note(e.what()); }
    catch (...) { note("an exception that is not a std::exception"); }
  }
  std::string failure() { std::lock_guard<std::mutex> lk(m); return "a lane of the search ended with an exception: " + (what.empty() ? std::string("(no text)") : what); }
  ~LaneThreads() {
    { std::lock_guard<std::mutex> lk(m); stop = true; gen++; }
    cv.notify_all();
    for (auto& t : threads) t.join();
  }
  void ensure(size_t lanes) {                      // workers for lanes 1 .. lanes-1
    while (threads.size() + 1 < lanes) {
      const size_t i = threads.size() + 1;
      threads.emplace_back([this, i] {
        unsigned long seen = 0;
        for (;;) {
          {
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return gen != seen; });
            seen = gen;
            if (stop) return;
            if (i >= k) continue;
          }
          guarded([&] { job(i); });
          if (remaining.fetch_sub(1, std::memory_order_acq_rel) == 1) {   // the last lane: wake a caller that has stopped spinning
            { std::lock_guard<std::mutex> lk(m); }
            done_cv.notify_all();
          }
        }
      });
    }
  }
  void start(size_t lanes, std::function<void(size_t)> fn) {
    { std::lock_guard<std::mutex> lk(m); job = std::move(fn); k = lanes; remaining.store(lanes - 1, std::memory_order_relaxed); threw.store(false, std::memory_order_relaxed); gen++; }
    cv.notify_all();
  }
  std::condition_variable done_cv;
  void wait() {                                    // the last lane to finish is the end of the call: watch for it (spinning, then yielding) for 2 ms, then sleep until it says so
    Backoff spin;
    while (remaining.load(std::memory_order_acquire) != 0) {
      // (the caller's own lane of an hg38-sized call ends 0.3 ms before the last one: woken from a condition variable it returned
      // 35 us after that lane had finished)
      if (spin.spins < 256 || spin.waited_us() < 2000) { spin.pause(); continue; }
      std::unique_lock<std::mutex> lk(m);
      done_cv.wait(lk, [&] { return remaining.load(std::memory_order_acquire) == 0; });
    }
  }
};

// Child contexts of a chunked search: own stream (high priority), buffers and scratch; the parent's reference.
static int ensure_lanes(calitas_ctx* ctx, size_t k) {
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  int least = 0, greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
  // (Reserving CUs for the lanes with hipExtStreamCreateWithCUMask on the scan stream was tried: 8 of 256 CUs masked out cost the
  // scan 9 %, 32 cost 80 %, and the lanes' small kernels did not get faster.)
  if (!ctx->scan_stream) HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->scan_stream, hipStreamNonBlocking, least));   // (two levels on this device: 0 and -1)
  // The runtime performs these device-to-host copies with a blit kernel (rocprofv3: __amd_rocclr_copyBuffer) that shares the CUs
  // with everything else.  On a high-priority stream it held up the other lane's small kernels for the whole copy (rocprofv3
  // timeline: a 5 us merge pass took 370 us); on the lowest priority the lanes' kernels get their slots first.  A copy kernel of
  // our own with a small grid reached the same 55 GB/s but slowed the other lane more, so the runtime's copy stays.
  if (!ctx->copy_stream) HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->copy_stream, hipStreamNonBlocking, least));
  if (!ctx->lane_threads) ctx->lane_threads = new LaneThreads();
  ctx->lane_threads->ensure(k);
  while (ctx->lanes.size() < k) {
    calitas_ctx* c = new calitas_ctx();
    c->device = ctx->device; c->parent = ctx;
    int lane_prio = greatest;
    if (const char* e = TUNE_GET("CALITAS_LANE_PRIO")) {       // experiment: the tails do not outrank the scan ("low": none does; "low0" / "low01": the first / the first two lanes)
      const size_t idx = ctx->lanes.size();
      if (std::strcmp(e, "low") == 0 || (std::strcmp(e, "low0") == 0 && idx == 0) || (std::strcmp(e, "low01") == 0 && idx <= 1)) lane_prio = least;
    }
    bool ok = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, lane_prio) == hipSuccess;
    for (int i = 0; i < 8; i++) ok = ok && hipEventCreateWithFlags(&c->ev[i], i < 6 ? hipEventReleaseToDevice : hipEventDefault) == hipSuccess;   // as in calitas_create
    ok = ok && hipEventCreateWithFlags(&c->scan_done, hipEventReleaseToDevice) == hipSuccess;   // timed: it also brackets the scan
    ok = ok && hipEventCreateWithFlags(&c->rows_ready, hipEventDisableTiming | hipEventReleaseToDevice) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&c->inputs_ready, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->d_counters, 8 * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&c->h_counters, 8 * sizeof(uint32_t), hipHostMallocDefault) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->d_guides, sizeof(GuideDev) * MAX_GUIDES) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&c->h_guides, sizeof(GuideDev) * MAX_GUIDES, hipHostMallocDefault) == hipSuccess;
    ctx->lanes.push_back(c);
    if (!ok) { calitas_destroy_lanes(ctx); return fail(ctx, CALITAS_EHIP, "could not create a search lane"); }
  }
  return CALITAS_OK;
}

// ctx->side: a child context with a stream and buffers of its own, the parent's reference and worker pool (see ctx.hpp).
int calitas_side_context(calitas_ctx* ctx, calitas_ctx** side, int which) {
  *side = nullptr;
  if (ctx->device < 0) return fail(ctx, CALITAS_ENODEV, "host-only context");
  calitas_ctx*& slot = which ? ctx->side2 : ctx->side;
  if (!slot) {
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    calitas_ctx* c = new calitas_ctx();
    c->device = ctx->device; c->parent = ctx;
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; i < 8; i++) ok = ok && hipEventCreateWithFlags(&c->ev[i], i < 6 ? hipEventReleaseToDevice : hipEventDefault) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->d_counters, 8 * sizeof(uint32_t)) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&c->h_counters, 8 * sizeof(uint32_t), hipHostMallocDefault) == hipSuccess;
    ok = ok && hipMalloc((void**)&c->d_guides, sizeof(GuideDev) * MAX_GUIDES) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&c->h_guides, sizeof(GuideDev) * MAX_GUIDES, hipHostMallocDefault) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); calitas_destroy(c); return fail(ctx, CALITAS_EHIP, "could not create the side context"); }
    slot = c;
  }
  *side = slot;
  return CALITAS_OK;
}

void calitas_destroy_lanes(calitas_ctx* ctx) {
  delete ctx->lane_threads; ctx->lane_threads = nullptr;
  for (calitas_ctx* c : ctx->lanes) calitas_destroy(c);
  ctx->lanes.clear();
  if (ctx->scan_stream) { (void)hipStreamDestroy(ctx->scan_stream); ctx->scan_stream = nullptr; }
  for (auto& st : ctx->scan_more) if (st) { (void)hipStreamDestroy(st); st = nullptr; }
  if (ctx->copy_stream) { (void)hipStreamDestroy(ctx->copy_stream); ctx->copy_stream = nullptr; }
}

// Contig ranges [first, last) of a chunked search: cut at contig boundaries (removeOverlaps groups and the final sort never
// cross a contig), sized by `weights`.
static std::vector<std::pair<int, int>> chunk_ranges(const PackedRef& ref, const std::vector<double>& weights) {
  const int n = (int)ref.contigs.size();
  std::vector<std::pair<int, int>> out;
  double wsum = 0;
  for (double w : weights) wsum += w;
  uint64_t total = ref.total_bases, acc = 0;
  double target = 0;
  int first = 0;
  size_t k = 0;
  for (int c = 0; c < n && k + 1 < weights.size(); c++) {
    acc += ref.contigs[c].len;
    const double goal = (target + weights[k]) / wsum * (double)total;
    const uint64_t next = c + 1 < n ? ref.contigs[c + 1].len : 0;
    // close the chunk after contig c when that lands nearer to the goal than taking one more contig would
    if ((double)acc >= goal || (double)acc + (double)next / 2 > goal) {
      if (c + 1 < n) { out.emplace_back(first, c + 1); first = c + 1; target += weights[k]; k++; }
    }
  }
  out.emplace_back(first, n);
  return out;
}

// Frees every scratch buffer of the context and its lanes (not the reference): the state a memory-bounded retry starts from.
static void release_scratch(calitas_ctx* ctx) {
  (void)hipDeviceSynchronize();
  calitas_destroy_lanes(ctx);
  (void)hipFree(ctx->d_recs); (void)hipFree(ctx->d_raw); (void)hipFree(ctx->d_slab); (void)hipFree(ctx->d_items);
  ctx->d_recs = nullptr; ctx->d_raw = nullptr; ctx->d_slab = nullptr; ctx->d_items = nullptr;
  ctx->rec_cap = ctx->raw_cap = ctx->item_cap = 0; ctx->slab_cap = 0;
  if (ctx->h_raw) { (void)hipHostFree(ctx->h_raw); ctx->h_raw = nullptr; ctx->h_raw_cap = 0; }
  select_destroy(ctx->select); ctx->select = nullptr;
  hits_destroy(ctx->hits); ctx->hits = nullptr; ctx->hits_names_serial = ~0ull;
  hits_destroy(ctx->hits_alt); ctx->hits_alt = nullptr; ctx->hits_alt_names_serial = ~0ull;
  binned_destroy(ctx->binned); ctx->binned = nullptr;
}

// calitas_search_hits when one pass does not fit the device (a PAM-less search at max-guide-diffs 8 on a whole genome keeps 2.4 KB of
// strip per scan record and yields ~27 rows per kilobase): one pass per contig, one after the other on this context's own stream,
// every contig's text copied to the host before the next one starts; the texts are concatenated at the end (removeOverlaps groups
// and the final sort never cross a contig).
// With a sink the pieces (header, then every contig's rows in at most 1 GB portions) are handed over as they arrive instead of being
// collected: no text block at all, *tsv stays NULL.
// user_dst (round 5): the text goes to user_cap bytes of the caller's -- page-locked (calitas_pin_host), so that every contig's rows
// cross the bus straight to their place: no bounce buffer, no memcpy into fresh pages (0.4-0.6 s per 22 GB of rows), *tsv = user_dst.
static int search_hits_sequential(calitas_ctx* ctx, const calitas_guide_t* guide, const std::string& guide_id, const calitas_params_t* params,
                                  const char* aligner_version, const char* time_stamp, char** tsv, uint64_t* tsv_bytes, uint64_t* n_rows,
                                  calitas_text_sink_t sink = nullptr, void* sink_user = nullptr, const HitsExtSource* ext_source = nullptr,
                                  char* user_dst = nullptr, uint64_t user_cap = 0) {
  const auto t_call = std::chrono::steady_clock::now();
  SearchPlan pl;
  int rc = plan_search(ctx, 1, guide, params, pl);
  if (rc) return rc;
  if (user_dst && sink) return fail(ctx, CALITAS_EINVAL, "a text sink and a destination buffer at once");
  const PackedRef& ref = ctx->ref;
  std::string version, stamp;
  calitas_default_version_and_stamp(aligner_version, time_stamp, version, stamp);
  const RowStrings rs = make_row_strings(ref, pl.gh[0], guide_id, pl.p, version, stamp);
  // Round 5: the per-contig texts cross PCIe compact (post.hpp) when the call builds one block and a caller's entries, if any, are
  // compact as well -- 21.8 GB of rows at BASELINE config 5's size were 0.42-0.47 s of the reference passes on the bus.  genome_build
  // stays in the rows (the variant branch's rows have one of their own); a text sink still gets whole rows as they come.
  std::string cut_head;
  bool compact = !sink && (!ext_source || ext_source->compact_rows);
  if (const char* e = TUNE_GET("CALITAS_COMPACT_ROWS")) compact = compact && std::atoi(e) != 0;
  const RowStrings rs_dev = compact ? compact_row_strings_keep_build(rs, &cut_head) : rs;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  rc = ensure_bin_base(ctx, pl, ctx->stream);
  if (rc) return rc;
  if (!ctx->copy_stream) {   // the fallback of the SDMA copy must not share ctx->stream with the helper thread's next pass (text_to_host)
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->copy_stream, hipStreamNonBlocking, least));
  }
  // The text grows in one pageable block (realloc: pages move, bytes are not copied); every contig's rows come over PCIe into a
  // reused page-locked bounce buffer and from there into the block on the worker pool (page-locking 40+ GB of pieces and
  // concatenating them afterwards took longer than the search).
  const size_t hlen = rs.header.size();
  if (user_dst && user_cap < hlen + 1) return fail(ctx, CALITAS_EINVAL, "the destination buffer does not hold the header line");
  // (a block of the library's: the one the last such search's caller handed back, if it is still parked -- its pages are there)
  char* text = sink ? nullptr : user_dst ? user_dst : (char*)calitas_out_take_big(hlen + (64u << 20));
  if (!sink && !text) text = (char*)calitas_out_grow(nullptr, 0, hlen + (64u << 20));
  if (!sink && !text) return fail(ctx, CALITAS_EINVAL, "out of memory");
  if (text) std::memcpy(text, rs.header.data(), hlen);
  else if (sink(rs.header.data(), hlen, sink_user) != 0) return fail(ctx, CALITAS_EIO, "the text sink reported an error");
  size_t total = hlen;
  char* bounce = nullptr;
  size_t bounce_cap = 0;
  auto drop = [&] { if (!user_dst) calitas_free(text); calitas_free(bounce); };
  calitas_timing_t tm{};
  uint64_t rows = 0;
  std::mutex copy_mu;
  const int n_contigs = (int)ref.contigs.size();
  uint64_t bases_done = 0;
  uint32_t n_passes = 0;
  double ms_rows = 0;              // inside lane_rows: kernels, their host round trips and every (re)allocation of scratch
  double ms_grow = 0, ms_land = 0; // this thread: the text block grown, the texts copied from the bounce buffer to their place
  // The passes: one plan per selected contig.
  std::vector<SearchPlan> passes;
  std::vector<int> pass_contig;
  {
    uint64_t win_lo = 0;
    for (int c = 0; c < n_contigs; c++) {
      const uint64_t win_n = window_count(ref.contigs[c].len, pl.step);
      if (pl.p.chrom_index < 0 || pl.p.chrom_index == c) {
        SearchPlan q = pl;
        q.tile_lo = (uint32_t)(ref.contigs[c].gbase / ref.tile);
        const uint32_t tile_hi = c + 1 < n_contigs ? (uint32_t)(ref.contigs[c + 1].gbase / ref.tile) : (uint32_t)ref.tiles.size();
        q.n_tiles = tile_hi - q.tile_lo;
        q.bases = ref.contigs[c].len; q.win_lo = win_lo; q.win_n = win_n;
        plan_bins(ctx, q, c, c + 1);
        // buffers sized from the estimate that sent this search here: no retry round per contig
        if (ctx->seq_recs_per_tile > 0 && ctx->seq_pams == pl.gd[0].n_pams && ctx->seq_L == pl.gd[0].L && pl.gd[0].min_guide_score == ctx->seq_min_score)
          q.rec_hint = (uint64_t)(ctx->seq_recs_per_tile * (double)q.n_tiles) + 1;
        passes.push_back(q); pass_contig.push_back(c);
      }
      win_lo += win_n;
    }
  }
  n_passes = (uint32_t)passes.size();
  g_pass_ms[0].store(0, std::memory_order_relaxed); g_pass_ms[1].store(0, std::memory_order_relaxed);
  // Two row-stage scratch sets (ctx->hits / hits_alt) take turns: a helper thread runs the device stages of pass i+1 while this thread
  // copies the text of pass i over PCIe and hands it on -- the copy is 1.5 of the 2.7 s of a PAM-less d = 8 search on an hg38-sized
  // genome, the device stages 1.0.  The sink is only ever called from this (the caller's) thread.
  struct Slot { LaneText lt; int rc = CALITAS_OK; double ms = 0; hipEvent_t rows_done = nullptr; int state = 0; };   // 0 free, 1 rows queued
  Slot slots[2];
  for (auto& sl : slots)
    if (hipEventCreateWithFlags(&sl.rows_done, hipEventDisableTiming) != hipSuccess) {
      for (auto& s2 : slots) if (s2.rows_done) (void)hipEventDestroy(s2.rows_done);
      drop();
      return fail(ctx, CALITAS_EHIP, "hipEventCreateWithFlags failed");
    }
  std::mutex mu;
  std::condition_variable cv;
  bool abort_passes = false;
  std::thread producer([&] {
    (void)hipSetDevice(ctx->device);
    HitsWork* work[2] = {ctx->hits, ctx->hits_alt};
    uint64_t serial[2] = {ctx->hits_names_serial, ctx->hits_alt_names_serial};
    for (size_t i = 0; i < passes.size(); i++) {
      Slot& sl = slots[i & 1];
      {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return sl.state == 0 || abort_passes; });
        if (abort_passes) break;
      }
      ctx->hits = work[i & 1]; ctx->hits_names_serial = serial[i & 1];
      sl.lt = LaneText();
      const auto t_rows = std::chrono::steady_clock::now();
      passes[i].general_tail = ext_source != nullptr;
      try {
        sl.rc = lane_rows(ctx, passes[i], false, rs_dev, guide_id, version, stamp, sl.lt, false, nullptr, ext_source, pass_contig[i]);
      } catch (const std::exception& e) {                      // (this thread has no caller to unwind to)
        sl.rc = fail(ctx, CALITAS_EHIP, std::string("a contig pass ended with an exception: ") + e.what());
      }
      if (sl.rc == CALITAS_OK && hipEventRecord(sl.rows_done, ctx->stream) != hipSuccess) sl.rc = fail(ctx, CALITAS_EHIP, "hipEventRecord failed");
      sl.ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_rows).count();
      work[i & 1] = ctx->hits; serial[i & 1] = ctx->hits_names_serial;
      {
        std::lock_guard<std::mutex> lk(mu);
        sl.state = 1;
      }
      cv.notify_all();
      if (sl.rc) break;
    }
    ctx->hits = work[0]; ctx->hits_names_serial = serial[0];
    ctx->hits_alt = work[1]; ctx->hits_alt_names_serial = serial[1];
  });
  auto stop_producer = [&] {
    { std::lock_guard<std::mutex> lk(mu); abort_passes = true; }
    cv.notify_all();
    producer.join();
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& sl : slots) (void)hipEventDestroy(sl.rows_done);
  };
  rc = CALITAS_OK;
  for (size_t i = 0; i < passes.size() && !rc; i++) {
    Slot& sl = slots[i & 1];
    {
      std::unique_lock<std::mutex> lk(mu);
      cv.wait(lk, [&] { return sl.state == 1; });
    }
    if (sl.rc) { rc = sl.rc; break; }
    LaneText& lt = sl.lt;
    const int c = pass_contig[i];
    const bool trace_contigs = TUNE_GET("CALITAS_TRACE") && std::atoi(TUNE_GET("CALITAS_TRACE")) >= 3;
    const double ms_rows_at = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count();
    struct Said { bool on; int c; double at; uint64_t bytes; std::chrono::steady_clock::time_point t0;
                  ~Said() { if (on) std::fprintf(stderr, "[calitas] search_hits: contig %d: rows queued at %.1f ms, %llu bytes on the host at %.1f ms\n", c, at, (unsigned long long)bytes,
                                                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); } } said{trace_contigs, c, ms_rows_at, lt.bytes, t_call};
    ms_rows += sl.ms;
    bases_done += ref.contigs[c].len;
    if (lt.bytes) {
      // room for this contig, and -- extrapolating from the bases done so far -- for the rest
      const bool expand = compact && !lt.on_host;             // (rows the host stages built are whole already)
      const size_t full_bytes = expand ? (size_t)lt.bytes + (size_t)lt.rows * (cut_head.size() + rs.tail.size() - 1) : (size_t)lt.bytes;
      const double per_base = (double)(total - hlen + full_bytes) / (double)std::max<uint64_t>(1, bases_done);
      const size_t guess = pl.p.chrom_index >= 0 ? 0 : (size_t)(per_base * 1.05 * (double)(ref.total_bases - bases_done));
      if (user_dst) {
        if ((uint64_t)total + full_bytes + 1 > user_cap) {
          rc = fail(ctx, CALITAS_EINVAL, "the destination buffer is too small for the text (" + std::to_string(user_cap) + " bytes; " +
                                         std::to_string(total + full_bytes + 1) + " needed after " + std::to_string(i + 1) + " of " + std::to_string(passes.size()) + " contigs)");
          break;
        }
      } else if (!sink) {
        const auto t_grow = std::chrono::steady_clock::now();
        char* grown = (char*)calitas_out_grow(text, total, total + full_bytes + 1 + guess);
        if (!grown) { rc = fail(ctx, CALITAS_EINVAL, "out of memory"); break; }
        text = grown;
        ms_grow += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_grow).count();
      }
      if (expand) {
        // the contig's compact text over the bus in pieces into the page-locked block, the worker pool puts guide_id, protospacer and
        // the tail back while the rest is still on its way (compact_rows_to_host); the next contig's kernels run meanwhile
        const size_t n = (size_t)lt.bytes;
        if (n > bounce_cap) { calitas_free(bounce); bounce = (char*)calitas_out_alloc_pinned(n); bounce_cap = bounce ? n : 0; }
        if (!bounce) { rc = fail(ctx, CALITAS_EINVAL, "out of memory"); break; }
        size_t wrote = 0;
        rc = compact_rows_to_host(ctx, ctx, lt, n, bounce, cut_head, rs.tail, text + total, &copy_mu, &wrote, sl.rows_done);
        if (rc) break;
        if (wrote != full_bytes) { rc = fail(ctx, CALITAS_EHIP, "a contig's compact rows do not expand to the row count the device reported (internal error)"); break; }
        if ((rc = rows_late_check(ctx, lt)) != CALITAS_OK) break;
        total += full_bytes;
        rows += lt.rows;
        tm.scan_kernel_ms += lt.tm.scan_kernel_ms; tm.align_kernel_ms += lt.tm.align_kernel_ms; tm.gpu_total_ms += lt.tm.gpu_total_ms;
        tm.host_post_ms += lt.tm.host_post_ms; tm.bases_scanned += lt.tm.bases_scanned; tm.packed_bytes += lt.tm.packed_bytes;
        tm.scan_records += lt.tm.scan_records; tm.candidate_columns += lt.tm.candidate_columns; tm.raw_alignments += lt.tm.raw_alignments;
        tm.accepted_alignments += lt.tm.accepted_alignments; tm.retries += lt.tm.retries; tm.hits_copy_ms += lt.tm.hits_copy_ms;
        tm.binned_lanes += lt.tm.binned_lanes; tm.owned_general_lanes += lt.tm.owned_general_lanes;
        { std::lock_guard<std::mutex> lk(mu); sl.state = 0; }
        cv.notify_all();
        continue;
      }
      if (lt.on_host) {
        if (!sink) std::memcpy(text + total, lt.host_rows.data(), (size_t)lt.bytes);
        else if (sink(lt.host_rows.data(), lt.bytes, sink_user) != 0) { rc = fail(ctx, CALITAS_EIO, "the text sink reported an error"); break; }
      } else if (user_dst) {                            // straight to its place in the caller's page-locked buffer
        // (in portions: one copy of a whole contig's rows -- 1.7 GB for chr1 of BASELINE config 5's shape -- held up everybody else's
        // copies for 20-65 ms at a time: the variant half's aligner batches took 15 instead of 3 ms)
        const size_t kPortion = 128u << 20;
        for (size_t off = 0; off < (size_t)lt.bytes && !rc; off += kPortion) {
          double ms = 0;
          rc = text_to_host(ctx, ctx, text + total + off, lt.d_text + off, std::min(kPortion, (size_t)lt.bytes - off), &copy_mu, &ms, sl.rows_done);
          lt.tm.hits_copy_ms += ms;
        }
        if (rc) break;
        if ((rc = rows_late_check(ctx, lt)) != CALITAS_OK) break;
      } else {
        const size_t kPiece = 1ull << 30;             // bounce buffer: at most 1 GB page-locked
        for (size_t off = 0; off < (size_t)lt.bytes && !rc; off += kPiece) {
          const size_t n = std::min(kPiece, (size_t)lt.bytes - off);
          if (n > bounce_cap) { calitas_free(bounce); bounce = (char*)calitas_out_alloc_pinned(n); bounce_cap = bounce ? n : 0; }
          if (!bounce) { rc = fail(ctx, CALITAS_EINVAL, "out of memory"); break; }
          double ms = 0;
          dma_open_once(ctx);
          if (!sink && ctx->dma.usable() && n >= (64u << 20)) {
            // The portion crosses the bus in 32 MB pieces queued back to back on the DMA engine, and every piece goes from the bounce
            // buffer to its place while the ones behind it are still on their way (round 5: one copy, then one memcpy of the whole
            // portion, was 0.4 s on the bus + 0.4-0.5 s of memcpy into fresh pages, one after the other, per 22 GB of rows -- the
            // longest chain of a search with variants at BASELINE config 5's size).
            if (hipError_t e = calitas_spin_sync(sl.rows_done); e != hipSuccess) { rc = fail(ctx, CALITAS_EHIP, std::string("waiting for a contig's rows: ") + hipGetErrorString(e)); break; }
            const size_t piece = 32u << 20;
            std::vector<unsigned long long> tickets;
            const auto t_dma = std::chrono::steady_clock::now();
            if (ctx->dma.start_pieces(bounce, lt.d_text + off, n, piece, tickets)) {
              char* dst = text + total + off;
              bool ok = true;
              double ms_wait_dma = 0;
              for (size_t k = 0; k < tickets.size(); k++) {
                const auto t_w = std::chrono::steady_clock::now();
                if (!ctx->dma.finish(tickets[k])) ok = false;     // (every ticket is waited for: nothing may land in a freed block)
                ms_wait_dma += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_w).count();
                if (!ok) continue;
                const size_t b0 = k * piece, nb = std::min(piece, n - b0);
                const auto t_land = std::chrono::steady_clock::now();
                std::lock_guard<std::mutex> host_lock(ctx->host_mu);   // the helper thread's host stages (if any) use the same pool
                ctx->pool->for_blocks(nb, [&](size_t b, size_t e, int) { stream_copy(dst + b0 + b, bounce + b0 + b, e - b); });
                ms_land += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_land).count();
              }
              if (!ok) { rc = fail(ctx, CALITAS_EHIP, "SDMA copy failed"); break; }
              (void)t_dma;
              lt.tm.hits_copy_ms += ms_wait_dma;                  // (what this thread waited for the bus; the rest of the copy hid behind the memcpy)
              continue;
            }
            if (TUNE_GET("CALITAS_TRACE")) std::fprintf(stderr, "[calitas] SDMA copy declined (%s), using one copy per portion\n", DmaCopier::last_reason());
          }
          rc = text_to_host(ctx, ctx, bounce, lt.d_text + off, n, &copy_mu, &ms, sl.rows_done);
          if (rc) break;
          lt.tm.hits_copy_ms += ms;
          if (sink) {
            if (sink(bounce, n, sink_user) != 0) rc = fail(ctx, CALITAS_EIO, "the text sink reported an error");
            continue;
          }
          char* dst = text + total + off;
          const char* src = bounce;
          const auto t_land = std::chrono::steady_clock::now();
          std::lock_guard<std::mutex> host_lock(ctx->host_mu);      // the helper thread's host stages (if any) use the same pool
          ctx->pool->for_blocks(n, [&](size_t b, size_t e, int) { std::memcpy(dst + b, src + b, e - b); });
          ms_land += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_land).count();
        }
        if (rc) break;
        if ((rc = rows_late_check(ctx, lt)) != CALITAS_OK) break;
      }
      if (lt.ext_place) {                                 // the caller's own rows into the holes the rows kernel left for them
        if (sink || !lt.ext || !lt.ext->fill) { rc = fail(ctx, CALITAS_EINVAL, "rows left to the caller, but nowhere to write them (internal error)"); break; }
        if (lt.ext->fill(lt.ext_place, text + total, user_dst != nullptr) != 0) { rc = fail(ctx, CALITAS_EINVAL, "the caller's rows could not be written into the text"); break; }
      }
      total += (size_t)lt.bytes;
    }
    rows += lt.rows;
    tm.scan_kernel_ms += lt.tm.scan_kernel_ms; tm.align_kernel_ms += lt.tm.align_kernel_ms; tm.gpu_total_ms += lt.tm.gpu_total_ms;
    tm.host_post_ms += lt.tm.host_post_ms; tm.bases_scanned += lt.tm.bases_scanned; tm.packed_bytes += lt.tm.packed_bytes;
    tm.scan_records += lt.tm.scan_records; tm.candidate_columns += lt.tm.candidate_columns; tm.raw_alignments += lt.tm.raw_alignments;
    tm.accepted_alignments += lt.tm.accepted_alignments; tm.retries += lt.tm.retries; tm.hits_copy_ms += lt.tm.hits_copy_ms;
    tm.binned_lanes += lt.tm.binned_lanes; tm.owned_general_lanes += lt.tm.owned_general_lanes;
    {
      std::lock_guard<std::mutex> lk(mu);
      sl.state = 0;
    }
    cv.notify_all();
  }
  stop_producer();
  if (rc) { drop(); return rc; }
  calitas_free(bounce);
  if (user_dst) {
    if ((uint64_t)total + 1 > user_cap) return fail(ctx, CALITAS_EINVAL, "the destination buffer is too small for the text");
    text[total] = 0;
  } else if (!sink) {
    char* grown = (char*)calitas_out_grow(text, total, total + 1);
    if (!grown) { calitas_free(text); return fail(ctx, CALITAS_EINVAL, "out of memory"); }
    text = (char*)calitas_out_shrink(grown, total + 1);           // (a parked block taken for a much smaller text)
    text[total] = 0;
  }
  tm.hit_rows = rows; tm.hits_bytes = total; tm.lanes = 1; tm.contig_passes = n_passes;
  ctx->timing = tm;
  ctx->last_text_bytes = total;
  if (TUNE_GET("CALITAS_TRACE"))
    std::fprintf(stderr, "[calitas] search_hits: one pass per contig (%d), scan %.3f ms, align %.3f ms, all device stages incl. allocation %.3f ms (with a caller's hits: %.3f ms up to the row stage, %.3f ms waiting for the hits), text copy %.3f ms + %.3f ms from the bounce buffer to its place + %.3f ms growing the block (sums), call %.3f ms (%llu rows, %zu bytes)\n",
                 n_contigs, tm.scan_kernel_ms, tm.align_kernel_ms, ms_rows, g_pass_ms[0].load(std::memory_order_relaxed), g_pass_ms[1].load(std::memory_order_relaxed), tm.hits_copy_ms, ms_land, ms_grow,
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count(), (unsigned long long)rows, total);
  *tsv = text;
  if (tsv_bytes) *tsv_bytes = total;
  if (n_rows) *n_rows = rows;
  return CALITAS_OK;
}

// owned: {first window, windows} of a window range (calitas_params_t::first_window / n_windows) whose rows the call returns, all of it
// on the per-bin kernels; *owned_declined: they could not decide it (the caller then takes the slow path), nothing is returned.
static int search_hits_attempt(calitas_ctx* ctx, const calitas_guide_t* guide, const std::string& guide_id, const calitas_params_t* params,
                               const char* aligner_version, const char* time_stamp, char** tsv, uint64_t* tsv_bytes, uint64_t* n_rows,
                               char* user_dst = nullptr, uint64_t user_cap = 0, const uint64_t* owned = nullptr, bool* owned_declined = nullptr);

// Whether this search is known not to fit one pass: forced (CALITAS_SEQUENTIAL, tests), or at least as permissive as the last one on
// this context that did not.  remember = true records the search as such.
static bool known_not_to_fit(calitas_ctx* ctx, const calitas_guide_t* guide, const calitas_params_t* params, bool remember) {
  if (!remember && TUNE_GET("CALITAS_SEQUENTIAL")) return true;
  SearchPlan pl;
  if (!guide || !params || plan_search(ctx, 1, guide, params, pl) != CALITAS_OK) return false;
  const GuideDev& g = pl.gd[0];
  if (remember) { ctx->seq_L = g.L; ctx->seq_pams = g.n_pams; ctx->seq_min_score = g.min_guide_score; ctx->seq_recs_per_tile = 0; return true; }
  return params->chrom_index < 0 && ctx->seq_pams == g.n_pams && ctx->seq_L == g.L && g.min_guide_score <= ctx->seq_min_score;
}

// Scan records per live tile this search produces, from a scan of every k-th tile with a record capacity of 0 (counted, not kept):
// a few hundred tiles, tens of microseconds.
static int estimate_scan_records(calitas_ctx* ctx, const SearchPlan& pl, double* recs_per_tile, uint64_t* live_tiles) {
  const PackedRef& ref = ctx->ref;
  const uint32_t stride = std::max<uint32_t>(1, pl.n_tiles / 512);
  const uint32_t n_sample = (pl.n_tiles + stride - 1) / stride;
  auto live = [&](uint32_t t) {
    const TileInfo& ti = ref.tiles[t];
    return ti.flag != 2u && ti.contig != 0xFFFFFFFFu && (pl.p.chrom_index < 0 || ti.contig == (uint32_t)pl.p.chrom_index);
  };
  uint64_t live_all = 0, live_sample = 0;
  for (uint32_t t = 0; t < pl.n_tiles; t++) if (live(pl.tile_lo + t)) { live_all++; if (t % stride == 0) live_sample++; }
  *live_tiles = live_all; *recs_per_tile = 0;
  if (live_sample == 0) return CALITAS_OK;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  std::memcpy(ctx->h_guides, pl.gd.data(), sizeof(GuideDev) * pl.n_guides);
  HIP_TRY(ctx, hipMemcpyAsync(ctx->d_guides, ctx->h_guides, sizeof(GuideDev) * pl.n_guides, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemsetAsync(ctx->d_counters, 0, 8 * sizeof(uint32_t), ctx->stream));
  ScanArgs sa; AlignArgs aa;
  fill_kernel_args(ctx, pl, sa, aa);
  sa.rec_capacity = 0; sa.tile_stride = stride;
  HIP_TRY(ctx, launch_scan_rows(sa, ref.chunk, pl.warm_words, n_sample, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(ctx->h_counters, ctx->d_counters, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  *recs_per_tile = (double)ctx->h_counters[0] / (double)live_sample;
  return CALITAS_OK;
}

// Whether one pass of this search over the whole reference would overrun the device (or CALITAS_DEVICE_BUDGET_MB): decided from the
// worst case when that is harmless, from the memory of the last search that fit, and otherwise from a sampled record count -- not
// from a failed allocation of hundreds of gigabytes.  true: the search is remembered as one for per-contig passes.
static bool predicted_not_to_fit(calitas_ctx* ctx, const calitas_guide_t* guide, const calitas_params_t* params) {
  SearchPlan pl;
  if (!guide || !params || plan_search(ctx, 1, guide, params, pl) != CALITAS_OK) return false;   // the attempt reports the error
  const GuideDev& g = pl.gd[0];
  if (ctx->fit_pams == g.n_pams && ctx->fit_L == g.L && g.min_guide_score >= ctx->fit_min_score) return false;
  uint64_t limit = 0;
  if (const char* e = TUNE_GET("CALITAS_DEVICE_BUDGET_MB")) limit = (uint64_t)std::atoll(e) << 20;
  else {
    size_t mem_free = 0, mem_total = 0;
    if (hipSetDevice(ctx->device) != hipSuccess || hipMemGetInfo(&mem_free, &mem_total) != hipSuccess) return false;
    limit = (uint64_t)mem_total * 7 / 10;
  }
  // per scan record: its strips and itself, with the growth margin of the buffers; per raw alignment (about 1.5 per record): the
  // record, the filter's and the row stage's scratch and its share of the text (~110 + ~1100 bytes)
  const uint64_t per_rec = (pl.slab_per_rec + sizeof(ScanRecord)) * 5 / 4 + (sizeof(RawAln) + 110 + 1100) * 3 / 2;
  const uint64_t worst = (pl.bases / 16 + 1) * 2;
  if (worst <= limit / per_rec) return false;
  double per_tile = 0;
  uint64_t live = 0;
  if (estimate_scan_records(ctx, pl, &per_tile, &live) != CALITAS_OK) return false;
  const double n_rec = per_tile * (double)live;
  if (TUNE_GET("CALITAS_TRACE"))
    std::fprintf(stderr, "[calitas] search_hits: about %.3g scan records expected (%.1f per tile), %.1f GB of scratch for one pass, limit %.1f GB\n",
                 n_rec, per_tile, n_rec * (double)per_rec / 1e9, (double)limit / 1e9);
  if (n_rec * (double)per_rec <= (double)limit) return false;
  ctx->seq_L = g.L; ctx->seq_pams = g.n_pams; ctx->seq_min_score = g.min_guide_score; ctx->seq_recs_per_tile = per_tile;
  return true;
}

int calitas_search_hits_impl(calitas_ctx* ctx, const calitas_guide_t* guide, const std::string& guide_id, const calitas_params_t* params,
                            const char* aligner_version, const char* time_stamp, char** tsv, uint64_t* tsv_bytes, uint64_t* n_rows) {
  g_marks.start();
  struct Dump { ~Dump() { g_marks.mark("return"); g_marks.dump(); } } dump_at_exit;
  int rc = CALITAS_ENOMEM;
  if (params && (params->first_window != 0 || params->n_windows != 0))     // a process's stretch of a multi-GPU job: one pass, no per-contig mode
    return search_hits_attempt(ctx, guide, guide_id, params, aligner_version, time_stamp, tsv, tsv_bytes, n_rows);
  if (!known_not_to_fit(ctx, guide, params, false) && !predicted_not_to_fit(ctx, guide, params)) {
    rc = search_hits_attempt(ctx, guide, guide_id, params, aligner_version, time_stamp, tsv, tsv_bytes, n_rows);
    if (rc != CALITAS_ENOMEM) return rc;
    if (TUNE_GET("CALITAS_TRACE")) std::fprintf(stderr, "[calitas] search_hits: %s -- retrying with one pass per contig\n", ctx->err.c_str());
    release_scratch(ctx);
    (void)known_not_to_fit(ctx, guide, params, true);
  }
  *tsv = nullptr;
  rc = search_hits_sequential(ctx, guide, guide_id, params, aligner_version, time_stamp, tsv, tsv_bytes, n_rows);
  if (rc == CALITAS_ENOMEM) release_scratch(ctx);   // leave the context usable for smaller searches
  return rc;
}

// calitas_search_hits with hits of the caller's own brought into every contig's row stage (the variant branch, variants.cpp): one pass per
// contig on the general kernels.  kExtDeclined is returned as CALITAS_ESTATE + *declined: a stage left the device path, the caller
// merges on the host instead.
int calitas_search_hits_ext_impl(calitas_ctx* ctx, const calitas_guide_t* guide, const std::string& guide_id, const calitas_params_t* params,
                                 const char* aligner_version, const char* time_stamp, const HitsExtSource& source, char** tsv,
                                 uint64_t* tsv_bytes, uint64_t* n_rows, bool* declined, char* user_dst, uint64_t user_cap) {
  *declined = false;
  *tsv = nullptr;
  if (!known_not_to_fit(ctx, guide, params, false)) (void)predicted_not_to_fit(ctx, guide, params);   // (sizes the passes' buffers when the search is a dense one)
  int rc = search_hits_sequential(ctx, guide, guide_id, params, aligner_version, time_stamp, tsv, tsv_bytes, n_rows, nullptr, nullptr, &source, user_dst, user_cap);
  if (rc == kExtDeclined) { *declined = true; *tsv = nullptr; return CALITAS_ESTATE; }
  if (rc == CALITAS_ENOMEM) release_scratch(ctx);
  return rc;
}

// calitas_search_hits_into: one pass (with lanes), text straight into the caller's buffer -- or, since round 5, one pass per contig when
// the search does not fit the device (a PAM-less search at eight differences on a whole genome: tens of gigabytes of text): every
// contig's rows then cross the bus straight to their place in the buffer (search_hits_sequential's user_dst).  A window range
// (first_window / n_windows) stays one pass.
int calitas_search_hits_into_impl(calitas_ctx* ctx, const calitas_guide_t* guide, const std::string& guide_id, const calitas_params_t* params,
                                  const char* aligner_version, const char* time_stamp, char* dst, uint64_t dst_capacity, uint64_t* tsv_bytes,
                                  uint64_t* n_rows) {
  if (!dst || dst_capacity < 2) return fail(ctx, CALITAS_EINVAL, "no destination buffer");
  char* text = nullptr;
  if (params && (params->first_window != 0 || params->n_windows != 0))
    return search_hits_attempt(ctx, guide, guide_id, params, aligner_version, time_stamp, &text, tsv_bytes, n_rows, dst, dst_capacity);
  int rc = CALITAS_ENOMEM;
  if (!known_not_to_fit(ctx, guide, params, false) && !predicted_not_to_fit(ctx, guide, params)) {
    rc = search_hits_attempt(ctx, guide, guide_id, params, aligner_version, time_stamp, &text, tsv_bytes, n_rows, dst, dst_capacity);
    if (rc != CALITAS_ENOMEM) return rc;
    if (TUNE_GET("CALITAS_TRACE")) std::fprintf(stderr, "[calitas] search_hits_into: %s -- retrying with one pass per contig\n", ctx->err.c_str());
    release_scratch(ctx);
    (void)known_not_to_fit(ctx, guide, params, true);
  }
  text = nullptr;
  rc = search_hits_sequential(ctx, guide, guide_id, params, aligner_version, time_stamp, &text, tsv_bytes, n_rows, nullptr, nullptr, nullptr, dst, dst_capacity);
  if (rc == CALITAS_ENOMEM) release_scratch(ctx);
  return rc;
}

// calitas_search_hits_stream: the text goes to `sink` -- in one piece when the search fits one call, header and per-contig pieces
// otherwise (no block of the size of the whole text is ever allocated then).
int calitas_search_hits_stream_impl(calitas_ctx* ctx, const calitas_guide_t* guide, const std::string& guide_id, const calitas_params_t* params,
                                    const char* aligner_version, const char* time_stamp, calitas_text_sink_t sink, void* user,
                                    uint64_t* tsv_bytes, uint64_t* n_rows) {
  char* text = nullptr;
  uint64_t bytes = 0, rows = 0;
  int rc = CALITAS_ENOMEM;
  if (!known_not_to_fit(ctx, guide, params, false) && !predicted_not_to_fit(ctx, guide, params)) {
    rc = search_hits_attempt(ctx, guide, guide_id, params, aligner_version, time_stamp, &text, &bytes, &rows);
    if (rc == CALITAS_OK) {
      const int s = sink(text, bytes, user);
      calitas_free(text);
      if (s != 0) return fail(ctx, CALITAS_EIO, "the text sink reported an error");
      if (tsv_bytes) *tsv_bytes = bytes;
      if (n_rows) *n_rows = rows;
      return CALITAS_OK;
    }
    if (rc != CALITAS_ENOMEM) return rc;
    if (TUNE_GET("CALITAS_TRACE")) std::fprintf(stderr, "[calitas] search_hits: %s -- retrying with one pass per contig\n", ctx->err.c_str());
    release_scratch(ctx);
    (void)known_not_to_fit(ctx, guide, params, true);
  }
  rc = search_hits_sequential(ctx, guide, guide_id, params, aligner_version, time_stamp, &text, tsv_bytes, n_rows, sink, user);
  if (rc == CALITAS_ENOMEM) release_scratch(ctx);
  return rc;
}

// ---- calitas_search_hits on a window range ------------------------------------------------------------------------------
// A process of a multi-GPU job owns a stretch of the genome: the rows whose coordinate_start lies at or behind the start of window
// first_window and before the start of window first_window + n_windows (windowIterator's sequence over the whole reference,
// SearchReference.scala:39-71).  coordinate_start is the first key of ReferenceHit.sort, so the stretches of consecutive ranges are
// consecutive pieces of hits.txt, wherever the cuts fall -- inside a contig, inside a repeat.  The per-bin kernels decide a bin's
// hits from the bin and the edges of its neighbours (binned.hip), so the call aligns the windows the stretch's bins (plus one on
// either side) reach and keeps the rows of the stretch; nothing is exchanged between the processes.  When a bin declines (crowded,
// a chain of hits longer than the halo) the contigs the stretch touches are searched whole on the general kernels and their rows
// filtered by position on the host: slower, same rows.

// The plan of a stretch: bins, the windows their contexts reach, the tiles those windows lie in.
static bool plan_owned_range(const calitas_ctx* ctx, SearchPlan& pl, uint64_t first, uint64_t count) {
  const PackedRef& ref = ctx->ref;
  const int nc = (int)ref.contigs.size();
  if (!pl.bin_shift || ctx->bin_base.size() != (size_t)nc + 1) return false;
  std::vector<uint64_t> wb((size_t)nc + 1, 0);
  for (int c = 0; c < nc; c++) wb[c + 1] = wb[c] + window_count(ref.contigs[c].len, pl.step);
  const uint64_t total = wb[nc];
  if (count == 0 || first + count > total) return false;
  auto locate = [&](uint64_t w, int& c, uint64_t& pos) {       // start of global window w; w == total: the end of the reference
    if (w >= total) { c = nc; pos = 0; return; }
    c = (int)(std::upper_bound(wb.begin(), wb.end(), w) - wb.begin()) - 1;
    pos = (w - wb[c]) * (uint64_t)pl.step;
  };
  int c_lo = 0, c_hi = 0;
  uint64_t p_lo = 0, p_hi = 0;
  locate(first, c_lo, p_lo);
  locate(first + count, c_hi, p_hi);
  // a stretch that starts with the first window of a contig owns the contig from base 0, one that ends at a contig's first window
  // owns the contig before it to its end -- (c, 0) keys say exactly that
  pl.owned = true;
  pl.own_lo = ((uint64_t)c_lo << 32) | p_lo;
  pl.own_hi = ((uint64_t)c_hi << 32) | p_hi;
  // last owned position
  int c_last = c_hi;
  uint64_t p_last = p_hi;
  if (p_hi == 0) { c_last = c_hi - 1; while (c_last > c_lo && ref.contigs[c_last].len == 0) c_last--; p_last = ref.contigs[c_last].len; }
  if (p_last > 0) p_last--;
  const uint64_t bin = 1ull << pl.bin_shift;
  uint32_t b_lo = ctx->bin_base[c_lo] + (uint32_t)(p_lo >> pl.bin_shift), b_hi = ctx->bin_base[c_last] + (uint32_t)(p_last >> pl.bin_shift);
  if (b_lo > ctx->bin_base[c_lo]) b_lo--;                      // one bin of context on either side, inside the contig
  if (b_hi + 1 < ctx->bin_base[c_last + 1]) b_hi++;
  pl.bin_first = b_lo; pl.n_bins = b_hi - b_lo + 1;
  // the windows that start in those bins: the context of the first owned bin (two windows to the left) lies in the bin before it, that
  // of the last one (the longest hit to the right) in the bin behind it (binned.hip) -- and trace_kernel lists an alignment in the
  // bin its window starts in, which must be one of the lane's
  const int64_t ctx_lo = (int64_t)((uint64_t)(b_lo - ctx->bin_base[c_lo]) << pl.bin_shift);
  const uint64_t ctx_hi = std::min<uint64_t>(ref.contigs[c_last].len, ((uint64_t)(b_hi - ctx->bin_base[c_last]) + 1) << pl.bin_shift);
  const uint64_t k_lo = ((uint64_t)ctx_lo + (uint64_t)pl.step - 1) / (uint64_t)pl.step;
  const uint64_t nw_last = wb[c_last + 1] - wb[c_last];
  const uint64_t k_hi = ctx_hi == 0 ? 0 : std::min<uint64_t>(nw_last, (ctx_hi - 1) / (uint64_t)pl.step + 1);
  pl.gw_lo = std::min(wb[c_lo] + k_lo, wb[c_lo + 1]);
  pl.gw_hi = wb[c_last] + k_hi;
  if (pl.gw_hi < pl.gw_lo) pl.gw_hi = pl.gw_lo;
  // tiles those windows lie in
  const uint64_t g_lo = ref.contigs[c_lo].gbase + (uint64_t)ctx_lo;
  const uint64_t g_hi = ref.contigs[c_last].gbase + std::min<uint64_t>(ref.contigs[c_last].len, ctx_hi + (uint64_t)pl.p.window_size);
  pl.tile_lo = (uint32_t)(g_lo / ref.tile);
  pl.n_tiles = (uint32_t)((g_hi + ref.tile - 1) / ref.tile) - pl.tile_lo;
  uint64_t bases = 0;
  for (int c = c_lo; c <= c_last; c++) bases += ref.contigs[c].len;
  if (c_lo == c_last) bases = std::min<uint64_t>(ref.contigs[c_lo].len, ctx_hi + (uint64_t)pl.p.window_size) - (uint64_t)ctx_lo;
  pl.bases = bases;
  pl.win_lo = pl.gw_lo; pl.win_n = pl.gw_hi - pl.gw_lo;
  (void)bin;
  return true;
}


static int search_hits_owned(calitas_ctx* ctx, const calitas_guide_t* guide, const std::string& guide_id, const calitas_params_t* params,
                             const char* aligner_version, const char* time_stamp, char** tsv, uint64_t* tsv_bytes, uint64_t* n_rows,
                             char* user_dst, uint64_t user_cap) {
  *tsv = nullptr;
  if (tsv_bytes) *tsv_bytes = 0;
  if (n_rows) *n_rows = 0;
  calitas_params_t whole = *params;
  whole.first_window = 0; whole.n_windows = 0;
  if (whole.chrom_index >= 0) return fail(ctx, CALITAS_EINVAL, "a window range and chrom_index exclude each other");
  SearchPlan pl;
  int rc = plan_search(ctx, 1, guide, &whole, pl);
  if (rc) return rc;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  rc = ensure_bin_base(ctx, pl, ctx->stream);
  if (rc) return rc;
  const PackedRef& ref = ctx->ref;
  if (params->first_window < 0 || params->n_windows <= 0 || (uint64_t)params->first_window + (uint64_t)params->n_windows > pl.win_n)
    return fail(ctx, CALITAS_EINVAL, "first_window / n_windows outside the window table (" + std::to_string(pl.win_n) + " windows)");
  std::string version, stamp;
  calitas_default_version_and_stamp(aligner_version, time_stamp, version, stamp);
  const RowStrings rs = make_row_strings(ref, pl.gh[0], guide_id, pl.p, version, stamp);
  const size_t hlen = rs.header.size();
  {
    // the stretch on the per-bin kernels, cut into lanes like any other call (search_hits_attempt): everything it returns is final
    const uint64_t range[2] = {(uint64_t)params->first_window, (uint64_t)params->n_windows};
    bool declined = false;
    rc = search_hits_attempt(ctx, guide, guide_id, &whole, aligner_version, time_stamp, tsv, tsv_bytes, n_rows, user_dst, user_cap, range, &declined);
    if (rc || !declined) return rc;
    HIP_TRY(ctx, calitas_spin_sync(ctx->stream));
  }
  // ---- the contigs the stretch touches, whole, on the general kernels; their rows filtered by position ----
  if (TUNE_GET("CALITAS_TRACE")) std::fprintf(stderr, "[calitas] search_hits on a window range: the bins declined, searching the touched contigs whole\n");
  std::vector<uint64_t> wb(ref.contigs.size() + 1, 0);
  for (size_t c = 0; c < ref.contigs.size(); c++) wb[c + 1] = wb[c] + window_count(ref.contigs[c].len, pl.step);
  const uint64_t first = (uint64_t)params->first_window, last = first + (uint64_t)params->n_windows;
  auto key_of = [&](uint64_t w) {
    if (w >= wb.back()) return (uint64_t)ref.contigs.size() << 32;
    const size_t c = (size_t)(std::upper_bound(wb.begin(), wb.end(), w) - wb.begin()) - 1;
    return ((uint64_t)c << 32) | ((w - wb[c]) * (uint64_t)pl.step);
  };
  const uint64_t own_lo = key_of(first), own_hi = key_of(last);
  std::string body;
  uint64_t rows = 0;
  calitas_timing_t tm{};
  for (size_t c = 0; c < ref.contigs.size(); c++) {
    if (wb[c + 1] <= first || wb[c] >= last || wb[c + 1] == wb[c]) continue;
    calitas_params_t pc = whole;
    pc.chrom_index = (int32_t)c;
    char* t = nullptr;
    uint64_t tb = 0, tr = 0;
    rc = search_hits_attempt(ctx, guide, guide_id, &pc, version.c_str(), stamp.c_str(), &t, &tb, &tr, nullptr, 0);
    if (rc) return rc;
    // rows: chromosome is column 4, coordinate_start column 5 (RH:99-132); the contig is c, so only the position decides
    const char* q = t + hlen;
    const char* end = t + tb;
    while (q < end) {
      const char* nl = (const char*)std::memchr(q, '\n', (size_t)(end - q));
      const char* row_end = nl ? nl + 1 : end;
      const char* f = q;
      for (int k = 0; k < 4 && f < row_end; k++) { const char* tab = (const char*)std::memchr(f, '\t', (size_t)(row_end - f)); f = tab ? tab + 1 : row_end; }
      const uint64_t pos = std::strtoull(f, nullptr, 10);
      const uint64_t key = ((uint64_t)c << 32) | pos;
      if (key >= own_lo && key < own_hi) { body.append(q, (size_t)(row_end - q)); rows++; }
      q = row_end;
    }
    calitas_free(t);
    tm.scan_kernel_ms += ctx->timing.scan_kernel_ms; tm.align_kernel_ms += ctx->timing.align_kernel_ms; tm.gpu_total_ms += ctx->timing.gpu_total_ms;
    tm.bases_scanned += ctx->timing.bases_scanned; tm.packed_bytes += ctx->timing.packed_bytes; tm.scan_records += ctx->timing.scan_records;
    tm.raw_alignments += ctx->timing.raw_alignments; tm.accepted_alignments += ctx->timing.accepted_alignments;
  }
  const size_t total = hlen + body.size();
  if (user_dst && user_cap < total + 1) return fail(ctx, CALITAS_EINVAL, "the caller's buffer is too small for the text");
  char* text = user_dst ? user_dst : (char*)calitas_out_alloc_pinned(total + 1);
  if (!text) return fail(ctx, CALITAS_EINVAL, "out of memory");
  std::memcpy(text, rs.header.data(), hlen);
  std::memcpy(text + hlen, body.data(), body.size());
  text[total] = 0;
  tm.hit_rows = rows; tm.hits_bytes = total; tm.lanes = 1;
  ctx->timing = tm;
  *tsv = text;
  if (tsv_bytes) *tsv_bytes = total;
  if (n_rows) *n_rows = rows;
  return CALITAS_OK;
}

// user_dst: the text goes into this caller-owned buffer of user_cap bytes (calitas_search_hits_into) instead of a block of the
// library; CALITAS_EINVAL when it is too small.
static int search_hits_attempt(calitas_ctx* ctx, const calitas_guide_t* guide, const std::string& guide_id, const calitas_params_t* params,
                               const char* aligner_version, const char* time_stamp, char** tsv, uint64_t* tsv_bytes, uint64_t* n_rows,
                               char* user_dst, uint64_t user_cap, const uint64_t* owned, bool* owned_declined) {
  if (!owned && params && (params->first_window != 0 || params->n_windows != 0))
    return search_hits_owned(ctx, guide, guide_id, params, aligner_version, time_stamp, tsv, tsv_bytes, n_rows, user_dst, user_cap);
  if (owned_declined) *owned_declined = false;
  const auto t_call = std::chrono::steady_clock::now();
  *tsv = nullptr;
  if (tsv_bytes) *tsv_bytes = 0;
  if (n_rows) *n_rows = 0;
  SearchPlan pl;
  int rc = plan_search(ctx, 1, guide, params, pl);
  if (rc) return rc;
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  rc = ensure_bin_base(ctx, pl, ctx->stream);                // (built once per reference and window size)
  if (rc) return rc;
  const PackedRef& ref = ctx->ref;
  const SearchPlan whole = pl;
  if (owned && (!plan_owned_range(ctx, pl, owned[0], owned[1]) || !binned_possible(ctx, pl))) { if (owned_declined) *owned_declined = true; return CALITAS_OK; }
  // The constant pieces of a row.  A chunked search builds them after its scans are queued: nothing on the device needs them before
  // the first range's rows, and the first scan should not wait for string formatting on the host.
  std::string version, stamp;
  RowStrings rs, rs_compact;
  size_t hlen = 0;
  auto make_rows = [&] {
    calitas_default_version_and_stamp(aligner_version, time_stamp, version, stamp);
    rs = make_row_strings(ref, pl.gh[0], guide_id, pl.p, version, stamp);
    rs_compact = compact_row_strings(rs);
    hlen = rs.header.size();
  };
  // Compact rows (post.hpp) for the ranges of a chunked call whose text is copied while later ranges are still at work: half the bytes
  // on the bus, head and tail put back by the worker pool.  Not the last range: nothing hides its expansion, and where the per-bin
  // kernels run its rows kernel writes the text straight to its final place.
  std::vector<char> lane_compact;
  auto rs_lane = [&](size_t c) -> const RowStrings& { return c < lane_compact.size() && lane_compact[c] ? rs_compact : rs; };
  const bool trace = TUNE_GET("CALITAS_TRACE") != nullptr;
  HIP_TRY(ctx, hipSetDevice(ctx->device));

  // ---- how many lanes: one pass over the whole reference, or contig ranges pipelined against each other ----
  std::vector<double> weights;
  if (const char* e = TUNE_GET("CALITAS_CHUNKS")) {
    // "3" = three equal chunks, "5:3:2" = relative sizes
    for (const char* q = e; *q;) {
      char* end = nullptr;
      double v = std::strtod(q, &end);
      if (end == q) break;
      weights.push_back(v);
      q = *end == ':' ? end + 1 : end;
    }
    if (weights.size() == 1) { int k = std::max(1, std::min(16, (int)weights[0])); weights.assign((size_t)k, 1.0); }
    for (double w : weights) if (!(w > 0)) { weights.clear(); break; }
  } else if ((owned ? owned[1] * (uint64_t)pl.step : ref.total_bases) >= (2048ull << 20)) {
    // measured on hg38-sized input: the last range small, its tail is what nothing hides.  5:3:2 while all of the text
    // crossed the bus behind the first range's rows; with compact rows for the first two ranges (round 4) the copies are half as long
    // and the first range can be larger, the last smaller: 5.5:3:1.5 2.196 ms against 2.268 (tools/sweep_env.py, interleaved;
    // 6:3:1 2.253, 5:3.5:1.5 2.207, four ranges 2.28).  With the last range on the per-bin kernels, its rows compact too and every
    // expansion fed by its copy, the last range shrinks again: 5.8:2.9:1.3 2.016 / 2.025 ms against 2.073 for 5.5:3:1.5 (6:2.8:1.2
    // 2.021, 6.2:2.8:1 2.013, 6.4:2.6:1 2.022, 5.6:3.2:1.2 2.083, 5:3:2 2.104; the cuts fall on contig boundaries)
    weights = {5.8, 2.9, 1.3};
  } else if ((owned ? owned[1] * (uint64_t)pl.step : ref.total_bases) >= (256ull << 20)) {
    // a half, a quarter or an eighth of it (a rank's share on 2, 4 or 8 GPUs).  Round 3, with the per-bin tail: 1.41 / 0.86 / 0.58 ms
    // for two equal ranges against 1.46 / 0.89 / 0.62 for 3:2 and 1.53 / 0.96 / 0.72 for three; one pass: - / 0.90 / 0.60.
    // Round 4, with the last range's text written in place by its rows kernel (nothing of it is left to copy when its tail ends) the
    // second range shrinks: interleaved on one box (tools/owned_cut_sweep.py) a half 1.315 -> 1.23 ms at 5:3 (3:2 1.25, 2:1 1.31), a
    // quarter 0.75 -> 0.72 at 5:3 (3:2 0.73), an eighth 0.48 -> 0.46 at 3:2 (5:3 0.46-0.49, 2:1 0.53)
    if ((owned ? owned[1] * (uint64_t)pl.step : ref.total_bases) >= (600ull << 20)) weights = {5, 3};
    else weights = {3, 2};
  }
  std::vector<std::pair<int, int>> ranges;
  if (!owned && weights.size() > 1 && pl.p.chrom_index < 0 && ref.contigs.size() > 1) ranges = chunk_ranges(ref, weights);
  // A window range of a multi-GPU job is cut into consecutive window ranges the same way (a rank of two searches half the genome: one
  // pass took 1.68 ms, scan, tail and copy one after the other): each piece owns its stretch, the texts concatenate (coordinate_start
  // is the first sort key), and a piece whose bins decline declines the call.
  std::vector<SearchPlan> owned_plans;
  if (owned && weights.size() > 1) {
    double wsum = 0, acc = 0;
    for (double w : weights) wsum += w;
    uint64_t first = owned[0];
    for (size_t c = 0; c < weights.size(); c++) {
      acc += weights[c];
      const uint64_t end = c + 1 == weights.size() ? owned[0] + owned[1] : owned[0] + (uint64_t)((double)owned[1] * acc / wsum);
      if (end <= first) continue;
      SearchPlan q = whole;
      if (!plan_owned_range(ctx, q, first, end - first) || !binned_possible(ctx, q)) { if (owned_declined) *owned_declined = true; return CALITAS_OK; }
      owned_plans.push_back(q);
      first = end;
    }
    if (owned_plans.size() < 2) owned_plans.clear();
  }
  const size_t K = !owned_plans.empty() ? owned_plans.size() : ranges.size() > 1 ? ranges.size() : 1;

  std::vector<LaneText> parts(K);
  std::vector<calitas_ctx*> lanes(K, ctx);
  char* text = nullptr;
  char* text_dev = nullptr;                                    // the same memory as the device addresses it (null: it cannot)
  size_t capacity = 0;
  std::mutex copy_mu;
  auto alloc_text = [&](size_t body) {
    text_dev = nullptr;
    if (user_dst) {                                   // the caller's buffer: as much room as it has
      if (user_cap < hlen + body + 1 && user_cap < hlen + 1) return false;
      capacity = (size_t)user_cap - hlen - 1;
      text = user_dst;
    } else {
      capacity = body;
      text = (char*)calitas_out_alloc_pinned(hlen + body + 1);
    }
    if (text) std::memcpy(text, rs.header.data(), hlen);
    if (text && !TUNE_GET("CALITAS_TEXT_IN_PLACE_OFF")) {
      void* dp = nullptr;
      if (hipHostGetDevicePointer(&dp, text, 0) == hipSuccess) text_dev = static_cast<char*>(dp); else (void)hipGetLastError();
    }
    return text != nullptr;
  };
  auto free_text = [&] { if (!user_dst) calitas_free(text); text = nullptr; };
  const char* no_room = "the caller's buffer is too small for the text";
  // copies lane c's rows to their place (offset = header + rows of the lanes before it)
  auto place = [&](size_t c, size_t offset) -> int {
    LaneText& lt = parts[c];
    if (!lt.bytes) return CALITAS_OK;
    if (lt.on_host) { std::memcpy(text + hlen + offset, lt.host_rows.data(), lt.bytes); return CALITAS_OK; }
    calitas_ctx* lane = lanes[c];
    if (lt.in_place) {                                         // written by the rows kernel where it belongs: wait for the kernel
      if (lt.d_text != text_dev + hlen + offset) return fail(lane, CALITAS_EHIP, "a lane's text was written to another place than the one it belongs to (internal error)");
      HIP_TRY(lane, calitas_spin_sync(lane->stream));
      g_marks.mark("rows-done");
      if (lane->binned_late_check && lane->mbox.host && lane->mbox.host[BIN_BOX_LATE] != 0)
        return fail(lane, CALITAS_EHIP, "binned rows kernel: a row's length differs between the two kernels (internal error)");
      lt.tm.hits_copy_ms = 0;
      lt.tm.hits_kernel_ms = rows_stage_ms(lane, lt.tm);
      return CALITAS_OK;
    }
    if (lt.compact_bytes) {                                    // compact rows: over the bus into a staging block, head and tail put back on the pool
      char* staging = (char*)calitas_out_alloc_pinned((size_t)lt.compact_bytes);
      if (!staging) return fail(lane, CALITAS_EINVAL, "out of memory");
      g_marks.mark("staging");
      size_t wrote = 0;
      // (Copy and expansion in pieces were no faster while every piece cost two passes of the whole pool -- ~100 us of fixed latency, 2.164
      // against 2.170 ms per hg38-sized call with 4 MB pieces, 3.4 ms with 1 MB; as one job fed by the copy's pieces: see DESIGN.md 4.5.)
      int r = compact_rows_to_host(ctx, lane, lt, (size_t)lt.compact_bytes, staging, rs.head, rs.tail, text + hlen + offset, &copy_mu, &wrote);
      calitas_free(staging);
      if (r) return r;
      if ((r = rows_late_check(lane, lt)) != CALITAS_OK) return r;
      if (wrote != (size_t)lt.bytes) return fail(lane, CALITAS_EHIP, "a lane's compact rows do not expand to the row count the device reported (internal error)");
      lt.tm.hits_kernel_ms = rows_stage_ms(lane, lt.tm);
      return CALITAS_OK;
    }
    int r = text_to_host(ctx, lane, text + hlen + offset, lt.d_text, (size_t)lt.bytes, &copy_mu, &lt.tm.hits_copy_ms);
    if (r) return r;
    if ((r = rows_late_check(lane, lt)) != CALITAS_OK) return r;
    lt.tm.hits_kernel_ms = rows_stage_ms(lane, lt.tm);   // recorded around hits_run by lane_rows
    return CALITAS_OK;
  };

  bool chunked = K > 1;
  g_marks.mark("planned");
  if (chunked) {
    rc = ensure_lanes(ctx, K);
    if (rc) return rc;
    std::vector<SearchPlan> plans(K, pl);
    for (size_t c = 0; c < K && !rc; c++) {
      lanes[c] = ctx->lanes[c];
      SearchPlan& q = plans[c];
      if (!owned_plans.empty()) q = owned_plans[c];            // (a piece of a window range: planned above)
      else {
      q.tile_lo = (uint32_t)(ref.contigs[ranges[c].first].gbase / ref.tile);
      const uint32_t tile_hi = ranges[c].second < (int)ref.contigs.size() ? (uint32_t)(ref.contigs[ranges[c].second].gbase / ref.tile) : (uint32_t)ref.tiles.size();
      q.n_tiles = tile_hi - q.tile_lo;
      q.bases = 0; q.win_lo = 0; q.win_n = 0;
      for (int k = 0; k < ranges[c].first; k++) q.win_lo += window_count(ref.contigs[k].len, q.step);
      for (int k = ranges[c].first; k < ranges[c].second; k++) { q.bases += ref.contigs[k].len; q.win_n += window_count(ref.contigs[k].len, q.step); }
      plan_bins(ctx, q, ranges[c].first, ranges[c].second);
      }
      q.narrow_tail = c + 1 < K; q.three_ranges = K >= 3; q.last_range = K >= 3 && c + 1 == K; q.range_index = (int)c;
      rc = lane_prepare(lanes[c], q);
      if (rc) ctx->err = lanes[c]->err;
    }
    if (rc) return rc;
    // all scans go to one low-priority stream in chunk order; each lane's own (high-priority) stream picks its chunk up
    // when its scan is done, so the tail of chunk c runs while chunk c+1 is still being scanned
    rc = ensure_window_table(ctx, pl, ctx->scan_stream);
    if (rc) return rc;
    const bool device_rows = !TUNE_GET("CALITAS_HOST_HITS");
    {
      bool compact_on = device_rows;
      if (const char* e = TUNE_GET("CALITAS_COMPACT_ROWS")) compact_on = compact_on && std::atoi(e) != 0;
      lane_compact.assign(K, 0);
      // Which ranges move compact rows: the leading ones always (their expansion hides behind the later ranges' scans); the last one
      // where the text is long -- a call cut into three: its rows kernel then writes 3 MB into device memory and the text's pieces are
      // expanded as they land, instead of 9 MB written across PCIe by the kernel itself, 2.072 against 2.098 ms per hg38-sized call;
      // the last of two ranges keeps its rows kernel writing in place (a rank of eight: 0.48 against 0.50 ms).
      size_t n_compact = K >= 3 ? K : K - 1;
      if (const char* e = TUNE_GET("CALITAS_COMPACT_LANES")) n_compact = std::min<size_t>(K, (size_t)std::max(0, std::atoi(e)));
      for (size_t c = 0; c < n_compact; c++) lane_compact[c] = compact_on ? 1 : 0;
    }
    // (no early return inside this loop: the scans of the earlier lanes are already in flight and every exit waits for them)
    auto hip_rc = [&](hipError_t e, const char* what) {
      if (e == hipSuccess) return (int)CALITAS_OK;
      if (e == hipErrorOutOfMemory) (void)hipGetLastError();
      return fail(ctx, e == hipErrorOutOfMemory ? CALITAS_ENOMEM : CALITAS_EHIP, std::string(what) + ": " + hipGetErrorString(e));
    };
    // (The inputs of all ranges queued ahead of the first scan, so that the scans run back to back: tried again with the row-wise
    // scan, 2.71 vs 2.68-2.72 ms per pass -- the scans then take 6 % longer beside the tails and nothing is won.)
    g_marks.mark("lanes-ready");
    // (Holding the scan of a range back until the aligner kernels of the range before it are done -- they take half as long again
    // beside a scan, the scan twice as long beside them -- was tried: 2.77 against 2.55 ms per pass.)
    // The inputs of all ranges (guide constants, cleared counters: a 272-byte upload and a fill per lane, 60-140 us of the scan stream
    // each when they sit between two scans) are queued ahead of the scans; CALITAS_INPUTS_FIRST=0: each before its own scan, 1: all of
    // them ahead of the first scan on the scan stream (round 2), 2 (default, round 3: 2.320 against 2.335 ms per hg38-sized call, 0.510
    // against 0.517 for an eighth, interleaved):
    bool inputs_first = true;
    int inputs_mode = 2;
    if (const char* e = TUNE_GET("CALITAS_INPUTS_FIRST")) { inputs_mode = std::atoi(e); inputs_first = inputs_mode != 0; }
    // mode 2: the first range's inputs ahead of its scan on the scan stream, the later ranges' on their own streams (which have nothing
    // else to do yet); the scan stream waits for each with an event that has long fired when its turn comes.  A range's small inputs
    // are ONE launch (queue_lane_setup) where they used to be two stream commands for the scan and three or four for the row stage.
    std::vector<char> rows_queued(K, 0);                      // the lane's row constants went out with its scan inputs (one launch for both)
    const bool device_rows_early = !TUNE_GET("CALITAS_HOST_HITS");
    if (inputs_mode == 2) {
      // the first range: its scan inputs (one launch: queue_lane_setup) and its scan, before anything else is prepared
      bool one = false;
      rc = queue_lane_setup(lanes[0], plans[0], nullptr, ctx->scan_stream, &one);
      if (!rc && !one) rc = queue_scan_inputs(lanes[0], plans[0], ctx->scan_stream);
      if (!rc) rc = launch_scan_stage(lanes[0], plans[0], ctx->scan_stream, true);
      if (rc) ctx->err = lanes[0]->err;
      g_marks.mark("scan-queued");
      if (!rc) make_rows();
      g_marks.mark("row-strings");
      // the later ranges: scan inputs and row constants in one launch on the range's own stream, then its scan behind the event
      for (size_t c = 1; c < K && !rc; c++) {
        one = false;
        if (device_rows_early) rc = queue_lane_setup(lanes[c], plans[c], &rs_lane(c), lanes[c]->stream, &one);
        if (!rc && one) rows_queued[c] = 1;
        if (!rc && !one) rc = queue_scan_inputs(lanes[c], plans[c], lanes[c]->stream);
        if (!rc) rc = hip_rc(hipEventRecord(lanes[c]->inputs_ready, lanes[c]->stream), "hipEventRecord");
        if (!rc) rc = hip_rc(hipStreamWaitEvent(ctx->scan_stream, lanes[c]->inputs_ready, 0), "hipStreamWaitEvent");
        if (!rc) rc = launch_scan_stage(lanes[c], plans[c], ctx->scan_stream, true);           // records lanes[c]->scan_done
        if (rc && ctx->err.empty()) ctx->err = lanes[c]->err;
        g_marks.mark("scan-queued");
      }
      // ... and the first range's row constants (its tail starts when its scan ends)
      if (!rc && device_rows_early) {
        one = false;
        rc = queue_lane_setup(lanes[0], plans[0], &rs_lane(0), lanes[0]->stream, &one, false);
        if (!rc && one) rows_queued[0] = 1;
        if (rc) ctx->err = lanes[0]->err;
      }
    } else {
      if (inputs_first)
        for (size_t c = 0; c < K && !rc; c++) { rc = queue_scan_inputs(lanes[c], plans[c], ctx->scan_stream); if (rc) ctx->err = lanes[c]->err; }
      for (size_t c = 0; c < K && !rc; c++) {
        rc = launch_scan_stage(lanes[c], plans[c], ctx->scan_stream, inputs_first);           // records lanes[c]->scan_done
        if (rc) ctx->err = lanes[c]->err;
        g_marks.mark("scan-queued");
      }
      if (!rc) make_rows();
      g_marks.mark("row-strings");
    }
    for (size_t c = 0; c < K && !rc; c++) {
      // the row constants of a range go onto its stream before the wait for its scan: in place while the scan runs
      if (device_rows && !rows_queued[c]) rc = hip_rc(queue_row_constants(lanes[c], plans[c], rs_lane(c)), "hits_prepare");
      if (!rc) rc = hip_rc(hipStreamWaitEvent(lanes[c]->stream, lanes[c]->scan_done, 0), "hipStreamWaitEvent");
    }
    g_marks.mark("rows-prepared");
    if (rc) { (void)hipDeviceSynchronize(); return rc; }
    auto guess = [](size_t last) { return last + last / 4 + (1u << 20); };   // the next call's text is about as long as the last one's
    if (!alloc_text(guess(ctx->last_text_bytes))) {
      (void)hipDeviceSynchronize();
      return fail(ctx, CALITAS_EINVAL, "out of memory");
    }
    std::mutex mu;
    std::condition_variable cv;
    std::vector<char> done(K, 0);
    std::vector<char> placed(K, 0);
    auto lane_body = [&](size_t c) {
      {
        (void)hipSetDevice(ctx->device);
        if (c) g_marks.start_at(t_call);
        LaneText& lt = parts[c];
        // (whatever happens to this lane -- an exception included --, the lanes behind it must not wait for it forever)
        struct DoneGuard {
          std::mutex& mu; std::condition_variable& cv; std::vector<char>& done; size_t c;
          ~DoneGuard() { std::lock_guard<std::mutex> lk(mu); done[c] = 1; cv.notify_all(); }
        } done_guard{mu, cv, done, c};
        // the last range's text is what nothing hides: its rows kernel writes it to its final place -- 0.475 against 0.512 ms for an eighth
        // of the genome, 1.285 against 1.346 for a half.  (For the earlier ranges too: 0.531 / 1.50 ms -- their row kernels then sit on
        // the CUs waiting for the bus while the next range is being scanned; their copies run beside the later ranges' kernels anyway.)
        LaneDest dest;
        dest.get = [&, c](char** dst, uint64_t* cap) {
          if (!text_dev) return false;
          size_t before = 0;
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { for (size_t i = 0; i < c; i++) if (!done[i]) return false; return true; });
          for (size_t i = 0; i < c; i++) { if (parts[i].rc != CALITAS_OK) return false; before += parts[i].bytes; }
          if (before >= capacity) return false;
          *dst = text_dev + hlen + before; *cap = capacity - before;
          return true;
        };
        lt.rc = lane_rows(lanes[c], plans[c], true, rs_lane(c), guide_id, version, stamp, lt, device_rows, c + 1 == K && device_rows && !lane_compact[c] ? &dest : nullptr);
        if (lt.rc == CALITAS_OK && lane_compact[c] && !lt.on_host && !lt.in_place && lt.bytes) {   // what the lanes behind it place their text by: the expanded size
          lt.compact_bytes = lt.bytes;
          lt.bytes += lt.rows * (uint64_t)(rs.head.size() + rs.tail.size() - 1);
        }
        size_t offset = 0;
        bool ok = lt.rc == CALITAS_OK;
        {
          std::unique_lock<std::mutex> lk(mu);
          done[c] = 1;
          cv.notify_all();
          cv.wait(lk, [&] { for (size_t i = 0; i < c; i++) if (!done[i]) return false; return true; });
          for (size_t i = 0; i < c; i++) { offset += parts[i].bytes; ok = ok && parts[i].rc == CALITAS_OK; }
        }
        if (ok && offset + lt.bytes <= capacity) {
          int r = place(c, offset);
          if (r) lt.rc = r; else placed[c] = 1;
        }
        if (c) g_marks.dump((int)c);
      }
    };
    g_marks.mark("text-allocated");
    ctx->lane_threads->start(K, lane_body);
    g_marks.mark("threads-started");
    ctx->lane_threads->guarded([&] { lane_body(0); });         // the calling thread drives the first lane itself
    ctx->lane_threads->wait();
    g_marks.mark("joined");
    if (g_marks.on && !ctx->lane_threads->threw.load()) {       // the scan stream's idle time between the ranges' scans
      std::string gaps;
      for (size_t c = 0; c + 1 < K; c++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, lanes[c]->t_scan1, lanes[c + 1]->t_scan0) == hipSuccess) gaps += " " + std::to_string((int)(ms * 1e3f));
        if (hipEventElapsedTime(&ms, lanes[c]->t_scan0, lanes[c]->t_scan1) == hipSuccess) gaps += " (scan " + std::to_string((int)(ms * 1e3f)) + ")";
      }
      std::fprintf(stderr, "[calitas] scan stream idle between ranges (us):%s\n", gaps.c_str());
    }
    if (ctx->lane_threads->threw.load()) {
      (void)hipDeviceSynchronize();
      free_text();
      return fail(ctx, CALITAS_EHIP, ctx->lane_threads->failure());
    }
    rc = CALITAS_OK;
    bool overflow = false;
    for (size_t c = 0; c < K; c++) {
      if (parts[c].rc == CALITAS_ESTATE) overflow = true;
      else if (parts[c].rc && !rc) { rc = parts[c].rc; ctx->err = lanes[c]->err; }
    }
    if (rc == kOwnedDeclined || (owned && overflow)) {          // a piece of the window range could not be decided bin by bin: the caller's slow path
      (void)hipDeviceSynchronize();
      free_text();
      if (owned_declined) *owned_declined = true;
      return CALITAS_OK;
    }
    if (rc || overflow) {
      (void)hipDeviceSynchronize();
      free_text();
      if (rc) return rc;
      {
        // the lanes counted their scan records and alignments even where they could not keep them: would one pass over everything fit?
        uint64_t n_rec = 0, n_raw = 0;
        for (size_t c = 0; c < K; c++) { n_rec += lanes[c]->h_counters[0]; n_raw += std::max(lanes[c]->h_counters[1], lanes[c]->h_counters[3]); }
        // strips + records + alignments with the filter's and the row stage's scratch (~110 + ~1100 bytes each, text included)
        const uint64_t need = n_rec * (pl.slab_per_rec + sizeof(ScanRecord)) * 5 / 4 + n_raw * (sizeof(RawAln) + 110 + 1100);
        size_t mem_free = 0, mem_total = 0;
        if (hipMemGetInfo(&mem_free, &mem_total) == hipSuccess && need > (uint64_t)mem_total * 7 / 10)
          return fail(ctx, CALITAS_ENOMEM, "one pass would need about " + std::to_string(need >> 30) + " GB of scratch on the device");
      }
      if (trace) std::fprintf(stderr, "[calitas] search_hits: a lane's buffers overflowed, rerunning in one pass\n");
      chunked = false;
      parts.assign(1, LaneText()); lanes.assign(1, ctx);
    } else {
      size_t total = 0;
      for (auto& lt : parts) total += lt.bytes;
      bool all = true;
      for (size_t c = 0; c < K; c++) all = all && (placed[c] || parts[c].bytes == 0);
      if (!all) {   // the guess was too small: place everything again in a buffer of the right size
        if (user_dst) return fail(ctx, CALITAS_EINVAL, no_room);
        free_text();
        if (!alloc_text(guess(total))) return fail(ctx, CALITAS_EINVAL, "out of memory");   // big enough for the next call's guess as well
        size_t off = 0;
        for (size_t c = 0; c < K; c++) { rc = place(c, off); if (rc) { ctx->err = lanes[c]->err; free_text(); return rc; } off += parts[c].bytes; }
      }
    }
  }
  if (!chunked) {
    if (rs.header.empty()) make_rows();
    rc = lane_rows(ctx, pl, false, rs, guide_id, version, stamp, parts[0]);
    if (rc == kOwnedDeclined) { if (owned_declined) *owned_declined = true; return CALITAS_OK; }
    if (rc) return rc;
    g_marks.mark("lane-done");
    if (!alloc_text((size_t)parts[0].bytes) || parts[0].bytes > capacity) return fail(ctx, CALITAS_EINVAL, user_dst ? no_room : "out of memory");
    rc = place(0, 0);
    if (rc) { free_text(); return rc; }
    g_marks.mark("text-copied");
  }
  size_t total = hlen;
  calitas_timing_t tm{};
  uint64_t rows = 0;
  for (auto& lt : parts) {
    total += lt.bytes; rows += lt.rows;
    tm.scan_kernel_ms += lt.tm.scan_kernel_ms; tm.align_kernel_ms += lt.tm.align_kernel_ms; tm.gpu_total_ms += lt.tm.gpu_total_ms;
    tm.host_post_ms += lt.tm.host_post_ms; tm.bases_scanned += lt.tm.bases_scanned; tm.packed_bytes += lt.tm.packed_bytes;
    tm.scan_records += lt.tm.scan_records; tm.candidate_columns += lt.tm.candidate_columns; tm.raw_alignments += lt.tm.raw_alignments;
    tm.accepted_alignments += lt.tm.accepted_alignments; tm.retries += lt.tm.retries;
    tm.hits_kernel_ms += lt.tm.hits_kernel_ms; tm.hits_copy_ms += lt.tm.hits_copy_ms; tm.binned_lanes += lt.tm.binned_lanes; tm.owned_general_lanes += lt.tm.owned_general_lanes;
  }
  text[total] = 0;
  tm.hit_rows = rows; tm.hits_bytes = total; tm.lanes = (uint32_t)parts.size();
  ctx->timing = tm;
  ctx->last_text_bytes = total;
  if (trace && parts.size() > 1) {
    std::string per;
    for (auto& lt : parts) { char b[96]; std::snprintf(b, sizeof b, " [scan %.3f align+trace %.3f rows %.3f copy %.3f]", lt.tm.scan_kernel_ms, lt.tm.align_kernel_ms, lt.tm.hits_kernel_ms, lt.tm.hits_copy_ms); per += b; }
    std::fprintf(stderr, "[calitas] search_hits lanes (ms):%s\n", per.c_str());
    if (chunked) {                                             // what the scan stream lost between two scans: end of one .. start of the next
      std::string gaps;
      for (size_t c = 0; c + 1 < lanes.size() && c + 1 < K; c++) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, lanes[c]->t_scan1, lanes[c + 1]->t_scan0) != hipSuccess) { (void)hipGetLastError(); continue; }
        char b[32]; std::snprintf(b, sizeof b, " %.1f", ms * 1e3); gaps += b;
      }
      std::fprintf(stderr, "[calitas] search_hits: scan stream idle between the scans (us):%s\n", gaps.c_str());
    }
  }
  if (trace)
    std::fprintf(stderr, "[calitas] search_hits: %zu lane(s), scan %.3f ms, align %.3f ms, hits kernels %.3f ms, text copy %.3f ms (sums over lanes), call %.3f ms (%llu accepted, %llu rows, %zu bytes)\n",
                 parts.size(), tm.scan_kernel_ms, tm.align_kernel_ms, tm.hits_kernel_ms, tm.hits_copy_ms,
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count(),
                 (unsigned long long)tm.accepted_alignments, (unsigned long long)rows, total);
  if (pl.p.chrom_index < 0 && (ctx->fit_pams != pl.gd[0].n_pams || ctx->fit_L != pl.gd[0].L || pl.gd[0].min_guide_score < ctx->fit_min_score)) {
    ctx->fit_L = pl.gd[0].L; ctx->fit_pams = pl.gd[0].n_pams; ctx->fit_min_score = pl.gd[0].min_guide_score;   // the most permissive search seen to fit
  }
  *tsv = text;
  if (tsv_bytes) *tsv_bytes = total;
  if (n_rows) *n_rows = rows;
  return CALITAS_OK;
}

// ---- calitas_search_hits_batch ----------------------------------------------------------------------------------------
// Guides flow through the lanes as a pipeline: every lane thread queues the scan of its next guide on the shared low-priority
// scan stream and then runs that guide's tail (align ... rows, copy) on its own stream, so guide g+1 is being scanned while
// guide g's tail runs.  Each guide's text goes to its own pinned buffer.
int calitas_search_hits_batch_impl(calitas_ctx* ctx, int32_t n_guides, const calitas_guide_t* guides, const char* const* guide_ids,
                                  const calitas_params_t* params, const char* aligner_version, const char* time_stamp, char** tsv,
                                  uint64_t* tsv_bytes, uint64_t* n_rows) {
  const auto t_call = std::chrono::steady_clock::now();
  for (int i = 0; i < n_guides; i++) { tsv[i] = nullptr; if (tsv_bytes) tsv_bytes[i] = 0; if (n_rows) n_rows[i] = 0; }
  std::string version, stamp;
  calitas_default_version_and_stamp(aligner_version, time_stamp, version, stamp);
  auto release = [&]() { for (int i = 0; i < n_guides; i++) { calitas_free(tsv[i]); tsv[i] = nullptr; } };
  // Five guides in flight: with three the bus idled a sixth of the time between the texts of a 96-guide batch on an hg38-sized genome
  // (15.3 GB per batch: 328.6 ms; four lanes 298.0, five 292.1 = 52 GB/s, six 300.6, eight 295.5).
  int n_lanes = 5;
  if (const char* e = TUNE_GET("CALITAS_BATCH_LANES")) n_lanes = std::max(1, std::min(8, std::atoi(e)));
  n_lanes = std::min(n_lanes, (int)n_guides);
  if (n_lanes < 2) {   // nothing to pipeline
    for (int i = 0; i < n_guides; i++) {
      int rc = calitas_search_hits_impl(ctx, &guides[i], guide_ids && guide_ids[i] ? guide_ids[i] : "", params, version.c_str(), stamp.c_str(), &tsv[i],
                                tsv_bytes ? &tsv_bytes[i] : nullptr, n_rows ? &n_rows[i] : nullptr);
      if (rc) { release(); return rc; }
    }
    return CALITAS_OK;
  }
  // plans first: every guide is validated before anything is queued, and all must share one window tiling
  // A window range (a process's stretch of a multi-GPU job): every guide's plan is the stretch's -- the rows it owns, decided by the
  // per-bin kernels (plan_owned_range); a guide whose bins decline goes through calitas_search_hits on the range afterwards.
  const bool ranged = params && (params->first_window != 0 || params->n_windows != 0);
  calitas_params_t whole = *params;
  whole.first_window = 0; whole.n_windows = 0;
  if (ranged && whole.chrom_index >= 0) return fail(ctx, CALITAS_EINVAL, "a window range and chrom_index exclude each other");
  std::vector<SearchPlan> plans((size_t)n_guides);
  for (int i = 0; i < n_guides; i++) {
    int rc = plan_search(ctx, 1, &guides[i], &whole, plans[i]);
    if (rc) return rc;
    if (plans[i].step != plans[0].step)
      return fail(ctx, CALITAS_EINVAL, "all guides of one batch must have the same length (same window tiling, SearchReference.scala:529)");
  }
  HIP_TRY(ctx, hipSetDevice(ctx->device));
  for (int i = 0; i < n_guides; i++) { int rc = ensure_bin_base(ctx, plans[i], ctx->stream); if (rc) return rc; }
  // Which tail: every guide's tail runs beside the scans of the guides behind it, where the per-bin kernels cost the scans more than
  // they save the tail (binned_possible: the three-range call's finding) -- the general kernels, then, unless the batch
  // runs on a stretch, which only the bins can own.  (Round 3 got there by accident: the first guide that crowded a bin switched the
  // bins off for every guide behind it; with the decline remembered per guide the batch took 328 instead of 295 ms per 96 guides.)
  // (A reference below 2 Gb keeps the bins, as a single call on it does: fewer launches, and its scans are short.)
  if (!ranged && ctx->ref.total_bases >= (2048ull << 20) && !TUNE_GET("CALITAS_BATCH_BINNED")) for (auto& q : plans) q.three_ranges = true;
  std::vector<char> owned_ok((size_t)n_guides, 1);
  if (ranged) {
    if (params->first_window < 0 || params->n_windows <= 0 || (uint64_t)params->first_window + (uint64_t)params->n_windows > plans[0].win_n)
      return fail(ctx, CALITAS_EINVAL, "first_window / n_windows outside the window table (" + std::to_string(plans[0].win_n) + " windows)");
    for (int i = 0; i < n_guides; i++)
      owned_ok[(size_t)i] = plan_owned_range(ctx, plans[i], (uint64_t)params->first_window, (uint64_t)params->n_windows) && binned_possible(ctx, plans[i]);
  }
  int rc = ensure_lanes(ctx, (size_t)n_lanes);
  if (rc) return rc;
  for (int l = 0; l < n_lanes && !rc; l++) { rc = lane_prepare(ctx->lanes[l], plans[0]); if (rc) ctx->err = ctx->lanes[l]->err; }
  if (rc) return rc;
  rc = ensure_window_table(ctx, plans[0], ctx->scan_stream);
  if (rc) return rc;
  int n_scan_streams = 1;
  if (const char* e = TUNE_GET("CALITAS_BATCH_SCAN_STREAMS")) n_scan_streams = std::max(1, std::min(4, std::atoi(e)));
  if (n_scan_streams > 1) {
    int least = 0, greatest = 0;
    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
    for (int k = 0; k + 1 < n_scan_streams; k++)
      if (!ctx->scan_more[k]) HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->scan_more[k], hipStreamNonBlocking, least));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->scan_stream));         // (the window table, if it was just built, is there for all of them)
  }
  const PackedRef& ref = ctx->ref;
  std::mutex scan_mu, copy_mu;
  const bool device_rows = !TUNE_GET("CALITAS_HOST_HITS");
  std::atomic<uint64_t> expand_us{0};                           // lane threads' time in expand_rows (their turn on the pool included)
  bool compact_rows = device_rows;
  if (const char* e = TUNE_GET("CALITAS_COMPACT_ROWS")) compact_rows = compact_rows && std::atoi(e) != 0;
  std::vector<int> rcs((size_t)n_guides, CALITAS_OK);
  std::vector<std::string> errs((size_t)n_guides);   // a failed guide's message, kept apart from its lane (a retry below destroys the lanes)
  std::vector<calitas_timing_t> tms((size_t)n_guides);
  // A guide on a lane, in three parts: its scan queued on the scan stream; its tail (aligner, filter, rows) on the lane's stream with
  // the host answering the tail's counters; its text brought to the host (copy + expansion) and handed over.
  struct InFlight {
    int g = -1;
    LaneText lt;
    RowStrings rs_full, rs;
  };
  auto queue_scan = [&](calitas_ctx* lane, int g, InFlight& f) -> int {
    const SearchPlan& pl = plans[g];
    const std::string gid = guide_ids && guide_ids[g] ? guide_ids[g] : "";
    f.g = g;
    f.lt = LaneText();
    f.rs_full = make_row_strings(ref, pl.gh[0], gid, pl.p, version, stamp);
    // the device writes compact rows (post.hpp): a batch is bound by its texts on the bus (15.3 GB per 96 guides on an hg38-sized
    // genome), and 270 of a row's ~520 bytes are the call's constants
    f.rs = compact_rows ? compact_row_strings(f.rs_full) : f.rs_full;
    // the previous guide of this lane is done on the device (its tail ended before this is queued), so the lane's buffers are free
    // for this scan
    if (device_rows) HIP_TRY(lane, queue_row_constants(lane, pl, f.rs));   // before the wait for the scan is queued
    std::lock_guard<std::mutex> lk(scan_mu);
    const int turn = g % n_scan_streams;
    return launch_scan_stage(lane, pl, turn == 0 ? ctx->scan_stream : ctx->scan_more[turn - 1]);       // records lane->scan_done
  };
  auto run_tail = [&](calitas_ctx* lane, InFlight& f, hipEvent_t scans_done) -> int {
    HIP_TRY(lane, hipStreamWaitEvent(lane->stream, scans_done, 0));
    const std::string gid = guide_ids && guide_ids[f.g] ? guide_ids[f.g] : "";
    return lane_rows(lane, plans[f.g], true, f.rs, gid, version, stamp, f.lt, device_rows);
  };
  auto finish = [&](calitas_ctx* lane, InFlight& f) -> int {
    const int g = f.g;
    LaneText& lt = f.lt;
    const RowStrings& rs = f.rs;
    const RowStrings& rs_full = f.rs_full;
    const bool expand = compact_rows && !lt.on_host;           // (rows the host stages built are whole already)
    const size_t add = rs_full.head.size() + rs_full.tail.size() - 1;
    const size_t hlen = rs.header.size(), total = hlen + (size_t)lt.bytes + (expand ? (size_t)lt.rows * add : 0);
    char* text = (char*)(expand ? calitas_out_alloc(total + 1) : calitas_out_alloc_pinned(total + 1));
    if (!text) return fail(lane, CALITAS_EINVAL, "out of memory");
    std::memcpy(text, rs.header.data(), hlen);
    if (lt.bytes && lt.on_host) std::memcpy(text + hlen, lt.host_rows.data(), (size_t)lt.bytes);
#ifdef CALITAS_EXPERIMENTS
    else if (const char* mode = TUNE_GET("CALITAS_BATCH_TEXT")) {   // what the batch costs without its texts' way home (the texts are wrong)
      if (!std::strcmp(mode, "copy") && lt.bytes) {
        char* staging = (char*)calitas_out_alloc_pinned((size_t)lt.bytes);
        if (!staging) { calitas_free(text); return fail(lane, CALITAS_EINVAL, "out of memory"); }
        int cr = text_to_host(ctx, lane, staging, lt.d_text, (size_t)lt.bytes, &copy_mu, &lt.tm.hits_copy_ms);
        calitas_free(staging);
        if (cr) { calitas_free(text); return cr; }
      } else HIP_TRY(lane, calitas_spin_sync(lane->stream));
    }
#endif
    else if (lt.bytes && expand) {
      char* staging = (char*)calitas_out_alloc_pinned((size_t)lt.bytes);
      if (!staging) { calitas_free(text); return fail(lane, CALITAS_EINVAL, "out of memory"); }
      size_t wrote = 0;
      const auto t_exp = std::chrono::steady_clock::now();
      int cr = compact_rows_to_host(ctx, lane, lt, (size_t)lt.bytes, staging, rs_full.head, rs_full.tail, text + hlen, &copy_mu, &wrote);
      expand_us.fetch_add((uint64_t)std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_exp).count());
      calitas_free(staging);
      if (!cr) cr = rows_late_check(lane, lt);
      if (cr) { calitas_free(text); return cr; }
      if (wrote != total - hlen) { calitas_free(text); return fail(lane, CALITAS_EHIP, "the compact rows of a guide do not expand to the row count the device reported (internal error)"); }
    } else if (lt.bytes) {
      int cr = text_to_host(ctx, lane, text + hlen, lt.d_text, (size_t)lt.bytes, &copy_mu, &lt.tm.hits_copy_ms);
      if (!cr) cr = rows_late_check(lane, lt);
      if (cr) { calitas_free(text); return cr; }
    }
    text[total] = 0;
    tsv[g] = text;
    if (tsv_bytes) tsv_bytes[g] = total;
    if (n_rows) n_rows[g] = lt.rows;
    tms[g] = lt.tm; tms[g].hit_rows = lt.rows; tms[g].hits_bytes = total;
    return CALITAS_OK;
  };
  auto failed = [&](calitas_ctx* lane, int g, int r) {
    rcs[g] = r;
    errs[g] = lane->err;
    (void)hipStreamSynchronize(lane->stream);                   // leave the lane quiet before its next guide
  };
  // (Phases -- the guides in groups of one per lane, the group's scans back to back with nothing beside them, then the group's tails side
  // by side with no scan beside them, the texts of the group before brought in meanwhile -- were tried in round 4: the scans then take
  // 1.27 instead of 2.05 ms per guide, and the 96 guides take as long as before, 249 ms.  A guide's tail is 1.3 ms of the whole chip
  // whatever runs beside it (five tails side by side: 6.6 ms); it is the tail's own kernels that have to get cheaper, not their place.
  // tools/batch_probe.py)
  // one host thread per lane: the caller drives lane 0, the context's lane threads (they live as long as the lanes: starting a thread
  // per lane and call cost ~100 us) the others
  auto lane_job = [&](size_t l_) {
    const int l = (int)l_;
    (void)hipSetDevice(ctx->device);
    calitas_ctx* lane = ctx->lanes[l];
    InFlight f;
    for (int g = l; g < n_guides; g += n_lanes) {
      if (!owned_ok[(size_t)g]) { rcs[g] = kOwnedDeclined; continue; }   // (the stretch is not one for the bins: below, one guide at a time)
      int r = queue_scan(lane, g, f);
      if (!r) r = run_tail(lane, f, lane->scan_done);
      if (!r) r = finish(lane, f);
      if (r) failed(lane, g, r);
    }
  };
  ctx->lane_threads->start((size_t)n_lanes, lane_job);
  ctx->lane_threads->guarded([&] { lane_job(0); });
  ctx->lane_threads->wait();
  if (ctx->lane_threads->threw.load()) {
    (void)hipDeviceSynchronize();
    release();
    return fail(ctx, CALITAS_EHIP, ctx->lane_threads->failure());
  }
  for (int g = 0; g < n_guides; g++) {
    if (rcs[g] == CALITAS_OK) continue;
    if (rcs[g] == CALITAS_ESTATE || rcs[g] == CALITAS_ENOMEM || rcs[g] == kOwnedDeclined) {   // a lane's buffers overflowed / did not fit / the bins declined a stretch: this guide again through calitas_search_hits (retry logic, per-contig passes, the whole-contig path of a stretch)
      int r = calitas_search_hits_impl(ctx, &guides[g], guide_ids && guide_ids[g] ? guide_ids[g] : "", params, version.c_str(), stamp.c_str(), &tsv[g],
                               tsv_bytes ? &tsv_bytes[g] : nullptr, n_rows ? &n_rows[g] : nullptr);
      if (r) { release(); return r; }
      tms[g] = ctx->timing;
      continue;
    }
    ctx->err = errs[g];
    release();
    return rcs[g];
  }
  calitas_timing_t tm{};
  for (auto& t : tms) {
    tm.scan_kernel_ms += t.scan_kernel_ms; tm.align_kernel_ms += t.align_kernel_ms; tm.gpu_total_ms += t.gpu_total_ms;
    tm.bases_scanned += t.bases_scanned; tm.packed_bytes += t.packed_bytes; tm.scan_records += t.scan_records;
    tm.candidate_columns += t.candidate_columns; tm.raw_alignments += t.raw_alignments; tm.accepted_alignments += t.accepted_alignments;
    tm.retries += t.retries; tm.hit_rows += t.hit_rows; tm.hits_bytes += t.hits_bytes; tm.binned_lanes += t.binned_lanes; tm.owned_general_lanes += t.owned_general_lanes;
  }
  tm.lanes = (uint32_t)n_lanes;
  ctx->timing = tm;
  if (TUNE_GET("CALITAS_TRACE"))
    std::fprintf(stderr, "[calitas] search_hits_batch: %d guides on %d lanes, scan %.3f ms, align %.3f ms, rows expanded on the host %.3f ms (sums), call %.3f ms (%llu rows, %llu bytes)\n",
                 n_guides, n_lanes, tm.scan_kernel_ms, tm.align_kernel_ms, (double)expand_us.load() * 1e-3,
                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count(),
                 (unsigned long long)tm.hit_rows, (unsigned long long)tm.hits_bytes);
  return CALITAS_OK;
}

