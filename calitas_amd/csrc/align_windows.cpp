// align_windows.cpp -- calitas_align_windows: SequentialGuideAligner.align / alignBest on explicit (guide, target) pairs
// through the same GPU kernels as SearchReference.  This is the seam PairwiseAlignSequences (PairwiseAlignSequences.scala:64)
// and AlignToReference (AlignToReference.scala:114-135 via SequentialGuideAligner.scala:359-418) call per task.
//
// Every task becomes one contig of a temporary packed reference with a single window [0, len) (no N trimming, no
// upper-casing requirement: SGA.align works on the bytes it is given and scoring is case-insensitive).  The scan kernel is
// skipped: with alignBest's limits (d = protospacer length) practically every column reaches minGuideScore, so all
// columns of both strands go to align_kernel, whose threshold test is the exact enumeration rule anyway.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <numeric>
#include <vector>

#include "ctx.hpp"
#include "tuning.hpp"

namespace {

struct TempDevice {
  uint32_t* codes = nullptr; uint32_t* mask = nullptr; Run* runs = nullptr; ContigInfo* contigs = nullptr; TileInfo* tiles = nullptr;
  uint64_t* win_base = nullptr; int2* win = nullptr; GuideDev* guides = nullptr;
};

// Slot k of the context's scratch (ctx.hpp, AlignScratch) with room for `bytes`, grown with a margin when it is too small.
template <typename T>
hipError_t scratch(calitas_ctx* ctx, int k, size_t bytes, T** out) {
  if (bytes > ctx->aw.cap[k]) {
    (void)hipFree(ctx->aw.p[k]); ctx->aw.p[k] = nullptr; ctx->aw.cap[k] = 0;
    const size_t want = bytes + bytes / 4 + 256;
    hipError_t e = hipMalloc(&ctx->aw.p[k], want);
    if (e != hipSuccess) return e;
    ctx->aw.cap[k] = want;
  }
  *out = static_cast<T*>(ctx->aw.p[k]);
  return hipSuccess;
}

}  // namespace

extern "C" int calitas_align_windows(calitas_ctx* ctx, int32_t n_tasks, const calitas_guide_t* guides, const uint8_t* const* targets,
                                     const uint32_t* target_lengths, const int32_t* target_offsets, const calitas_params_t* params,
                                     calitas_aln_t** out, uint64_t* n_out, uint32_t** counts) {
  if (!ctx) return CALITAS_EINVAL;
  if (!guides || !targets || !target_lengths || !params || !out || !n_out || n_tasks < 0) return calitas_fail(ctx, CALITAS_EINVAL, "NULL argument");
  *out = nullptr; *n_out = 0;
  if (counts) *counts = nullptr;
  if (ctx->device < 0) return calitas_fail(ctx, CALITAS_ENODEV, "host-only context: calitas_align_windows needs a GPU (there is no CPU fallback)");
  const calitas_params_t& p = *params;
  if (p.max_gaps_between_guide_and_pam < 0 || p.max_gaps_between_guide_and_pam > 16) return calitas_fail(ctx, CALITAS_EINVAL, "max-gaps-between-guide-and-pam must be 0..16");
  const bool best_mode = p.max_guide_diffs < 0;   // alignBest / alignToRefBest: limits derived from each guide (SGA:339-342, 412-416)
  const Scores sc = derive_scores(p.guide_mismatch_net_cost, p.pam_mismatch_net_cost, p.genome_gap_net_cost, p.guide_gap_net_cost);
  HIP_TRY(ctx, hipSetDevice(ctx->device));

  using clk = std::chrono::steady_clock;
  auto ms_since = [](clk::time_point t) { return std::chrono::duration<double, std::milli>(clk::now() - t).count(); };
  double ms_prep = 0, ms_pack = 0, ms_recs = 0, ms_upload = 0, ms_gpu = 0, ms_post = 0;
  auto t_sec = clk::now();
  std::vector<std::vector<calitas_aln_t>> pieces;             // the kept alignments, block by block in task order (assembled once, at the end)
  std::vector<uint32_t> per_task((size_t)std::max(n_tasks, 0), 0u);
  // guides: consecutive tasks that pass the same guide (the variant branch: one guide, a million windows) share one GuideHost
  std::vector<GuideHost> uniq;
  std::vector<int> gh_of((size_t)n_tasks);
  struct GhView { const std::vector<GuideHost>& u; const std::vector<int>& of; const GuideHost& operator[](int t) const { return u[(size_t)of[(size_t)t]]; } };
  const GhView gh{uniq, gh_of};
  std::vector<int> task_d(n_tasks), task_p(n_tasks), task_D(n_tasks), task_O(n_tasks);
  for (int t = 0; t < n_tasks; t++) {
    const bool same = t > 0 && guides[t].protospacer == guides[t - 1].protospacer && guides[t].pams == guides[t - 1].pams &&
                      guides[t].n_pams == guides[t - 1].n_pams && guides[t].pam_is_5prime == guides[t - 1].pam_is_5prime &&
                      guides[t].cli_length == guides[t - 1].cli_length;
    if (!same) {
      uniq.emplace_back();
      std::string e = make_guide_host(guides[t], uniq.back());
      if (!e.empty()) return calitas_fail(ctx, CALITAS_EINVAL, "task " + std::to_string(t) + ": " + e);
    }
    gh_of[(size_t)t] = (int)uniq.size() - 1;
    if (target_lengths[t] > 60000) return calitas_fail(ctx, CALITAS_EINVAL, "targets longer than 60000 bases are not supported here");
    const int L = (int)gh[t].protospacer.size();
    int pamlen = 0;
    for (auto& pm : gh[t].pams) pamlen = std::max<int>(pamlen, (int)pm.size());
    if (best_mode) { task_d[t] = L; task_p[t] = pamlen; task_D[t] = L + p.max_gaps_between_guide_and_pam + pamlen; task_O[t] = 0; }
    else {
      task_d[t] = p.max_guide_diffs; task_p[t] = p.max_pam_mismatches; task_O[t] = p.max_overlap;
      task_D[t] = p.max_total_diffs >= 0 ? p.max_total_diffs : p.max_guide_diffs + p.max_gaps_between_guide_and_pam + p.max_pam_mismatches;
    }
  }

  ms_prep += ms_since(t_sec);
  // ---- process the tasks in chunks of at most MAX_GUIDES distinct (guide, limits) configurations and 8192 tasks ----
  int t0 = 0;
  while (t0 < n_tasks) {
    t_sec = clk::now();
    std::map<std::string, int> slot_of;
    std::vector<int> slot_task;              // representative task of each slot
    std::vector<int> task_slot;
    int t1 = t0;
    while (t1 < n_tasks && t1 - t0 < 32768) {
      if (t1 > t0 && gh_of[(size_t)t1] == gh_of[(size_t)t1 - 1] && task_d[t1] == task_d[t1 - 1] && task_p[t1] == task_p[t1 - 1]) {
        task_slot.push_back(task_slot.back());     // same guide and limits as the task before: same slot
        t1++;
        continue;
      }
      std::string key = gh[t1].q + "|" + std::to_string(gh[t1].pam5) + "|" + std::to_string(task_d[t1]) + "|" + std::to_string(task_p[t1]);
      for (auto& pm : gh[t1].pams_q) key += "|" + pm;
      auto it = slot_of.find(key);
      if (it == slot_of.end()) {
        if ((int)slot_task.size() == MAX_GUIDES) break;
        it = slot_of.emplace(key, (int)slot_task.size()).first;
        slot_task.push_back(t1);
      }
      task_slot.push_back(it->second);
      t1++;
    }
    const int nt = t1 - t0, ns = (int)slot_task.size();

    std::vector<GuideDev> gd(ns);
    uint32_t slab_bytes = 0;
    for (int s = 0; s < ns; s++) {
      const int t = slot_task[s];
      std::string e = build_guide_dev(gh[t], p, sc, task_d[t], task_p[t], gd[s]);
      if (!e.empty()) return calitas_fail(ctx, CALITAS_EINVAL, "task " + std::to_string(t) + ": " + e);
      gd[s].cli_length = 0;     // SGA.align itself has no length filter (that is SearchReference.scala:536)
      const uint32_t ncols_max = 16 + gd[s].span + 1, stride_max = (ncols_max + 4) & ~3u;
      const uint32_t ntb_max = (ncols_max + p.max_gaps_between_guide_and_pam + MAX_PAM_LEN + 3) & ~3u;
      slab_bytes = std::max<uint32_t>(slab_bytes, (uint32_t)((sizeof(SlabHeader) + ntb_max + gd[s].L * stride_max + 15) & ~15u));
    }

    ms_prep += ms_since(t_sec); t_sec = clk::now();
    // temporary packed reference: one contig per task, back to back (no scan here, so no tile padding or halos)
    std::vector<uint64_t> lens(nt);
    std::vector<const uint8_t*> bases(nt);
    uint32_t W = 16;
    for (int i = 0; i < nt; i++) {
      lens[i] = target_lengths[t0 + i]; bases[i] = targets[t0 + i];
      W = std::max<uint32_t>(W, target_lengths[t0 + i]);
    }
    PackedRef ref;
    WorkerPool* const owners_pool = (ctx->parent ? ctx->parent : ctx)->pool;   // (a child context works on its parent's pool)
    try { pack_targets_dense(ref, nt, lens.data(), bases.data(), owners_pool); }
    catch (std::exception& e) { return calitas_fail(ctx, CALITAS_EINVAL, e.what()); }

    ms_pack += ms_since(t_sec); t_sec = clk::now();
    // every column of every task, both directions
    std::vector<uint64_t> rec_at((size_t)nt + 1, 0);
    for (int i = 0; i < nt; i++) rec_at[(size_t)i + 1] = rec_at[(size_t)i] + 2 * ((lens[i] + 15) / 16);
    if (rec_at[(size_t)nt] > 0xFFFFFFF0ull) return calitas_fail(ctx, CALITAS_EINVAL, "too many target columns for one batch");
    std::vector<ScanRecord> recs((size_t)rec_at[(size_t)nt]);
    WorkerPool one_thread(1);
    (owners_pool && nt >= 4096 ? owners_pool : &one_thread)->for_blocks((size_t)nt, [&](size_t tb, size_t te, int) {
      for (size_t i = tb; i < te; i++) {
        const uint32_t g0 = (uint32_t)(ref.contigs[i].gbase / 16);
        ScanRecord* r = recs.data() + rec_at[i];
        for (uint32_t w = 0; w * 16 < lens[i]; w++) {
          const uint32_t nb = (uint32_t)std::min<uint64_t>(16, lens[i] - (uint64_t)w * 16);
          const uint32_t m = nb == 16 ? 0xFFFFu : ((1u << nb) - 1u);
          *r++ = ScanRecord{g0 + w, m | ((uint32_t)task_slot[i] << 17)};
          *r++ = ScanRecord{g0 + w, m | (1u << 16) | ((uint32_t)task_slot[i] << 17)};
        }
      }
    });
    const uint32_t n_rec = (uint32_t)recs.size();
    const uint32_t slots_per_rec = (W + 14) / W + 1;
    const uint64_t slab_per_rec = (uint64_t)slab_bytes * slots_per_rec;
    {
      int rc = ensure_buffers(ctx, std::max<uint32_t>(n_rec, 1u << 12), std::max<uint32_t>(ctx->raw_cap, std::max<uint32_t>(n_rec * 4, 1u << 16)), slab_per_rec,
                              std::max<uint32_t>(1u << 12, n_rec * slots_per_rec * 16));   // every column of every task can pass
      if (rc) return rc;
    }

    ms_recs += ms_since(t_sec); t_sec = clk::now();
    TempDevice td;
    const size_t nruns = std::max<size_t>(1, ref.runs.size());
    std::vector<uint64_t> wbase(nt + 1);
    std::iota(wbase.begin(), wbase.end(), 0ull);
    std::vector<int2> wins(nt);
    for (int i = 0; i < nt; i++) wins[i] = make_int2(0, (int)lens[i]);
    HIP_TRY(ctx, scratch(ctx, 0, ref.codes.size() * 4, &td.codes));
    HIP_TRY(ctx, scratch(ctx, 1, ref.mask.size() * 4, &td.mask));
    HIP_TRY(ctx, scratch(ctx, 2, nruns * sizeof(Run), &td.runs));
    HIP_TRY(ctx, scratch(ctx, 3, ref.contigs.size() * sizeof(ContigInfo), &td.contigs));
    HIP_TRY(ctx, scratch(ctx, 4, ref.tiles.size() * sizeof(TileInfo), &td.tiles));
    HIP_TRY(ctx, scratch(ctx, 5, wbase.size() * sizeof(uint64_t), &td.win_base));
    HIP_TRY(ctx, scratch(ctx, 6, std::max<size_t>(1, wins.size()) * sizeof(int2), &td.win));
    HIP_TRY(ctx, scratch(ctx, 7, sizeof(GuideDev) * ns, &td.guides));
    HIP_TRY(ctx, hipMemcpyAsync(td.codes, ref.codes.data(), ref.codes.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(td.mask, ref.mask.data(), ref.mask.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    if (!ref.runs.empty()) HIP_TRY(ctx, hipMemcpyAsync(td.runs, ref.runs.data(), ref.runs.size() * sizeof(Run), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(td.contigs, ref.contigs.data(), ref.contigs.size() * sizeof(ContigInfo), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(td.tiles, ref.tiles.data(), ref.tiles.size() * sizeof(TileInfo), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(td.win_base, wbase.data(), wbase.size() * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream));
    if (!wins.empty()) HIP_TRY(ctx, hipMemcpyAsync(td.win, wins.data(), wins.size() * sizeof(int2), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(td.guides, gd.data(), sizeof(GuideDev) * ns, hipMemcpyHostToDevice, ctx->stream));
    if (n_rec) HIP_TRY(ctx, hipMemcpyAsync(ctx->d_recs, recs.data(), (size_t)n_rec * sizeof(ScanRecord), hipMemcpyHostToDevice, ctx->stream));

    ms_upload += ms_since(t_sec); t_sec = clk::now();
    uint32_t n_raw = 0;
    for (;;) {
      uint32_t zero[8] = {n_rec, 0, 0, 0, 0, 0, 0, 0};
      HIP_TRY(ctx, hipMemcpyAsync(ctx->d_counters, zero, sizeof(zero), hipMemcpyHostToDevice, ctx->stream));
      AlignArgs aa{};
      aa.codes = td.codes; aa.mask = td.mask; aa.runs = td.runs; aa.n_runs = (int64_t)ref.runs.size();
      aa.contigs = td.contigs; aa.tiles = td.tiles; aa.win_base = td.win_base; aa.win = td.win; aa.guides = td.guides; aa.recs = ctx->d_recs;
      aa.rec_count = ctx->d_counters; aa.out = ctx->d_raw; aa.out_count = ctx->d_counters + 1; aa.anomalies = ctx->d_counters + 2; aa.trace_done = ctx->d_counters + 5; aa.job_count = ctx->d_counters + 6;
      aa.rec_capacity = ctx->rec_cap; aa.out_capacity = ctx->raw_cap; aa.tile_words = (uint32_t)(ref.tile / 16);
      aa.slab = ctx->d_slab; aa.cand_count = ctx->d_counters + 4; aa.slab_bytes = slab_bytes; aa.slots_per_rec = slots_per_rec; aa.gw_lo = 0; aa.gw_hi = ~0ull;
      aa.items = ctx->d_items; aa.item_count = ctx->d_counters + 3; aa.item_capacity = ctx->item_cap;
      aa.sp.window_size = (int)W; aa.sp.step = (int)W; aa.sp.n_guides = ns;
      aa.sp.max_guide_diffs = 0; aa.sp.max_pam_mismatches = 0; aa.sp.max_diffs_filtering = 0;   // per-guide values live in GuideDev
      aa.sp.max_gaps = p.max_gaps_between_guide_and_pam;
      aa.sp.match = sc.match; aa.sp.mismatch = sc.mismatch; aa.sp.pam_match = sc.pam_match; aa.sp.pam_mismatch = sc.pam_mismatch;
      aa.sp.query_gap = sc.query_gap; aa.sp.target_gap = sc.target_gap; aa.sp.eqx_by_score = p.eqx_by_score & 1; aa.sp.per_matrix = (p.eqx_by_score >> 1) & 1; aa.sp.chrom_index = -1;
      HIP_TRY(ctx, launch_align(aa, 1024, ctx->stream));
      HIP_TRY(ctx, launch_trace(aa, 2048, ctx->stream));
      HIP_TRY(ctx, hipMemcpyAsync(ctx->h_counters, ctx->d_counters, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      if (ctx->h_counters[2] != 0) return calitas_fail(ctx, CALITAS_EHIP, "aligner kernel reported an inconsistent traceback (internal error)");
      n_raw = ctx->h_counters[1];
      if (n_raw > ctx->raw_cap) {
        int rc = ensure_buffers(ctx, ctx->rec_cap, n_raw + n_raw / 8, slab_per_rec, ctx->item_cap);
        if (rc) return rc;
        continue;
      }
      break;
    }
    std::vector<RawAln> raw(n_raw);
    if (n_raw) HIP_TRY(ctx, hipMemcpy(raw.data(), ctx->d_raw, (size_t)n_raw * sizeof(RawAln), hipMemcpyDeviceToHost));

    ms_gpu += ms_since(t_sec); t_sec = clk::now();
    // ---- host: enumeration order per task, conversion, SGA:315-320 ----
    // The records are grouped by task (a counting sort: a record's contig is its task's place in this chunk); the workers then take
    // consecutive blocks of tasks -- order within the task, conversion, filter -- and their results are joined in block order.
    auto list_of = [&](const RawAln& r) { const bool pam5 = gh[t0 + (int)r.contig].pam5; return pam5 ? (r.dir == 1 ? 0 : 1) : (r.dir == 0 ? 0 : 1); };
    std::vector<uint32_t> first((size_t)nt + 1, 0);
    for (uint32_t i = 0; i < n_raw; i++) first[(size_t)raw[i].contig + 1]++;
    for (int c = 0; c < nt; c++) first[(size_t)c + 1] += first[(size_t)c];
    std::vector<uint32_t> order(n_raw);
    {
      std::vector<uint32_t> cur(first.begin(), first.end() - 1);
      for (uint32_t i = 0; i < n_raw; i++) order[cur[raw[i].contig]++] = i;
    }
    WorkerPool serial(1);
    WorkerPool* const owners = (ctx->parent ? ctx->parent : ctx)->pool;   // (a child context works on its parent's pool)
    WorkerPool* pool = (owners && n_raw >= 4096) ? owners : &serial;
    std::vector<std::vector<calitas_aln_t>> part((size_t)pool->size());
    pool->for_blocks((size_t)nt, [&](size_t cb, size_t ce, int tid) {
      std::vector<calitas_aln_t> win;
      std::vector<int> kept;
      auto& mine = part[(size_t)tid];
      for (size_t c = cb; c < ce; c++) {
        const uint32_t i = first[c], j = first[c + 1];
        if (i == j) continue;
        std::sort(order.begin() + i, order.begin() + j, [&](uint32_t x, uint32_t y) {
          const RawAln &a = raw[x], &b = raw[y];
          const int la = list_of(a), lb = list_of(b);
          if (la != lb) return la < lb;
          if (a.t_end_guide != b.t_end_guide) return a.t_end_guide < b.t_end_guide;
          if (a.pad != b.pad) return a.pad < b.pad;               // per-matrix enumeration: Diag, Left, Up
          if (a.pam != b.pam) return a.pam < b.pam;
          return x < y;
        });
        const int t = t0 + (int)c;
        const int64_t off = target_offsets ? target_offsets[t] : 0;
        win.resize(j - i);
        for (uint32_t k = i; k < j; k++) {
          raw_to_aln(raw[order[k]], gh[t], off, off + (int64_t)lens[c], win[k - i]);
          win[k - i].contig_index = t;       // task index
          win[k - i].guide_index = t;
        }
        window_filter(win.data(), (int)win.size(), task_D[t], task_O[t], kept);
        for (int k : kept) mine.push_back(win[k]);
        per_task[t] = (uint32_t)kept.size();
      }
    });
    for (auto& v : part) if (!v.empty()) pieces.emplace_back(std::move(v));
    t0 = t1;
    ms_post += ms_since(t_sec);
  }
  if (TUNE_GET("CALITAS_TRACE") && n_tasks >= 1024)
    std::fprintf(stderr, "[calitas] align_windows: %d tasks: guides %.1f ms, pack %.1f ms, records %.1f ms, alloc+upload %.1f ms, kernels+copy %.1f ms, filter %.1f ms\n",
                 n_tasks, ms_prep, ms_pack, ms_recs, ms_upload, ms_gpu, ms_post);

  std::vector<size_t> piece_at(pieces.size() + 1, 0);
  for (size_t k = 0; k < pieces.size(); k++) piece_at[k + 1] = piece_at[k] + pieces[k].size();
  const size_t n_result = piece_at[pieces.size()];
  *n_out = n_result;
  *out = (calitas_aln_t*)calitas_out_alloc(std::max<size_t>(1, n_result) * sizeof(calitas_aln_t));
  if (!*out) return calitas_fail(ctx, CALITAS_EINVAL, "out of memory");
  {
    // (one copy, by the pool: a batch of the variant branch returns 7 MB of records, and three single-threaded copies of them were
    // 4 of the call's 14 ms)
    WorkerPool one_thread(1);
    WorkerPool* const owners_pool = (ctx->parent ? ctx->parent : ctx)->pool;
    calitas_aln_t* const dst = *out;
    (owners_pool && n_result >= 4096 ? owners_pool : &one_thread)->for_blocks(pieces.size(), [&](size_t b, size_t e, int) {
      for (size_t k = b; k < e; k++) std::memcpy(dst + piece_at[k], pieces[k].data(), pieces[k].size() * sizeof(calitas_aln_t));
    });
  }
  if (counts) {
    *counts = (uint32_t*)calitas_out_alloc(std::max<size_t>(1, per_task.size()) * sizeof(uint32_t));
    if (!*counts) return calitas_fail(ctx, CALITAS_EINVAL, "out of memory");
    if (!per_task.empty()) std::memcpy(*counts, per_task.data(), per_task.size() * sizeof(uint32_t));
  }
  return CALITAS_OK;
}

// Padded strings of an alignment returned by calitas_align_windows, built from the caller's own target bytes (case kept,
// as Alignment.paddedString does at SequentialGuideAligner.scala:511; '-' strand: Sequences.revcomp of the window).
extern "C" int calitas_padded_strings_target(const calitas_guide_t* guide, const calitas_aln_t* aln, const uint8_t* target, uint32_t target_length,
                                             int32_t target_offset, char* padded_guide, char* padded_alignment, char* padded_target) {
  if (!guide || !aln || !target || !padded_guide || !padded_alignment || !padded_target) return CALITAS_EINVAL;
  GuideHost gh;
  if (!make_guide_host(*guide, gh).empty()) return CALITAS_EINVAL;
  const int64_t s = (int64_t)aln->start_offset - target_offset, e = (int64_t)aln->end_offset - target_offset;
  if (s < 0 || e > (int64_t)target_length || s > e) return CALITAS_EINVAL;
  std::string t(reinterpret_cast<const char*>(target) + s, (size_t)(e - s));
  if (aln->strand == '-') t = revcomp_str(t);
  const std::string q = gh.query_for(aln->pam_index);
  size_t qi = 0, ti = 0;
  int n = aln->n_ops;
  for (int i = 0; i < n; i++) {
    switch (aln->ops[i]) {
      case 'I': padded_guide[i] = q[qi++]; padded_alignment[i] = '~'; padded_target[i] = '-'; break;
      case 'D': padded_guide[i] = '-'; padded_alignment[i] = '~'; padded_target[i] = t[ti++]; break;
      case '=': padded_guide[i] = q[qi++]; padded_alignment[i] = '|'; padded_target[i] = t[ti++]; break;
      default:  padded_guide[i] = q[qi++]; padded_alignment[i] = '.'; padded_target[i] = t[ti++]; break;
    }
  }
  padded_guide[n] = padded_alignment[n] = padded_target[n] = 0;
  return CALITAS_OK;
}
