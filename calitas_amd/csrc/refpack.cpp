// refpack.cpp -- ASCII contigs -> 2-bit codes + exception mask + run table + scan-tile table (layout: common.hpp).
#include "refpack.hpp"
#include "tuning.hpp"
#include "parallel.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <thread>

namespace calitas {

namespace {

struct ByteClass {
  uint8_t code[256];
  uint8_t exc[256];
  ByteClass() {
    for (int b = 0; b < 256; b++) {
      exc[b] = 1;
      int m = iupac_mask((unsigned char)b);
      bool letter = (b >= 'A' && b <= 'Z') || (b >= 'a' && b <= 'z');
      // exception code 1 = IUPAC ambiguity other than N (also U/u, so the original letter survives a round trip)
      code[b] = (letter && m != 0 && (b & 0xDF) != 'N') ? 1 : 0;
    }
    const char* plain = "ACGTacgt";
    for (int i = 0; i < 8; i++) { exc[(unsigned char)plain[i]] = 0; code[(unsigned char)plain[i]] = (uint8_t)(i & 3); }
  }
};
const ByteClass kClass;

int pick_chunk(uint64_t total_bases) {
  // Aim for >= 2048 tiles so 256 CUs see several waves of workgroups; a lane re-scans 32 warm-up bases per direction,
  // so larger chunks waste less (32/chunk) but give fewer workgroups -- and a longer chain per lane costs registers: at 512 bases
  // (17-word chains) scan_rows_kernel holds three waves per SIMD, at 256 (9 words, 89 VGPRs) five, which more than pays for the
  // 6 % of extra warm-up (1.286 against 1.314 ms per hg38 pass; CALITAS_CHUNK=512 still selects the longer lane).
  int chunk = 256;
  if (const char* e = TUNE_GET("CALITAS_CHUNK")) { int v = std::atoi(e); if (v == 64 || v == 128 || v == 256 || v == 512) return v; }
  while (chunk > 64 && total_bases / ((uint64_t)chunk * LANES_PER_TILE) < 2048) chunk >>= 1;
  return chunk;
}

}  // namespace

const Run* PackedRef::run_at(uint64_t gpos) const {
  int64_t r = run_floor(runs.data(), (int64_t)runs.size(), gpos);
  if (r >= 0 && gpos < runs[r].start + runs[r].len) return &runs[r];
  return nullptr;
}

char PackedRef::base_upper(uint64_t gpos) const {
  if ((mask[gpos >> 5] >> (gpos & 31)) & 1) {
    const Run* r = run_at(gpos);
    if (!r || r->ch == 0) return 'N';
    return (char)((r->ch >= 'a' && r->ch <= 'z') ? r->ch - 32 : r->ch);
  }
  return "ACGT"[(codes[gpos >> 4] >> ((gpos & 15) * 2)) & 3];
}

// n bases at src -> code words cw / mask words mw of packed position g0 (a multiple of 32) and the exception runs among them.
static void pack_span(const uint8_t* src, uint64_t n, uint64_t g0, uint32_t* cw, uint32_t* mw, std::vector<Run>& runs) {
  int runCh = -1; uint64_t runStart = 0;
  for (uint64_t w = 0; w * 32 < n; w++) {
    uint32_t m = 0, c0 = 0, c1 = 0;
    uint64_t lim = std::min<uint64_t>(32, n - w * 32);
    for (uint64_t k = 0; k < lim; k++) {
      uint8_t b = src[w * 32 + k];
      uint32_t code = kClass.code[b];
      uint32_t e = kClass.exc[b];
      m |= e << k;
      if (k < 16) c0 |= code << (2 * k); else c1 |= code << (2 * (k - 16));
      if (e) {
        if (runCh != (int)b) {
          if (runCh >= 0) runs.push_back(Run{g0 + runStart, (uint32_t)(w * 32 + k - runStart), (uint8_t)runCh, {0, 0, 0}});
          runCh = b; runStart = w * 32 + k;
        }
      } else if (runCh >= 0) {
        runs.push_back(Run{g0 + runStart, (uint32_t)(w * 32 + k - runStart), (uint8_t)runCh, {0, 0, 0}});
        runCh = -1;
      }
    }
    if (lim < 32) m |= 0xFFFFFFFFu << lim;  // beyond the contig end: padding
    mw[w] = m; cw[2 * w] = c0; cw[2 * w + 1] = c1;
  }
  if (runCh >= 0) runs.push_back(Run{g0 + runStart, (uint32_t)(n - runStart), (uint8_t)runCh, {0, 0, 0}});
}

// Explicit targets (calitas_align_windows) packed back to back, each starting on a 32-base boundary: the align / trace kernels read
// inside a target only, so none of the tile padding and halos the scan kernel needs.  "Tiles" are 32 bases (tile -> target lookup).
void pack_targets_dense(PackedRef& out, int n, const uint64_t* lengths, const uint8_t* const* bases, WorkerPool* pool) {
  out = PackedRef();
  out.genome_build = "windows";
  out.chunk = 32; out.tile = 32;
  uint64_t g = 0, total = 0;
  out.contigs.resize((size_t)n);
  for (int i = 0; i < n; i++) {
    out.contigs[i].gbase = g; out.contigs[i].len = lengths[i];
    g += (lengths[i] + 31) / 32 * 32;
    total += lengths[i];
  }
  if (g == 0) g = 32;
  if (g / 16 > 0xFFFFFFFFull) throw std::invalid_argument("targets too large for 32-bit packed word indices");
  out.total_bases = total; out.total_packed = g;
  out.codes.assign(g / 16, 0u);
  out.mask.assign(g / 32, 0xFFFFFFFFu);
  out.tiles.assign(g / 32, TileInfo{0xFFFFFFFFu, 1u});
  // every target starts on a 32-base boundary: no code or mask word is shared, blocks of targets can be packed side by side; the runs
  // of exception bases of a block are in position order, and so are the blocks
  auto pack_block = [&](size_t b, size_t e, std::vector<Run>& runs) {
    for (size_t i = b; i < e; i++) {
      const uint64_t g0 = out.contigs[i].gbase;
      pack_span(bases[i], lengths[i], g0, out.codes.data() + g0 / 16, out.mask.data() + g0 / 32, runs);
      for (uint64_t t = g0 / 32; t < (g0 + lengths[i] + 31) / 32; t++) out.tiles[t].contig = (uint32_t)i;
    }
  };
  if (!pool || pool->size() < 2 || n < 4096) { pack_block(0, (size_t)n, out.runs); return; }
  std::vector<std::vector<Run>> runs((size_t)pool->size());
  pool->for_blocks((size_t)n, [&](size_t b, size_t e, int tid) { pack_block(b, e, runs[(size_t)tid]); });
  for (auto& r : runs) out.runs.insert(out.runs.end(), r.begin(), r.end());
}

void pack_reference(PackedRef& out, int n_contigs, const char* const* names, const uint64_t* lengths,
                    const uint8_t* const* bases, const char* genome_build, int threads) {
  out = PackedRef();
  if (genome_build && *genome_build) out.genome_build = genome_build;
  uint64_t total = 0;
  for (int i = 0; i < n_contigs; i++) total += lengths[i];
  out.total_bases = total;
  out.chunk = pick_chunk(total);
  out.tile = (uint64_t)out.chunk * LANES_PER_TILE;
  const uint64_t T = out.tile;

  // bases[i] == nullptr: contig i is absent (refpack.hpp): name and length as given, one dead tile of packed space
  bool any_absent = false;
  for (int i = 0; i < n_contigs; i++) any_absent = any_absent || (bases[i] == nullptr && lengths[i] != 0);
  if (any_absent) { out.absent.assign((size_t)n_contigs, 0); for (int i = 0; i < n_contigs; i++) out.absent[(size_t)i] = bases[i] == nullptr && lengths[i] != 0; }
  uint64_t g = T;  // tile 0 is padding: the left halo of the first real tile reads it
  for (int i = 0; i < n_contigs; i++) {
    out.names.emplace_back(names[i]);
    ContigInfo ci;
    ci.gbase = g; ci.len = lengths[i];
    out.contigs.push_back(ci);
    uint64_t padded = ((lengths[i] + (uint64_t)out.chunk + T - 1) / T) * T;  // >= one chunk of padding after the contig
    if (padded == 0 || out.is_absent((size_t)i)) padded = T;
    g += padded;
  }
  g += T;  // trailing padding tile: right halo of the last real tile
  out.total_packed = g;
  if (g / 16 > 0xFFFFFFFFull) throw std::invalid_argument("reference too large for 32-bit packed word indices");
  out.codes.assign(g / 16, 0u);
  out.mask.assign(g / 32, 0xFFFFFFFFu);  // padding = exception, code 0
  out.tiles.assign(g / T, TileInfo{0xFFFFFFFFu, 2u});

  // Segment the work: each segment is a 32-base aligned slice of one contig (so mask/code words are never shared).
  struct Seg { int contig; uint64_t off, len; std::vector<Run> runs; };
  std::vector<Seg> segs;
  const uint64_t SEG = 1ull << 22;
  for (int i = 0; i < n_contigs; i++)
    for (uint64_t off = 0; off < lengths[i] && !out.is_absent((size_t)i); off += SEG) segs.push_back(Seg{i, off, std::min(SEG, lengths[i] - off), {}});

  int nt = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
  if (nt < 1) nt = 1;
  if (nt > 64) nt = 64;
  if ((size_t)nt > segs.size()) nt = (int)std::max<size_t>(1, segs.size());

  auto work = [&](int tid) {
    for (size_t si = tid; si < segs.size(); si += nt) {
      Seg& s = segs[si];
      const uint64_t g0 = out.contigs[s.contig].gbase + s.off;  // multiple of 32 (gbase multiple of tile, off of 2^22)
      pack_span(bases[s.contig] + s.off, s.len, g0, out.codes.data() + g0 / 16, out.mask.data() + g0 / 32, s.runs);
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < nt; t++) pool.emplace_back(work, t);
  work(0);
  for (auto& t : pool) t.join();

  // Merge per-segment runs (segments are in packed order); join runs of the same byte that touch across a boundary.
  for (auto& s : segs) {
    for (auto& r : s.runs) {
      if (!out.runs.empty()) {
        Run& p = out.runs.back();
        if (p.ch == r.ch && p.start + p.len == r.start && (uint64_t)p.len + r.len <= 0xFFFFFFFFull &&
            r.start != out.contigs[s.contig].gbase) {
          p.len += r.len;
          continue;
        }
      }
      out.runs.push_back(r);
    }
    std::vector<Run>().swap(s.runs);
  }

  // Tile table.  A tile is "dead" (flag 2) when it and both halo chunks hold nothing but upper-case N / padding.
  const uint64_t ntiles = out.total_packed / T;
  for (int i = 0; i < n_contigs; i++) {
    uint64_t t0 = out.contigs[i].gbase / T;
    uint64_t nt_c = ((lengths[i] + (uint64_t)out.chunk + T - 1) / T);
    if (nt_c == 0 || out.is_absent((size_t)i)) nt_c = 1;
    for (uint64_t t = 0; t < nt_c; t++) out.tiles[t0 + t].contig = (uint32_t)i;
  }
  // Coverage of [lo, hi) by exception bases / by dead bases (N or padding), from the mask words and the run table.
  auto classify = [&](uint64_t lo, uint64_t hi, bool& anyExc, bool& allDead) {
    anyExc = false; allDead = true;
    for (uint64_t w = lo / 32; w < hi / 32; w++) {
      uint32_t m = out.mask[w];
      if (m) anyExc = true;
      if (m != 0xFFFFFFFFu) allDead = false;
    }
    if (allDead) {
      // all exception bases: dead only if every run inside is 'N' (padding has no run entry)
      int64_t r = run_floor(out.runs.data(), (int64_t)out.runs.size(), lo);
      if (r < 0) r = 0;
      for (; r < (int64_t)out.runs.size() && out.runs[r].start < hi; r++) {
        const Run& rr = out.runs[r];
        if (rr.start + rr.len <= lo) continue;
        if (rr.ch != 'N') { allDead = false; break; }
      }
    }
  };
  for (uint64_t t = 0; t < ntiles; t++) {
    uint64_t lo = t * T, hi = lo + T;
    uint64_t hlo = lo >= (uint64_t)out.chunk ? lo - out.chunk : lo;
    uint64_t hhi = std::min(out.total_packed, hi + out.chunk);
    bool anyExc, allDead;
    classify(hlo, hhi, anyExc, allDead);
    out.tiles[t].flag = allDead ? 2u : (anyExc ? 1u : 0u);
    if (out.tiles[t].flag == 1u && out.tiles[t].contig != 0xFFFFFFFFu) out.masked_tiles.push_back((uint32_t)t);
  }
}

}  // namespace calitas
