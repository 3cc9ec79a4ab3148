// variants.cpp -- the variant branch of SearchReference.execute (SearchReference.scala:101-400, 570-630) behind one C entry point.
//
// calitas_search_variants = the reference hits of calitas_search + the hits of every variant window, merged by removeOverlaps /
// ReferenceHit.sort (calitas_hits_tsv_ext).  Variant windows are produced on the host exactly as variantWindowIterator does
// (nextChunk / reChunk SR:326-347, alleleCombos SR:351-399, buildVariantWindow SR:263-323), aligned on the GPU through the
// calitas_align_windows path in batches, lifted back with refOffsetAtBaseOffset (SR:133-156) and turned into rows with window-local
// flanks (SR:598-613) and the variant columns (RH:211-233).  calitas_amd/variants.py holds the same logic in Python (the parity
// tests run both); this file exists because BASELINE config 5 has three million variants.
//
// RESTATEMENT, NOT DESIGN: four host functions below follow the reference statement by statement, because what they compute IS the
// contract -- the order in which allele combinations are enumerated decides the order of the variant windows (and so SR:622's arrival
// order of their hits), and the shape of a window's CIGAR decides every lifted coordinate:
//   ref_offset_at          = VariantWindow.refOffsetAtBaseOffset   SearchReference.scala:133-156
//   is_valid               = VariantSet.isValid                     SearchReference.scala:182-193
//   build_window           = buildVariantWindow                     SearchReference.scala:263-323  (windowStart / windowEnd, the right-to-left
//                            patch, refPos / baseOffset / precedingMatch, the M / I / D case split, the same `require`)
//   allele_combos_counts   = alleleCombos(Seq[Int])                 SearchReference.scala:377-399  (denominators, group size, (allele + 1) % n)
// They are pinned by the reference's own vectors V1-V9 (SearchReferenceTest.scala:150-295) through tests/test_variants_host.py.
// Everything around them -- arenas, the VCF reader, the pipeline of stages, keys and rows, the merge on the device -- has no
// counterpart in the reference.
//
// THE THREADS OF ONE CALL (round 5), and what each of them owns.  A stage is a thread that runs jobs in the order they are handed
// over (StageThread: two jobs waiting at most, a failed stage drops what is behind it but still pays its turns); all of them share the
// context's worker pool for the parallel part of a job.
//   caller       walks the VCF's records as they are published (VarTable::have), lists what every window is made of (Spec), hands a
//                full list to the builder and, at a contig's end, the contig's "finish" behind its last batch
//   vcf reader   maps the file, parses it in waves of 16 MB on the pool, publishes the records wave by wave
//   md5          the VCF's identifier "name:md5" (RH:175-183); whoever needs it first joins it (need_vid)
//   builder      build_window for a list (pool), the batch to one of the aligners
//   aligner x2   calitas_align_windows of a batch on a side context each (device); the batches reach the lifter in the order they were built
//   lifter       lifts a batch's alignments back, lists them as hits (HitList: pieces that never move); at a contig's end the groups' own
//                walks, the entries' tie order and keys (finish_contig)
//   finisher     the rows of the contig's placed entries (with a placeholder where the identifier goes), then publishes the contig
//   helper       the reference's per-contig passes (search.cpp, calitas_search_hits_ext_impl): asks for a contig's entries when its row stage
//                is due (HitsExtSource::get, blocks until published), makes the rows of the plain entries the device's walk kept
//                (HitsExt::rows_for), and its copying thread hands every contig's text to ...
//   filler       ... which waits for the identifier once and writes the kept entries' rows into the holes the rows kernel left
//                (HitsExt::fill; on the copying thread itself when the text is in a block of the library's, which may still move)
// cx[c] (ContigExt) is the lifter's until finish_contig(c) returns, the finisher's until it publishes c, then the helper's and the
// filler's; hits[] grows on the lifter only, everybody else reads published contigs through the pointers in cx[c].entry.
//
// VCF support is the subset the reference's path needs (fgbio vcf.api): CHROM POS ID REF ALT FILTER INFO(AF, END); plain or gzip.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <string>
#include <vector>

#include "ctx.hpp"
#include "tuning.hpp"

namespace {

// One element in place, more on the heap: a VCF record has one ALT and one AF value nearly always, and three million records with two
// small heap blocks each were six million allocations per call to make -- and to hand back.
template <typename T>
struct Few {
  T first{};
  std::vector<T> rest;
  uint32_t n = 0;
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
  const T& operator[](size_t i) const { return i == 0 ? first : rest[i - 1]; }
  template <class... A>
  void emplace_back(A&&... a) { if (n == 0) first = T(std::forward<A>(a)...); else rest.emplace_back(std::forward<A>(a)...); n++; }
  void push_back(const T& v) { emplace_back(v); }
  void clear() { n = 0; rest.clear(); }
};

struct Var {
  std::string chrom, id, ref;
  int pos = 0, end = 0;                    // 1-based; fgbio Variant.end
  Few<std::string> alts;
  Few<float> afs;
};

struct Allele {                            // VariantAllele SR:105-110
  const Var* v;
  int alt;                                 // index into v->alts
  float af;
};

struct CigarEl { char op; int n; };

// VariantWindow SR:118-157.  A view: alleles, cigar and bases live in the arena of the worker that built the window (three million
// windows with three small heap blocks each cost more to allocate and free than to align).
struct Window {
  int contig = 0, start = 0;               // start: 1-based reference position of the first base
  uint32_t chunk = 0;                      // serial number of the nextChunk() cluster it came from: windows of two chunks share no variant
  const Allele* variants = nullptr; int nv = 0;
  const CigarEl* cigar = nullptr; int nc = 0;
  const char* bases = nullptr; int len = 0;
};
struct Arena { std::vector<char> bases; std::vector<Allele> alleles; std::vector<CigarEl> cigars; };
struct ArenaMark { size_t bases, alleles, cigars; };   // where a window's pieces start in its arena (pointers are set once the arena is complete)

// strtod of p[0..n) for the numbers a VCF's AF holds.  Plain decimals of at most 15 significant digits and 22 decimal places are an
// integer below 2^53 divided by a power of ten that a double holds exactly: one correctly rounded division, the very double strtod
// returns (Clinger's fast path).  Everything else -- exponents, longer digit strings, inf / nan, blanks -- goes to strtod itself.
double parse_decimal(const char* p, size_t n) {
  static const double kPow10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
  size_t i = 0;
  bool neg = false;
  if (i < n && (p[i] == '-' || p[i] == '+')) { neg = p[i] == '-'; i++; }
  uint64_t m = 0;
  int digits = 0, frac = 0;
  bool dot = false, any = false, simple = true;
  for (; i < n; i++) {
    const char c = p[i];
    if (c >= '0' && c <= '9') {
      any = true;
      if (m != 0 || c != '0') digits++;
      if (digits > 15) { simple = false; break; }
      m = m * 10 + (uint64_t)(c - '0');
      if (dot) frac++;
    } else if (c == '.' && !dot) dot = true;
    else { simple = false; break; }
  }
  if (simple && any && frac <= 22) {
    const double v = (double)m / kPow10[frac];
    return neg ? -v : v;
  }
  char num[64];
  const size_t cl = std::min(n, sizeof(num) - 1);
  std::memcpy(num, p, cl); num[cl] = 0;
  return std::strtod(num, nullptr);
}

// One VCF record (a line without its newline) -> v; false for headers, short lines and other chromosomes (read_vcf of variants.py).
bool parse_record(const char* b, const char* e, const char* chrom, size_t chrom_len, Var& v) {
  if (b >= e || *b == '#') return false;
  // fields 0-4 and 7 (CHROM POS ID REF ALT . . INFO), located in place
  const char* f0[9]; size_t fl[9]; int nf = 0;
  while (nf < 9) {
    const char* t = (const char*)std::memchr(b, '\t', (size_t)(e - b));
    f0[nf] = b; fl[nf] = (size_t)((t ? t : e) - b); nf++;
    if (!t) break;
    b = t + 1;
  }
  if (nf < 5 || (chrom && (fl[0] != chrom_len || std::memcmp(f0[0], chrom, fl[0]) != 0))) return false;
  // (everything in place: three million records per call at full size, and a temporary string per field -- the INFO column's entries
  // above all -- was most of the quarter second the file took)
  auto to_int = [](const char* p, size_t n) -> int {            // atoi of p[0..n): blanks, a sign, digits
    size_t i = 0;
    while (i < n && (p[i] == ' ' || (p[i] >= '\t' && p[i] <= '\r'))) i++;
    bool neg = false;
    if (i < n && (p[i] == '-' || p[i] == '+')) { neg = p[i] == '-'; i++; }
    long v = 0;
    while (i < n && p[i] >= '0' && p[i] <= '9') { v = v * 10 + (p[i] - '0'); i++; }
    return (int)(neg ? -v : v);
  };
  v.chrom.assign(f0[0], fl[0]);
  v.pos = to_int(f0[1], fl[1]);
  if (!(fl[2] == 1 && f0[2][0] == '.')) v.id.assign(f0[2], fl[2]);
  v.ref.assign(f0[3], fl[3]);
  {
    const char* a0 = f0[4];
    const char* const ae = f0[4] + fl[4];
    for (;;) {                                                  // split(ALT, ','): an empty ALT is one empty allele
      const char* c = (const char*)std::memchr(a0, ',', (size_t)(ae - a0));
      v.alts.emplace_back(a0, (size_t)((c ? c : ae) - a0));
      if (!c) break;
      a0 = c + 1;
    }
  }
  bool have_end = false;
  if (nf > 7) {
    const char* k0 = f0[7];
    const char* const ie = f0[7] + fl[7];
    for (;;) {                                                  // the INFO column's entries, ';' between them
      const char* sc = (const char*)std::memchr(k0, ';', (size_t)(ie - k0));
      const char* const ke = sc ? sc : ie;
      const size_t kl = (size_t)(ke - k0);
      if (kl >= 3 && std::memcmp(k0, "AF=", 3) == 0) {
        v.afs.clear();
        const char* x0 = k0 + 3;
        for (;;) {                                              // values between commas; "." and nothing are no value
          const char* c = (const char*)std::memchr(x0, ',', (size_t)(ke - x0));
          const char* const xe = c ? c : ke;
          const size_t xl = (size_t)(xe - x0);
          if (xl != 0 && !(xl == 1 && x0[0] == '.')) v.afs.push_back((float)parse_decimal(x0, xl));
          if (!c) break;
          x0 = c + 1;
        }
      } else if (kl >= 4 && std::memcmp(k0, "END=", 4) == 0) {
        v.end = to_int(k0 + 4, kl - 4); have_end = true;
      }
      if (!sc) break;
      k0 = sc + 1;
    }
  }
  if (!have_end) v.end = v.pos + (int)v.ref.size() - 1;
  return true;
}

// The whole file in memory (gzip through zlib), then the lines parsed on the worker pool: every worker takes the lines that
// start in its byte range, and the per-worker lists are joined in file order.
// The records of a VCF in file order.  They stay in the blocks the workers parsed them into (one contiguous table of three million
// records is 460 MB touched for the first time by ONE thread: 0.18 of the file's 0.3 s); at[i] finds record i.
struct VarTable {
  std::vector<std::vector<Var>> parts;
  std::vector<Var*> at;                    // room for every line of the file; at[0, ready) are there
  // The file is parsed in waves (read_vcf) while the caller already walks the records of the waves before: have(i) waits until record i
  // is there or the file is done.  (Behind a pointer: the table itself moves -- into the call's garbage, at the end.)
  struct Sync { std::mutex mu; std::condition_variable cv; std::atomic<size_t> ready{0}; std::atomic<bool> done{false}; };
  std::unique_ptr<Sync> sync{new Sync()};
  size_t seen = 0;                         // (the consumer's copy of ready: no atomic load per record)
  size_t size() const { return sync->ready.load(std::memory_order_acquire); }
  Var& operator[](size_t i) { return *at[i]; }
  const Var& operator[](size_t i) const { return *at[i]; }
  bool have(size_t i) {
    if (i < seen) return true;
    seen = sync->ready.load(std::memory_order_acquire);
    if (i < seen) return true;
    std::unique_lock<std::mutex> lk(sync->mu);
    sync->cv.wait(lk, [&] { return i < sync->ready.load(std::memory_order_acquire) || sync->done.load(std::memory_order_acquire); });
    seen = sync->ready.load(std::memory_order_acquire);
    return i < seen;
  }
  void publish(size_t ready, bool done) {
    { std::lock_guard<std::mutex> lk(sync->mu); sync->ready.store(ready, std::memory_order_release); if (done) sync->done.store(true, std::memory_order_release); }
    sync->cv.notify_all();
  }
};

std::string read_vcf(const char* path, const char* chrom, calitas::WorkerPool* pool, VarTable& out) {
  const auto t_read = std::chrono::steady_clock::now();
  auto ms_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
  std::string data;
  bool plain = false;
  // a plain file is mapped and parsed where the page cache has it (reading it into a block of the call's own was 37 ms of one thread
  // per 127 MB before the first record was looked at; zlib's transparent mode copies at ~1 GB/s)
  struct Mapping { void* p = MAP_FAILED; size_t n = 0; ~Mapping() { if (p != MAP_FAILED) (void)munmap(p, n); } } map;
  {
    const int fd = ::open(path, O_RDONLY);
    if (fd >= 0) {
      unsigned char magic[2] = {0, 0};
      struct stat st{};
      if (::fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 2 && ::pread(fd, magic, 2, 0) == 2 && !(magic[0] == 0x1f && magic[1] == 0x8b)) {
        map.p = ::mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (map.p != MAP_FAILED) { map.n = (size_t)st.st_size; plain = true; (void)::madvise(map.p, map.n, MADV_WILLNEED); }
      }
      ::close(fd);
    }
  }
  if (!plain) {
    gzFile f = gzopen(path, "rb");           // transparent for plain text
    if (!f) return std::string("cannot read ") + path;
    gzbuffer(f, 1 << 20);
    std::vector<char> buf(8u << 20);
    for (;;) {
      const int got = gzread(f, buf.data(), (unsigned)buf.size());
      if (got < 0) { gzclose(f); return std::string("cannot read ") + path; }
      if (got == 0) break;
      data.append(buf.data(), (size_t)got);
    }
    gzclose(f);
  }
  const double ms_read = ms_since(t_read);
  const auto t_parse = std::chrono::steady_clock::now();
  const char* const text = plain ? (const char*)map.p : data.data();
  const size_t n = plain ? map.n : data.size(), chrom_len = chrom ? std::strlen(chrom) : 0;
  const size_t T = (size_t)pool->size();
  // room for a pointer per line, so that the table never moves while the caller reads it
  {
    std::vector<size_t> lines(T, 0);
    pool->for_blocks(n, [&](size_t b, size_t e, int tid) {
      size_t c = 0;
      for (const char* p = text + b; p < text + e;) {
        const char* nl = (const char*)std::memchr(p, '\n', (size_t)(text + e - p));
        if (!nl) break;
        c++; p = nl + 1;
      }
      lines[(size_t)tid] += c;
    });
    size_t total_lines = 1;
    for (size_t c : lines) total_lines += c;
    out.at.assign(total_lines, nullptr);
  }
  // Waves of 16 MB (at least eight): every worker takes the lines that start in its share of the wave, the wave's records are listed
  // in file order and published, and the caller walks them while the next wave is parsed (the walk used to start when the last of
  // three million records was in: 0.06 s into the call at BASELINE config 5's size).
  const size_t wave = std::max<size_t>(1u << 20, std::min<size_t>(16u << 20, (n + 7) / 8));
  const size_t n_waves = n ? (n + wave - 1) / wave : 0;
  out.parts.assign(n_waves * T, std::vector<Var>());
  size_t total = 0;
  for (size_t w = 0; w < n_waves; w++) {
    const size_t w_lo = w * wave, w_hi = std::min(n, w_lo + wave);
    pool->for_blocks(w_hi - w_lo, [&](size_t b0, size_t e0, int tid) {
      const size_t b = w_lo + b0, e = w_lo + e0;
      const char* const base = text;
      const char* const end = base + n;
      const char* p = base + b;
      if (b > 0) { const char* nl = (const char*)std::memchr(base + b - 1, '\n', n - (b - 1)); p = nl ? nl + 1 : end; }   // first line start >= b
      std::vector<Var>& mine = out.parts[w * T + (size_t)tid];
      // (a record is parsed where it stays: a Var built aside and moved in, into a vector that doubled its way up, was a third of the
      // 0.11 s the records took -- room for a record per 24 bytes, which no line with an INFO column undercuts)
      mine.reserve((e - b) / 24 + 16);
      while (p < base + e) {
        const char* nl = (const char*)std::memchr(p, '\n', (size_t)(end - p));
        const char* le = nl ? nl : end;
        if (p < le && *p != '#') {
          mine.emplace_back();
          if (!parse_record(p, le, chrom, chrom_len, mine.back())) mine.pop_back();
        }
        p = le + 1;
      }
    });
    for (size_t t = 0; t < T; t++) {
      std::vector<Var>& mine = out.parts[w * T + t];
      if (total + mine.size() > out.at.size()) { out.publish(total, true); return "the VCF holds more records than lines (internal error)"; }
      for (size_t k = 0; k < mine.size(); k++) out.at[total + k] = &mine[k];
      total += mine.size();
    }
    out.publish(total, w + 1 == n_waves);
  }
  if (n_waves == 0) out.publish(0, true);
  if (TUNE_GET("CALITAS_TRACE") && total >= 100000)
    std::fprintf(stderr, "[calitas] read_vcf: %zu bytes read in %.1f ms, %zu records parsed in %.1f ms (%zu waves)\n", n, ms_read, total, ms_since(t_parse), n_waves);
  return "";
}

// alleleCombos(counts) SR:377-399: every combination of allele indices, the first variant varying slowest
std::vector<std::vector<int>> allele_combos_counts(const std::vector<int>& counts) {
  size_t total = 1;
  for (int c : counts) total *= (size_t)c;
  std::vector<std::vector<int>> results(total, std::vector<int>(counts.size(), 0));
  size_t denom = 1;
  for (size_t i = 0; i < counts.size(); i++) {
    denom *= (size_t)counts[i];
    const size_t group = total / denom;
    size_t j = 0;
    int allele = 0;
    while (j < total) {
      for (size_t k = 0; k < group; k++) results[j++][i] = allele;
      allele = (allele + 1) % counts[i];
    }
  }
  return results;
}

bool is_valid(const std::vector<const Var*>& vs) {   // VariantSet.isValid SR:182-193
  for (size_t i = 0; i + 1 < vs.size(); i++) {
    const Var &a = *vs[i], &b = *vs[i + 1];
    const int s1 = a.pos, e1 = a.pos + (int)a.ref.size() - 1, s2 = b.pos, e2 = b.pos + (int)b.ref.size() - 1;
    if (a.chrom == b.chrom && s1 <= e2 && e1 >= s2) return false;
  }
  return true;
}

// Upper-cased bases [s, e) of a contig (what the reference reads after toUpperCase): 2-bit decode, exceptions through base_upper.
void upper_span(const PackedRef& ref, int contig, long s, long e, std::string& out) {
  const ContigInfo& c = ref.contigs[contig];
  out.resize((size_t)std::max(0L, e - s));
  for (long q = s; q < e; q++) {
    const uint64_t gpos = c.gbase + (uint64_t)q;
    out[(size_t)(q - s)] = ((ref.mask[gpos >> 5] >> (gpos & 31)) & 1u) ? ref.base_upper(gpos) : "ACGT"[(ref.codes[gpos >> 4] >> ((gpos & 15) * 2)) & 3u];
  }
}

// buildVariantWindow SR:263-323.  The window's pieces are appended to A (w gets the counts, `mark` where they start); tmp / ctmp are scratch.
std::string build_window(const Var* const* variants, const int* alleles, size_t nv, int contig, const PackedRef& ref, int padding, Arena& A,
                         std::string& tmp, std::vector<CigarEl>& ctmp, Window& w, ArenaMark& mark) {
  const int window_start = std::max(1, variants[0]->pos - padding);
  const int window_end = std::min((int)ref.contigs[contig].len, variants[nv - 1]->end + padding);
  w.contig = contig; w.start = window_start;
  mark = ArenaMark{A.bases.size(), A.alleles.size(), A.cigars.size()};
  upper_span(ref, contig, window_start - 1, std::max(window_start - 1, window_end), tmp);
  for (size_t i = 0; i < nv; i++) {
    const Var* v = variants[i];
    const int a = alleles[i] - 1;
    A.alleles.push_back(Allele{v, a, (size_t)a < v->afs.size() ? v->afs[a] : 0.0f});
  }
  const Allele* const wv = A.alleles.data() + mark.alleles;
  w.nv = (int)nv;
  for (size_t k = nv; k-- > 0;) {                     // right to left, so earlier offsets stay valid
    const Allele& al = wv[k];
    const int i = al.v->pos - window_start;
    if (i < 0 || (size_t)i > tmp.size()) return "variant outside its window";
    tmp.replace((size_t)i, std::min(al.v->ref.size(), tmp.size() - (size_t)i), al.v->alts[al.alt]);
  }
  ctmp.clear();
  int ref_pos = window_start, base_off = 0;
  for (size_t k = 0; k < nv; k++) {
    const Allele& al = wv[k];
    const int pm = al.v->pos - ref_pos;
    if (pm > 0) { ctmp.push_back({'M', pm}); ref_pos += pm; base_off += pm; }
    const int rl = (int)al.v->ref.size(), alen = (int)al.v->alts[al.alt].size();
    if (rl == alen) ctmp.push_back({'M', rl});
    else if (rl == 1 && alen > 1) { ctmp.push_back({'M', 1}); ctmp.push_back({'I', alen - 1}); }
    else if (rl > 1 && alen == 1) { ctmp.push_back({'M', 1}); ctmp.push_back({'D', rl - 1}); }
    else { ctmp.push_back({'D', rl}); ctmp.push_back({'I', alen}); }
    ref_pos += rl; base_off += alen;
  }
  ctmp.push_back({'M', (int)tmp.size() - base_off});
  for (const CigarEl& e : ctmp) {                      // Cigar.coalesce
    if (A.cigars.size() > mark.cigars && A.cigars.back().op == e.op) A.cigars.back().n += e.n; else A.cigars.push_back(e);
  }
  w.nc = (int)(A.cigars.size() - mark.cigars);
  long on_query = 0;
  for (size_t k = mark.cigars; k < A.cigars.size(); k++) if (A.cigars[k].op == 'M' || A.cigars[k].op == 'I') on_query += A.cigars[k].n;
  if (on_query != (long)tmp.size()) return "requirement failed: cigar length on query != bases";
  A.bases.insert(A.bases.end(), tmp.begin(), tmp.end());
  w.len = (int)tmp.size();
  return "";
}

// refOffsetAtBaseOffset SR:133-156
bool ref_offset_at(const Window& w, int offset, bool preceding, int& out) {
  auto on_q = [](const CigarEl& e) { return (e.op == 'M' || e.op == 'I') ? e.n : 0; };
  auto on_t = [](const CigarEl& e) { return (e.op == 'M' || e.op == 'D') ? e.n : 0; };
  if (offset == w.len) {
    int t = 0;
    for (int k = 0; k < w.nc; k++) t += on_t(w.cigar[k]);
    out = w.start - 1 + t;
    return true;
  }
  int ref_off = w.start - 1, base_off = 0;
  int k = 0;
  while (k < w.nc && offset >= base_off + on_q(w.cigar[k])) { ref_off += on_t(w.cigar[k]); base_off += on_q(w.cigar[k]); k++; }
  if (k >= w.nc) return false;
  const char op = w.cigar[k].op;
  if (op == 'I') { out = preceding ? ref_off - 1 : ref_off; return true; }
  if (op == 'M') { out = ref_off + (offset - base_off); return true; }
  return false;                                       // "Query bases can't be present at operator D."
}

std::string format_metric_double(double d) {          // fgbio Metric.formatValue(Double), as variants.py states it
  char b[64];
  auto strip = [](std::string s) {
    while (!s.empty() && s.back() == '0') s.pop_back();
    if (!s.empty() && s.back() == '.') s.pop_back();
    return s;
  };
  if (d == 0) return "0";
  if (std::fabs(d) < 0.00001) {
    const int ex = (int)std::floor(std::log10(std::fabs(d)));
    std::snprintf(b, sizeof b, "%.5f", d / std::pow(10.0, ex));
    return strip(b) + "E" + std::to_string(ex);
  }
  std::snprintf(b, sizeof b, "%.6f", d);
  return strip(b);
}

std::string display_string(const Allele& a) {         // VariantAllele.displayString SR:108
  char b[64];
  std::snprintf(b, sizeof b, ":%d:", a.v->pos - 1);
  std::string s = (a.v->id.empty() ? std::string(".") : a.v->id) + b + a.v->ref + ">" + a.v->alts[a.alt];
  std::snprintf(b, sizeof b, ":%.3f", (double)a.af);
  return s + b;
}

std::string revcomp(const std::string& s) { return calitas::revcomp_str(s); }

int ga_count(const char* pg, const char* pa, int len, bool lower, bool both_sides, bool mms, bool gaps) {   // GA:139-163
  auto is_lower = [](char c) { return c >= 'a' && c <= 'z'; };
  auto is_letter = [](char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); };
  int n = 0;
  for (int i = 0; i < len; i++) {
    if (mms && pa[i] == '.' && is_lower(pg[i]) == lower) { n++; continue; }
    if (!(gaps && pa[i] == '~')) continue;
    const char gb = pg[i];
    bool me = gb != '-' && is_lower(gb) == lower;
    if (!me) {
      int pi = i; while (pi > 0 && pg[pi] == '-') pi--;
      int ni = i; while (ni < len - 1 && pg[ni] == '-') ni++;
      const char prev = pg[pi], next = pg[ni];
      if (both_sides) me = (prev == '-' || is_lower(prev) == lower) && (next == '-' || is_lower(next) == lower);
      else me = (is_letter(prev) && is_lower(prev) == lower) || (is_letter(next) && is_lower(next) == lower);
    }
    if (me) n++;
  }
  return n;
}


// MD5 (RFC 1321) of a file, hex: the second half of ReferenceHit's VCF identifier "name:md5" (RH:175-183).  One chain of dependent
// additions and rotations from the first byte to the last: 0.21 s per 127 MB as a loop over a step table, 0.14 s with the 64 steps
// written out (constants and rotations as immediates, the selection functions in their three-operation forms) -- and the first row that
// names a variant cannot be final before it is done, so at BASELINE config 5's size this is what the first contig's text waits for.
#define CALITAS_MD5_ROL(x, s) (((x) << (s)) | ((x) >> (32 - (s))))
#define CALITAS_MD5_F1(b, c, d) ((d) ^ ((b) & ((c) ^ (d))))
#define CALITAS_MD5_F2(b, c, d) ((c) ^ ((d) & ((b) ^ (c))))
#define CALITAS_MD5_F3(b, c, d) ((b) ^ (c) ^ (d))
#define CALITAS_MD5_F4(b, c, d) ((c) ^ ((b) | ~(d)))
#define CALITAS_MD5_STEP(f, a, b, c, d, g, k, s) a += f(b, c, d) + m[g] + (k); a = b + CALITAS_MD5_ROL(a, s);
static void md5_block(uint32_t* h, const unsigned char* p) {
  uint32_t m[16];
  std::memcpy(m, p, 64);                              // (little-endian words, as on every machine this library is built for)
  uint32_t a = h[0], b = h[1], c = h[2], d = h[3];
#define S1(a, b, c, d, g, k, s) CALITAS_MD5_STEP(CALITAS_MD5_F1, a, b, c, d, g, k, s)
#define S2(a, b, c, d, g, k, s) CALITAS_MD5_STEP(CALITAS_MD5_F2, a, b, c, d, g, k, s)
#define S3(a, b, c, d, g, k, s) CALITAS_MD5_STEP(CALITAS_MD5_F3, a, b, c, d, g, k, s)
#define S4(a, b, c, d, g, k, s) CALITAS_MD5_STEP(CALITAS_MD5_F4, a, b, c, d, g, k, s)
  S1(a,b,c,d,0,0xd76aa478u,7) S1(d,a,b,c,1,0xe8c7b756u,12) S1(c,d,a,b,2,0x242070dbu,17) S1(b,c,d,a,3,0xc1bdceeeu,22)
  S1(a,b,c,d,4,0xf57c0fafu,7) S1(d,a,b,c,5,0x4787c62au,12) S1(c,d,a,b,6,0xa8304613u,17) S1(b,c,d,a,7,0xfd469501u,22)
  S1(a,b,c,d,8,0x698098d8u,7) S1(d,a,b,c,9,0x8b44f7afu,12) S1(c,d,a,b,10,0xffff5bb1u,17) S1(b,c,d,a,11,0x895cd7beu,22)
  S1(a,b,c,d,12,0x6b901122u,7) S1(d,a,b,c,13,0xfd987193u,12) S1(c,d,a,b,14,0xa679438eu,17) S1(b,c,d,a,15,0x49b40821u,22)
  S2(a,b,c,d,1,0xf61e2562u,5) S2(d,a,b,c,6,0xc040b340u,9) S2(c,d,a,b,11,0x265e5a51u,14) S2(b,c,d,a,0,0xe9b6c7aau,20)
  S2(a,b,c,d,5,0xd62f105du,5) S2(d,a,b,c,10,0x02441453u,9) S2(c,d,a,b,15,0xd8a1e681u,14) S2(b,c,d,a,4,0xe7d3fbc8u,20)
  S2(a,b,c,d,9,0x21e1cde6u,5) S2(d,a,b,c,14,0xc33707d6u,9) S2(c,d,a,b,3,0xf4d50d87u,14) S2(b,c,d,a,8,0x455a14edu,20)
  S2(a,b,c,d,13,0xa9e3e905u,5) S2(d,a,b,c,2,0xfcefa3f8u,9) S2(c,d,a,b,7,0x676f02d9u,14) S2(b,c,d,a,12,0x8d2a4c8au,20)
  S3(a,b,c,d,5,0xfffa3942u,4) S3(d,a,b,c,8,0x8771f681u,11) S3(c,d,a,b,11,0x6d9d6122u,16) S3(b,c,d,a,14,0xfde5380cu,23)
  S3(a,b,c,d,1,0xa4beea44u,4) S3(d,a,b,c,4,0x4bdecfa9u,11) S3(c,d,a,b,7,0xf6bb4b60u,16) S3(b,c,d,a,10,0xbebfbc70u,23)
  S3(a,b,c,d,13,0x289b7ec6u,4) S3(d,a,b,c,0,0xeaa127fau,11) S3(c,d,a,b,3,0xd4ef3085u,16) S3(b,c,d,a,6,0x04881d05u,23)
  S3(a,b,c,d,9,0xd9d4d039u,4) S3(d,a,b,c,12,0xe6db99e5u,11) S3(c,d,a,b,15,0x1fa27cf8u,16) S3(b,c,d,a,2,0xc4ac5665u,23)
  S4(a,b,c,d,0,0xf4292244u,6) S4(d,a,b,c,7,0x432aff97u,10) S4(c,d,a,b,14,0xab9423a7u,15) S4(b,c,d,a,5,0xfc93a039u,21)
  S4(a,b,c,d,12,0x655b59c3u,6) S4(d,a,b,c,3,0x8f0ccc92u,10) S4(c,d,a,b,10,0xffeff47du,15) S4(b,c,d,a,1,0x85845dd1u,21)
  S4(a,b,c,d,8,0x6fa87e4fu,6) S4(d,a,b,c,15,0xfe2ce6e0u,10) S4(c,d,a,b,6,0xa3014314u,15) S4(b,c,d,a,13,0x4e0811a1u,21)
  S4(a,b,c,d,4,0xf7537e82u,6) S4(d,a,b,c,11,0xbd3af235u,10) S4(c,d,a,b,2,0x2ad7d2bbu,15) S4(b,c,d,a,9,0xeb86d391u,21)
#undef S1
#undef S2
#undef S3
#undef S4
  h[0] += a; h[1] += b; h[2] += c; h[3] += d;
}
std::string md5_file(const char* path, std::string& hex) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return std::string("cannot read ") + path;
  uint32_t h[4] = {0x67452301u, 0xefcdab89u, 0x98badcfeu, 0x10325476u};
  std::vector<unsigned char> buf(1 << 20);
  uint64_t total = 0;
  size_t have = 0;                                   // bytes of an incomplete block at the start of buf
  for (;;) {
    const size_t got = std::fread(buf.data() + have, 1, buf.size() - have, f);
    total += got;
    const size_t n = have + got;
    size_t off = 0;
    for (; off + 64 <= n; off += 64) md5_block(h, buf.data() + off);
    have = n - off;
    std::memmove(buf.data(), buf.data() + off, have);
    if (got == 0) break;
  }
  std::fclose(f);
  unsigned char tail[128] = {0};
  std::memcpy(tail, buf.data(), have);
  tail[have] = 0x80;
  const size_t tl = have < 56 ? 64 : 128;
  const uint64_t bits = total * 8;
  for (int i = 0; i < 8; i++) tail[tl - 8 + i] = (unsigned char)(bits >> (8 * i));
  for (size_t off = 0; off < tl; off += 64) md5_block(h, tail + off);
  char out[33];
  for (int i = 0; i < 16; i++) std::snprintf(out + 2 * i, 3, "%02x", (h[i / 4] >> (8 * (i % 4))) & 0xFFu);
  hex = out;
  return "";
}

// A thread that runs jobs in the order they are handed over (a stage of the variant branch's pipeline).  After a job has failed the
// ones behind it are dropped -- but a dropped job's `skipped` handler still runs, in the job's place: whatever a job owes OTHER threads
// (its turn in the order in which batches reach the lifter) is paid there, so nobody waits for a job that will never run.  drain()
// reports the failure.  A stage that is destroyed with jobs still queued (the calling thread left through an exception) drops them
// the same way before it joins its thread.
struct StageThread {
  struct Job { std::function<int(std::string&)> run; std::function<void()> skipped; };
  std::mutex mu;
  std::condition_variable cv;
  std::deque<Job> jobs;
  bool busy = false, quit = false;
  int rc = CALITAS_OK;
  std::string err;
  std::thread t;
  void start(int device) {
    t = std::thread([this, device] {
      if (device >= 0) (void)hipSetDevice(device);
      for (;;) {
        Job job;
        bool skip = false;
        {
          std::unique_lock<std::mutex> lk(mu);
          cv.wait(lk, [&] { return !jobs.empty() || quit; });
          if (jobs.empty()) return;
          job = std::move(jobs.front());
          jobs.pop_front();
          busy = true;
          skip = quit || rc != CALITAS_OK || !err.empty();
        }
        cv.notify_all();
        int r = CALITAS_OK;
        std::string e;
        try {
          if (!skip) r = job.run(e);
          else if (job.skipped) job.skipped();
        } catch (const std::exception& x) { r = CALITAS_EHIP; e = std::string("a stage of the variant branch ended with an exception: ") + x.what(); }
        catch (...) { r = CALITAS_EHIP; e = "a stage of the variant branch ended with an exception"; }
        job = Job();
        {
          std::lock_guard<std::mutex> lk(mu);
          busy = false;
          if (r && rc == CALITAS_OK) rc = r;
          if (!e.empty() && err.empty()) err = e;
        }
        cv.notify_all();
      }
    });
  }
  // hands a job over; waits while max_waiting jobs are waiting (ms_wait: that time is added to it)
  // (a job refused here -- the stage has failed -- has NOT been queued: its `skipped` handler runs on the calling thread, now)
  int enqueue(std::function<int(std::string&)> job, size_t max_waiting, double* ms_wait, std::function<void()> skipped = nullptr) {
    const auto t0 = std::chrono::steady_clock::now();
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return jobs.size() < max_waiting; });
    if (ms_wait) *ms_wait += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rc != CALITAS_OK || !err.empty()) {                      // (the caller learns the reason from drain())
      const int r = rc != CALITAS_OK ? rc : CALITAS_EINVAL;
      lk.unlock();
      if (skipped) skipped();
      return r;
    }
    jobs.push_back(Job{std::move(job), std::move(skipped)});
    lk.unlock();
    cv.notify_all();
    return CALITAS_OK;
  }
  // every job handed over has run
  int drain(double* ms_wait, std::string* err_out) {
    const auto t0 = std::chrono::steady_clock::now();
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return jobs.empty() && !busy; });
    if (ms_wait) *ms_wait += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (err_out && !err.empty() && err_out->empty()) *err_out = err;
    return rc;
  }
  ~StageThread() {
    if (!t.joinable()) return;
    { std::lock_guard<std::mutex> lk(mu); quit = true; }
    cv.notify_all();
    t.join();
  }
};

}  // namespace

// user_dst / user_cap: calitas_search_variants_into -- the text goes to the caller's (page-locked) buffer, *tsv = user_dst on success.
static int search_variants_impl(calitas_ctx* ctx, const calitas_guide_t* guide, const char* guide_id, const calitas_params_t* params,
                                const char* vcf_path, const char* chrom, const char* vcf_id, const char* aligner_version,
                                const char* time_stamp, char** tsv, uint64_t* tsv_bytes, uint64_t* n_rows, uint64_t* n_windows,
                                char* user_dst, uint64_t user_cap) {
  if (!ctx) return CALITAS_EINVAL;
  if (!guide || !params || !vcf_path || !tsv) return calitas_fail(ctx, CALITAS_EINVAL, "NULL argument");
  *tsv = nullptr;
  if (tsv_bytes) *tsv_bytes = 0;
  if (n_rows) *n_rows = 0;
  if (n_windows) *n_windows = 0;
  if (!ctx->has_ref) return calitas_fail(ctx, CALITAS_ESTATE, "calitas_set_reference has not been called");
  if (params->first_window != 0 || params->n_windows != 0)
    return calitas_fail(ctx, CALITAS_EINVAL, "a window range (first_window / n_windows) is for calitas_search only: removeOverlaps needs every alignment of a contig");
  const PackedRef& ref = ctx->ref;
  const calitas_params_t& p = *params;
  GuideHost gh;
  {
    std::string e = make_guide_host(*guide, gh);
    if (!e.empty()) return calitas_fail(ctx, CALITAS_EINVAL, e);
  }
  const std::string gid = guide_id ? guide_id : "";
  std::string vid = vcf_id ? vcf_id : "";
  std::string md5_err;
  std::thread md5_thread;                                                                          // ReferenceHit.scala:175-183: file name and md5,
  if (!vcf_id)                                                                                     // needed when the first row is written
    md5_thread = std::thread([&] {
      std::string hex;
      md5_err = md5_file(vcf_path, hex);
      const char* slash = std::strrchr(vcf_path, '/');
      vid = std::string(slash ? slash + 1 : vcf_path) + ":" + hex;
    });
  struct JoinMd5 { std::thread& t; ~JoinMd5() { if (t.joinable()) t.join(); } } join_md5{md5_thread};
  std::mutex vid_mu;                                               // (several stages may ask; one of them joins the thread)
  auto need_vid = [&]() -> bool { std::lock_guard<std::mutex> lk(vid_mu); if (md5_thread.joinable()) md5_thread.join(); return md5_err.empty(); };
  // what a row holds in the identifier's place until the MD5 is known: as long as the identifier will be (name : 32 hex digits)
  const std::string vid_placeholder = vcf_id ? std::string(vcf_id) : [&] { const char* slash = std::strrchr(vcf_path, '/'); return std::string(slash ? slash + 1 : vcf_path) + ":" + std::string(32, '0'); }();
  std::string version, stamp;
  calitas_default_version_and_stamp(aligner_version, time_stamp, version, stamp);
  const int d = p.max_guide_diffs, g = p.max_gaps_between_guide_and_pam;
  const int max_pam = [&] { size_t m = 0; for (auto& q : gh.pams) m = std::max(m, q.size()); return (int)m; }();
  const int padding = (int)gh.protospacer.size() + max_pam - 1 + d + g;                          // SR:575 (query.length - 1 + d + g)

  const auto t_call = std::chrono::steady_clock::now();
  auto cpu_seconds = [] {                                                                          // (CALITAS_TRACE: how busy the call kept the process's threads)
    rusage u{};
    (void)getrusage(RUSAGE_SELF, &u);
    return (double)u.ru_utime.tv_sec + (double)u.ru_stime.tv_sec + 1e-6 * ((double)u.ru_utime.tv_usec + (double)u.ru_stime.tv_usec);
  };
  const double cpu0 = cpu_seconds();
  const bool trace_stages = TUNE_GET("CALITAS_TRACE") && std::atoi(TUNE_GET("CALITAS_TRACE")) >= 3;   // (every batch's way through the stages)
  auto ms_since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count(); };
  std::atomic<int> batch_serial{0};                               // (batches through align_part, for CALITAS_FAIL_ALIGN_BATCH)
  std::atomic<long long> ns_align{0};                             // (two aligner threads add to it)
  double ms_ref = 0, ms_parse = 0, ms_rows = 0, ms_merge = 0, ms_build = 0, ms_wait = 0, ms_finish = 0;
  calitas_aln_t* ref_alns = nullptr;                                                               // (host merge only, below)
  uint64_t n_ref = 0;
  int rc = CALITAS_OK;
  calitas_ctx* actx = ctx;                                                                         // where the variant windows are aligned (below)

  VarTable vcf;                                                    // (read below, once the reference passes are under way)
  std::vector<std::string> order;                                                                 // contigs the iterator walks
  for (auto& n : ref.names) if (!chrom || n == chrom) order.push_back(n);

  // the pieces of a row that do not depend on the hit
  const RowStrings rs = make_row_strings(ref, gh, gid, p, version, stamp);
  calitas_params_t ap = p;                                                                         // the explicit-target pass
  ap.chrom_index = -1;

  // Every alignment of every variant window is a hit as far as removeOverlaps goes, but most of them repeat a reference hit (the part
  // of the window the variant does not touch) and lose against it there: a hit gets its key here -- lifted coordinates, score,
  // variant_description -- on the worker pool, and its row only if it is kept (make_row, called back from the row stage of hits_tsv).
  struct ExtHit { const Window* w; const calitas_aln_t* a; int gstart, tlen; std::string desc; };
  // (in pieces that never move: the helper thread makes rows from entries of a published contig -- through pointers taken when the contig was
  // finished -- while the lifter thread appends the next contigs' hits)
  struct HitList {
    enum : size_t { kPiece = 1u << 16 };
    std::vector<std::unique_ptr<ExtHit[]>> pieces;
    size_t n = 0;
    size_t size() const { return n; }
    ExtHit& operator[](size_t i) { return pieces[i >> 16][i & (kPiece - 1)]; }
    const ExtHit& operator[](size_t i) const { return pieces[i >> 16][i & (kPiece - 1)]; }
    void resize(size_t m) {                                                                         // (grows only)
      while (pieces.size() * kPiece < m) pieces.emplace_back(new ExtHit[kPiece]);
      n = m;
    }
    void release() { std::vector<std::unique_ptr<ExtHit[]>>().swap(pieces); n = 0; }
  };
  HitList hits;
  struct Batch { std::vector<Window> wins; std::vector<Arena> arenas; };
  std::deque<Batch> kept_windows;                                                                    // the windows and alignment records behind the hits
  std::vector<calitas_aln_t*> kept_out;
  const size_t kBatch = 65536;
  uint64_t windows_total = 0;
  std::string err;

  auto fetch_ref = [&](int ci, long s1, long e1, bool minus) {                                     // fetchBases RH:261-266, 1-based closed
    const long clen = (long)ref.contigs[ci].len;
    const long as = std::max(1L, s1), ae = std::min(clen, e1);
    std::string b((size_t)std::max(0L, as - s1), 'N');
    for (long q = as; q <= ae; q++) b += ref.base_upper(ref.contigs[ci].gbase + (uint64_t)(q - 1));
    b.append((size_t)std::max(0L, e1 - ae), 'N');
    return minus ? revcomp(b) : b;
  };

  // lifted coordinates of an alignment of a window (SR:615-620); false: "Query bases can't be present at operator D."
  auto lift = [](const Window& w, const calitas_aln_t& a, int& start, int& end, int& gstart, int& gend) {
    return ref_offset_at(w, a.start_offset, true, start) && ref_offset_at(w, a.end_offset, false, end) &&
           ref_offset_at(w, a.guide_start_offset, true, gstart) && ref_offset_at(w, a.guide_end_offset, false, gend);
  };

  // the row of a kept hit (RH:210-254 with the window's own bases, SR:598-613)
  const std::string build_with_variants = ref.genome_build + "+variants";
  // compact: without guide_id and protospacer and with "\n" for a tail (post.hpp, compact_row_strings_keep_build) -- what the device's
  // row stage is given when the per-contig texts cross PCIe compact; genome_build stays: a row with a variant has "<build>+variants"
  // vid_use / vid_at: the identifier's text to put into the row (null: the real one -- the caller has waited for it) and where, counted
  // from the row's first byte, it went (0: the row names no variant) -- for rows that are made before the VCF's MD5 is known.
  auto make_row = [&](const ExtHit& h, std::string& row, bool compact = false, const std::string* vid_use = nullptr, uint32_t* vid_at = nullptr) {
        const std::string& vid_text = vid_use ? *vid_use : vid;
        if (vid_at) *vid_at = 0;
        const Window& w = *h.w;
        const calitas_aln_t& a = *h.a;
        const int wl = w.len;
        const int gs = a.guide_start_offset, ge = a.guide_end_offset, as = a.start_offset, ae = a.end_offset;   // window-local
        int start = 0, end = 0, gstart = 0, gend = 0;
        (void)lift(w, a, start, end, gstart, gend);                                                 // (succeeded when the hit was keyed)
        auto flank = [&](int from, int to, bool have) { return have ? std::string(w.bases + from, (size_t)(to - from)) : std::string(); };
        const bool minus = a.strand == '-';
        const bool h_l10 = gs >= 10, h_r10 = wl - ge >= 10, h_l8 = as >= 8, h_r8 = wl - ae >= 8;
        std::string l10 = flank(gs - 10, gs, h_l10), r10 = flank(ge, ge + 10, h_r10), l8 = flank(as - 8, as, h_l8), r8 = flank(ae, ae + 8, h_r8);
        bool v_l10 = h_l10, v_r10 = h_r10, v_l8 = h_l8, v_r8 = h_r8;
        if (minus) {
          std::string t10 = l10, t8 = l8;
          l10 = h_r10 ? revcomp(r10) : std::string(); r10 = h_l10 ? revcomp(t10) : std::string();
          l8 = h_r8 ? revcomp(r8) : std::string();   r8 = h_l8 ? revcomp(t8) : std::string();
          v_l10 = h_r10; v_r10 = h_l10; v_l8 = h_r8; v_r8 = h_l8;
        }
        auto ten_left = [&] { return fetch_ref(w.contig, gstart + 1 - 10, gstart, minus); };
        auto ten_right = [&] { return fetch_ref(w.contig, gend + 1, gend + 10, minus); };
        auto eight_left = [&] { return fetch_ref(w.contig, start + 1 - 8, start, minus); };
        auto eight_right = [&] { return fetch_ref(w.contig, end + 1, end + 8, minus); };
        const std::string c5_10 = v_l10 ? l10 : (!minus ? ten_left() : ten_right());
        const std::string c3_10 = v_r10 ? r10 : (!minus ? ten_right() : ten_left());
        const std::string c5_8 = v_l8 ? l8 : (!minus ? eight_left() : eight_right());
        const std::string c3_8 = v_r8 ? r8 : (!minus ? eight_right() : eight_left());
        // padded strings from the window's own bases (SGA:511; '-' strand: revcomp of the window span)
        const std::string& q = rs.query[a.pam_index + 1];
        // (two million rows per call at full size: the pieces of a row are put together in buffers on the stack, not in strings of their own)
        char t[CALITAS_MAX_OPS + 8];
        {
          const int tl = std::min(ae - as, (int)CALITAS_MAX_OPS);
          if (!minus) std::memcpy(t, w.bases + as, (size_t)tl);
          else for (int i = 0; i < tl; i++) t[i] = complement_base(w.bases[ae - 1 - i]);
        }
        const int n_ops = a.n_ops;
        char pg[CALITAS_MAX_OPS + 1], pa[CALITAS_MAX_OPS + 1], pt[CALITAS_MAX_OPS + 1];
        size_t qi = 0, ti = 0;
        int mm = 0, gp = 0;
        for (int i = 0; i < n_ops; i++) {
          switch (a.ops[i]) {
            case 'I': pg[i] = q[qi++]; pa[i] = '~'; pt[i] = '-'; gp++; break;
            case 'D': pg[i] = '-'; pa[i] = '~'; pt[i] = t[ti++]; gp++; break;
            case '=': pg[i] = q[qi++]; pa[i] = '|'; pt[i] = t[ti++]; break;
            default:  pg[i] = q[qi++]; pa[i] = '.'; pt[i] = t[ti++]; mm++; break;
          }
        }
        int ps = -1, pe = -1;                                                                       // GA:111-115
        for (int i = 0; i < n_ops; i++) if (pg[i] >= 'A' && pg[i] <= 'Z') { if (ps < 0) ps = i; pe = i; }
        char unpadded_target[CALITAS_MAX_OPS + 1];
        size_t n_unpadded = 0;
        for (int i = ps; i >= 0 && i <= pe; i++) if (pt[i] != '-') unpadded_target[n_unpadded++] = pt[i];
        // variants under the hit (RH:211) and their columns (RH:211-233)
        const Allele* vs_few[8];
        std::vector<const Allele*> vs_many;
        size_t n_vs = 0;
        for (int k = 0; k < w.nv; k++) {
          const Allele& al = w.variants[k];
          if (!(start <= al.v->pos - 1 && al.v->pos - 1 <= end)) continue;
          if (n_vs < 8) vs_few[n_vs] = &al;
          else { if (n_vs == 8) vs_many.assign(vs_few, vs_few + 8); vs_many.push_back(&al); }
          n_vs++;
        }
        const Allele* const* vs_p = n_vs <= 8 ? vs_few : vs_many.data();
        struct VsView { const Allele* const* p; size_t n; bool empty() const { return n == 0; } size_t size() const { return n; }
                        const Allele* operator[](size_t i) const { return p[i]; } const Allele* const* begin() const { return p; } const Allele* const* end() const { return p + n; } };
        const VsView vs{vs_p, n_vs};
        std::string ids, descs, af;
        if (!vs.empty()) {
          const Allele* mn = vs[0];
          for (const Allele* al : vs) if (al->af < mn->af) mn = al;                                 // minBy keeps the first minimum
          af = format_metric_double((double)mn->af);
          for (size_t i = 0; i < vs.size(); i++) { if (i) { ids += ';'; descs += ';'; } ids += vs[i]->v->id; descs += display_string(*vs[i]); }
        }
        const int gmm = ga_count(pg, pa, n_ops, false, false, true, false), ggp = ga_count(pg, pa, n_ops, false, false, false, true);
        char cigar[4 * CALITAS_MAX_OPS + 8];
        size_t n_cigar = 0;
        auto put_int = [](char* at, long v) -> size_t {            // decimal digits of v at `at`; returns how many
          char d[24]; int nd = 0; const bool neg = v < 0; unsigned long u = neg ? (unsigned long)(-v) : (unsigned long)v;
          do { d[nd++] = (char)('0' + u % 10); u /= 10; } while (u);
          size_t k = 0;
          if (neg) at[k++] = '-';
          while (nd) at[k++] = d[--nd];
          return k;
        };
        for (int i = 0; i < n_ops;) { int j = i; while (j < n_ops && a.ops[j] == a.ops[i]) j++; n_cigar += put_int(cigar + n_cigar, j - i); cigar[n_cigar++] = (char)a.ops[i]; i = j; }
        // the row is put together in place: room for the longest it can be, one pointer walking through it (sixty appends to a string,
        // each with its capacity check, were a microsecond per row -- two seconds of the workers' time per call at full size)
        const size_t row_at = row.size();                                                           // (appends to what is there)
        const std::string& build = vs.empty() ? ref.genome_build : build_with_variants;
        const std::string& pam_used = rs.pam_used[a.pam_index + 1];
        const size_t room = gid.size() + gh.protospacer.size() + build.size() + ref.names[w.contig].size() + n_unpadded + c5_10.size() + c3_10.size() +
                            pam_used.size() + ids.size() + descs.size() + vid_text.size() + af.size() + 3 * (size_t)n_ops + c5_8.size() + c3_8.size() + n_cigar +
                            rs.proto_len.size() + rs.tail.size() + 9 * 24 + 40;
        row.resize(row_at + room);
        char* wp = &row[row_at];
        auto add = [&](const std::string& s) { std::memcpy(wp, s.data(), s.size()); wp += s.size(); *wp++ = '\t'; };
        auto add_mem = [&](const char* m, size_t len) { std::memcpy(wp, m, len); wp += len; *wp++ = '\t'; };
        auto add_int = [&](long v) { wp += put_int(wp, v); *wp++ = '\t'; };
        if (!compact) { add(gid); add(gh.protospacer); }
        add(build); add(ref.names[w.contig]);
        add_int(gstart); add_int(gend); *wp++ = (char)a.strand; *wp++ = '\t'; add_mem(unpadded_target, n_unpadded);
        add(c5_10); add(c3_10); add(pam_used); add(ids); add(descs); if (vs.empty()) *wp++ = '\t'; else { if (vid_at) *vid_at = (uint32_t)(wp - &row[row_at]); add(vid_text); } add(af);
        add_int(a.score); add_int(gmm); add_int(ggp); add_int(gmm + ggp);
        add_int(ga_count(pg, pa, n_ops, true, true, true, false)); add_int(mm + gp);
        add_mem(pg, (size_t)n_ops); add_mem(pa, (size_t)n_ops); add_mem(pt, (size_t)n_ops);
        add(c5_8); add(c3_8); add_mem(cigar, n_cigar); add(rs.proto_len); add_int((long)n_unpadded);
        if (!compact) { std::memcpy(wp, rs.tail.data(), rs.tail.size()); wp += rs.tail.size(); }   // aligner .. time_stamp + '\n'
        else *wp++ = '\n';                                       // (the compact tail; the cell before it keeps its tab: the full tail starts with the next field)
        if (wp > &row[row_at] && wp[-1] == '\n') wp--;
        row.resize((size_t)(wp - row.data()));
  };

  // A built batch of windows through the aligner (device) and its alignments lifted back and listed as hits (worker pool).  Runs on the
  // aligner thread (below) while this thread builds the next batch; `err_out` is that thread's own.
  struct Aligned { calitas_aln_t* out = nullptr; uint64_t n_out = 0; uint32_t* counts = nullptr; };
  auto align_part = [&](calitas_ctx* actx, Batch& batch, const size_t n, Aligned& res) -> int {
    std::vector<calitas_guide_t> guides(n, *guide);
    std::vector<const uint8_t*> targets(n);
    std::vector<uint32_t> lens(n);
    std::vector<int32_t> offs(n, 0);
    for (size_t i = 0; i < n; i++) { targets[i] = reinterpret_cast<const uint8_t*>(batch.wins[i].bases); lens[i] = (uint32_t)batch.wins[i].len; }
    const auto t0 = std::chrono::steady_clock::now();
    if (const char* inj = TUNE_GET("CALITAS_FAIL_ALIGN_BATCH"))        // tests: the error path of the stages (a failed batch must fail the call, not hang it)
      if (std::atoi(inj) == batch_serial++) return calitas_fail(ctx, CALITAS_EHIP, "injected failure of an aligner batch (CALITAS_FAIL_ALIGN_BATCH)");
    int r = calitas_align_windows(actx, (int32_t)n, guides.data(), targets.data(), lens.data(), offs.data(), &ap, &res.out, &res.n_out, &res.counts);
    if (r) { if (actx != ctx) calitas_fail(ctx, r, calitas_last_error(actx)); return r; }
    ns_align += (long long)(ms_since(t0) * 1e6);
    if (trace_stages) std::fprintf(stderr, "[calitas] search_variants: a batch (contig %d ..) aligned %.1f .. %.1f ms\n", batch.wins[0].contig, ms_since(t_call) - ms_since(t0), ms_since(t_call));
    return CALITAS_OK;
  };
  auto lift_part = [&](Batch& batch, const size_t n, const Aligned& res, std::string& err) -> int {
    calitas_aln_t* const out = res.out;
    const uint64_t n_out = res.n_out;
    uint32_t* const counts = res.counts;
    const auto t1 = std::chrono::steady_clock::now();
    // the batch's windows and records stay until the rows are written
    kept_windows.emplace_back(std::move(batch));                  // (vectors move: the views keep pointing into the arenas)
    const std::vector<Window>& wins = kept_windows.back().wins;
    kept_out.push_back(out);
    std::vector<uint64_t> first(n + 1, 0);
    for (size_t t = 0; t < n; t++) first[t + 1] = first[t] + counts[t];
    const size_t base = hits.size();
    hits.resize(base + (size_t)n_out);
    std::vector<std::string> errs((size_t)ctx->pool->size());
    ctx->pool->for_blocks(n, [&](size_t tb, size_t te, int tid) {
      for (size_t t = tb; t < te && errs[(size_t)tid].empty(); t++) {
        const Window& w = wins[t];
        for (uint64_t k = first[t]; k < first[t + 1]; k++) {
          const calitas_aln_t& a = out[k];
          ExtHit& h = hits[base + (size_t)k];
          h.w = &w; h.a = &a;
          int start = 0, end = 0, gend = 0;
          if (!lift(w, a, start, end, h.gstart, gend)) { errs[(size_t)tid] = "Query bases can't be present at operator D."; break; }
          h.tlen = 0;
          for (int i = 0; i < a.n_ops; i++) if (a.ops[i] != 'I') h.tlen++;
          // variants under the hit (RH:211): their display strings are the hit's removeOverlaps group (SR:656)
          bool any = false;
          for (int q = 0; q < w.nv; q++) { const Allele& al = w.variants[q]; if (start <= al.v->pos - 1 && al.v->pos - 1 <= end) { if (any) h.desc += ';'; h.desc += display_string(al); any = true; } }
        }
      }
    });
    for (auto& e : errs) if (!e.empty() && err.empty()) err = e;
    calitas_free(counts);
    ms_rows += ms_since(t1);
    if (trace_stages) std::fprintf(stderr, "[calitas] search_variants: a batch (contig %d ..) lifted %.1f .. %.1f ms\n", wins[0].contig, ms_since(t_call) - ms_since(t1), ms_since(t_call));
    return CALITAS_OK;
  };
  auto align_stage = [&](Batch& batch, const size_t n, std::string& err) -> int {
    if (n == 0) return CALITAS_OK;
    Aligned res;
    const int r = align_part(actx, batch, n, res);
    return r ? r : lift_part(batch, n, res, err);
  };
  // where a built batch goes: to the aligner thread once it runs (hand_over), through align_stage on this thread before that
  std::function<int(Batch&&, size_t)> hand_over;
  // variantWindowIterator SR:217-256 with nextChunk / reChunk SR:326-347.  The iterator itself only lists what each window is made
  // of (variants and alleles: a Spec); a full batch of windows is then built on the worker pool and handed to the GPU -- by a stage
  // thread of its own once the stages run (round 5: this thread used to wait for every batch's build, 0.15-0.3 s per call at BASELINE
  // config 5's size, with the list of the next batch standing still meanwhile).
  struct Spec { std::vector<uint32_t> off{0}; std::vector<const Var*> v; std::vector<int> a, contig; std::vector<uint32_t> chunk; };
  Spec spec;
  uint32_t chunk_serial = 0;
  StageThread* builder_p = nullptr;                               // (set once the stage threads run)
  double ms_wait_builder = 0;                                     // this thread's waits for the builder stage (ms_wait: the later stages')
  auto build_spec = [&](const Spec& sp, std::string& e_out) -> int {
    const size_t n = sp.contig.size();
    if (n == 0) return CALITAS_OK;
    const auto t_build = std::chrono::steady_clock::now();
    Batch b;
    b.wins.resize(n + 1);
    b.arenas.resize((size_t)ctx->pool->size());
    std::vector<std::string> errs((size_t)ctx->pool->size());
    ctx->pool->for_blocks(n, [&](size_t lo, size_t hi, int tid) {
      Arena& A = b.arenas[(size_t)tid];
      std::string tmp;
      std::vector<CigarEl> ctmp;
      std::vector<ArenaMark> marks(hi - lo);
      A.bases.reserve((hi - lo) * (size_t)(2 * padding + 8));
      for (size_t k = lo; k < hi && errs[(size_t)tid].empty(); k++)
        errs[(size_t)tid] = build_window(sp.v.data() + sp.off[k], sp.a.data() + sp.off[k], sp.off[k + 1] - sp.off[k], sp.contig[k], ref,
                                         padding, A, tmp, ctmp, b.wins[k], marks[k - lo]);
      for (size_t k = lo; k < hi; k++) {                         // the arena is complete: the views get their pointers
        Window& w = b.wins[k];
        w.chunk = sp.chunk[k];
        w.bases = A.bases.data() + marks[k - lo].bases; w.variants = A.alleles.data() + marks[k - lo].alleles; w.cigar = A.cigars.data() + marks[k - lo].cigars;
      }
    });
    for (auto& e : errs) if (!e.empty() && e_out.empty()) e_out = e;
    ms_build += ms_since(t_build);
    if (trace_stages) std::fprintf(stderr, "[calitas] search_variants: a batch of %zu windows (contig %d ..) built %.1f .. %.1f ms\n", n, sp.contig[0], ms_since(t_call) - ms_since(t_build), ms_since(t_call));
    if (!e_out.empty()) return CALITAS_EINVAL;
    if (hand_over) return hand_over(std::move(b), n);
    return align_stage(b, n, e_out);
  };
  auto build_and_flush = [&]() -> int {
    if (spec.contig.empty()) return CALITAS_OK;
    auto held = std::make_shared<Spec>(std::move(spec));
    spec = Spec();
    if (builder_p) return builder_p->enqueue([&, held](std::string& e) { return build_spec(*held, e); }, 2, &ms_wait_builder);
    std::string e;
    const int r = build_spec(*held, e);
    if (!e.empty()) { if (err.empty()) err = e; return CALITAS_OK; }   // (the walk stops at err; the call's code is set where it ends)
    return r;
  };
  auto emit = [&](const Var* const* vs, const int* al, size_t nv, int contig) -> int {
    spec.v.insert(spec.v.end(), vs, vs + nv);
    spec.a.insert(spec.a.end(), al, al + nv);
    spec.off.push_back((uint32_t)spec.v.size());
    spec.contig.push_back(contig);
    spec.chunk.push_back(chunk_serial);
    windows_total++;
    return spec.contig.size() >= kBatch ? build_and_flush() : CALITAS_OK;
  };
  // ---- the reference windows (SR:527-561) and the merge (SR:641-648) on the device, beside the variant windows --------------------
  // The reference's own hits never leave the device.  A hit of a variant window that touches no variant joins the removeOverlaps
  // group of the reference hits of its chromosome and strand (SR:656) -- most of them repeat a reference hit and lose against it there,
  // the ones an edge of their window cut short do not -- so every one of them goes into the device's walk of that group (hits.hpp,
  // HitsExt), behind the reference hits with the same sort key as SR:622 has them arrive.  The groups of the hits that do touch variants
  // hold nothing else: they are walked here, and what they keep is handed to the device for its place in ReferenceHit.sort's order only.
  // The device then writes every surviving row, its own and these, into one text per contig.  Ties between rows of different groups
  // follow calitas_hits_tsv_ext (the reference leaves them to a hash map): the reference group first, then the variant groups in order
  // of first appearance.
  // The two halves run side by side: this thread produces, aligns (on a side context: a stream and buffers of its own) and keys the
  // variant windows contig by contig -- host work, mostly -- while a helper thread drives the reference's per-contig passes (device work
  // and the text over PCIe); the row stage of contig c waits until this thread has published the contig's entries.
  const char* force_host = TUNE_GET("CALITAS_VARIANTS_HOST");
  const bool device_merge = !(force_host && std::atoi(force_host) != 0) && p.max_overlap >= 1;
  calitas_ctx* actx2 = nullptr;                                                                    // (a second aligner, below)
  if (device_merge) { rc = calitas_side_context(ctx, &actx); if (rc) return rc; rc = calitas_side_context(ctx, &actx2, 1); if (rc) return rc; }
  const size_t nc = ref.contigs.size();
  struct ContigExt {
    std::vector<HitsExtKey> keys; std::vector<uint64_t> row_off;
    std::vector<std::string> segs;                                // the rows' text as the workers wrote it: a block of rows each
    std::vector<const char*> seg_ptr; std::vector<uint64_t> seg_off;
    std::vector<const ExtHit*> entry;                              // the entries in tie order: the plain ones, then the placed ones
    std::vector<uint32_t> row_len;
    std::vector<uint32_t> vid_off;                                 // rows filled in on the host: where a row holds the VCF's identifier (0: nowhere)
    std::vector<const char*> row_ptr;                              // ... and where the row stands in its buffer
    size_t n_plain = 0;
    std::vector<std::string> segs_placed;                          // the placed entries' rows (segs: the plain entries')
    HitsExt ext;
  };
  std::vector<ContigExt> cx(nc);
  std::mutex pub_mu;
  std::condition_variable pub_cv;
  size_t published = 0;                                                                            // contigs [0, published) have their entries
  bool give_up = false;
  struct HelperResult { int rc = CALITAS_OK; bool declined = false; char* tsv = nullptr; uint64_t bytes = 0, rows = 0; double ms = 0; } hr;
  // Rows written into the text on the host (hits.hpp, HitsExtRows::fill_on_host): when the text goes to a buffer of the caller's the
  // copying thread of the reference passes only hands the job over -- this stage waits for the VCF's MD5 once and fills the holes.
  // (Declared before the helper thread, whose calls hand it work: it outlives it.)
  StageThread filler;
  std::atomic<long long> ns_fill{0};
  std::atomic<uint64_t> rows_filled{0};
  std::thread helper;
  const bool trace_contigs = trace_stages;
  auto publish = [&](size_t upto, bool quit) {
    { std::lock_guard<std::mutex> lk(pub_mu); published = std::max(published, upto); give_up = give_up || quit; }
    pub_cv.notify_all();
    if (trace_contigs) std::fprintf(stderr, "[calitas] search_variants: contigs before %zu published at %.1f ms\n", upto, ms_since(t_call));
  };
  HitsExtSource source;
  // Compact rows on the per-contig stream of this call: OFF unless asked for.  Measured at BASELINE config 5's size (round 5): the texts'
  // time on the bus halves (0.40 -> 0.21 s) and the call does not get shorter (1.10-1.14 against 1.12-1.17 s) -- this branch is bound by
  // its host threads (a 16-core quota), and putting guide_id, protospacer and the tail back into 41 million rows is more work for them
  // (window building 0.31-0.39 -> 0.42-0.51 s, the entries' rows 0.37-0.42 -> 0.41-0.51 s).
  source.compact_rows = device_merge && TUNE_ON("CALITAS_VARIANTS_COMPACT");
  // Rows on demand (hits.hpp, HitsExt::rows_for): an entry's row is made when the device's walk has kept it -- one in ten at BASELINE config
  // 5's size; the rows of all two million entries were 0.37-0.42 s of the lifter thread's 0.7 s per call, and 1.1 GB on their way to the
  // device.  CALITAS_VARIANTS_ROWS=all: every entry's row up front, as before (the two give the same bytes: tests/test_gpu_variants.py).
  bool rows_on_demand = device_merge;
  if (const char* e = TUNE_GET("CALITAS_VARIANTS_ROWS")) rows_on_demand = rows_on_demand && std::strcmp(e, "all") != 0;
  // ... and the rows of the entries the device keeps never go to the device: the rows kernel leaves holes, the host fills them once the
  // text is there (HitsExtRows::fill_on_host).  CALITAS_VARIANTS_ROWS=device: the kept rows go up and the rows kernel copies them, as in
  // the first half of round 5; compact rows (CALITAS_VARIANTS_COMPACT) imply it -- a hole's place is known in the text the device wrote.
  bool fill_on_host = rows_on_demand && !source.compact_rows;
  if (const char* e = TUNE_GET("CALITAS_VARIANTS_ROWS")) fill_on_host = fill_on_host && std::strcmp(e, "device") != 0;
  if (fill_on_host) filler.start(-1);
  struct JoinHelper {                                                                               // (declared behind everything the helper thread uses)
    std::thread& t; decltype(publish)& pub; size_t all;
    ~JoinHelper() { if (t.joinable()) { pub(all, true); t.join(); } }
  } join_helper{helper, publish, nc};
  source.get = [&](int c, const HitsExt** e) -> int {
    std::unique_lock<std::mutex> lk(pub_mu);
    pub_cv.wait(lk, [&] { return published > (size_t)c || give_up; });
    if (published <= (size_t)c) return 1;
    *e = cx[(size_t)c].ext.n ? &cx[(size_t)c].ext : nullptr;
    return 0;
  };
  if (device_merge)
    helper = std::thread([&] {
      const auto t0 = std::chrono::steady_clock::now();
      (void)hipSetDevice(ctx->device);
      try {
        hr.rc = calitas_search_hits_ext_impl(ctx, guide, gid, params, version.c_str(), stamp.c_str(), source, &hr.tsv, &hr.bytes, &hr.rows, &hr.declined, user_dst, user_cap);
      } catch (const std::exception& e) {                                                           // (a thread of its own has no caller to unwind to)
        hr.rc = calitas_fail(ctx, CALITAS_EHIP, std::string("the reference passes ended with an exception: ") + e.what());
      }
      hr.ms = ms_since(t0);
    });

  // The rows of contig c's entries [lo, hi) -- of those with kept[i] != 0, or of all (kept null) -- as consecutive blocks of entries, each
  // written into a buffer of its own (segs) by whichever worker takes it next (the entries with variants stand at the end of the order and
  // their rows cost 2.5 times a plain one: equal shares per worker left seven workers with all of them, 25 ms against 10 per contig); the
  // device takes the buffers piece by piece.  x.row_len[i] = the row's length with its newline, 0 for an entry that is not wanted.
  std::atomic<long long> ns_demand{0};                             // (rows on demand: the helper thread's time here)
  std::atomic<uint64_t> rows_made{0};
  auto make_rows = [&](size_t c, size_t lo, size_t hi, const uint8_t* kept, std::vector<std::string>& segs) {
    ContigExt& x = cx[c];
    const size_t n = hi - lo;
    const size_t T = (size_t)ctx->pool->size();
    const size_t S = std::max<size_t>(1, std::min<size_t>(4 * T, (n + 255) / 256));
    segs.assign(S, std::string());
    if (n == 0) return;
    std::atomic<size_t> next_seg{0};
    ctx->pool->run([&](int) {
      for (;;) {
        const size_t sg = next_seg.fetch_add(1, std::memory_order_relaxed);
        if (sg >= S) return;
        const size_t b = lo + n * sg / S, e = lo + n * (sg + 1) / S;
        size_t wanted = e - b;
        if (kept) { wanted = 0; for (size_t i = b; i < e; i++) wanted += kept[i] != 0; }
        if (!wanted) continue;
        std::string& buf = segs[sg];
        buf.reserve(wanted * 700 + 2048);                        // (+ the room make_row asks for before it knows the last row's length)
        for (size_t i = b; i < e; i++) {
          if (kept && !kept[i]) continue;
          const size_t at = buf.size();
          if (fill_on_host) make_row(*x.entry[i], buf, source.compact_rows, &vid_placeholder, &x.vid_off[i]);   // (appends; the MD5 may not be there yet)
          else make_row(*x.entry[i], buf, source.compact_rows);
          buf += '\n';
          x.row_len[i] = (uint32_t)(buf.size() - at);
        }
        rows_made.fetch_add(wanted, std::memory_order_relaxed);
        if (fill_on_host) {                                      // (the buffer is complete: where each of its rows stands)
          size_t acc = 0;
          for (size_t i = b; i < e; i++) if (x.row_len[i] && (!kept || kept[i])) { x.row_ptr[i] = buf.data() + acc; acc += x.row_len[i]; }
        }
      }
    });
  };
  // ... and what the device is given: the offsets of all entries' rows in the text that the buffers -- the plain entries', then the placed
  // ones' -- make in this order.
  auto rows_of = [&](size_t c, HitsExtRows* out) -> int {
    ContigExt& x = cx[c];
    const size_t n = x.entry.size();
    x.row_off.resize(n + 1);
    x.row_off[0] = 0;
    for (size_t i = 0; i < n; i++) x.row_off[i + 1] = x.row_off[i] + x.row_len[i];
    if (fill_on_host) { *out = HitsExtRows(); out->row_off = x.row_off.data(); out->fill_on_host = true; return CALITAS_OK; }
    x.seg_ptr.clear(); x.seg_off.assign(1, 0);
    for (std::vector<std::string>* group : {&x.segs, &x.segs_placed})
      for (const std::string& sg : *group) {
        if (sg.empty()) continue;
        x.seg_ptr.push_back(sg.data());
        x.seg_off.push_back(x.seg_off.back() + sg.size());
      }
    if (x.seg_off.back() != x.row_off[n])
      return calitas_fail(ctx, CALITAS_EINVAL, "the rows of a contig's entries are not where their offsets say (internal error)");
    if (x.seg_ptr.empty()) { x.seg_ptr.push_back(""); x.seg_off.push_back(0); }   // (no row at all: one empty piece)
    out->row_off = x.row_off.data(); out->rows = nullptr;
    out->n_seg = (uint32_t)x.seg_ptr.size(); out->seg = x.seg_ptr.data(); out->seg_off = x.seg_off.data();
    return CALITAS_OK;
  };
  // The entries of contig c -- hits[h0, h1), in arrival order -- for the device: the groups' walks, every entry's key and (unless the
  // device asks for them later: rows on demand) row.
  double ms_groups = 0, ms_make = 0, ms_blob = 0;
  auto finish_contig = [&](size_t c, size_t h0, size_t h1) -> int {
    if (h1 == h0) return CALITAS_OK;
    if (h1 - h0 >= 0xFFFFFFF0ull) return calitas_fail(ctx, CALITAS_EINVAL, "more than 2^32 hits of variant windows on one contig");
    const auto t0 = std::chrono::steady_clock::now();
    const size_t T = (size_t)ctx->pool->size();
    // blocks of hits cut where the chunk changes: hits of two chunks share no variant, hence no group
    std::vector<size_t> cut(T + 1, h1);
    cut[0] = h0;
    for (size_t t = 1; t < T; t++) {
      size_t k = std::max(cut[t - 1], h0 + (h1 - h0) * t / T);
      while (k < h1 && k > h0 && hits[k].w->chunk == hits[k - 1].w->chunk) k++;
      cut[t] = k;
    }
    struct Lite { int start, end, score; uint32_t idx; };
    std::vector<std::vector<uint32_t>> plain(T), kept(T);
    ctx->pool->run([&](int tid) {
      const size_t b = cut[(size_t)tid], e = cut[(size_t)tid + 1];
      if (b >= e) return;
      std::unordered_map<std::string, uint32_t> group_of;
      std::vector<std::vector<Lite>> groups;
      std::string key;
      for (size_t k = b; k < e; k++) {
        const ExtHit& h = hits[k];
        if (h.desc.empty()) { plain[(size_t)tid].push_back((uint32_t)(k - h0)); continue; }
        key.assign(1, (char)h.a->strand);
        key += h.desc;
        auto it = group_of.find(key);
        if (it == group_of.end()) { it = group_of.emplace(key, (uint32_t)groups.size()).first; groups.emplace_back(); }
        groups[it->second].push_back(Lite{h.gstart, h.gstart + h.tlen - 1, h.a->score, (uint32_t)(k - h0)});
      }
      for (auto& hs : groups) {                                                                    // removeOverlaps SR:653-675 on one group
        std::stable_sort(hs.begin(), hs.end(), [](const Lite& x, const Lite& y) { return x.start != y.start ? x.start < y.start : -x.score < -y.score; });
        auto overlap = [](const Lite& x, const Lite& y) { return std::max(0, std::min(x.end, y.end) - std::max(x.start, y.start)); };   // RH:141-144
        size_t i = 0;
        while (i < hs.size()) {
          const Lite hit = hs[i++];
          while (i < hs.size() && overlap(hs[i], hit) >= p.max_overlap && hs[i].score <= hit.score) i++;
          if (i >= hs.size() || overlap(hs[i], hit) < p.max_overlap) kept[(size_t)tid].push_back(hit.idx);
        }
      }
    });
    std::vector<uint32_t> order;                                                                    // the entries in tie order (as offsets from h0)
    for (auto& v : plain) order.insert(order.end(), v.begin(), v.end());
    const size_t n_plain = order.size();
    for (auto& v : kept) order.insert(order.end(), v.begin(), v.end());
    const size_t n = order.size();
    ContigExt& x = cx[c];
    x.entry.resize(n);
    x.keys.resize(n);
    for (size_t i = 0; i < n; i++) {
      const ExtHit& h = hits[h0 + order[i]];
      x.entry[i] = &h;
      x.keys[i] = HitsExtKey{h.gstart, h.gstart + h.tlen - 1, h.a->score, (h.a->strand == '-' ? HITS_EXT_MINUS : 0u) | (i >= n_plain ? HITS_EXT_PLACED : 0u)};
    }
    x.ext.contig = (int32_t)c; x.ext.n = (uint32_t)n; x.ext.keys = x.keys.data();
    x.n_plain = n_plain;
    x.row_len.assign(n, 0);
    if (fill_on_host) { x.vid_off.assign(n, 0); x.row_ptr.assign(n, nullptr); }
    ms_groups += ms_since(t0);
    if (trace_stages) std::fprintf(stderr, "[calitas] search_variants: contig %zu: groups and keys %.1f .. %.1f ms\n", c, ms_since(t_call) - ms_since(t0), ms_since(t_call));
    return CALITAS_OK;
  };
  // ... and the rows (the finisher stage, behind the lifter: the lifter carried keys, groups and rows one after the other, 0.43-0.47 s
  // per call at BASELINE config 5's size, and every other stage of the variant half waited for it).
  // The placed entries -- kept by the walks of their own groups, so their rows are wanted whatever the device decides -- get their rows
  // now; the plain ones when the device's walk has kept them (next to none: they repeat reference hits), on the helper thread inside the
  // contig's row stage.
  auto finish_rows = [&](size_t c) -> int {
    ContigExt& x = cx[c];
    const size_t n = x.entry.size(), n_plain = x.n_plain;
    if (n == 0) return CALITAS_OK;
    const auto t1 = std::chrono::steady_clock::now();
    if (!fill_on_host && !need_vid()) return calitas_fail(ctx, CALITAS_EIO, md5_err);
    make_rows(c, n_plain, n, nullptr, x.segs_placed);
    if (fill_on_host)
      x.ext.fill = [&, c](const uint64_t* place, char* text, bool stays) -> int {
        auto work = [&, c, text](const uint64_t* pl) -> int {
          const auto t_f = std::chrono::steady_clock::now();
          if (!need_vid()) return calitas_fail(ctx, CALITAS_EIO, md5_err);
          if (vid.size() != vid_placeholder.size()) return calitas_fail(ctx, CALITAS_EINVAL, "the VCF's identifier is not as long as its placeholder (internal error)");
          ContigExt& y = cx[c];
          std::atomic<uint64_t> done{0};
          ctx->pool->for_blocks(y.entry.size(), [&](size_t b, size_t e, int) {
            uint64_t k = 0;
            for (size_t i = b; i < e; i++) {
              if (pl[i] == ~0ull) continue;
              if (!y.row_ptr[i]) continue;                       // (counted below: a kept entry without a row is an error)
              char* dst = text + pl[i];
              std::memcpy(dst, y.row_ptr[i], y.row_len[i]);
              if (y.vid_off[i]) std::memcpy(dst + y.vid_off[i], vid.data(), vid.size());
              k++;
            }
            done += k;
          });
          uint64_t want = 0;
          for (size_t i = 0; i < y.entry.size(); i++) want += pl[i] != ~0ull;
          if (done.load() != want) return calitas_fail(ctx, CALITAS_EINVAL, "an entry the device kept has no row (internal error)");
          rows_filled += want;
          ns_fill += (long long)(ms_since(t_f) * 1e6);
          return CALITAS_OK;
        };
        if (!stays) return work(place);
        auto held = std::make_shared<std::vector<uint64_t>>(place, place + cx[c].entry.size());   // (place[] is the device stage's: valid during this call only)
        return filler.enqueue([work, held](std::string&) -> int { return work(held->data()); }, 64, nullptr);
      };
    if (rows_on_demand) {
      // (runs on the helper thread; cx[c] is this contig's alone from here on)
      x.ext.rows_for = [&, c, n_plain](const uint8_t* kept, HitsExtRows* out) -> int {
        const auto t_d = std::chrono::steady_clock::now();
        std::fill(cx[c].row_len.begin(), cx[c].row_len.begin() + (std::ptrdiff_t)n_plain, 0u);   // (a second row stage of the same contig starts afresh)
        make_rows(c, 0, n_plain, kept, cx[c].segs);
        const int r = rows_of(c, out);
        ns_demand += (long long)(ms_since(t_d) * 1e6);
        return r;
      };
      ms_make += ms_since(t1);
      return CALITAS_OK;
    }
    make_rows(c, 0, n_plain, nullptr, x.segs);
    HitsExtRows made;
    const int r = rows_of(c, &made);
    if (r) return r;
    x.ext.row_off = made.row_off; x.ext.rows = nullptr; x.ext.n_seg = made.n_seg; x.ext.seg = made.seg; x.ext.seg_off = made.seg_off;
    ms_make += ms_since(t1);
    return CALITAS_OK;
  };
  // The contigs before `upto` have all their windows emitted: align what is pending, finish and publish them.
  size_t contigs_done = 0, hits_done = 0;
  // Two stage threads: batch k is on the device (aligner), the alignments of batch k - 1 are lifted back and listed as hits (lifter),
  // while this thread walks the VCF and builds batch k + 1 -- 48 batches of 65 536 windows at full size: 8-10 ms each in the aligner, 3 to
  // lift, 5 to walk and build.  Jobs run in the order they were handed over; two wait per stage at most.
  // (Two aligners when the call has side contexts: a batch is 8-10 ms in calitas_align_windows and 5 ms to walk and build, so one
  // aligner was the pipeline's slowest stage; the batches alternate between them and reach the lifter in their own order.)
  // Declared BEFORE the stages: what the stages' jobs capture must outlive the stage threads, which are joined by the stages'
  // destructors -- also when this thread leaves through an exception (std::bad_alloc while building a batch) with jobs still queued.
  std::mutex order_mu;
  std::condition_variable order_cv;
  uint64_t batches_handed = 0, batches_lifting = 0;               // (batches_lifting: under order_mu)
  std::function<int(size_t)> finish_upto_fn;                      // (finish_upto, defined below: the lifter runs it behind a contig's last batch)
  // Batch k's turn at the lifter: every batch takes it exactly once, in the order the batches were built -- whether its job ran, failed,
  // threw or was dropped because the stage had failed before (StageThread's `skipped` handler) -- so a job of the other aligner that
  // waits for "batches_lifting == k" is never left waiting for a job that will not run.  at_turn (may be empty) runs inside the turn.
  auto pass_turn = [&](uint64_t k, const std::function<int()>& at_turn) -> int {
    std::unique_lock<std::mutex> lk(order_mu);
    order_cv.wait(lk, [&] { return batches_lifting == k; });
    int r = CALITAS_OK;
    try { if (at_turn) r = at_turn(); }
    catch (...) { batches_lifting = k + 1; lk.unlock(); order_cv.notify_all(); throw; }
    batches_lifting = k + 1;
    lk.unlock();
    order_cv.notify_all();
    return r;
  };
  // (the lifter first: the aligners' jobs hand work to it, so it is destroyed -- joined -- after them)
  // (... the finisher before it: the lifter's jobs hand the contigs' rows to it; and the builder last: its jobs hand work to the aligners)
  StageThread finisher, lifter, aligner, aligner2, builder;
  const bool two_aligners = actx2 != nullptr;
  aligner.start(ctx->device);
  if (two_aligners) aligner2.start(ctx->device);
  lifter.start(-1);
  finisher.start(-1);
  builder.start(-1);
  builder_p = &builder;
  // a contig's entries for the device are made on the lifter thread, behind the lift of the contig's last batch, while this thread is
  // already walking the next contig (waiting for the stages to run dry at every one of 25 contig ends was 0.22 s of the variant half)
  auto hand_over_finish = [&](size_t upto) -> int {
    const uint64_t k = batches_handed++;
    const bool second = two_aligners && (k & 1);
    return (second ? aligner2 : aligner).enqueue([&, k, upto](std::string&) -> int {
      return pass_turn(k, [&]() -> int { return lifter.enqueue([&, upto](std::string&) { return finish_upto_fn(upto); }, 2, nullptr); });
    }, 2, &ms_wait, [&, k] { (void)pass_turn(k, nullptr); });
  };
  hand_over = [&](Batch&& b, size_t n) -> int {
    auto held = std::make_shared<Batch>(std::move(b));
    const uint64_t k = batches_handed++;
    const bool second = two_aligners && (k & 1);
    calitas_ctx* const where = second ? actx2 : actx;
    return (second ? aligner2 : aligner).enqueue([&, held, n, k, where](std::string& e) -> int {
      auto res = std::make_shared<Aligned>();
      int r = CALITAS_OK;
      try { if (n) r = align_part(where, *held, n, *res); }
      catch (const std::exception& x) { r = CALITAS_EHIP; e = std::string("the aligner stage of the variant branch ended with an exception: ") + x.what(); }
      // the lifter takes the batches in the order they were built, whichever aligner is done first
      return pass_turn(k, [&]() -> int {
        if (r || !n) return r;
        return lifter.enqueue([&, held, n, res](std::string& le) { return lift_part(*held, n, *res, le); }, 2, nullptr);
      });
    }, 2, &ms_wait, [&, k] { (void)pass_turn(k, nullptr); });
  };
  auto drain = [&]() -> int {                                     // everything handed over is in hits[]
    const int r0 = builder.drain(&ms_wait_builder, &err);
    const int ra = aligner.drain(&ms_wait, &err);
    const int rb = two_aligners ? aligner2.drain(&ms_wait, &err) : CALITAS_OK;
    const int rl = lifter.drain(&ms_wait, &err);
    const int rf = finisher.drain(&ms_wait, &err);
    return r0 ? r0 : ra ? ra : rb ? rb : rl ? rl : rf;
  };
  // (every batch handed over before it has been through the aligner: finish_contigs drains first)
  auto finish_upto = [&](size_t upto) -> int {
    if (upto <= contigs_done) return CALITAS_OK;
    const auto t_fin = std::chrono::steady_clock::now();
    struct Fin { double& ms; std::chrono::steady_clock::time_point t0; ~Fin() { ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } } fin{ms_finish, t_fin};
    if (device_merge) {
      // hits[hits_done, ...) lie on contigs [contigs_done, upto), in contig order
      size_t h = hits_done;
      for (size_t c = contigs_done; c < upto; c++) {
        size_t e = h;
        while (e < hits.size() && (size_t)hits[e].w->contig == c) e++;
        int r = finish_contig(c, h, e);
        if (r) return r;
        h = e;
        r = finisher.enqueue([&, c](std::string&) -> int { const int rr = finish_rows(c); if (!rr) publish(c + 1, false); return rr; }, 4, nullptr);
        if (r) return r;
      }
    }
    hits_done = hits.size();
    contigs_done = upto;
    return CALITAS_OK;
  };
  size_t contigs_asked = 0;                                       // (this thread's side of contigs_done)
  auto finish_contigs = [&](size_t upto) -> int {
    if (upto <= contigs_asked) return CALITAS_OK;
    int r = build_and_flush();
    if (r || !err.empty()) return r;
    contigs_asked = upto;
    // The contig's entries are made on the lifter thread, behind the lift of the contig's last batch, while this thread goes on with
    // the next contig -- no waiting for the stages to run dry at each of the 25 contig ends.  (Measured three times: on the one aligner
    // thread there was at first, that thread carried 1.26 s of host work one after the other, variant half 1.54 against 1.38 s; on the
    // lifter with two aligners but a pool that let one caller in at a time, the same 1.35-1.41 s; with the pool's shares, 0.95-1.04
    // against 1.25-1.32 s, step 1.36-1.38 against 1.60-1.62 s on one box, alternating.)
    // (through the builder stage, behind the contig's last batch: the batches are numbered where they are handed on)
    return builder.enqueue([&, upto](std::string&) -> int { return hand_over_finish(upto); }, 2, &ms_wait_builder);
  };
  finish_upto_fn = finish_upto;

  // The VCF, beside the first of the reference passes (they need nothing of it before their first row stage: 0.15 s at BASELINE config
  // 5's size that the helper thread used to sit out) -- and beside the walk below: a thread of its own parses the file wave by wave,
  // the walk follows it record by record (VarTable::have).
  std::string vcf_err;
  std::thread vcf_reader([&] {
    const auto t0 = std::chrono::steady_clock::now();
    try { vcf_err = read_vcf(vcf_path, chrom, ctx->pool, vcf); }
    catch (const std::exception& x) { vcf_err = std::string("reading the VCF ended with an exception: ") + x.what(); }
    ms_parse = ms_since(t0);
    vcf.publish(vcf.size(), true);                                  // (whatever happened: the walk must not wait for more)
  });
  struct JoinReader { std::thread& t; ~JoinReader() { if (t.joinable()) t.join(); } } join_reader{vcf_reader};
  const int max_variants = p.max_variants;
  size_t ci = 0, i = 0;
  const auto t_walk = std::chrono::steady_clock::now();
  // (three million chunks per call at full size: the vectors are reused, and a chunk's contig is looked up when the contig changes --
  // a search through the 25 names per chunk was a third of this thread's 0.38 s in the loop)
  std::vector<const Var*> chunk, sub;
  size_t ci_of_contig = (size_t)-1;
  int contig = -1;
  while (vcf.have(i) && err.empty() && rc == CALITAS_OK) {
    chunk.assign(1, &vcf[i]);
    const Var* last = &vcf[i];
    i++;
    while (vcf.have(i) && vcf[i].chrom == last->chrom && vcf[i].pos <= last->end + padding) { last = &vcf[i]; chunk.push_back(last); i++; }
    while (ci < order.size() && order[ci] != chunk[0]->chrom) ci++;
    if (ci >= order.size()) { err = "next on empty iterator (VCF contig " + chunk[0]->chrom + " not in reference order)"; break; }
    if (ci != ci_of_contig) {
      contig = -1;
      for (size_t k = 0; k < ref.names.size(); k++) if (ref.names[k] == order[ci]) { contig = (int)k; break; }
      ci_of_contig = ci;
    }
    chunk_serial++;
    if ((size_t)contig > contigs_asked) { rc = finish_contigs((size_t)contig); if (rc || !err.empty()) break; }   // the contigs before this one are complete: their entries are made behind their last batch
    for (size_t s = 0; s < chunk.size() && err.empty() && rc == CALITAS_OK; s++) {
      sub.clear();
      for (size_t k = s; k < chunk.size(); k++) { if (chunk[k]->pos - chunk[s]->end > padding) break; sub.push_back(chunk[k]); }
      // alleleCombos SR:351-369
      if ((int)sub.size() > max_variants || sub.size() == 1) {       // (a single variant: the same windows, without the tables)
        const Var* v = sub[0];
        for (size_t a = 0; a < v->alts.size() && err.empty() && rc == CALITAS_OK; a++) { const int al = (int)a + 1; rc = emit(&v, &al, 1, contig); }
      } else {
        std::vector<int> counts;
        for (const Var* v : sub) counts.push_back(1 + (int)v->alts.size());
        for (const std::vector<int>& alleles : allele_combos_counts(counts)) {
          std::vector<const Var*> sv; std::vector<int> sa;
          for (size_t k = 0; k < sub.size(); k++) if (alleles[k] != 0) { sv.push_back(sub[k]); sa.push_back(alleles[k]); }
          if (sv.empty() || !is_valid(sv)) continue;
          rc = emit(sv.data(), sa.data(), sv.size(), contig);
          if (rc || !err.empty()) break;
        }
      }
    }
  }
  vcf_reader.join();
  if (!vcf_err.empty() && err.empty()) { err = vcf_err; if (rc == CALITAS_OK) rc = CALITAS_EIO; }
  if (rc == CALITAS_OK && err.empty()) rc = finish_contigs(nc);
  const double ms_walk = ms_since(t_walk);                         // (this thread from the first variant to the last window handed over)
  const auto t_drain = std::chrono::steady_clock::now();
  { const int dr = drain(); if (rc == CALITAS_OK) rc = dr; }
  const double ms_drain = ms_since(t_drain);
  if (rc != CALITAS_OK || !err.empty()) {
    publish(nc, true);
    if (helper.joinable()) helper.join();
    if (hr.tsv != user_dst) calitas_free(hr.tsv);                 // (never the caller's own block)
    if (!err.empty()) return calitas_fail(ctx, rc != CALITAS_OK ? rc : CALITAS_EINVAL, err);   // (a stage's own text, or this thread's)
    return rc;                                                    // (the context's error text was set where the call failed)
  }
  const double ms_variant_half = ms_since(t_call);

  // What the variant half built is millions of small heap blocks (descriptions, VCF records, arenas): handed back by all workers, not
  // by the one thread that happens to leave the function.
  auto teardown = [&] {
    const auto t0 = std::chrono::steady_clock::now();
    ctx->pool->for_blocks(hits.size(), [&](size_t b, size_t e, int) { for (size_t k = b; k < e; k++) std::string().swap(hits[k].desc); });
    ctx->pool->for_blocks(vcf.parts.size(), [&](size_t b, size_t e, int) { for (size_t k = b; k < e; k++) std::vector<Var>().swap(vcf.parts[k]); });
    ctx->pool->for_blocks(kept_windows.size(), [&](size_t b, size_t e, int) { for (size_t k = b; k < e; k++) kept_windows[k] = Batch(); });
    ctx->pool->for_blocks(kept_out.size(), [&](size_t b, size_t e, int) { for (size_t k = b; k < e; k++) { calitas_free(kept_out[k]); kept_out[k] = nullptr; } });
    // the contigs' row blobs (1.1 GB at full size) and entry tables, and the big tables themselves: the kernel clears pages as they are
    // handed back, on the thread that hands them back -- one thread per block instead of this one for all of them on the way out
    ctx->pool->for_blocks(cx.size() + 2, [&](size_t b, size_t e, int) {
      for (size_t k = b; k < e; k++) {
        if (k < cx.size()) {
          std::vector<std::string>().swap(cx[k].segs);
          std::vector<std::string>().swap(cx[k].segs_placed);
          std::vector<uint32_t>().swap(cx[k].row_len);
          std::vector<uint32_t>().swap(cx[k].vid_off);
          std::vector<const char*>().swap(cx[k].row_ptr);
          std::vector<HitsExtKey>().swap(cx[k].keys);
          std::vector<uint64_t>().swap(cx[k].row_off);
          std::vector<const ExtHit*>().swap(cx[k].entry);
        } else if (k == cx.size()) {
          hits.release();
        } else {
          std::vector<Var*>().swap(vcf.at);
        }
      }
    });
    if (TUNE_GET("CALITAS_TRACE")) std::fprintf(stderr, "[calitas] search_variants: teardown %.1f ms\n", ms_since(t0));
  };
  if (device_merge) {
    helper.join();
    if (fill_on_host) {                                           // the rows the filler stage still owes the text
      std::string fe;
      const int fr = filler.drain(nullptr, &fe);
      if (hr.rc == CALITAS_OK && (fr || !fe.empty())) { hr.rc = fr ? fr : CALITAS_EINVAL; if (!fe.empty()) calitas_fail(ctx, hr.rc, fe); if (hr.tsv != user_dst) calitas_free(hr.tsv); hr.tsv = nullptr; }
    }
    if (hr.rc == CALITAS_OK) {
      const size_t n_hits = hits.size(), n_vcf = vcf.size();
      if (TUNE_GET("CALITAS_FREE_NOW")) teardown();
      else {
        // millions of small heap blocks and a few gigabytes of tables: nobody waits for them (0.17 s per call at full size even with
        // every worker handing them back) -- they go to the library's own thread as they are
        struct Garbage { decltype(hits) h; decltype(vcf) v; decltype(kept_windows) kw; decltype(kept_out) ko; decltype(cx) c; };
        auto g = std::make_shared<Garbage>();
        g->h = std::move(hits); g->v = std::move(vcf); g->kw = std::move(kept_windows); g->ko = std::move(kept_out); g->c = std::move(cx);
        calitas_reap_later([g]() mutable { for (auto* o : g->ko) calitas_free(o); g.reset(); });
      }
      *tsv = hr.tsv;
      if (tsv_bytes) *tsv_bytes = hr.bytes;
      if (n_rows) *n_rows = hr.rows;
      if (n_windows) *n_windows = windows_total;
      if (TUNE_GET("CALITAS_TRACE"))
        std::fprintf(stderr, "[calitas] search_variants: VCF %.1f ms (%zu records), %llu windows: walked and handed over in %.1f ms (waiting for the builder stage %.1f ms; there: built in %.1f ms, waiting for the aligner threads %.1f ms), stages drained in %.1f ms (align %.1f ms, keys %.1f ms there), "
                             "contigs finished in %.1f ms (groups %.1f + rows %.1f + blobs %.1f ms) of %zu hits (%llu rows made, %.1f ms of them on demand; %llu written into the text on the host in %.1f ms), "
                             "variant half done at %.1f ms; beside it the reference search with those hits on the device %.1f ms; call %.1f ms, %.2f s of CPU time\n",
                     ms_parse, n_vcf, (unsigned long long)windows_total, ms_walk, ms_wait_builder, ms_build, ms_wait, ms_drain, (double)ns_align.load() / 1e6, ms_rows, ms_finish, ms_groups, ms_make, ms_blob, n_hits, (unsigned long long)rows_made.load(), (double)ns_demand.load() / 1e6, (unsigned long long)rows_filled.load(), (double)ns_fill.load() / 1e6, ms_variant_half, hr.ms, ms_since(t_call), cpu_seconds() - cpu0);
      return CALITAS_OK;
    }
    if (hr.tsv != user_dst) calitas_free(hr.tsv);
    if (!hr.declined) { teardown(); return hr.rc; }
    if (TUNE_GET("CALITAS_TRACE")) std::fprintf(stderr, "[calitas] search_variants: the device row stage declined, merging on the host\n");
  }
  // On the host (a stage the device declines: -O 0, a window beyond the device filter, an overlap cluster beyond one lane's walk):
  // reference windows on the GPU, their alignment records back, removeOverlaps + sort over everything.
  {
    const auto t0 = std::chrono::steady_clock::now();
    rc = calitas_search_impl(ctx, 1, guide, params, &ref_alns, &n_ref);
    if (rc) { teardown(); return rc; }
    ms_ref = ms_since(t0);
  }
  // the rows of the kept variant-window hits are made on the way out
  std::vector<calitas_ext_hit_t> ext(hits.size());
  for (size_t k = 0; k < hits.size(); k++) {
    const ExtHit& h = hits[k];
    ext[k].contig_index = h.w->contig; ext[k].coordinate_start = h.gstart; ext[k].end = h.gstart + h.tlen - 1; ext[k].score = h.a->score;
    ext[k].strand = (int8_t)h.a->strand; ext[k].variant_description = h.desc.empty() ? nullptr : h.desc.c_str(); ext[k].row = nullptr;
  }
  if (TUNE_GET("CALITAS_TWIN_STATS")) {   // how many hits of variant windows that touch no variant repeat a reference hit exactly
    std::vector<std::array<int64_t, 3>> keys(n_ref);
    for (uint64_t i = 0; i < n_ref; i++) {
      const calitas_aln_t& a = ref_alns[i];
      int tl = 0;
      for (int k = 0; k < a.n_ops; k++) if (a.ops[k] != 'I') tl++;
      keys[i] = {((int64_t)a.contig_index << 32) | (uint32_t)a.guide_start_offset, ((int64_t)(a.guide_start_offset + tl - 1) << 8) | (uint8_t)a.strand, a.score};
    }
    std::sort(keys.begin(), keys.end());
    uint64_t plain = 0, twins = 0, with_desc = 0, shown = 0;
    for (size_t hk = 0; hk < hits.size(); hk++) {
      const ExtHit& h = hits[hk];
      if (!h.desc.empty()) { with_desc++; continue; }
      plain++;
      const std::array<int64_t, 3> k{((int64_t)h.w->contig << 32) | (uint32_t)h.gstart, ((int64_t)(h.gstart + h.tlen - 1) << 8) | (uint8_t)h.a->strand, h.a->score};
      if (std::binary_search(keys.begin(), keys.end(), k)) twins++;
      else if (shown++ < 8)
        std::fprintf(stderr, "[calitas] no twin: contig %d start %d len %d strand %c score %d, window start %d len %zu, aln offsets %d..%d\n", h.w->contig, h.gstart, h.tlen,
                     (char)h.a->strand, h.a->score, h.w->start, (size_t)h.w->len, h.a->start_offset, h.a->end_offset);
    }
    std::fprintf(stderr, "[calitas] variant-window hits: %zu, %llu with a description, %llu without, of those %llu repeat a reference hit\n", hits.size(),
                 (unsigned long long)with_desc, (unsigned long long)plain, (unsigned long long)twins);
  }
  if (!need_vid()) { calitas_free(ref_alns); teardown(); return calitas_fail(ctx, CALITAS_EIO, md5_err); }   // (the rows below name the VCF)
  struct RowMaker { decltype(make_row)* fn; const HitList* hits; } maker{&make_row, &hits};
  uint64_t nr = 0;
  const auto t_merge = std::chrono::steady_clock::now();
  *tsv = hits_tsv(ref, gh, gid, p, ref_alns, n_ref, version, stamp, &nr, ctx->pool, calitas_out_alloc, ext.data(), (uint64_t)ext.size(),
                  [](void* user, uint64_t e, std::string& row) { auto* m = static_cast<RowMaker*>(user); row.clear(); (*m->fn)((*m->hits)[(size_t)e], row); }, &maker);
  calitas_free(ref_alns);
  const size_t n_vcf_records = vcf.size();
  teardown();
  if (!*tsv) return calitas_fail(ctx, CALITAS_EINVAL, "out of memory");
  if (user_dst) {                                                 // (the merge on the host built a block of the library's: into the caller's buffer)
    const size_t len = std::strlen(*tsv);
    if ((uint64_t)len + 1 > user_cap) {
      calitas_free(*tsv); *tsv = nullptr;
      return calitas_fail(ctx, CALITAS_EINVAL, "the destination buffer is too small for the text (" + std::to_string(user_cap) + " bytes; " + std::to_string(len + 1) + " needed)");
    }
    std::memcpy(user_dst, *tsv, len + 1);
    calitas_free(*tsv);
    *tsv = user_dst;
  }
  if (tsv_bytes) *tsv_bytes = std::strlen(*tsv);
  if (n_rows) *n_rows = nr;
  if (n_windows) *n_windows = windows_total;
  ms_merge = ms_since(t_merge);
  if (TUNE_GET("CALITAS_TRACE"))
    std::fprintf(stderr, "[calitas] search_variants: reference search %.1f ms, VCF %.1f ms (%zu records), %llu windows: align %.1f ms, rows %.1f ms, merge %.1f ms, call %.1f ms\n",
                 ms_ref, ms_parse, n_vcf_records, (unsigned long long)windows_total, (double)ns_align.load() / 1e6, ms_rows, ms_merge, ms_since(t_call));
  return CALITAS_OK;
}

extern "C" int calitas_search_variants(calitas_ctx* ctx, const calitas_guide_t* guide, const char* guide_id, const calitas_params_t* params,
                                       const char* vcf_path, const char* chrom, const char* vcf_id, const char* aligner_version,
                                       const char* time_stamp, char** tsv, uint64_t* tsv_bytes, uint64_t* n_rows, uint64_t* n_windows) {
  return search_variants_impl(ctx, guide, guide_id, params, vcf_path, chrom, vcf_id, aligner_version, time_stamp, tsv, tsv_bytes, n_rows, n_windows, nullptr, 0);
}

extern "C" int calitas_search_variants_into(calitas_ctx* ctx, const calitas_guide_t* guide, const char* guide_id, const calitas_params_t* params,
                                            const char* vcf_path, const char* chrom, const char* vcf_id, const char* aligner_version,
                                            const char* time_stamp, char* dst, uint64_t dst_capacity, uint64_t* tsv_bytes, uint64_t* n_rows,
                                            uint64_t* n_windows) {
  if (!ctx) return CALITAS_EINVAL;
  if (!dst || dst_capacity < 2) return calitas_fail(ctx, CALITAS_EINVAL, "no destination buffer");
  char* text = nullptr;
  return search_variants_impl(ctx, guide, guide_id, params, vcf_path, chrom, vcf_id, aligner_version, time_stamp, &text, tsv_bytes, n_rows, n_windows, dst, dst_capacity);
}

// ---- what the variant search knows about a VCF, on its own (callers that search many guides against one VCF; the CPU tests) ----------

extern "C" int calitas_vcf_identifier(calitas_ctx* ctx, const char* vcf_path, char** id) {
  if (!vcf_path || !id) return calitas_fail(ctx, CALITAS_EINVAL, "NULL argument");
  *id = nullptr;
  std::string hex;
  const std::string e = md5_file(vcf_path, hex);
  if (!e.empty()) return calitas_fail(ctx, CALITAS_EIO, e);
  const char* slash = std::strrchr(vcf_path, '/');
  const std::string v = std::string(slash ? slash + 1 : vcf_path) + ":" + hex;
  char* out = (char*)calitas_out_alloc(v.size() + 1);
  if (!out) return calitas_fail(ctx, CALITAS_EINVAL, "out of memory");
  std::memcpy(out, v.c_str(), v.size() + 1);
  *id = out;
  return CALITAS_OK;
}

extern "C" int calitas_vcf_records(calitas_ctx* ctx, const char* vcf_path, const char* chrom, char** text, uint64_t* n_records) {
  if (!ctx) return CALITAS_EINVAL;
  if (!vcf_path || !text) return calitas_fail(ctx, CALITAS_EINVAL, "NULL argument");
  *text = nullptr;
  if (n_records) *n_records = 0;
  VarTable vcf;
  const std::string e = read_vcf(vcf_path, chrom, ctx->pool, vcf);
  vcf.publish(vcf.size(), true);
  if (!e.empty()) return calitas_fail(ctx, CALITAS_EIO, e);
  std::string out;
  char num[64];
  for (size_t i = 0; vcf.have(i); i++) {                          // (through have(), as the search walks the table)
    const Var& v = vcf[i];
    out += v.chrom; out += '\t';
    out += std::to_string(v.pos); out += '\t';
    out += std::to_string(v.end); out += '\t';
    out += v.id; out += '\t';
    out += v.ref; out += '\t';
    for (size_t a = 0; a < v.alts.size(); a++) { if (a) out += ','; out += v.alts[a]; }
    out += '\t';
    for (size_t a = 0; a < v.afs.size(); a++) { if (a) out += ','; std::snprintf(num, sizeof(num), "%.9g", (double)v.afs[a]); out += num; }
    out += '\n';
  }
  char* block = (char*)calitas_out_alloc(out.size() + 1);
  if (!block) return calitas_fail(ctx, CALITAS_EINVAL, "out of memory");
  std::memcpy(block, out.c_str(), out.size() + 1);
  *text = block;
  if (n_records) *n_records = vcf.size();
  return CALITAS_OK;
}
