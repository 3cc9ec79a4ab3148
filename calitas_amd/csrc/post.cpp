// post.cpp -- host-side stages of the product path (see post.hpp).  Reference citations: SGA = SequentialGuideAligner.scala,
// GA = GuideAlignment.scala, RH = ReferenceHit.scala, SR = SearchReference.scala.
#include "post.hpp"
#include "tuning.hpp"

#include <immintrin.h>

#include <algorithm>
#include <cctype>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <map>
#include <memory>

namespace calitas {

char complement_base(char c) {  // fgbio Sequences.complement: IUPAC aware, case preserving (SGA:532)
  static const char* from = "ACGTUMKRYVBHDWSN";
  static const char* to   = "TGCAAKMYRBVDHWSN";
  char u = (char)std::toupper((unsigned char)c);
  const char* p = std::strchr(from, u);
  if (!p || u == 0) return c;
  char r = to[p - from];
  return std::islower((unsigned char)c) ? (char)std::tolower((unsigned char)r) : r;
}

std::string revcomp_str(const std::string& s) {
  std::string o(s.rbegin(), s.rend());
  for (auto& c : o) c = complement_base(c);
  return o;
}

std::string GuideHost::query_for(int pam_index) const {
  if (pam_index < 0 || pams.empty()) return protospacer;
  return pam5 ? pams[pam_index] + protospacer : protospacer + pams[pam_index];
}

std::string make_guide_host(const calitas_guide_t& g, GuideHost& out) {
  if (!g.protospacer || !*g.protospacer) return "guide has no protospacer";
  out = GuideHost();
  out.protospacer = g.protospacer;
  for (auto& c : out.protospacer) c = (char)std::toupper((unsigned char)c);  // SGA:64
  if ((int)out.protospacer.size() > CALITAS_MAX_PROTOSPACER) return "protospacer longer than 32 nt is not supported by the scan kernel";
  for (char c : out.protospacer) if (iupac_mask((unsigned char)c) == 0) return std::string("protospacer has a non-IUPAC character: ") + c;
  if (g.n_pams < 0 || g.n_pams > CALITAS_MAX_PAMS) return "too many PAMs";
  for (int i = 0; i < g.n_pams; i++) {
    std::string p = g.pams[i] ? g.pams[i] : "";
    for (auto& c : p) c = (char)std::tolower((unsigned char)c);          // SGA:65-66
    if (p.size() > CALITAS_MAX_PAM_LEN) return "PAM longer than 16 nt is not supported";
    for (char c : p) if (iupac_mask((unsigned char)c) == 0) return std::string("PAM has a non-IUPAC character: ") + c;
    out.pams.push_back(p);
  }
  // SGA:446 treats a single empty PAM as "no PAM"
  if (out.pams.size() == 1 && out.pams[0].empty()) out.pams.clear();
  for (auto& p : out.pams) if (p.empty()) return "empty PAM among several PAMs is not supported";
  out.pam5 = g.pam_is_5prime != 0 && !out.pams.empty();
  out.cli_length = g.cli_length > 0 ? g.cli_length : (int)(out.protospacer.size() + (out.pams.empty() ? 0 : out.pams[0].size()));
  out.q = out.pam5 ? revcomp_str(out.protospacer) : out.protospacer;
  for (auto& p : out.pams) out.pams_q.push_back(out.pam5 ? revcomp_str(p) : p);
  return "";
}

static inline bool consumes_target(uint8_t op) { return op == '=' || op == 'X' || op == 'D'; }
static inline bool consumes_query(uint8_t op) { return op == '=' || op == 'X' || op == 'I'; }

void raw_to_aln(const RawAln& r, const GuideHost& g, int64_t win_a, int64_t win_b, calitas_aln_t& out) {
  static const char OPC[4] = {'=', 'X', 'I', 'D'};
  // aligner-space ops: guide part (stored in traceback order), then the gap to the PAM, then the PAM (SGA:472-476)
  uint8_t ops[CALITAS_MAX_OPS];
  int n = 0;
  for (int i = r.n_ops - 1; i >= 0; i--) ops[n++] = (uint8_t)OPC[(r.ops[i >> 2] >> ((i & 3) * 2)) & 3];
  const int n_guide_ops = n;
  int pam_len = 0;
  if (r.pam >= 0) {
    pam_len = (int)g.pams_q[r.pam].size();
    for (int i = 0; i < r.offset; i++) ops[n++] = 'D';
    for (int i = 0; i < pam_len; i++) ops[n++] = ((r.pam_x >> i) & 1) ? 'X' : '=';
  }
  // GuideAlignment.apply (GA:21-31), always evaluated with the '+' rule in aligner space (SGA:264,281,297,302):
  // target letters left of the first / right of the last protospacer column.
  int first_q = -1, last_q = -1;
  for (int i = 0; i < n_guide_ops; i++) if (consumes_query(ops[i])) { if (first_q < 0) first_q = i; last_q = i; }
  int left_delta = 0, right_delta = 0;
  for (int i = 0; i < first_q; i++) if (consumes_target(ops[i])) left_delta++;
  for (int i = last_q + 1; i < n; i++) if (consumes_target(ops[i])) right_delta++;
  const int64_t start_s = (int64_t)r.t_start - 1;                        // SGA:515
  const int64_t end_s = (int64_t)r.t_end_guide + r.offset + pam_len;     // SGA:516 (targetEnd is inclusive)
  const int64_t gstart_s = start_s + left_delta, gend_s = end_s - right_delta;
  const int64_t wn = win_b - win_a;

  std::memset(&out, 0, sizeof(out));
  out.guide_index = r.guide;
  out.contig_index = (int32_t)r.contig;
  out.window_start = (int32_t)win_a;
  out.score = r.score;
  out.pam_index = (int8_t)r.pam;
  out.n_ops = (int16_t)n;
  if (r.dir == 0) {   // target as is: SGA:297 (3' PAM, '+') and SGA:281 (5' PAM, '-')
    out.start_offset = (int32_t)(win_a + start_s);  out.end_offset = (int32_t)(win_a + end_s);
    out.guide_start_offset = (int32_t)(win_a + gstart_s);  out.guide_end_offset = (int32_t)(win_a + gend_s);
  } else {            // reverse-complemented target: SGA:303-309 (3' PAM, '-') and SGA:271-274 (5' PAM, '+')
    out.start_offset = (int32_t)(win_a + wn - end_s);  out.end_offset = (int32_t)(win_a + wn - start_s);
    out.guide_start_offset = (int32_t)(win_a + wn - gend_s);  out.guide_end_offset = (int32_t)(win_a + wn - gstart_s);
  }
  const bool plus = g.pam5 ? (r.dir == 1) : (r.dir == 0);
  out.strand = plus ? '+' : '-';
  if (g.pam5) for (int i = 0; i < n; i++) out.ops[i] = ops[n - 1 - i];   // cigar.reverse / paddedAlignment.reverse SGA:267-269
  else std::memcpy(out.ops, ops, n);
}


void window_filter(const calitas_aln_t* alns, int n, int max_total_diffs, int max_overlap, std::vector<int>& kept) {
  kept.clear();
  // The caller passes the forward-strand list followed by the reverse-strand list (SGA:316); each is sorted on its own.
  // Scratch is per thread and reused: this runs once per window with candidates (~10^5 times per pass).
  static thread_local std::vector<int> idx[2], gaps, edits;
  idx[0].clear(); idx[1].clear();
  gaps.resize(n); edits.resize(n);
  for (int i = 0; i < n; i++) {
    int g = 0, e = 0;
    for (int k = 0; k < alns[i].n_ops; k++) { const uint8_t op = alns[i].ops[k]; e += op != '='; g += (op == 'I' || op == 'D'); }   // GA:100-101
    gaps[i] = g; edits[i] = e;
    idx[alns[i].strand == '-' ? 1 : 0].push_back(i);
  }
  for (int s = 0; s < 2; s++) {
    auto& v = idx[s];
    auto before = [&](int x, int y) {   // GA:125-129
      if (alns[x].score != alns[y].score) return alns[x].score > alns[y].score;
      return gaps[x] < gaps[y];
    };
    if (v.size() > 32) {
      std::stable_sort(v.begin(), v.end(), before);
    } else {                            // insertion sort: stable, and no temporary buffer from the heap
      for (size_t a = 1; a < v.size(); a++) {
        const int x = v[a];
        size_t b = a;
        while (b > 0 && before(x, v[b - 1])) { v[b] = v[b - 1]; b--; }
        v[b] = x;
      }
    }
    for (int i : v) {
      const calitas_aln_t& a = alns[i];
      if (edits[i] > max_total_diffs) continue;
      bool clash = false;
      for (int k : kept) {
        const calitas_aln_t& b = alns[k];
        if (b.strand != a.strand || b.contig_index != a.contig_index) continue;
        int o = std::min(a.end_offset, b.end_offset) - std::max(a.start_offset, b.start_offset);   // GA:119-122
        if (o > max_overlap) { clash = true; break; }
      }
      if (!clash) kept.push_back(i);
    }
  }
}

// Upper-cased target bases [start, end) in guide orientation: as is for '+', reverse complement for '-'.
static std::string target_bases(const PackedRef& ref, int contig, int64_t start, int64_t end, bool minus) {
  std::string s;
  const ContigInfo& c = ref.contigs[contig];
  for (int64_t p = start; p < end; p++) s += (p >= 0 && (uint64_t)p < c.len) ? ref.base_upper(c.gbase + (uint64_t)p) : 'N';
  return minus ? revcomp_str(s) : s;
}

void padded_strings(const PackedRef& ref, const GuideHost& g, const calitas_aln_t& a, std::string& pg, std::string& pa, std::string& pt) {
  const std::string q = g.query_for(a.pam_index);
  const std::string t = target_bases(ref, a.contig_index, a.start_offset, a.end_offset, a.strand == '-');
  pg.clear(); pa.clear(); pt.clear();
  size_t qi = 0, ti = 0;
  for (int i = 0; i < a.n_ops; i++) {
    switch (a.ops[i]) {
      case 'I': pg += q[qi++]; pa += '~'; pt += '-'; break;
      case 'D': pg += '-'; pa += '~'; pt += t[ti++]; break;
      case '=': pg += q[qi++]; pa += '|'; pt += t[ti++]; break;
      default:  pg += q[qi++]; pa += '.'; pt += t[ti++]; break;
    }
  }
}

namespace {

struct Lite {          // what removeOverlaps and the sort look at
  int contig; int start; int end; char strand; int score; uint64_t idx;
};

std::string core_parameters(const calitas_params_t& p, int max_total) {  // SR:496-508
  std::vector<std::string> kv = {
    "max-variants=" + std::to_string(p.max_variants), "window-size=" + std::to_string(p.window_size),
    "max-guide-diffs=" + std::to_string(p.max_guide_diffs), "max-pam-mismatches=" + std::to_string(p.max_pam_mismatches),
    "max-gaps-between-guide-and-pam=" + std::to_string(p.max_gaps_between_guide_and_pam),
    "max-total-diffs=" + std::to_string(max_total), "max-overlap=" + std::to_string(p.max_overlap),
    "guide-mismatch-net-cost=" + std::to_string(p.guide_mismatch_net_cost),
    "pam-mismatch-net-cost=" + std::to_string(p.pam_mismatch_net_cost),
    "genome-gap-net-cost=" + std::to_string(p.genome_gap_net_cost), "guide-gap-net-cost=" + std::to_string(p.guide_gap_net_cost)};
  std::sort(kv.begin(), kv.end());
  std::string s;
  for (size_t i = 0; i < kv.size(); i++) { if (i) s += ';'; s += kv[i]; }
  return s;
}

const char* const kColumns[34] = {
  "guide_id", "unpadded_guide_sequence", "genome_build", "chromosome", "coordinate_start", "coordinate_end", "strand",
  "unpadded_target_sequence", "ten_bases_5_prime", "ten_bases_3_prime", "pam_used", "variant_id", "variant_description",
  "variant_vcf", "allele_frequency", "score", "guide_mm", "guide_gaps", "guide_mm_plus_gaps", "pam_mm", "total_mm_plus_gaps",
  "padded_guide", "padded_alignment", "padded_target", "padded_extra_8_bases_5_prime", "padded_extra_8_bases_3_prime", "cigar",
  "unpadded_guide_sequence_length", "unpadded_target_sequence_length", "aligner", "aligner_version", "aligner_search_pam",
  "aligner_other_parameters", "time_stamp"};

}  // namespace

// Upper-cased bases [start, end) of a contig into out (N beyond the contig ends, RH:262-264); reverse complement when minus.
// Fast path: 2-bit decode; any exception bit in the covered mask words sends the span through base_upper().
static int fetch_span(const PackedRef& ref, int contig, int64_t start, int64_t end, bool minus, char* out) {
  const ContigInfo& c = ref.contigs[contig];
  const int n = (int)(end - start);
  bool plain = start >= 0 && (uint64_t)end <= c.len;
  if (plain) {
    const uint64_t g0 = c.gbase + (uint64_t)start, g1 = c.gbase + (uint64_t)end;
    for (uint64_t w = g0 >> 5; w <= (g1 - 1) >> 5 && n > 0; w++) if (ref.mask[w]) { plain = false; break; }
    if (plain) {
      for (int i = 0; i < n; i++) { const uint64_t g = g0 + i; out[i] = "ACGT"[(ref.codes[g >> 4] >> ((g & 15) * 2)) & 3]; }
    }
  }
  if (!plain)
    for (int i = 0; i < n; i++) { const int64_t p = start + i; out[i] = (p >= 0 && (uint64_t)p < c.len) ? ref.base_upper(c.gbase + (uint64_t)p) : 'N'; }
  if (minus) {
    for (int i = 0, j = n - 1; i < j; i++, j--) { char t = out[i]; out[i] = out[j]; out[j] = t; }
    for (int i = 0; i < n; i++) out[i] = complement_base(out[i]);
  }
  return n;
}

// GuideAlignment.count (GA:139-163) on padded char arrays of length len.
static int ga_count_raw(const char* pg, const char* pa, int len, bool lower, bool bothSides, bool mms, bool gaps) {
  auto is_lower = [](char c) { return c >= 'a' && c <= 'z'; };
  auto is_letter = [](char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z'); };
  int n = 0;
  for (int i = 0; i < len; i++) {
    if (mms && pa[i] == '.' && is_lower(pg[i]) == lower) { n++; continue; }
    if (!(gaps && pa[i] == '~')) continue;
    const char gb = pg[i];
    bool me = gb != '-' && is_lower(gb) == lower;
    if (!me) {
      int pi = i; while (pi > 0 && pg[pi] == '-') pi--;            // previousNonDash GA:168-172
      int ni = i; while (ni < len - 1 && pg[ni] == '-') ni++;      // nextNonDash GA:177-182
      const char prev = pg[pi], next = pg[ni];
      if (bothSides) me = (prev == '-' || is_lower(prev) == lower) && (next == '-' || is_lower(next) == lower);
      else me = (is_letter(prev) && is_lower(prev) == lower) || (is_letter(next) && is_lower(next) == lower);
    }
    if (me) n++;
  }
  return n;
}


RowStrings make_row_strings(const PackedRef& ref, const GuideHost& g, const std::string& guide_id, const calitas_params_t& p,
                            const std::string& version, const std::string& time_stamp) {
  const int max_total = p.max_total_diffs >= 0 ? p.max_total_diffs
                                               : p.max_guide_diffs + p.max_gaps_between_guide_and_pam + p.max_pam_mismatches;  // SR:493
  std::string search_pam;
  for (size_t i = 0; i < g.pams.size(); i++) { if (i) search_pam += ','; search_pam += g.pams[i]; }   // RH:207
  RowStrings rc;
  rc.head = guide_id + "\t" + g.protospacer + "\t" + ref.genome_build + "\t";
  rc.tail = std::string("CALITAS:SearchReference") + "\t" + version + "\t" + search_pam + "\t" + core_parameters(p, max_total) + "\t" +
            time_stamp + "\n";                                                                       // SR:522
  rc.proto_len = std::to_string(g.protospacer.size());
  for (int pi = -1; pi < (int)g.pams.size(); pi++) {
    std::string q = g.query_for(pi), used;
    for (char c : q) if (c >= 'a' && c <= 'z') used += c;
    rc.query.push_back(q); rc.pam_used.push_back(used);
  }
  for (int i = 0; i < 34; i++) { if (i) rc.header += '\t'; rc.header += kColumns[i]; }
  rc.header += '\n';
  return rc;
}

RowStrings compact_row_strings(const RowStrings& full) {
  RowStrings c = full;
  c.head.clear();
  c.tail = "\n";
  return c;
}

RowStrings compact_row_strings_keep_build(const RowStrings& full, std::string* cut) {
  RowStrings c = full;
  size_t at = full.head.find('\t');                             // head = guide_id \t protospacer \t genome_build \t
  if (at != std::string::npos) at = full.head.find('\t', at + 1);
  const size_t n = at == std::string::npos ? 0 : at + 1;
  if (cut) *cut = full.head.substr(0, n);
  c.head = full.head.substr(n);
  c.tail = "\n";
  return c;
}

namespace {

// Rows are assembled in a buffer that stays in the first-level cache and leave it as whole 64-byte lines through non-temporal stores:
// the output is written once and not read again by this thread (no read for ownership: a 176 MB text is otherwise 352 MB of memory
// traffic plus the 84 MB read).  The bytes before the first line boundary of the destination and behind the last go out as plain stores.
static void stream_lines_sse2(char* w, const char* src, size_t lines) {
  for (size_t i = 0; i < lines; i++, src += 64, w += 64) {
    __m128i* dst = reinterpret_cast<__m128i*>(w);
    _mm_stream_si128(dst, _mm_loadu_si128(reinterpret_cast<const __m128i*>(src)));
    _mm_stream_si128(dst + 1, _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + 16)));
    _mm_stream_si128(dst + 2, _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + 32)));
    _mm_stream_si128(dst + 3, _mm_loadu_si128(reinterpret_cast<const __m128i*>(src + 48)));
  }
}
__attribute__((target("avx2"))) static void stream_lines_avx2(char* w, const char* src, size_t lines) {
  for (size_t i = 0; i < lines; i++, src += 64, w += 64) {
    __m256i* dst = reinterpret_cast<__m256i*>(w);
    _mm256_stream_si256(dst, _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src)));
    _mm256_stream_si256(dst + 1, _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + 32)));
  }
}
static const bool wide_stores = __builtin_cpu_supports("avx2");


struct RowWriter {
  char* w;                                                    // where buf[0] belongs
  size_t fill = 0;
  alignas(64) char buf[4096];
  explicit RowWriter(char* at) : w(at) {}
  void put(const char* p, size_t len) {
    while (len) {
      const size_t c = std::min(len, sizeof(buf) - fill);
      std::memcpy(buf + fill, p, c);
      fill += c; p += c; len -= c;
      if (fill == sizeof(buf)) lines_out();
    }
  }
  void lines_out() {
    size_t off = 0;
    if ((uintptr_t)w & 63) {                                  // (once per writer: up to the first line boundary)
      off = std::min(fill, (size_t)(64 - ((uintptr_t)w & 63)));
      std::memcpy(w, buf, off);
      w += off;
    }
    const size_t lines = (fill - off) / 64;
    if (wide_stores) stream_lines_avx2(w, buf + off, lines); else stream_lines_sse2(w, buf + off, lines);
    w += lines * 64;
    off += lines * 64;
    if (off < fill) std::memmove(buf, buf + off, fill - off);
    fill -= off;
  }
  void finish() {
    lines_out();
    if (fill) { std::memcpy(w, buf, fill); w += fill; fill = 0; }
    _mm_sfence();
  }
};

}  // namespace

// memcpy with non-temporal stores: for tens of megabytes that nobody reads back soon (a contig's text on its way from the bounce buffer
// to its block), so that the destination's lines are written without being fetched first.
void stream_copy(char* dst, const char* src, size_t n) {
  const size_t head = std::min(n, (size_t)((64 - ((uintptr_t)dst & 63)) & 63));   // up to the destination's next 64-byte line
  if (head) { std::memcpy(dst, src, head); dst += head; src += head; n -= head; }
  const size_t lines = n / 64;
  if (lines) { if (wide_stores) stream_lines_avx2(dst, src, lines); else stream_lines_sse2(dst, src, lines); }
  const size_t done = lines * 64;
  if (n > done) std::memcpy(dst + done, src + done, n - done);
  _mm_sfence();
}

// The expansion as a job of pieces (WorkerPool::offer) over a text that may still be arriving.  A piece is the rows that START in one
// 64 KB stretch of the compact text.  Whoever arrives takes the next piece, waits until its bytes are there, counts its rows, learns
// from the piece before it how many rows precede it (and tells the piece behind it), and places its rows -- a row's place is its
// compact offset plus (head + tail - 1) bytes for every row before it.  Only the counting is a chain from piece to piece (a few
// microseconds each); the placing runs on as many workers as have arrived.  Nobody waits for a worker that has not arrived: it
// finds the pieces gone.  (Two WorkerPool::run passes with a static share per worker made one hg38-sized call in forty take 6-10 ms,
// profiles/r04_slow_calls.txt.)
struct RowExpansion : SharedJob {
  static constexpr size_t PIECE = 64u << 10;
  const char* compact = nullptr;
  size_t n = 0;
  char* out = nullptr;
  const char* hp = nullptr; const char* tp = nullptr;
  size_t H = 0, TL = 0, add = 0;
  std::string tail_head;                                      // what stands between the middles of two consecutive rows
  uint64_t max_rows = 0;
  size_t limit = 1, P = 0;
  std::atomic<size_t> avail{0};                               // bytes of compact[] that are there
  std::atomic<bool> gave_up{false};                           // the feeder's copy failed: nothing more will arrive
  std::atomic<bool> too_many{false};                          // more rows than max_rows: the rows beyond are not written
  std::atomic<size_t> joined{0}, next{0}, done{0};
  std::unique_ptr<std::atomic<uint64_t>[]> before;            // [P + 1] rows before piece k, plus one (0: not known yet)

  // first byte of the first row that starts at or after b (b > 0); waits for the bytes it has to look at.  n when there is none.
  size_t row_start(size_t b) {
    if (b >= n) return n;
    size_t from = b - 1;
    Backoff wait;
    for (;;) {
      const size_t have = avail.load(std::memory_order_acquire);
      if (have > from) {
        const char* nl = (const char*)std::memchr(compact + from, '\n', have - from);
        if (nl) return (size_t)(nl - compact) + 1;
        from = have;
      }
      if (have >= n || gave_up.load(std::memory_order_relaxed)) return n;
      wait.pause();
    }
  }

  bool exhausted() const override { return next.load(std::memory_order_relaxed) >= P || joined.load(std::memory_order_relaxed) >= limit; }
  void work() override {
    if (joined.fetch_add(1, std::memory_order_relaxed) >= limit) return;
    for (;;) {
      const size_t k = next.fetch_add(1, std::memory_order_relaxed);
      if (k >= P) return;
      const size_t first = k ? row_start(k * PIECE) : 0;
      const size_t last = row_start((k + 1) * PIECE);          // (every row of the piece ends at or before `last`, which has arrived)
      uint64_t c = 0;
      for (const char* p = compact + first; p < compact + last;) {
        const char* nl = (const char*)std::memchr(p, '\n', (size_t)(compact + last - p));
        if (!nl) break;                                       // (a text that does not end with a newline: the caller checks)
        c++; p = nl + 1;
      }
      Backoff wait;
      uint64_t b;
      while ((b = before[k].load(std::memory_order_acquire)) == 0) wait.pause();
      before[k + 1].store(b + c, std::memory_order_release);
      b -= 1;
      if (b + c > max_rows) too_many.store(true, std::memory_order_relaxed);
      else if (c && !gave_up.load(std::memory_order_relaxed)) {
        RowWriter rw(out + first + b * add);
        const char* p = compact + first;
        const char* const end = compact + last;
        rw.put(hp, H);
        while (p < end) {
          const char* nl = (const char*)std::memchr(p, '\n', (size_t)(end - p));
          if (!nl) break;
          rw.put(p, (size_t)(nl - p));
          p = nl + 1;
          if (p < end) rw.put(tail_head.data(), tail_head.size()); else rw.put(tp, TL);
        }
        rw.finish();
      }
      done.fetch_add(1, std::memory_order_release);
    }
  }
};

std::shared_ptr<RowExpansion> expand_rows_begin(const char* compact, size_t n, uint64_t rows, const std::string& head, const std::string& tail,
                                                char* out, WorkerPool* pool) {
  auto job = std::make_shared<RowExpansion>();
  job->compact = compact; job->n = n; job->out = out;
  job->hp = head.data(); job->tp = tail.data();
  job->H = head.size(); job->TL = tail.size(); job->add = head.size() + tail.size() - 1;
  job->tail_head = tail + head;
  job->max_rows = rows;
  // All workers (CALITAS_EXPAND_THREADS for experiments): 34.8 MB of rows in 3.8 / 2.0 / 1.0 / 0.55 / 0.35 ms on 1 / 2 / 4 / 8 / 16
  // of an MI355X box's cores (tools/expand_speed.py); a text below 1 MB is not worth the wake-ups.
  size_t T = pool && n >= (1u << 20) ? (size_t)pool->size() : 1;
  if (const char* e = TUNE_GET("CALITAS_EXPAND_THREADS"))    // (also for short texts: the tests run the job's hand-overs on small genomes)
    T = std::max<size_t>(1, std::min<size_t>(pool ? (size_t)pool->size() : 1, (size_t)std::atoi(e)));
  job->limit = T;
  job->P = (n + RowExpansion::PIECE - 1) / RowExpansion::PIECE;
  job->before.reset(new std::atomic<uint64_t>[job->P + 1]);
  for (size_t k = 0; k <= job->P; k++) job->before[k].store(k ? 0 : 1, std::memory_order_relaxed);
  if (T > 1 && job->P > 1) pool->offer(job);
  return job;
}

void expand_rows_arrived(RowExpansion& job, size_t bytes) { job.avail.store(std::min(bytes, job.n), std::memory_order_release); }

size_t expand_rows_end(RowExpansion& job, bool complete) {
  if (!complete) job.gave_up.store(true, std::memory_order_release);
  else job.avail.store(job.n, std::memory_order_release);
  job.work();
  Backoff wait;
  while (job.done.load(std::memory_order_acquire) < job.P) wait.pause();   // (pieces that were taken: their holders are running)
  if (!complete || job.too_many.load()) return (size_t)-1;
  if (job.n && job.compact[job.n - 1] != '\n') return (size_t)-1;
  const uint64_t rows = job.before[job.P].load(std::memory_order_acquire) - 1;
  if (rows != job.max_rows) return (size_t)-1;
  return job.n + (size_t)rows * job.add;
}

size_t expand_rows(const char* compact, size_t n, uint64_t rows, const std::string& head, const std::string& tail, char* out, WorkerPool* pool) {
  if (n && compact[n - 1] != '\n') return (size_t)-1;
  auto job = expand_rows_begin(compact, n, rows, head, tail, out, pool);
  return expand_rows_end(*job, true);
}

static inline char* put_int_p(char* w, long v) {
  char b[24]; int n = 0; const bool neg = v < 0; unsigned long u = neg ? (unsigned long)(-v) : (unsigned long)v;
  do { b[n++] = (char)('0' + u % 10); u /= 10; } while (u);
  if (neg) *w++ = '-';
  while (n) *w++ = b[--n];
  return w;
}
static inline char* put_mem(char* w, const char* p, size_t n) { std::memcpy(w, p, n); return w + n; }
static inline char* put_str(char* w, const std::string& s) { return put_mem(w, s.data(), s.size()); }

// Bases [from, to) of a forward-strand buffer that starts at contig offset lo, in guide orientation.
static inline char* put_bases(char* w, const char* fwd, int64_t lo, int64_t from, int64_t to, bool minus) {
  if (!minus) return put_mem(w, fwd + (from - lo), (size_t)(to - from));
  for (int64_t p = to - 1; p >= from; p--) *w++ = complement_base(fwd[p - lo]);
  return w;
}

// One hits.txt row (RH:210-254) written at w; returns the new write position.  The caller provides row_bound() bytes.
static char* write_row(char* w, const PackedRef& ref, const RowStrings& rc, const calitas_aln_t& a) {
  const bool minus = a.strand == '-';
  const std::string& q = rc.query[a.pam_index + 1];
  // one fetch covers the alignment and all four flanks (RH:213-216: 10 bases around the protospacer, 8 around the alignment)
  const int64_t lo = std::min<int64_t>((int64_t)a.start_offset - 8, (int64_t)a.guide_start_offset - 10);
  const int64_t hi = std::max<int64_t>((int64_t)a.end_offset + 8, (int64_t)a.guide_end_offset + 10);
  char fwd[CALITAS_MAX_OPS + 64];
  fetch_span(ref, a.contig_index, lo, hi, false, fwd);
  char t[CALITAS_MAX_OPS + 8], pg[CALITAS_MAX_OPS + 1], pa[CALITAS_MAX_OPS + 1], pt[CALITAS_MAX_OPS + 1];
  put_bases(t, fwd, lo, a.start_offset, a.end_offset, minus);
  const int n = a.n_ops;
  int qi = 0, ti = 0, mm = 0, gp = 0;
  for (int i = 0; i < n; i++) {   // Alignment.paddedString (SGA:511)
    switch (a.ops[i]) {
      case 'I': pg[i] = q[qi++]; pa[i] = '~'; pt[i] = '-'; gp++; break;
      case 'D': pg[i] = '-'; pa[i] = '~'; pt[i] = t[ti++]; gp++; break;
      case '=': pg[i] = q[qi++]; pa[i] = '|'; pt[i] = t[ti++]; break;
      default:  pg[i] = q[qi++]; pa[i] = '.'; pt[i] = t[ti++]; mm++; break;
    }
  }
  int ps = -1, pe = -1;           // unpaddedTargetWithoutPam GA:111-115
  for (int i = 0; i < n; i++) if (pg[i] >= 'A' && pg[i] <= 'Z') { if (ps < 0) ps = i; pe = i; }

  w = put_str(w, rc.head);
  w = put_str(w, ref.names[a.contig_index]); *w++ = '\t';
  w = put_int_p(w, a.guide_start_offset); *w++ = '\t';
  w = put_int_p(w, a.guide_end_offset); *w++ = '\t';
  *w++ = (char)a.strand; *w++ = '\t';
  int utn = 0;
  for (int i = ps; i >= 0 && i <= pe; i++) if (pt[i] != '-') { *w++ = pt[i]; utn++; }
  *w++ = '\t';
  // ten_bases_5_prime / 3_prime RH:227-228: left/right of the protospacer in genome orientation, swapped and reverse
  // complemented on the minus strand
  const int64_t gs = a.guide_start_offset, ge = a.guide_end_offset, as = a.start_offset, ae = a.end_offset;
  if (!minus) { w = put_bases(w, fwd, lo, gs - 10, gs, false); *w++ = '\t'; w = put_bases(w, fwd, lo, ge, ge + 10, false); }
  else        { w = put_bases(w, fwd, lo, ge, ge + 10, true);  *w++ = '\t'; w = put_bases(w, fwd, lo, gs - 10, gs, true); }
  *w++ = '\t';
  w = put_str(w, rc.pam_used[a.pam_index + 1]); *w++ = '\t';
  w = put_mem(w, "\t\t\t\t", 4);                        // variant_id, variant_description, variant_vcf, allele_frequency: None
  w = put_int_p(w, a.score); *w++ = '\t';
  const int gmm = ga_count_raw(pg, pa, n, false, false, true, false);    // guide_mm GA:103
  const int ggp = ga_count_raw(pg, pa, n, false, false, false, true);    // guide_gaps GA:104
  w = put_int_p(w, gmm); *w++ = '\t';
  w = put_int_p(w, ggp); *w++ = '\t';
  w = put_int_p(w, gmm + ggp); *w++ = '\t';             // guide_mm_plus_gaps GA:105 (every column is counted once)
  w = put_int_p(w, ga_count_raw(pg, pa, n, true, true, true, false)); *w++ = '\t';   // pam_mm GA:106
  w = put_int_p(w, mm + gp); *w++ = '\t';               // total_mm_plus_gaps = edits GA:101
  w = put_mem(w, pg, n); *w++ = '\t'; w = put_mem(w, pa, n); *w++ = '\t'; w = put_mem(w, pt, n); *w++ = '\t';
  if (!minus) { w = put_bases(w, fwd, lo, as - 8, as, false); *w++ = '\t'; w = put_bases(w, fwd, lo, ae, ae + 8, false); }   // RH:243-244
  else        { w = put_bases(w, fwd, lo, ae, ae + 8, true);  *w++ = '\t'; w = put_bases(w, fwd, lo, as - 8, as, true); }
  *w++ = '\t';
  for (int i = 0; i < n;) {                             // Cigar.coalesce + toString
    int j = i; while (j < n && a.ops[j] == a.ops[i]) j++;
    w = put_int_p(w, j - i); *w++ = (char)a.ops[i]; i = j;
  }
  *w++ = '\t';
  w = put_str(w, rc.proto_len); *w++ = '\t';
  w = put_int_p(w, utn); *w++ = '\t';
  w = put_str(w, rc.tail);
  return w;
}

static size_t row_bound(const PackedRef& ref, const RowStrings& rc) {
  size_t name = 0;
  for (auto& n : ref.names) name = std::max(name, n.size());
  return rc.head.size() + rc.tail.size() + name + 5 * (CALITAS_MAX_OPS + 4) + 256;
}

char* hits_tsv(const PackedRef& ref, const GuideHost& g, const std::string& guide_id, const calitas_params_t& p,
               const calitas_aln_t* alns, uint64_t n, const std::string& version, const std::string& time_stamp,
               uint64_t* n_rows, WorkerPool* pool, void* (*alloc)(size_t), const calitas_ext_hit_t* ext, uint64_t n_ext, ExtRowFn ext_row,
               void* ext_user) {
  WorkerPool serial(1);
  if (!pool) pool = &serial;
  const bool trace = TUNE_GET("CALITAS_TRACE") != nullptr;
  auto tnow = [] { return std::chrono::steady_clock::now(); };
  auto t_start = tnow();
  auto by_hit_order = [](const Lite& x, const Lite& y) {   // RH:284
    if (x.contig != y.contig) return x.contig < y.contig;
    if (x.start != y.start) return x.start < y.start;
    if (x.strand != y.strand) return x.strand < y.strand;   // '+' (0x2B) sorts before '-' (0x2D)
    return -x.score < -y.score;
  };
  auto overlap = [](const Lite& x, const Lite& y) {        // RH:141-144
    if (x.contig != y.contig) return 0;
    return std::max(0, std::min(x.end, y.end) - std::max(x.start, y.start));
  };
  // ---- removeOverlaps (SR:653-675): groups are (chromosome, strand); the contig index doubles as the dictionary
  // index.  Groups are independent, so they are processed in parallel; arrival order inside a group is preserved.
  const int n_contigs = (int)ref.contigs.size();
  std::vector<std::vector<Lite>> groups((size_t)n_contigs * 2);
  auto lite_of = [&](uint64_t i) {
    const calitas_aln_t& a = alns[i];
    int tlen = 0;
    for (int k = 0; k < a.n_ops; k++) if (consumes_target(a.ops[k])) tlen++;
    return Lite{a.contig_index, a.guide_start_offset, a.guide_start_offset + tlen - 1, (char)a.strand, a.score, i};   // RH:135-138
  };
  if (n < (1u << 16) || pool->size() == 1) {
    for (uint64_t i = 0; i < n; i++) { const Lite l = lite_of(i); groups[(size_t)l.contig * 2 + (l.strand == '-' ? 1 : 0)].push_back(l); }
  } else {
    // tens of millions of alignments (a PAM-less search at max-guide-diffs 8): every worker groups a consecutive block of them, then the
    // blocks' lists are joined per group in block order -- arrival order inside a group is what it was (1.7 s of a 2.0-s stage at 4e7)
    const size_t T = (size_t)pool->size();
    std::vector<std::vector<std::vector<Lite>>> local(T, std::vector<std::vector<Lite>>(groups.size()));
    pool->for_blocks((size_t)n, [&](size_t b, size_t e, int tid) {
      auto& mine = local[(size_t)tid];
      for (size_t i = b; i < e; i++) { const Lite l = lite_of(i); mine[(size_t)l.contig * 2 + (l.strand == '-' ? 1 : 0)].push_back(l); }
    });
    std::atomic<size_t> nextg(0);
    pool->run([&](int) {
      for (;;) {
        const size_t gi = nextg.fetch_add(1);
        if (gi >= groups.size()) break;
        size_t total = 0;
        for (size_t t = 0; t < T; t++) total += local[t][gi].size();
        groups[gi].reserve(total);
        for (size_t t = 0; t < T; t++) { groups[gi].insert(groups[gi].end(), local[t][gi].begin(), local[t][gi].end()); std::vector<Lite>().swap(local[t][gi]); }
      }
    });
  }
  // Hits built by the caller (variant windows, SR:570-630): idx >= n refers to ext[idx - n].  They arrive after the reference
  // hits (SR:622) and group by (chromosome, strand, variant_description) like every other hit (SR:656).
  std::map<std::string, size_t> desc_group;
  for (uint64_t e = 0; e < n_ext; e++) {
    const calitas_ext_hit_t& x = ext[e];
    if (x.contig_index < 0 || x.contig_index >= n_contigs) continue;
    const Lite l{x.contig_index, x.coordinate_start, x.end, (char)x.strand, x.score, n + e};
    if (!x.variant_description || !*x.variant_description) { groups[(size_t)x.contig_index * 2 + (x.strand == '-' ? 1 : 0)].push_back(l); continue; }
    const std::string desc = x.variant_description;
    const std::string key = std::to_string(x.contig_index) + ":" + (char)x.strand + ":" + desc;
    auto it = desc_group.find(key);
    if (it == desc_group.end()) { it = desc_group.emplace(key, groups.size()).first; groups.emplace_back(); }
    groups[it->second].push_back(l);
  }
  std::vector<std::vector<Lite>> kept_by_contig(n_contigs);
  {
    std::atomic<size_t> next(0);
    std::vector<std::vector<Lite>> kept_group(groups.size());
    pool->run([&](int) {
      for (;;) {
        size_t gi = next.fetch_add(1);
        if (gi >= groups.size()) break;
        auto& hs = groups[gi];
        if (hs.empty()) continue;
        std::stable_sort(hs.begin(), hs.end(), by_hit_order);
        auto& keep = kept_group[gi];
        size_t i = 0;
        while (i < hs.size()) {
          const Lite hit = hs[i++];
          while (i < hs.size() && overlap(hs[i], hit) >= p.max_overlap && hs[i].score <= hit.score) i++;
          if (i >= hs.size() || overlap(hs[i], hit) < p.max_overlap) keep.push_back(hit);
        }
      }
    });
    // final ReferenceHit.sort (SR:647): per contig, the groups' keeper lists are concatenated ('+', '-', then the variant
    // groups in order of first appearance) and stably sorted.  Equal keys can only meet across groups when one of them is
    // a variant group; the reference leaves that order to a hash map (SR:656), here it is the concatenation order.
    std::vector<std::vector<size_t>> groups_of_contig(n_contigs);
    for (size_t gi = 0; gi < groups.size(); gi++) if (!kept_group[gi].empty()) groups_of_contig[kept_group[gi][0].contig].push_back(gi);
    std::atomic<size_t> nextc(0);
    pool->run([&](int) {
      for (;;) {
        size_t c = nextc.fetch_add(1);
        if (c >= (size_t)n_contigs) break;
        auto& out = kept_by_contig[c];
        for (size_t gi : groups_of_contig[c]) out.insert(out.end(), kept_group[gi].begin(), kept_group[gi].end());
        std::stable_sort(out.begin(), out.end(), by_hit_order);
      }
    });
  }
  std::vector<Lite> keepers;
  for (auto& v : kept_by_contig) keepers.insert(keepers.end(), v.begin(), v.end());
  auto t_dedup = tnow();

  // ---- rows (RH:210-254), built in parallel blocks and concatenated in order ----
  const RowStrings rc = make_row_strings(ref, g, guide_id, p, version, time_stamp);
  const std::string& header = rc.header;
  const size_t BLOCK = 512;
  const size_t n_blocks = (keepers.size() + BLOCK - 1) / BLOCK;
  const size_t bound = row_bound(ref, rc);
  struct Part { std::unique_ptr<char[]> buf; size_t len = 0; };
  std::vector<Part> parts(n_blocks);
  {
    std::atomic<size_t> next(0);
    pool->run([&](int) {
      for (;;) {
        size_t b = next.fetch_add(1);
        if (b >= n_blocks) break;
        const size_t e = std::min(keepers.size(), (b + 1) * BLOCK);
        size_t need = 0;
        std::vector<std::string> made;                     // rows of the caller's hits that come without text
        for (size_t i = b * BLOCK; i < e; i++) {
          if (keepers[i].idx < n) { need += bound; continue; }
          const calitas_ext_hit_t& x = ext[keepers[i].idx - n];
          if (x.row) { need += std::strlen(x.row) + 1; continue; }
          made.emplace_back();
          if (ext_row) ext_row(ext_user, keepers[i].idx - n, made.back());
          need += made.back().size() + 1;
        }
        parts[b].buf.reset(new char[need]);
        char* w = parts[b].buf.get();
        size_t mi = 0;
        for (size_t i = b * BLOCK; i < e; i++) {
          if (keepers[i].idx < n) { w = write_row(w, ref, rc, alns[keepers[i].idx]); continue; }
          const char* r = ext[keepers[i].idx - n].row;
          if (r) w = put_mem(w, r, std::strlen(r));
          else { w = put_mem(w, made[mi].data(), made[mi].size()); mi++; }
          *w++ = '\n';
        }
        parts[b].len = (size_t)(w - parts[b].buf.get());
      }
    });
  }
  size_t total = header.size();
  std::vector<size_t> offs(n_blocks);
  for (size_t b = 0; b < n_blocks; b++) { offs[b] = total; total += parts[b].len; }
  auto t_rows = tnow();
  char* out = (char*)(alloc ? alloc(total + 1) : std::malloc(total + 1));
  if (!out) return nullptr;
  out[total] = 0;
  std::memcpy(out, header.data(), header.size());
  {
    std::atomic<size_t> next(0);
    pool->run([&](int) {
      for (;;) {
        size_t b = next.fetch_add(1);
        if (b >= n_blocks) break;
        std::memcpy(out + offs[b], parts[b].buf.get(), parts[b].len);
        parts[b].buf.reset();
      }
    });
  }
  if (n_rows) *n_rows = keepers.size();
  if (trace) {
    auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    std::fprintf(stderr, "[calitas] hits_tsv: removeOverlaps+sort %.2f ms, rows %.2f ms, concat %.2f ms (%zu rows, %zu bytes)\n",
                 ms(t_start, t_dedup), ms(t_dedup, t_rows), ms(t_rows, tnow()), keepers.size(), total);
  }
  return out;
}

}  // namespace calitas
