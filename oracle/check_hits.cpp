// check_hits.cpp -- TEST INFRASTRUCTURE (like the rest of oracle/): size-independent properties of a finished hits.txt that the
// reference's own post-processing guarantees, checked on texts too large to compare row by row against the CPU restatement
// (BASELINE config 5 at its stated size: 4.1e7 rows, 21.8 GB).  Only tests/ and bench.py's parity leg load this.
//
//   * ReferenceHit.sort (ReferenceHit.scala:284): rows ascend by (index of chromosome in the dictionary, coordinate_start, strand,
//     -score);
//   * removeOverlaps (SearchReference.scala:653-675): inside a group -- chromosome : strand : variant_description, SR:656 -- two
//     consecutive kept hits overlap by less than maxOverlap, with ReferenceHit.overlap / end as RH:135-144 define them
//     (end = coordinate_start + bases the cigar consumes on the target - 1; overlap = min(ends) - max(starts), not below 0);
//   * every row has the 34 columns of RH:99-132.
// The text is cut into byte ranges on line starts, one per thread; the ranges are stitched afterwards (order across a cut, the last
// hit of every group before a cut against the first one behind it).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace {

struct Hit { int contig = -1, start = 0, end = 0, score = 0; char strand = 0; };
struct Key { int contig, start; char strand; int neg_score; };
inline bool key_less(const Key& a, const Key& b) {
  if (a.contig != b.contig) return a.contig < b.contig;
  if (a.start != b.start) return a.start < b.start;
  if (a.strand != b.strand) return a.strand < b.strand;
  return a.neg_score < b.neg_score;
}
inline int overlap(const Hit& a, const Hit& b) { return a.contig != b.contig ? 0 : std::max(0, std::min(a.end, b.end) - std::max(a.start, b.start)); }

struct Range {
  uint64_t rows = 0, with_variant = 0, bad_order = 0, bad_overlap = 0, bad_columns = 0;
  bool any = false;
  Key first{}, last{};
  Hit ref_first[2], ref_last[2];                               // the group without a description, per strand ('+', '-')
  std::unordered_map<std::string, std::pair<Hit, Hit>> desc;   // contig \t strand \t description -> first and last hit of the range
};

inline int to_int(const char* b, const char* e) {
  bool neg = b < e && *b == '-';
  if (neg) b++;
  long v = 0;
  for (; b < e; b++) v = v * 10 + (*b - '0');
  return (int)(neg ? -v : v);
}

void scan(const char* text, const char* b, const char* e, const std::unordered_map<std::string, int>& contig_of, int max_overlap, Range& r) {
  (void)text;
  std::string name, key;
  while (b < e) {
    const char* nl = (const char*)std::memchr(b, '\n', (size_t)(e - b));
    const char* le = nl ? nl : e;
    // columns (0-based): 3 chromosome, 4 coordinate_start, 6 strand, 12 variant_description, 15 score, 26 cigar
    const char* f[35];
    int nf = 0;
    f[nf++] = b;
    for (const char* p = b; nf < 35;) {
      const char* t = (const char*)std::memchr(p, '\t', (size_t)(le - p));
      if (!t) break;
      p = t + 1;
      f[nf++] = p;
    }
    if (nf != 34) { r.bad_columns++; b = le + 1; continue; }
    auto fe = [&](int k) { return k + 1 < nf ? f[k + 1] - 1 : le; };
    name.assign(f[3], fe(3));
    auto it = contig_of.find(name);
    Hit h;
    h.contig = it == contig_of.end() ? -2 : it->second;
    if (h.contig < 0) r.bad_columns++;
    h.start = to_int(f[4], fe(4));
    h.strand = *f[6];
    h.score = to_int(f[15], fe(15));
    int tlen = 0;
    for (const char* p = f[26]; p < fe(26);) {                 // cigar: lengths of = X D (the ops that consume the target)
      int n = 0;
      while (p < fe(26) && *p >= '0' && *p <= '9') n = n * 10 + (*p++ - '0');
      const char op = p < fe(26) ? *p++ : 0;
      if (op != 'I') tlen += n;
    }
    h.end = h.start + tlen - 1;
    const Key k{h.contig, h.start, h.strand, -h.score};
    if (r.any && key_less(k, r.last)) r.bad_order++;
    if (!r.any) { r.first = k; r.any = true; }
    r.last = k;
    r.rows++;
    const bool has_desc = fe(12) > f[12];
    if (!has_desc) {
      const int s = h.strand == '-' ? 1 : 0;
      if (r.ref_last[s].contig >= 0 && overlap(h, r.ref_last[s]) >= max_overlap) r.bad_overlap++;
      if (r.ref_first[s].contig < 0) r.ref_first[s] = h;
      r.ref_last[s] = h;
    } else {
      r.with_variant++;
      key.assign(name); key += '\t'; key += h.strand; key += '\t'; key.append(f[12], fe(12));
      auto g = r.desc.find(key);
      if (g == r.desc.end()) r.desc.emplace(key, std::make_pair(h, h));
      else {
        if (overlap(h, g->second.second) >= max_overlap) r.bad_overlap++;
        g->second.second = h;
      }
    }
    b = le + 1;
  }
}

}  // namespace

extern "C" {

// text[0..n): a hits.txt with its header line.  names: the dictionary's contig names, '\n'-separated.  out[0..6) = rows, rows with a
// variant_description, rows out of ReferenceHit.sort order, pairs of consecutive kept hits of a group that overlap by >= max_overlap,
// rows with the wrong number of columns or an unknown chromosome, threads used.  Returns 0, or -1 without a header line.
int oracle_check_hits_text(const char* text, uint64_t n, const char* names, int max_overlap, int threads, uint64_t* out) {
  for (int i = 0; i < 6; i++) out[i] = 0;
  const char* end = text + n;
  const char* body = (const char*)std::memchr(text, '\n', (size_t)n);
  if (!body) return -1;
  body++;
  std::unordered_map<std::string, int> contig_of;
  {
    int idx = 0;
    for (const char* p = names; *p;) {
      const char* q = std::strchr(p, '\n');
      const size_t len = q ? (size_t)(q - p) : std::strlen(p);
      if (len) contig_of.emplace(std::string(p, len), idx++);
      p += len + (q ? 1 : 0);
    }
  }
  const int T = std::max(1, std::min(threads, 64));
  std::vector<const char*> cut((size_t)T + 1, end);
  cut[0] = body;
  for (int t = 1; t < T; t++) {
    const char* p = body + (uint64_t)(end - body) * (uint64_t)t / (uint64_t)T;
    p = std::max(p, cut[(size_t)t - 1]);
    if (p > body && p < end) { const char* nl = (const char*)std::memchr(p - 1, '\n', (size_t)(end - (p - 1))); p = nl ? nl + 1 : end; }
    cut[(size_t)t] = p;
  }
  std::vector<Range> ranges((size_t)T);
  std::vector<std::thread> th;
  for (int t = 0; t < T; t++) th.emplace_back([&, t] { scan(text, cut[(size_t)t], cut[(size_t)t + 1], contig_of, max_overlap, ranges[(size_t)t]); });
  for (auto& x : th) x.join();
  // stitch
  Range all;
  for (auto& r : ranges) {
    all.rows += r.rows; all.with_variant += r.with_variant; all.bad_order += r.bad_order; all.bad_overlap += r.bad_overlap; all.bad_columns += r.bad_columns;
    if (!r.any) continue;
    if (all.any && key_less(r.first, all.last)) all.bad_order++;
    all.any = true; all.last = r.last;
    for (int s = 0; s < 2; s++) {
      if (r.ref_first[s].contig < 0) continue;
      if (all.ref_last[s].contig >= 0 && overlap(r.ref_first[s], all.ref_last[s]) >= max_overlap) all.bad_overlap++;
      all.ref_last[s] = r.ref_last[s];
    }
    for (auto& kv : r.desc) {
      auto g = all.desc.find(kv.first);
      if (g == all.desc.end()) all.desc.emplace(kv.first, kv.second);
      else {
        if (overlap(kv.second.first, g->second.second) >= max_overlap) all.bad_overlap++;
        g->second.second = kv.second.second;
      }
    }
  }
  out[0] = all.rows; out[1] = all.with_variant; out[2] = all.bad_order; out[3] = all.bad_overlap; out[4] = all.bad_columns; out[5] = (uint64_t)T;
  return 0;
}

}  // extern "C"
