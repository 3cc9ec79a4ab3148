// fasta.cpp -- FASTA (+ sequence dictionary) reader for the host side of the product path.
// Replaces htsjdk's ReferenceSequenceFile / SAMSequenceDictionaryExtractor as used at SearchReference.scala:34-49,478-484
// and ReferenceHit.scala:168,208: whole contigs in file order, names cut at the first whitespace, the .dict (when present)
// supplies the assembly tag.  Unlike the reference, a missing .dict is tolerated (genome_build = "unknown").
#include "fasta.hpp"

#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

namespace calitas {

std::string read_fasta(const std::string& path, FastaData& out) {
  out = FastaData();
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return "cannot open " + path;
  std::vector<char> buf(1 << 24);
  std::string carry;      // partial header line across buffer boundaries
  bool in_header = false;
  size_t got;
  while ((got = std::fread(buf.data(), 1, buf.size(), f)) > 0) {
    const char* p = buf.data();
    const char* end = p + got;
    while (p < end) {
      if (in_header) {
        const char* nl = (const char*)std::memchr(p, '\n', end - p);
        carry.append(p, nl ? nl - p : end - p);
        if (!nl) { p = end; break; }
        p = nl + 1;
        in_header = false;
        if (!carry.empty() && carry.back() == '\r') carry.pop_back();
        size_t sp = carry.find_first_of(" \t");
        out.names.push_back(sp == std::string::npos ? carry : carry.substr(0, sp));
        out.seqs.emplace_back();
        carry.clear();
        continue;
      }
      if (*p == '>') { in_header = true; p++; continue; }
      const char* nl = (const char*)std::memchr(p, '\n', end - p);
      const char* stop = nl ? nl : end;
      // a '>' can only start a record at the beginning of a line; sequence lines never contain it
      if (stop > p && !out.seqs.empty()) {
        size_t len = stop - p;
        if (stop[-1] == '\r') len--;
        out.seqs.back().append(p, len);
      }
      p = nl ? nl + 1 : end;
    }
  }
  std::fclose(f);
  if (in_header) {  // header on the last line without a newline
    size_t sp = carry.find_first_of(" \t");
    out.names.push_back(sp == std::string::npos ? carry : carry.substr(0, sp));
    out.seqs.emplace_back();
  }
  if (out.names.empty()) return "no sequences in " + path;

  // sequence dictionary: <stem>.dict or <path>.dict
  std::vector<std::string> cands;
  size_t dot = path.find_last_of('.');
  if (dot != std::string::npos) cands.push_back(path.substr(0, dot) + ".dict");
  cands.push_back(path + ".dict");
  for (auto& d : cands) {
    std::ifstream di(d);
    if (!di) continue;
    std::string line;
    while (std::getline(di, line)) {
      if (line.compare(0, 3, "@SQ") != 0) continue;
      std::stringstream ss(line);
      std::string fld;
      while (std::getline(ss, fld, '\t'))
        if (fld.compare(0, 3, "AS:") == 0 && out.genome_build == "unknown") out.genome_build = fld.substr(3);
    }
    break;
  }
  return "";
}

}  // namespace calitas
