// post.hpp -- host-side stages above the kernels: raw alignment -> GuideAlignment record, the per-window filter,
// removeOverlaps / sort / hits.txt rows.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "../../include/calitas_hip.h"
#include "common.hpp"
#include "parallel.hpp"
#include "refpack.hpp"

namespace calitas {

struct GuideHost {
  std::string protospacer;            // upper case
  std::vector<std::string> pams;      // lower case, as given
  bool pam5 = false;
  int cli_length = 0;
  std::string q;                      // aligner-space query: protospacer, or its reverse complement for a 5' PAM
  std::vector<std::string> pams_q;    // aligner-space PAMs
  // query string of an alignment in guide orientation (GuideAlignment.guide): protospacer+pam or pam+protospacer
  std::string query_for(int pam_index) const;
};

char complement_base(char c);
std::string revcomp_str(const std::string& s);
// Validates and converts; returns an empty string on success, else the error text.
std::string make_guide_host(const calitas_guide_t& g, GuideHost& out);

// Converts one raw kernel record into a GuideAlignment record (SequentialGuideAligner.scala:260-313, GuideAlignment.scala:10-50).
void raw_to_aln(const RawAln& r, const GuideHost& g, int64_t win_a, int64_t win_b, calitas_aln_t& out);

// SequentialGuideAligner.scala:315-320 on one window's alignments (forward list then reverse list, enumeration order).
void window_filter(const calitas_aln_t* alns, int n, int max_total_diffs, int max_overlap, std::vector<int>& kept);

// Padded strings in guide orientation.
void padded_strings(const PackedRef& ref, const GuideHost& g, const calitas_aln_t& a, std::string& pg, std::string& pa, std::string& pt);

// The pieces of a hits.txt row that are the same for every hit of one guide (RH:205-254), and the header line.
struct RowStrings {
  std::string header;                   // the 34 column names + newline
  std::string head;                     // guide_id \t protospacer \t genome_build \t
  std::string tail;                     // aligner \t version \t search_pam \t parameters \t time_stamp \n
  std::string proto_len;
  std::vector<std::string> query;       // per PAM index + 1: the query in guide orientation (GuideAlignment.guide)
  std::vector<std::string> pam_used;    // per PAM index + 1: its lower-case part (RH:229)
};
RowStrings make_row_strings(const PackedRef& ref, const GuideHost& g, const std::string& guide_id, const calitas_params_t& p,
                            const std::string& version, const std::string& time_stamp);

// Compact rows.  270 of a row's ~520 bytes are the same in every row of a call: head (guide_id, protospacer, genome_build) and tail
// (aligner ... time_stamp, ReferenceHit.scala:99-132).  Given row constants with an empty head and "\n" for a tail, the device's row
// kernels write `chromosome \t middle \n` per row -- the same kernels, half the bytes over PCIe -- and the library puts head and tail
// back on the host's worker pool while the next text is on the bus.
void stream_copy(char* dst, const char* src, size_t n);   // memcpy with non-temporal stores (large blocks nobody reads back soon)
RowStrings compact_row_strings(const RowStrings& full);
// The same with genome_build left in the rows: the variant branch's rows have one of their own ("<build>+variants" for a hit that
// touches a variant, ReferenceHit.scala:208), so only guide_id and protospacer are cut; *cut = what was cut from the head.
RowStrings compact_row_strings_keep_build(const RowStrings& full, std::string* cut);
// n bytes of compact rows (`rows` of them) -> full rows at out, which has room for n + rows * (head.size() + tail.size() - 1) bytes.
// Returns the bytes written, or (size_t)-1 when the text does not hold exactly `rows` newline-terminated rows.
size_t expand_rows(const char* compact, size_t n, uint64_t rows, const std::string& head, const std::string& tail, char* out, WorkerPool* pool);
// The same over a text that is still arriving (the pieces of a copy from the device): begin() wakes the workers, which expand what
// arrived() has announced -- the first `bytes` bytes of compact[] are there -- and end() has the caller join in and wait for the
// pieces in hand (complete = false: the copy failed, nothing more will arrive; the result is then (size_t)-1).  head, tail, compact
// and out stay valid until end() has returned.
struct RowExpansion;
std::shared_ptr<RowExpansion> expand_rows_begin(const char* compact, size_t n, uint64_t rows, const std::string& head, const std::string& tail,
                                                char* out, WorkerPool* pool);
void expand_rows_arrived(RowExpansion& job, size_t bytes);
size_t expand_rows_end(RowExpansion& job, bool complete);

// Text of the row of ext[e] (without the newline), for hits whose calitas_ext_hit_t::row is NULL: called from the worker pool, only for
// the hits removeOverlaps kept -- a caller with millions of hits of its own (the variant branch) builds no text for the ones that go.
typedef void (*ExtRowFn)(void* user, uint64_t e, std::string& row);

// hits.txt text for one guide's alignments: malloc'd, NUL-terminated (nullptr when out of memory).
char* hits_tsv(const PackedRef& ref, const GuideHost& g, const std::string& guide_id, const calitas_params_t& p,
               const calitas_aln_t* alns, uint64_t n, const std::string& version, const std::string& time_stamp,
               uint64_t* n_rows, WorkerPool* pool = nullptr, void* (*alloc)(size_t) = nullptr, const calitas_ext_hit_t* ext = nullptr,
               uint64_t n_ext = 0, ExtRowFn ext_row = nullptr, void* ext_user = nullptr);

}  // namespace calitas
