// cli.cpp -- `calitas SearchReference ...` on the MI355X path: the flag surface of the reference tool
// (SearchReference.scala:452-470) over the C ABI of include/calitas_hip.h.  The Scala CLI stays the intended host in
// production; this binary is the same host logic for boxes without a JVM.
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/calitas_hip.h"

static void usage() {
  std::fprintf(stderr,
    "usage: calitas SearchReference -i GUIDEpam -I guide-id -r ref.fa [-o hits.txt] [-x aux-pam ...]\n"
    "         [-w window-size=1000] [-d max-guide-diffs=5] [-p max-pam-mismatches=1] [-g max-gaps-between-guide-and-pam=3]\n"
    "         [-D max-total-diffs] [-O max-overlap=10] [-m guide-mismatch-net-cost=-120] [-M pam-mismatch-net-cost=-260]\n"
    "         [-b genome-gap-net-cost=-122] [-B guide-gap-net-cost=-121] [-c chrom] [-t threads (ignored)]\n"
    "         [-v variants.vcf[.gz]] [-V max-variants=16]\n"
    "         [--device N]\n");
}

static std::string long_to_short(const std::string& a) {
  static const char* map[][2] = {
    {"--guide", "-i"}, {"--guide-id", "-I"}, {"--auxiliary-pams", "-x"}, {"--ref", "-r"}, {"--variants", "-v"}, {"--max-variants", "-V"},
    {"--output", "-o"}, {"--threads", "-t"}, {"--window-size", "-w"}, {"--max-guide-diffs", "-d"}, {"--max-pam-mismatches", "-p"},
    {"--max-gaps-between-guide-and-pam", "-g"}, {"--max-total-diffs", "-D"}, {"--max-overlap", "-O"},
    {"--guide-mismatch-net-cost", "-m"}, {"--pam-mismatch-net-cost", "-M"}, {"--genome-gap-net-cost", "-b"},
    {"--guide-gap-net-cost", "-B"}, {"--chrom", "-c"}};
  for (auto& m : map) if (a == m[0]) return m[1];
  return a;
}

int main(int argc, char** argv) {
  if (argc < 2 || std::strcmp(argv[1], "SearchReference") != 0) { usage(); return 2; }
  std::string guide, guide_id, ref, output, chrom, variants;
  std::vector<std::string> aux;
  calitas_params_t p{};
  p.window_size = 1000; p.max_guide_diffs = 5; p.max_pam_mismatches = 1; p.max_gaps_between_guide_and_pam = 3; p.max_total_diffs = -1;
  p.max_overlap = 10; p.guide_mismatch_net_cost = -120; p.pam_mismatch_net_cost = -260; p.genome_gap_net_cost = -122;
  p.guide_gap_net_cost = -121; p.chrom_index = -1; p.eqx_by_score = 0; p.max_variants = 16;
  int device = 0;
  for (int i = 2; i < argc; i++) {
    std::string a = argv[i], val;
    size_t eq = a.find('=');
    if (a.compare(0, 2, "--") == 0 && eq != std::string::npos) { val = a.substr(eq + 1); a = a.substr(0, eq); }
    a = long_to_short(a);
    auto next = [&]() -> std::string {
      if (!val.empty()) return val;
      if (i + 1 >= argc) { usage(); std::exit(2); }
      return argv[++i];
    };
    if (a == "-i") guide = next();
    else if (a == "-I") guide_id = next();
    else if (a == "-r") ref = next();
    else if (a == "-o") output = next();
    else if (a == "-c") chrom = next();
    else if (a == "-x") { aux.push_back(next()); while (i + 1 < argc && argv[i + 1][0] != '-') aux.push_back(argv[++i]); }
    else if (a == "-w") p.window_size = std::atoi(next().c_str());
    else if (a == "-d") p.max_guide_diffs = std::atoi(next().c_str());
    else if (a == "-p") p.max_pam_mismatches = std::atoi(next().c_str());
    else if (a == "-g") p.max_gaps_between_guide_and_pam = std::atoi(next().c_str());
    else if (a == "-D") p.max_total_diffs = std::atoi(next().c_str());
    else if (a == "-O") p.max_overlap = std::atoi(next().c_str());
    else if (a == "-m") p.guide_mismatch_net_cost = std::atoi(next().c_str());
    else if (a == "-M") p.pam_mismatch_net_cost = std::atoi(next().c_str());
    else if (a == "-b") p.genome_gap_net_cost = std::atoi(next().c_str());
    else if (a == "-B") p.guide_gap_net_cost = std::atoi(next().c_str());
    else if (a == "-V") p.max_variants = std::atoi(next().c_str());
    else if (a == "-t") (void)next();
    else if (a == "--device") device = std::atoi(next().c_str());
    else if (a == "-v") variants = next();
    else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); usage(); return 2; }
  }
  if (guide.empty() || guide_id.empty() || ref.empty()) { usage(); return 2; }

  // Guide.apply(sequence, auxPams): split by case (SequentialGuideAligner.scala:81-107)
  std::vector<std::string> parts;
  for (size_t i = 0; i < guide.size();) {
    bool lower = std::islower((unsigned char)guide[i]) != 0;
    size_t j = i;
    while (j < guide.size() && (std::islower((unsigned char)guide[j]) != 0) == lower) j++;
    parts.push_back(guide.substr(i, j - i));
    i = j;
  }
  if (parts.empty() || parts.size() > 2) { std::fprintf(stderr, "Invalid Guide sequence %s.\n", guide.c_str()); return 1; }
  if (parts.size() == 1 && !std::isupper((unsigned char)parts[0][0])) { std::fprintf(stderr, "Guide sequence cannot be all lower case.\n"); return 1; }
  if (!aux.empty() && parts.size() != 2) { std::fprintf(stderr, "Cannot provide auxiliary PAMs without providing a PAM in the guide sequence.\n"); return 1; }
  for (auto& x : aux) for (char c : x) if (std::isupper((unsigned char)c)) { std::fprintf(stderr, "All PAMs must be lower case.\n"); return 1; }
  std::string proto;
  std::vector<std::string> pams;
  int pam5 = 0;
  if (parts.size() == 1) proto = parts[0];
  else if (std::isupper((unsigned char)parts[0][0])) { proto = parts[0]; pams.push_back(parts[1]); }
  else { proto = parts[1]; pams.push_back(parts[0]); pam5 = 1; }
  for (auto& x : aux) pams.push_back(x);
  std::vector<const char*> pam_ptrs;
  for (auto& s : pams) pam_ptrs.push_back(s.c_str());
  calitas_guide_t g;
  g.protospacer = proto.c_str(); g.n_pams = (int32_t)pams.size(); g.pams = pam_ptrs.empty() ? nullptr : pam_ptrs.data();
  g.pam_is_5prime = pam5; g.cli_length = (int32_t)guide.size();

  calitas_ctx* ctx = nullptr;
  if (calitas_create(device, &ctx) != CALITAS_OK) { std::fprintf(stderr, "calitas: %s\n", calitas_last_error(nullptr)); return 1; }
  auto die = [&](const char* what) { std::fprintf(stderr, "calitas: %s: %s\n", what, calitas_last_error(ctx)); calitas_destroy(ctx); std::exit(1); };
  if (calitas_set_reference_fasta(ctx, ref.c_str()) != CALITAS_OK) die("reading reference");
  if (!chrom.empty()) {
    int32_t n = 0; calitas_reference_info(ctx, &n, nullptr, nullptr);
    for (int32_t i = 0; i < n; i++) { const char* nm; uint64_t len; calitas_contig_name(ctx, i, &nm, &len); if (chrom == nm) p.chrom_index = i; }
    if (p.chrom_index < 0) { std::fprintf(stderr, "Unknown chromosome: %s\n", chrom.c_str()); calitas_destroy(ctx); return 1; }
  }
  char* tsv = nullptr; uint64_t rows = 0, bytes = 0;
  if (!variants.empty()) {   // SearchReference.scala:570-630
    uint64_t windows = 0;
    if (calitas_search_variants(ctx, &g, guide_id.c_str(), &p, variants.c_str(), chrom.empty() ? nullptr : chrom.c_str(), nullptr, nullptr, nullptr,
                                &tsv, &bytes, &rows, &windows) != CALITAS_OK) die("search with variants");
    std::fprintf(stderr, "calitas: %llu variant windows\n", (unsigned long long)windows);
  }
  FILE* f = output.empty() ? stdout : std::fopen(output.c_str(), "w");
  if (!f) { std::fprintf(stderr, "cannot write %s\n", output.c_str()); return 1; }
  if (variants.empty()) {     // straight to the file: a hits.txt of tens of gigabytes (PAM-less, many diffs) is never held in memory
    auto to_file = [](const char* piece, uint64_t n, void* user) -> int { return std::fwrite(piece, 1, n, (FILE*)user) == n ? 0 : 1; };
    if (calitas_search_hits_stream(ctx, &g, guide_id.c_str(), &p, nullptr, nullptr, to_file, f, &bytes, &rows) != CALITAS_OK) die("search");
  } else std::fwrite(tsv, 1, bytes, f);
  if (f != stdout) std::fclose(f);
  calitas_timing_t tm; calitas_get_timing(ctx, &tm);
  std::fprintf(stderr, "calitas: %llu hits; scan %.3f ms, align %.3f ms, filter + rows %.3f ms, text copy %.3f ms (%u lane%s)\n",
               (unsigned long long)rows, tm.scan_kernel_ms, tm.align_kernel_ms, tm.hits_kernel_ms, tm.hits_copy_ms, tm.lanes, tm.lanes == 1 ? "" : "s");
  calitas_free(tsv); calitas_destroy(ctx);
  return 0;
}
