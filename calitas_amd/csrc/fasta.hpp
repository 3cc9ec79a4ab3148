// fasta.hpp -- see fasta.cpp
#pragma once
#include <string>
#include <vector>

namespace calitas {

struct FastaData {
  std::vector<std::string> names;
  std::vector<std::string> seqs;
  std::string genome_build = "unknown";
};

// Returns an empty string on success, else the error text.
std::string read_fasta(const std::string& path, FastaData& out);

}  // namespace calitas
