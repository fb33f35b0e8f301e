// filepath.hxx -- file-name helpers of the loaders / CLI.
// API parity: include/gunrock/util/filepath.hxx:18-25 (reference).
#pragma once

#include <string>

namespace gunrock {
namespace util {

inline std::string extract_filename(std::string path, std::string delim = "/") {
  const auto pos = path.rfind(delim);
  return pos == std::string::npos ? path : path.substr(pos + delim.size());
}
inline std::string extract_dataset(std::string filename) {
  const auto pos = filename.rfind('.');
  return pos == std::string::npos ? filename : filename.substr(0, pos);
}
inline bool has_extension(const std::string& f, const std::string& ext) {
  return f.size() >= ext.size() && f.compare(f.size() - ext.size(), ext.size(), ext) == 0;
}
inline bool is_market(std::string f) { return has_extension(f, ".mtx") || has_extension(f, ".mmio"); }
inline bool is_binary_csr(std::string f) { return has_extension(f, ".csr"); }

}  // namespace util
}  // namespace gunrock
