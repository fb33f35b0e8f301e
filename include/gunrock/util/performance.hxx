// performance.hxx -- --export_metrics JSON.
// API parity: include/gunrock/util/performance.hxx:82-283 (reference):
// util::stats::export_performance_stats(metrics, edges, vertices, run_times,
// primitive, filename, graph_type, json_dir, json_file, sources, tags, argc, argv)
// writing one JSON object with the reference's keys (schema "2022-10-28"):
// mteps = edges_visited / runtime_ms / 1000 (performance.hxx:225-229),
// avg/min/max/stdev process times, per-run arrays, sources, tags, gpuinfo.
// The reference serialises with nlohmann/json (not vendored); this writer emits
// the same keys by hand.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <ctime>
#include <fstream>
#include <numeric>
#include <sstream>
#include <string>
#include <vector>

#include <gunrock/framework/benchmark.hxx>
#include <gunrock/util/filepath.hxx>

namespace gunrock {
namespace util {
namespace stats {
namespace detail {
inline std::string quote(const std::string& s) {
  std::string o = "\"";
  for (char c : s) {
    if (c == '"' || c == '\\') o += '\\';
    o += c;
  }
  return o + "\"";
}
template <typename T>
inline std::string array(const std::vector<T>& v) {
  std::ostringstream o;
  o << "[";
  for (size_t i = 0; i < v.size(); ++i) o << (i ? ", " : "") << v[i];
  o << "]";
  return o.str();
}
inline std::string array(const std::vector<std::string>& v) {
  std::ostringstream o;
  o << "[";
  for (size_t i = 0; i < v.size(); ++i) o << (i ? ", " : "") << quote(v[i]);
  o << "]";
  return o.str();
}
}  // namespace detail

inline void export_performance_stats(std::vector<benchmark::host_benchmark_t>& benchmark_metrics, size_t edges,
                                     size_t vertices, std::vector<float>& run_times, std::string primitive,
                                     std::string filename, std::string graph_type, std::string json_dir,
                                     std::string json_file, std::vector<int>& sources,
                                     std::vector<std::string>& tags, int argc, char** argv) {
  const size_t n = run_times.size();
  float avg = 0, mn = 0, mx = 0, sd = 0;
  if (n) {
    avg = std::accumulate(run_times.begin(), run_times.end(), 0.0f) / (float)n;
    mn = *std::min_element(run_times.begin(), run_times.end());
    mx = *std::max_element(run_times.begin(), run_times.end());
    float ss = 0;
    for (float t : run_times) ss += (t - avg) * (t - avg);
    sd = n > 1 ? std::sqrt(ss / (float)(n - 1)) : 0.0f;
  }
  std::vector<unsigned long long> ev, nv, depth;
  std::vector<double> mteps;
  for (size_t i = 0; i < benchmark_metrics.size(); ++i) {
    ev.push_back(benchmark_metrics[i].edges_visited);
    nv.push_back(benchmark_metrics[i].vertices_visited);
    depth.push_back(benchmark_metrics[i].search_depth);
    const double ms = i < n ? run_times[i] : 0.0;
    mteps.push_back(ms > 0 ? (double)benchmark_metrics[i].edges_visited / ms / 1000.0 : 0.0);
  }
  double avg_mteps = mteps.empty() ? 0.0 : std::accumulate(mteps.begin(), mteps.end(), 0.0) / (double)mteps.size();

  std::time_t now = std::chrono::system_clock::to_time_t(std::chrono::system_clock::now());
  char stamp[64];
  std::strftime(stamp, sizeof stamp, "%a %b %d %H:%M:%S %Y", std::localtime(&now));
  std::string cmd;
  for (int i = 0; i < argc; ++i) cmd += std::string(i ? " " : "") + argv[i];

  hipDeviceProp_t prop{};
  int dev = 0, runtime = 0, driver = 0;
  (void)hipGetDevice(&dev);
  (void)hipGetDeviceProperties(&prop, dev);
  (void)hipRuntimeGetVersion(&runtime);
  (void)hipDriverGetVersion(&driver);

  const std::string dataset = extract_dataset(extract_filename(filename));
  if (json_file.empty()) json_file = primitive + "_" + dataset + ".json";
  std::ofstream out(json_dir + "/" + json_file);
  out << "{\n"
      << "  \"engine\": \"Essentials\",\n  \"json-schema\": \"2022-10-28\",\n"
      << "  \"primitive\": " << detail::quote(primitive) << ",\n"
      << "  \"graph-type\": " << detail::quote(graph_type) << ",\n"
      << "  \"graph-file\": " << detail::quote(filename) << ",\n"
      << "  \"dataset\": " << detail::quote(dataset) << ",\n"
      << "  \"num-vertices\": " << vertices << ",\n  \"num-edges\": " << edges << ",\n"
      << "  \"time\": " << detail::quote(stamp) << ",\n"
      << "  \"command-line\": " << detail::quote(cmd) << ",\n"
      << "  \"srcs\": " << detail::array(sources) << ",\n  \"tags\": " << detail::array(tags) << ",\n"
      << "  \"process-times\": " << detail::array(run_times) << ",\n"
      << "  \"avg-process-time\": " << avg << ",\n  \"min-process-time\": " << mn << ",\n"
      << "  \"max-process-time\": " << mx << ",\n  \"stddev-process-time\": " << sd << ",\n"
      << "  \"edges-visited\": " << detail::array(ev) << ",\n  \"nodes-visited\": " << detail::array(nv) << ",\n"
      << "  \"search-depth\": " << detail::array(depth) << ",\n"
      << "  \"mteps\": " << detail::array(mteps) << ",\n  \"avg-mteps\": " << avg_mteps << ",\n"
      << "  \"gpuinfo\": {\"name\": " << detail::quote(prop.name) << ", \"total_global_mem\": " << prop.totalGlobalMem
      << ", \"multi_processor_count\": " << prop.multiProcessorCount << ", \"clock_rate\": " << prop.clockRate
      << ", \"driver_api\": " << driver << ", \"runtime_api\": " << runtime << ", \"arch\": "
      << detail::quote(prop.gcnArchName) << "}\n}\n";
}

}  // namespace stats
}  // namespace util
}  // namespace gunrock
