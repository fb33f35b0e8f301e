// bfs.cu -- `bfs --market graph.mtx --src 0 [--validate] [--advance_load_balance ...]`
// CLI parity: examples/algorithms/bfs/bfs.cu (reference): same flags, same stdout
// lines ("Source : ", "GPU distances[:40] = ", "GPU Elapsed Time : ", and with
// --validate "CPU Distances[:40] = ", "CPU Elapsed Time : ", "Number of errors : ").
#include <gunrock/algorithms/bfs.hxx>
#include <gunrock/framework/benchmark.hxx>
#include <gunrock/util/performance.hxx>

#include "../driver_common.hxx"
#include "bfs_cpu.hxx"

using namespace gunrock;
using namespace driver;

int main(int argc, char** argv) {
  io::cli::parameters_t arguments(argc, argv, "Breadth First Search");
  csr_t csr;
  auto properties = driver::load(arguments, csr);
  auto G = graph::build<memory_space_t::device>(properties, csr);
  auto context = std::make_shared<gcuda::multi_context_t>(0);

  const size_t n_vertices = G.get_number_of_vertices();
  const size_t n_edges = G.get_number_of_edges();
  thrust::device_vector<vertex_t> distances(n_vertices), predecessors(n_vertices);

  std::vector<int> sources;
  io::cli::parse_source_string(arguments.source_string, &sources, (int)n_vertices, arguments.num_runs);
  std::vector<std::string> tags;
  io::cli::parse_tag_string(arguments.tag_string, &tags);
  const options_t options = arguments.get_options();

  std::vector<float> run_times;
  std::vector<benchmark::host_benchmark_t> metrics(sources.size());
  for (size_t i = 0; i < sources.size(); ++i) {
    context->get_context(0)->synchronize();
    benchmark::INIT_BENCH();
    bfs::param_t<vertex_t> param(sources[i], options);
    bfs::result_t<vertex_t> result(distances.data().get(), predecessors.data().get());
    run_times.push_back(bfs::run(G, param, result, context));
    metrics[i] = benchmark::EXTRACT();
    benchmark::DESTROY_BENCH();
    context->get_context(0)->synchronize();
  }

  if (arguments.export_metrics)
    util::stats::export_performance_stats(metrics, n_edges, n_vertices, run_times, "bfs", arguments.filename, "market",
                                          arguments.json_dir, arguments.json_file, sources, tags, argc, argv);

  std::cout << "Source : " << sources.back() << "\n";
  print::head(distances, 40, "GPU distances");
  std::cout << "GPU Elapsed Time : " << run_times.back() << " (ms)" << std::endl;

  if (arguments.validate) {
    thrust::host_vector<vertex_t> h_distances(n_vertices), h_predecessors(n_vertices);
    vertex_t last = sources.back();
    const float cpu_ms = bfs_cpu::run<csr_t, vertex_t, edge_t>(csr, last, h_distances.data(), h_predecessors.data());
    const size_t n_errors = util::compare(distances.data().get(), h_distances.data(), n_vertices);
    print::head(h_distances, 40, "CPU Distances");
    std::cout << "CPU Elapsed Time : " << cpu_ms << " (ms)" << std::endl;
    std::cout << "Number of errors : " << n_errors << std::endl;
    return n_errors == 0 ? 0 : 2;
  }
  return 0;
}
