// pr.cu -- `pr --market graph.mtx [-n runs]`
// CLI parity: examples/algorithms/pr/pr.cu (reference): alpha = 0.85, tol = 1e-6
// hard-coded (pr.cu:46-47), prints "GPU rank[:40] = " and "GPU Elapsed Time : ".
#include <gunrock/algorithms/pr.hxx>
#include <gunrock/framework/benchmark.hxx>
#include <gunrock/util/performance.hxx>

#include "../driver_common.hxx"

using namespace gunrock;
using namespace driver;

int main(int argc, char** argv) {
  io::cli::parameters_t arguments(argc, argv, "Page Rank");
  csr_t csr;
  auto properties = driver::load(arguments, csr);
  auto G = graph::build<memory_space_t::device>(properties, csr);
  auto context = std::make_shared<gcuda::multi_context_t>(0);

  const size_t n_vertices = G.get_number_of_vertices();
  const size_t n_edges = G.get_number_of_edges();
  thrust::device_vector<weight_t> p(n_vertices);
  const weight_t alpha = 0.85f, tol = 1e-6f;
  std::vector<std::string> tags;
  io::cli::parse_tag_string(arguments.tag_string, &tags);
  const options_t options = arguments.get_options();

  std::vector<float> run_times;
  std::vector<benchmark::host_benchmark_t> metrics((size_t)arguments.num_runs);
  int iterations = 0;
  for (int i = 0; i < arguments.num_runs; ++i) {
    benchmark::INIT_BENCH();
    pr::param_t<weight_t> param(alpha, tol, options);
    pr::result_t<weight_t> result(p.data().get());
    run_times.push_back(pr::run(G, param, result, context));
    iterations = result.iterations;
    metrics[(size_t)i] = benchmark::EXTRACT();
    benchmark::DESTROY_BENCH();
  }
  if (arguments.export_metrics) {
    std::vector<int> no_sources;
    util::stats::export_performance_stats(metrics, n_edges, n_vertices, run_times, "pr", arguments.filename, "market",
                                          arguments.json_dir, arguments.json_file, no_sources, tags, argc, argv);
  }
  print::head(p, 40, "GPU rank");
  std::cout << "GPU Elapsed Time : " << run_times.back() << " (ms)" << std::endl;
  std::cout << "Iterations : " << iterations << std::endl;
  return 0;
}
