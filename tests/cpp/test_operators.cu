// Operator-level parity checks against the reference SEMANTICS (not its code):
//   advance: neighbour k of input slot i lands at output[segments[i] + k], rejected -> -1
//            (thread_mapped.hxx:68-80, merge_path.hxx:218-279 of the reference), invalid
//            input slots are skipped, every load balance gives the same frontier;
//   filter : predicated/remove are STABLE compactions, compact keeps the same multiset,
//            bypass keeps positions and writes -1, `op` is never called on -1;
//   uniquify, parallel_for, frontier_t host API, bucketing's kernel choice.
// Prints "CHECK <name> ok|FAILED" lines and exits non-zero on any failure.
#include <gunrock/framework/operators/filter/predicated.hxx>
#include <gunrock/framework/operators/filter/remove.hxx>
#include <gunrock/framework/operators/filter/bypass.hxx>
#include <gunrock/framework/operators/filter/compact.hxx>
#include <gunrock/framework/operators/uniquify/unique.hxx>
#include <gunrock/framework/operators/uniquify/unique_copy.hxx>
#include <gunrock/algorithms/algorithms.hxx>

#include <algorithm>
#include <numeric>
#include <random>

using namespace gunrock;
using vertex_t = int;
using edge_t = int;
using weight_t = float;
using csr_t = format::csr_t<memory_space_t::device, vertex_t, edge_t, weight_t>;

static int failures = 0;
static void check(const char* name, bool ok) {
  printf("CHECK %s %s\n", name, ok ? "ok" : "FAILED");
  if (!ok) ++failures;
}

struct even_neighbors_t {  // keep neighbours with even id, count calls per source
  int* calls;
  __host__ __device__ bool operator()(vertex_t const& s, vertex_t const& n, edge_t const&, weight_t const&) const {
    math::atomic::add(calls + s, 1);
    return (n & 1) == 0;
  }
};
struct even_sources_t {  // backward advance: (source, destination) keep the edge's orientation; count calls per DESTINATION
  int* calls;
  __host__ __device__ bool operator()(vertex_t const& s, vertex_t const& n, edge_t const&, weight_t const&) const {
    math::atomic::add(calls + n, 1);
    return (s & 1) == 0;
  }
};
struct less_than_t {
  vertex_t bound;
  int* seen_invalid;
  __host__ __device__ bool operator()(vertex_t const& v) const {
    if (v < 0) math::atomic::add(seen_invalid, 1);
    return v < bound;
  }
};
struct mark_t {
  int* out;
  __host__ __device__ void operator()(vertex_t const& v) const { out[v] = v * 3; }
};

// launch_box (cuda/launch_box.hxx:194-360 of the reference): generic kernels + a grid-wide barrier kernel
struct add_index_t {
  int* out;
  __device__ void operator()(int const& tid, int const&) const { out[tid] += tid; }
};
__global__ void box_fill_kernel(int* out, int n, int value) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) out[i] = value;
}
// two phases separated by a hand-rolled grid barrier: only completes if the whole grid is resident,
// which is what launch_cooperative guarantees; the spin is bounded so a mistake fails instead of hanging
__global__ void box_coop_kernel(int* data, int n, unsigned* arrive, int* timed_out) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) data[i] = i;
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    atomicAdd(arrive, 1u);
    long long spins = 0;
    while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
      __builtin_amdgcn_s_sleep(8);
      if (++spins > 20000000ll) { *timed_out = 1; break; }
    }
    __threadfence();
  }
  __syncthreads();
  // phase 2 reads what OTHER workgroups wrote in phase 1
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int j = n - 1 - i;
    const int other = __hip_atomic_load(&data[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (other != j) *timed_out = 2;
  }
}

// Operators are asynchronous on the context's (non-blocking) stream -- unlike upstream,
// which synchronises after every operator (thread_mapped.hxx:94) -- so host reads of a
// frontier in the middle of a loop must synchronise the context first.
static std::shared_ptr<gcuda::multi_context_t> g_context;
template <typename E>
static std::vector<vertex_t> download(E& f) {
  if (g_context) g_context->get_context(0)->synchronize();
  std::vector<vertex_t> h(f.get_number_of_elements());
  if (!h.empty()) hipMemcpy(h.data(), f.data(), h.size() * sizeof(vertex_t), hipMemcpyDeviceToHost);
  return h;
}

struct dummy_problem_t : gunrock::problem_t<decltype(graph::build<memory_space_t::device>(
                             std::declval<graph::graph_properties_t>(), std::declval<csr_t&>()))> {
  using graph_type = decltype(graph::build<memory_space_t::device>(std::declval<graph::graph_properties_t>(),
                                                                  std::declval<csr_t&>()));
  dummy_problem_t(graph_type& G, std::shared_ptr<gcuda::multi_context_t> c) : gunrock::problem_t<graph_type>(G, c) {}
  void init() override {}
  void reset() override {}
};
struct dummy_enactor_t : gunrock::enactor_t<dummy_problem_t> {
  using gunrock::enactor_t<dummy_problem_t>::enactor_t;
  void loop(gcuda::multi_context_t&) override {}
};

int main() {
  // ---- a skewed random graph: one hub, many small rows, some empty rows -------------
  const int V = 3000;
  std::mt19937 rng(7);
  std::vector<std::vector<int>> adj(V);
  for (int v = 0; v < V; ++v) {
    int deg = (v == 5) ? 9000 : (v % 7 == 0 ? 0 : (int)(rng() % 12));
    for (int k = 0; k < deg; ++k) adj[v].push_back((int)(rng() % V));
  }
  format::coo_t<memory_space_t::host, vertex_t, edge_t, weight_t> coo(V, V, 0);
  std::vector<int> I, J;
  for (int v = 0; v < V; ++v)
    for (int n : adj[v]) { I.push_back(v); J.push_back(n); }
  coo = format::coo_t<memory_space_t::host, vertex_t, edge_t, weight_t>(V, V, (edge_t)I.size());
  for (size_t k = 0; k < I.size(); ++k) { coo.row_indices[k] = I[k]; coo.column_indices[k] = J[k]; coo.nonzero_values[k] = 1.0f; }
  csr_t csr;
  csr.from_coo(coo);
  graph::graph_properties_t props;
  props.directed = true;
  auto G = graph::build<memory_space_t::device>(props, csr);
  auto context = std::make_shared<gcuda::multi_context_t>(0);
  g_context = context;
  dummy_problem_t problem(G, context);
  dummy_enactor_t E(&problem, context);

  // ---- frontier host API ---------------------------------------------------------
  {
    frontier::frontier_t<vertex_t, edge_t> f;
    check("frontier.empty", f.is_empty() && f.get_number_of_elements() == 0);
    f.push_back(4); f.push_back(9);
    f.resize(5);  // new slots are invalid (-1)
    auto h = download(f);
    check("frontier.push_back+resize", h == std::vector<int>({4, 9, -1, -1, -1}));
    f.sequence(10, 6);
    check("frontier.sequence", download(f) == std::vector<int>({10, 11, 12, 13, 14, 15}));
    f.fill(2);
    check("frontier.fill", download(f) == std::vector<int>(6, 2));
    f.set_number_of_elements(3);
    check("frontier.set_number_of_elements", f.get_number_of_elements() == 3 && !f.is_empty());
    frontier::frontier_t<vertex_t, edge_t> g2 = f;  // copies share storage
    check("frontier.copy_shares_storage", g2.data() == f.data());
  }

  // ---- advance: same output for every load balance, reference slot semantics -----
  std::vector<int> input = {5, 17, -1, 21, 0, 5, 2999, 8};  // hub twice, an invalid slot, an empty row (0, 21)
  std::vector<int> expect;
  std::vector<int> expect_calls(V, 0);
  for (int v : input) {
    if (v < 0) continue;
    for (int n : adj[v]) { expect.push_back((n & 1) == 0 ? n : -1); expect_calls[v]++; }
  }
  thrust::device_vector<int> calls(V);
  auto run_advance = [&](operators::load_balance_t lb, const char* name) {
    thrust::fill(calls.begin(), calls.end(), 0);
    auto* in = E.get_input_frontier();
    in->set_number_of_elements(0);
    for (int v : input) in->push_back(v);
    operators::advance::execute_runtime(G, &E, even_neighbors_t{calls.data().get()}, lb, *context);
    auto out = download(*E.get_input_frontier());  // buffers were swapped (download synchronises)
    thrust::host_vector<int> hc = calls;
    bool ok = out == expect;
    for (int v = 0; v < V && ok; ++v) ok = hc[v] == expect_calls[v];
    check(name, ok);
  };
  run_advance(operators::load_balance_t::thread_mapped, "advance.thread_mapped");
  run_advance(operators::load_balance_t::warp_mapped, "advance.warp_mapped");
  run_advance(operators::load_balance_t::block_mapped, "advance.block_mapped");
  run_advance(operators::load_balance_t::merge_path, "advance.merge_path");
  run_advance(operators::load_balance_t::merge_path_v2, "advance.merge_path_v2");
  run_advance(operators::load_balance_t::bucketing, "advance.bucketing");
  // (round 6: that input is a frontier of HUBS -- its longest row is half of the level -- which advance::execute runs on the
  // merge-path kernel for every load balance.  The kernels' OWN long-row paths -- rows for the wave, rows of 2048+ for the
  // workgroup -- need a frontier whose long rows are the rule, not the exception: 1200 rows of 2100 neighbours each.)
  {
    const int V2 = 4000;
    std::vector<std::vector<int>> adj2(V2);
    for (int v = 0; v < V2; ++v) {
      const int deg = v < 1200 ? 2100 : (v % 5 == 0 ? 0 : (int)(rng() % 90));
      for (int k = 0; k < deg; ++k) adj2[v].push_back((int)(rng() % V2));
    }
    std::vector<int> I2, J2;
    for (int v = 0; v < V2; ++v)
      for (int n : adj2[v]) { I2.push_back(v); J2.push_back(n); }
    format::coo_t<memory_space_t::host, vertex_t, edge_t, weight_t> coo2(V2, V2, (edge_t)I2.size());
    for (size_t k = 0; k < I2.size(); ++k) { coo2.row_indices[k] = I2[k]; coo2.column_indices[k] = J2[k]; coo2.nonzero_values[k] = 1.0f; }
    csr_t csr2;
    csr2.from_coo(coo2);
    auto G2 = graph::build<memory_space_t::device>(props, csr2);
    dummy_problem_t problem2(G2, context);
    dummy_enactor_t E2(&problem2, context);
    std::vector<int> input2;
    for (int v = 0; v < 1200; ++v) input2.push_back(v);
    for (int v : {1203, -1, 1300, 3999, 1205, 7}) input2.push_back(v);
    std::vector<int> expect2, expect_calls2(V2, 0);
    for (int v : input2) {
      if (v < 0) continue;
      for (int n : adj2[v]) { expect2.push_back((n & 1) == 0 ? n : -1); expect_calls2[v]++; }
    }
    thrust::device_vector<int> calls2(V2);
    thrust::device_vector<int> d_in2(input2.begin(), input2.end());
    auto run2 = [&](operators::load_balance_t lb, const char* name) {
      thrust::fill(calls2.begin(), calls2.end(), 0);
      auto* in = E2.get_input_frontier();
      in->resize(input2.size());
      hipMemcpy(in->data(), d_in2.data().get(), input2.size() * sizeof(int), hipMemcpyDeviceToDevice);
      operators::advance::execute_runtime(G2, &E2, even_neighbors_t{calls2.data().get()}, lb, *context);
      auto out = download(*E2.get_input_frontier());
      thrust::host_vector<int> hc = calls2;
      bool ok = out == expect2;
      for (int v = 0; v < V2 && ok; ++v) ok = hc[v] == expect_calls2[v];
      check(name, ok);
    };
    run2(operators::load_balance_t::thread_mapped, "advance.long_rows.thread_mapped");
    run2(operators::load_balance_t::warp_mapped, "advance.long_rows.warp_mapped");
    run2(operators::load_balance_t::block_mapped, "advance.long_rows.block_mapped");
    run2(operators::load_balance_t::merge_path, "advance.long_rows.merge_path");
    run2(operators::load_balance_t::bucketing, "advance.long_rows.bucketing");
  }
  // fused advance + compact (extension): the kept neighbours of the same call, as a multiset, and one op call per edge
  {
    thrust::fill(calls.begin(), calls.end(), 0);
    auto* in = E.get_input_frontier();
    in->set_number_of_elements(0);
    for (int v : input) in->push_back(v);
    operators::advance::execute_compact(G, &E, even_neighbors_t{calls.data().get()}, *context);
    auto out = download(*E.get_input_frontier());
    std::vector<int> want;
    for (int x : expect) if (x >= 0) want.push_back(x);
    std::sort(want.begin(), want.end());
    std::sort(out.begin(), out.end());
    thrust::host_vector<int> hc = calls;
    bool ok = out == want;
    for (int v = 0; v < V && ok; ++v) ok = hc[v] == expect_calls[v];
    check("advance.execute_compact", ok);
  }
  // ---- advance_direction_t::backward over a CSC view: in-neighbours of the input, in CSC order ----------
  {
    format::csc_t<memory_space_t::device, vertex_t, edge_t, weight_t> csc;
    csc.from_csr(csr);
    auto G2 = graph::build<memory_space_t::device>(props, csr, csc);
    std::vector<std::vector<int>> radj(V);
    for (int r = 0; r < V; ++r)
      for (int n : adj[r]) radj[n].push_back(r);  // rows ascending, file order inside a row: what from_csr produces
    std::vector<int> want;
    std::vector<int> want_calls(V, 0);
    for (int v : input) {
      if (v < 0) continue;
      for (int u : radj[v]) { want.push_back((u & 1) == 0 ? u : -1); want_calls[v]++; }
    }
    auto run_backward = [&](auto lb_c, const char* name) {
      constexpr operators::load_balance_t lb = decltype(lb_c)::value;
      thrust::fill(calls.begin(), calls.end(), 0);
      auto* in = E.get_input_frontier();
      auto* out = E.get_output_frontier();
      in->set_number_of_elements(0);
      for (int v : input) in->push_back(v);
      operators::advance::execute<lb, operators::advance_direction_t::backward, operators::advance_io_type_t::vertices,
                                  operators::advance_io_type_t::vertices>(G2, even_sources_t{calls.data().get()}, in, out,
                                                                          E.scanned_work_domain, *context);
      auto got = download(*out);
      thrust::host_vector<int> hc = calls;
      bool ok = got == want;
      for (int v = 0; v < V && ok; ++v) ok = hc[v] == want_calls[v];
      check(name, ok);
    };
    run_backward(std::integral_constant<operators::load_balance_t, operators::load_balance_t::thread_mapped>{}, "advance.backward.thread_mapped");
    run_backward(std::integral_constant<operators::load_balance_t, operators::load_balance_t::block_mapped>{}, "advance.backward.block_mapped");
    run_backward(std::integral_constant<operators::load_balance_t, operators::load_balance_t::merge_path>{}, "advance.backward.merge_path");
    // forward on the same two-view graph is the CSR advance it always was
    {
      thrust::fill(calls.begin(), calls.end(), 0);
      auto* in = E.get_input_frontier();
      auto* out = E.get_output_frontier();
      in->set_number_of_elements(0);
      for (int v : input) in->push_back(v);
      operators::advance::execute<operators::load_balance_t::merge_path, operators::advance_direction_t::forward,
                                  operators::advance_io_type_t::vertices, operators::advance_io_type_t::vertices>(
          G2, even_neighbors_t{calls.data().get()}, in, out, E.scanned_work_domain, *context);
      check("advance.forward_on_two_view_graph", download(*out) == expect);
    }
  }
  {
    bool threw = false;
    try {
      operators::advance::execute_runtime(G, &E, even_neighbors_t{calls.data().get()},
                                          operators::load_balance_t::work_stealing, *context);
    } catch (error::exception_t& e) {
      threw = std::string(e.what()).find("Load balance type not supported.") != std::string::npos;
    }
    check("advance.work_stealing_throws", threw);
  }
  // whole graph as input, no output: op called exactly once per edge
  {
    thrust::fill(calls.begin(), calls.end(), 0);
    operators::advance::execute<operators::load_balance_t::merge_path, operators::advance_direction_t::forward,
                                operators::advance_io_type_t::graph, operators::advance_io_type_t::none>(
        G, &E, even_neighbors_t{calls.data().get()}, *context);
    context->get_context(0)->synchronize();
    thrust::host_vector<int> hc = calls;
    bool ok = true;
    for (int v = 0; v < V; ++v) ok = ok && hc[v] == (int)adj[v].size();
    check("advance.graph_input_no_output", ok);
  }
  // bucketing's choice follows the frontier's degree histogram
  {
    using operators::advance::bucketing::select;
    auto& ctx = *context->get_context(0);
    thrust::device_vector<int> seg;
    auto choice = [&](std::vector<int> degs) {
      std::vector<int> s(degs.size() + 1, 0);
      for (size_t i = 0; i < degs.size(); ++i) s[i + 1] = s[i] + degs[i];
      seg = s;
      return select(seg.data().get(), degs.size(), (size_t)s.back(), ctx);
    };
    check("bucketing.low_degree->thread_mapped", choice(std::vector<int>(5000, 3)) == operators::load_balance_t::thread_mapped);
    check("bucketing.uniform_high->warp_mapped", choice(std::vector<int>(500, 200)) == operators::load_balance_t::warp_mapped);
    std::vector<int> skew(4000, 4); skew[7] = 50000;
    check("bucketing.hub->merge_path", choice(skew) == operators::load_balance_t::merge_path);
    std::vector<int> mid(2000, 20); mid[3] = 300;
    check("bucketing.moderate->block_mapped", choice(mid) == operators::load_balance_t::block_mapped);
  }

  // ---- filter ---------------------------------------------------------------------
  std::vector<int> fin(20000);
  for (auto& x : fin) x = (rng() % 5 == 0) ? -1 : (int)(rng() % 1000);
  std::vector<int> stable;
  for (int x : fin) if (x >= 0 && x < 400) stable.push_back(x);
  thrust::device_vector<int> invalid_seen(1);
  auto run_filter = [&](operators::filter_algorithm_t alg) {
    invalid_seen[0] = 0;
    auto* in = E.get_input_frontier();
    in->resize(fin.size());
    hipMemcpy(in->data(), fin.data(), fin.size() * sizeof(int), hipMemcpyHostToDevice);
    operators::filter::execute_runtime(G, &E, less_than_t{400, invalid_seen.data().get()}, alg, *context);
    return download(*E.get_input_frontier());
  };
  check("filter.predicated_stable", run_filter(operators::filter_algorithm_t::predicated) == stable && invalid_seen[0] == 0);
  check("filter.remove_stable", run_filter(operators::filter_algorithm_t::remove) == stable);
  {
    auto got = run_filter(operators::filter_algorithm_t::compact);
    auto a = got, b = stable;
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    check("filter.compact_same_multiset", a == b && invalid_seen[0] == 0);
  }
  {
    auto got = run_filter(operators::filter_algorithm_t::bypass);
    bool ok = got.size() == fin.size();
    for (size_t i = 0; i < fin.size() && ok; ++i) ok = got[i] == ((fin[i] >= 0 && fin[i] < 400) ? fin[i] : -1);
    check("filter.bypass_keeps_positions", ok && invalid_seen[0] == 0);
  }
  // the reference's per-algorithm entry points (filter/predicated.hxx:12-39 etc.): filter::<algorithm>::execute(G, op, in*, out*,
  // standard_context) and the same for uniquify -- called directly, as a user TU that includes only those headers would
  {
    auto* in = E.get_input_frontier();
    auto* out = E.get_output_frontier();
    in->resize(fin.size());
    hipMemcpy(in->data(), fin.data(), fin.size() * sizeof(int), hipMemcpyHostToDevice);
    auto& sctx = *context->get_context(0);
    invalid_seen[0] = 0;
    operators::filter::predicated::execute(G, less_than_t{400, invalid_seen.data().get()}, in, out, sctx);
    bool ok = download(*out) == stable;
    operators::filter::remove::execute(G, less_than_t{400, invalid_seen.data().get()}, in, out, sctx);
    ok = ok && download(*out) == stable;
    operators::filter::compact::execute(G, less_than_t{400, invalid_seen.data().get()}, in, out, sctx);
    auto a = download(*out), b = stable;
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    ok = ok && a == b;
    operators::filter::bypass::execute(G, less_than_t{400, invalid_seen.data().get()}, in, sctx);  // in place
    auto got = download(*in);
    ok = ok && got.size() == fin.size();
    for (size_t i = 0; i < fin.size() && ok; ++i) ok = got[i] == ((fin[i] >= 0 && fin[i] < 400) ? fin[i] : -1);
    check("filter.per_algorithm_namespaces", ok && invalid_seen[0] == 0);
    std::vector<int> runs = {5, 5, -1, 5, 1, 1, 9};
    in->resize(runs.size());
    hipMemcpy(in->data(), runs.data(), runs.size() * sizeof(int), hipMemcpyHostToDevice);
    operators::uniquify::unique::execute(in, out, sctx);
    bool uok = download(*out) == std::vector<int>({5, 5, 1, 9});
    operators::uniquify::unique_copy::execute(in, out, sctx);
    check("uniquify.per_algorithm_namespaces", uok && download(*out) == std::vector<int>({5, 5, 1, 9}));
  }
  // ---- uniquify -------------------------------------------------------------------
  {
    auto* in = E.get_input_frontier();
    in->resize(fin.size());
    hipMemcpy(in->data(), fin.data(), fin.size() * sizeof(int), hipMemcpyHostToDevice);
    operators::uniquify::execute<operators::uniquify_algorithm_t::unique>(&E, *context, false, 100);
    auto got = download(*E.get_input_frontier());
    std::vector<int> want;
    for (int x : fin) if (x >= 0) want.push_back(x);
    std::sort(want.begin(), want.end());
    want.erase(std::unique(want.begin(), want.end()), want.end());
    check("uniquify.full", got == want);
    in = E.get_input_frontier();
    std::vector<int> runs = {3, 3, 3, 7, -1, 7, 7, 2, 2, 3};
    in->resize(runs.size());
    hipMemcpy(in->data(), runs.data(), runs.size() * sizeof(int), hipMemcpyHostToDevice);
    operators::uniquify::execute<operators::uniquify_algorithm_t::unique>(&E, *context, true, 100);
    check("uniquify.best_effort_adjacent_runs", download(*E.get_input_frontier()) == std::vector<int>({3, 7, 7, 2, 3}));
  }
  // ---- parallel_for ---------------------------------------------------------------
  {
    thrust::device_vector<int> marks(V, -5);
    operators::parallel_for::execute<operators::parallel_for_each_t::vertex>(G, mark_t{marks.data().get()}, *context);
    context->get_context(0)->synchronize();
    thrust::host_vector<int> h = marks;
    bool ok = true;
    for (int v = 0; v < V; ++v) ok = ok && h[v] == 3 * v;
    check("parallel_for.vertex", ok);
    thrust::fill(marks.begin(), marks.end(), -5);
    frontier::frontier_t<vertex_t, edge_t> f;
    f.push_back(4); f.push_back(-1); f.push_back(10);
    operators::parallel_for::execute<operators::parallel_for_each_t::element>(f, mark_t{marks.data().get()}, *context);
    context->get_context(0)->synchronize();
    h = marks;
    check("parallel_for.element_skips_invalid", h[4] == 12 && h[10] == 30 && h[0] == -5);
  }
  // ---- launch_box: blocked / strided / launch / cooperative / occupancy -----------------------
  {
    using namespace gcuda::launch_box;
    auto& sc = *context->get_context(0);
    const int n = 100000;
    thrust::device_vector<int> buf(n, 1);
    launch_box_t<launch_params_dynamic_grid_t<fallback, dim3_t<128>, 4>> blocked;
    blocked.launch_blocked(sc, add_index_t{buf.data().get()}, (std::size_t)n);
    launch_box_t<launch_params_dynamic_grid_t<fallback, dim3_t<256>>> strided;
    strided.launch_strided(sc, add_index_t{buf.data().get()}, (std::size_t)n);
    sc.synchronize();
    thrust::host_vector<int> h = buf;
    bool ok = true;
    for (int i = 0; i < n; ++i) ok = ok && h[i] == 1 + 2 * i;
    check("launch_box.blocked_and_strided", ok);
    launch_box_t<launch_params_t<fallback, dim3_t<64>, dim3_t<40>>> fixed;  // static grid of 40 workgroups
    fixed.launch(sc, box_fill_kernel, buf.data().get(), n, 7);
    sc.synchronize();
    h = buf;
    check("launch_box.launch_static_grid", fixed.grid_dimensions.x == 40 && h[0] == 7 && h[n - 1] == 7);
    using coop_t = launch_box_t<launch_params_dynamic_grid_t<fallback, dim3_t<256>>>;
    coop_t coop;
    thrust::device_vector<unsigned> arrive(1, 0u);
    thrust::device_vector<int> flag(1, 0);
    int* dptr = buf.data().get();
    unsigned* aptr = arrive.data().get();
    int* fptr = flag.data().get();
    int nn = n;
    coop.launch_cooperative(sc, box_coop_kernel, (std::size_t)(1 << 22), dptr, nn, aptr, fptr);  // grid clamped to residency
    sc.synchronize();
    thrust::host_vector<int> hf = flag;
    check("launch_box.cooperative_grid_barrier", hf[0] == 0 && coop.grid_dimensions.x > 1);
    const float occ = occupancy<coop_t>(box_coop_kernel);
    check("launch_box.occupancy", occ > 0.0f && occ <= 1.0f);
  }
  // ---- graph view accessors (graph/csr.hxx of the reference) -------------------------
  {
    thrust::host_vector<int> ro = csr.row_offsets;
    auto Gh_csr = format::csr_t<memory_space_t::host, vertex_t, edge_t, weight_t>(csr);
    auto Gh = graph::build<memory_space_t::host>(props, Gh_csr);
    bool ok = Gh.get_number_of_vertices() == V && Gh.get_number_of_edges() == (int)I.size();
    for (int v : {0, 5, 6, 2999}) {
      ok = ok && Gh.get_number_of_neighbors(v) == (int)adj[v].size() && Gh.get_starting_edge(v) == ro[v];
      if (!adj[v].empty()) {
        ok = ok && Gh.get_destination_vertex(ro[v]) == adj[v][0];
        ok = ok && Gh.get_source_vertex(ro[v]) == v && Gh.get_source_vertex(ro[v + 1] - 1) == v;
      }
    }
    check("graph.accessors", ok && Gh.is_directed());
  }
  printf(failures ? "FAILED %d checks\n" : "ALL CHECKS PASSED\n", failures);
  return failures ? 1 : 0;
}
