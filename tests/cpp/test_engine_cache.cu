// The C++ bridge (include/gunrock/algorithms/engine.hxx) caches graph handles -- and with them derived state: weight
// statistics (all-equal weights run on the BFS engine!), the transpose and its weights, bins -- keyed on the identity of the
// CSR arrays, because upstream's graph_t is a NON-OWNING view (graph/graph.hxx:187-214): editing the arrays in place between
// two run() calls is legal there.  Checked here, against a host Dijkstra:
//   default policy (validate_t::full): an in-place edit of ONE weight / ONE column is seen by the next run();
//   validate_t::identity: the edit is seen after engine::invalidate(context, G) (and not required otherwise);
//   a different graph allocated at the same addresses is a different graph.
// Prints "CHECK <name> ok|FAILED" lines, exits non-zero on any failure.
#include <gunrock/algorithms/sssp.hxx>
#include <gunrock/algorithms/pr.hxx>

#include <cfloat>
#include <queue>
#include <random>
#include <vector>

using namespace gunrock;
using vertex_t = int;
using edge_t = int;
using weight_t = float;
using csr_t = format::csr_t<memory_space_t::device, vertex_t, edge_t, weight_t>;
using csr_h = format::csr_t<memory_space_t::host, vertex_t, edge_t, weight_t>;

static int failures = 0;
static void check(const char* name, bool ok) {
  printf("CHECK %s %s\n", name, ok ? "ok" : "FAILED");
  if (!ok) ++failures;
}

// examples/algorithms/sssp/sssp_cpu.hxx:36-67 semantics: fp32 Dijkstra, FLT_MAX = unreached
static std::vector<float> dijkstra(const std::vector<int>& ro, const std::vector<int>& ci, const std::vector<float>& w, int src) {
  std::vector<float> d(ro.size() - 1, FLT_MAX);
  using item = std::pair<float, int>;
  std::priority_queue<item, std::vector<item>, std::greater<item>> pq;
  d[src] = 0.0f;
  pq.push({0.0f, src});
  while (!pq.empty()) {
    auto [du, u] = pq.top();
    pq.pop();
    if (du > d[u]) continue;
    for (int e = ro[u]; e < ro[u + 1]; ++e) {
      const float nd = du + w[e];
      if (nd < d[ci[e]]) { d[ci[e]] = nd; pq.push({nd, ci[e]}); }
    }
  }
  return d;
}

template <typename G_t>
static std::vector<float> run_sssp(G_t& G, int V, int src, std::shared_ptr<gcuda::multi_context_t> ctx) {
  thrust::device_vector<float> dist(V);
  thrust::device_vector<int> pred(V);
  gunrock::sssp::run(G, src, dist.data().get(), pred.data().get(), ctx);
  std::vector<float> h(V);
  hipMemcpy(h.data(), dist.data().get(), V * sizeof(float), hipMemcpyDeviceToHost);
  return h;
}

int main() {
  const int V = 20000;
  std::mt19937 rng(3);
  std::vector<int> ro(V + 1, 0), ci;
  for (int v = 0; v < V; ++v) {
    const int deg = 1 + (int)(rng() % 9);
    for (int k = 0; k < deg; ++k) ci.push_back((int)(rng() % V));
    ro[v + 1] = (int)ci.size();
  }
  const int E = (int)ci.size();
  std::vector<float> w(E, 1.0f);  // all equal: the engine's handle remembers that and searches breadth first
  csr_h h(V, V, E);
  for (int i = 0; i <= V; ++i) h.row_offsets[i] = ro[i];
  for (int e = 0; e < E; ++e) { h.column_indices[e] = ci[e]; h.nonzero_values[e] = w[e]; }
  csr_t csr(h);
  graph::graph_properties_t props;
  props.directed = true;
  auto G = graph::build<memory_space_t::device>(props, csr);
  auto ctx = std::make_shared<gcuda::multi_context_t>(0);
  const int src = 0;

  check("engine.default_policy_is_full", engine::policy().validate == engine::validate_t::full);
  check("engine.unit_weights", run_sssp(G, V, src, ctx) == dijkstra(ro, ci, w, src));
  // ONE weight edited in place (an out-edge of the source, so the answer must change)
  const int e0 = ro[src];
  w[e0] = 7.5f;
  hipMemcpy(csr.nonzero_values.data().get() + e0, &w[e0], sizeof(float), hipMemcpyHostToDevice);
  auto want = dijkstra(ro, ci, w, src);
  check("engine.full.one_weight_edit_is_seen", run_sssp(G, V, src, ctx) == want);
  // ONE column edited in place
  const int e1 = ro[src] + (ro[src + 1] - ro[src] > 1 ? 1 : 0);
  ci[e1] = (ci[e1] + 4321) % V;
  hipMemcpy(csr.column_indices.data().get() + e1, &ci[e1], sizeof(int), hipMemcpyHostToDevice);
  want = dijkstra(ro, ci, w, src);
  check("engine.full.one_column_edit_is_seen", run_sssp(G, V, src, ctx) == want);

  // identity policy: no validation on a cache hit -- the caller owes an invalidate() after an edit
  engine::policy().validate = engine::validate_t::identity;
  check("engine.identity.hit", run_sssp(G, V, src, ctx) == want);
  for (int e = 0; e < E; ++e) w[e] = 1.0f + (float)((e * 2654435761u) % 1000u);
  hipMemcpy(csr.nonzero_values.data().get(), w.data(), E * sizeof(float), hipMemcpyHostToDevice);
  engine::invalidate(*ctx, G);
  want = dijkstra(ro, ci, w, src);
  check("engine.identity.edit_then_invalidate", run_sssp(G, V, src, ctx) == want);
  engine::policy().validate = engine::validate_t::full;
  check("engine.full.after_identity", run_sssp(G, V, src, ctx) == want);

  // PageRank keeps a transposed copy of the weights in the handle: an in-place edit must reach it
  {
    thrust::device_vector<float> p(V);
    gunrock::pr::run(G, 0.85f, 1e-6f, p.data().get(), ctx);
    std::vector<float> a(V), b(V);
    hipMemcpy(a.data(), p.data().get(), V * sizeof(float), hipMemcpyDeviceToHost);
    for (int e = ro[5]; e < ro[6]; ++e) w[e] = (e == ro[5]) ? 5000.0f : 1.0f;  // vertex 5 now sends nearly all its rank along one edge
    hipMemcpy(csr.nonzero_values.data().get() + ro[5], &w[ro[5]], (ro[6] - ro[5]) * sizeof(float), hipMemcpyHostToDevice);
    gunrock::pr::run(G, 0.85f, 1e-6f, p.data().get(), ctx);
    hipMemcpy(b.data(), p.data().get(), V * sizeof(float), hipMemcpyDeviceToHost);
    check("engine.full.pr_sees_weight_edit", ro[6] - ro[5] < 2 || a != b);
  }
  // ---- csr_t<device>::from_coo(coo_t<device>): the device conversion (libgrx's stable radix sort) equals the host one --------
  {
    using namespace gunrock;
    std::mt19937 rng(99);
    const int R = 5000, NZ = 200003;
    format::coo_t<memory_space_t::host, int, int, float> hc(R, R, NZ);
    for (int k = 0; k < NZ; ++k) {
      hc.row_indices[k] = (int)(rng() % R);
      hc.column_indices[k] = (int)(rng() % R);
      hc.nonzero_values[k] = (float)(1 + rng() % 1000) * 0.25f;
    }
    for (int k = 0; k < 50; ++k) {  // duplicates and self loops
      hc.row_indices[k] = hc.row_indices[k + 50];
      hc.column_indices[k] = hc.column_indices[k + 50];
      hc.column_indices[k + 100] = hc.row_indices[k + 100];
    }
    format::csr_t<memory_space_t::host, int, int, float> want_csr;
    want_csr.from_coo(hc);
    format::coo_t<memory_space_t::device, int, int, float> dc(hc);
    format::csr_t<memory_space_t::device, int, int, float> got;
    got.from_coo(dc);
    thrust::host_vector<int> gro = got.row_offsets, gci = got.column_indices;
    thrust::host_vector<float> gnz = got.nonzero_values;
    bool ok = gro.size() == want_csr.row_offsets.size() && gci.size() == want_csr.column_indices.size();
    for (std::size_t i = 0; ok && i < gro.size(); ++i) ok = gro[i] == want_csr.row_offsets[i];
    for (std::size_t i = 0; ok && i < gci.size(); ++i)
      ok = gci[i] == want_csr.column_indices[i] && gnz[i] == want_csr.nonzero_values[i];
    check("formats.csr_from_coo_on_the_device", ok);
  }
  printf(failures ? "FAILED\n" : "ALL CHECKS PASSED\n");
  return failures ? 1 : 0;
}
