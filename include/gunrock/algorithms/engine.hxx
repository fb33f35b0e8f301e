// engine.hxx -- bridge from the C++ template API to the pre-compiled engine
// (libgrx.so, include/grx.h).  With -DGUNROCK_HEADER_ONLY the bridge is
// compiled out and every algorithm runs on the generic operators of these
// headers; otherwise bfs/sssp/pr::run() dispatch to the fused kernels unless
// options.engine_flags has bit 0 set.
#pragma once

#include <gunrock/cuda/context.hxx>
#include <gunrock/error.hxx>

#ifndef GUNROCK_HEADER_ONLY
#include <grx.h>

#include <map>
#include <mutex>
#include <tuple>

namespace gunrock {
namespace engine {

inline std::mutex& guard() {
  static std::mutex m;
  return m;
}

inline void check(grx_status_t st) {
  if (st != GRX_SUCCESS) throw error::exception_t(std::string(grx_last_error_string()));
}

// one engine context per (device, stream); lives for the process
inline grx_context_t context_for(gcuda::multi_context_t& mc) {
  static std::map<std::pair<int, void*>, grx_context_t> cache;
  auto* sc = mc.get_context(0);
  std::lock_guard<std::mutex> lock(guard());
  auto key = std::make_pair((int)sc->ordinal(), (void*)sc->stream());
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  grx_context_t c = nullptr;
  check(grx_context_create(sc->ordinal(), (void*)sc->stream(), &c));
  cache[key] = c;
  return c;
}

// graph handles are cached by the identity of the CSR arrays so per-graph
// preprocessing (the transpose behind pull PageRank) is paid once
template <typename graph_t>
inline grx_graph_t graph_for(grx_context_t ctx, graph_t& G) {
  using key_t = std::tuple<const void*, const void*, const void*, int, int>;
  static std::map<key_t, grx_graph_t> cache;
  std::lock_guard<std::mutex> lock(guard());
  key_t key{(const void*)G.get_row_offsets(), (const void*)G.get_column_indices(),
            (const void*)G.get_nonzero_values(), (int)G.get_number_of_vertices(), (int)G.get_number_of_edges()};
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  if (cache.size() > 64) {  // bounded: drop everything rather than grow without limit
    for (auto& kv : cache) grx_graph_destroy(kv.second);
    cache.clear();
  }
  grx_graph_t g = nullptr;
  check(grx_graph_create_csr(ctx, (int32_t)G.get_number_of_vertices(), (int32_t)G.get_number_of_edges(),
                             (const int32_t*)G.get_row_offsets(), (const int32_t*)G.get_column_indices(),
                             (const float*)G.get_nonzero_values(), G.is_directed(), G.is_weighted(),
                             G.is_symmetric(), &g));
  cache[key] = g;
  return g;
}

template <typename options_like_t>
inline grx_options_t to_c(const options_like_t& o) {
  grx_options_t c;
  grx_options_default(&c);
  c.advance_load_balance = (int32_t)o.advance_load_balance;
  c.filter_algorithm = (int32_t)o.filter_algorithm;
  c.enable_filter = o.enable_filter;
  c.enable_uniquify = o.enable_uniquify;
  c.uniquify_algorithm = (int32_t)o.uniquify_algorithm;
  c.best_effort_uniquify = o.best_effort_uniquify;
  c.uniquify_percent = o.uniquify_percent;
  c.advance_direction = (int32_t)o.advance_direction;
  c.engine_flags = (o.engine_flags & ~1);  // bit 0 selects the generic path on the C++ side
  return c;
}

template <typename graph_t>
constexpr bool supported_types() {
  return sizeof(typename graph_t::vertex_type) == 4 && sizeof(typename graph_t::edge_type) == 4 &&
         std::is_same<typename graph_t::weight_type, float>::value;
}

}  // namespace engine
}  // namespace gunrock
#endif  // GUNROCK_HEADER_ONLY
