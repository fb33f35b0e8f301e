// graph.hxx -- the non-owning device/host graph view handed to operators.
// API parity: include/gunrock/graph/graph.hxx:53-339 + graph/csr.hxx:61-235
// (reference): graph_t<space, V, E, W, graph_csr_t<...>> with vertex_type /
// edge_type / weight_type, get_number_of_vertices/edges, get_number_of_neighbors,
// get_starting_edge, get_destination_vertex, get_source_vertex (binary search
// over row offsets), get_edge_weight, get_row_offsets/column_indices/
// nonzero_values, is_directed/is_symmetric/is_weighted.  The view is a small
// trivially copyable struct (3 pointers, 2 counts, properties) captured BY VALUE
// in device lambdas and kernels.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include <gunrock/graph/properties.hxx>
#include <gunrock/memory.hxx>
#include <gunrock/util/load_store.hxx>

namespace gunrock {
namespace graph {

template <memory_space_t space, typename vertex_t, typename edge_t, typename weight_t>
struct graph_csr_t {
  using vertex_type = vertex_t;
  using edge_type = edge_t;
  using weight_type = weight_t;

  vertex_t number_of_vertices = 0;
  edge_t number_of_edges = 0;
  const edge_t* offsets = nullptr;
  const vertex_t* indices = nullptr;
  const weight_t* values = nullptr;

  template <typename csr_like_t>
  void set(csr_like_t& csr) {
    number_of_vertices = csr.number_of_rows;
    number_of_edges = csr.number_of_nonzeros;
    offsets = memory::raw_pointer_cast(csr.row_offsets.data());
    indices = memory::raw_pointer_cast(csr.column_indices.data());
    values = memory::raw_pointer_cast(csr.nonzero_values.data());
  }
};

// The in-edges of every vertex (reference: graph/csc.hxx:23-142): column_offsets[v] .. column_offsets[v + 1] index
// row_indices = the SOURCES of the edges that end in v.  A graph may carry it beside the CSR view
// (graph::build(properties, csr, csc)); it is what advance_direction_t::backward walks.
template <memory_space_t space, typename vertex_t, typename edge_t, typename weight_t>
struct graph_csc_t {
  vertex_t c_number_of_vertices = 0;
  edge_t c_number_of_edges = 0;
  const edge_t* c_offsets = nullptr;
  const vertex_t* c_indices = nullptr;
  const weight_t* c_values = nullptr;

  template <typename csc_like_t>
  void set(csc_like_t& csc) {
    c_number_of_vertices = csc.number_of_columns;
    c_number_of_edges = csc.number_of_nonzeros;
    c_offsets = memory::raw_pointer_cast(csc.column_offsets.data());
    c_indices = memory::raw_pointer_cast(csc.row_indices.data());
    c_values = memory::raw_pointer_cast(csc.nonzero_values.data());
  }
};

template <memory_space_t space, typename vertex_t, typename edge_t, typename weight_t, typename... views_t>
class graph_t : public views_t... {
  using first_view_t = graph_csr_t<space, vertex_t, edge_t, weight_t>;

 public:
  using vertex_type = vertex_t;
  using edge_type = edge_t;
  using weight_type = weight_t;
  using vertex_pointer_t = vertex_t*;
  using edge_pointer_t = edge_t*;
  using weight_pointer_t = weight_t*;

  using graph_csr_view_t = first_view_t;
  using graph_csc_view_t = graph_csc_t<space, vertex_t, edge_t, weight_t>;
  static constexpr bool has_csc_view = (std::is_same<views_t, graph_csc_view_t>::value || ...);

  graph_properties_t properties;

  __host__ __device__ graph_t() {}
  graph_t(graph_properties_t p) : properties(p) {}

  template <typename view_t = first_view_t>
  __host__ __device__ __forceinline__ vertex_t get_number_of_vertices() const {
    return first_view_t::number_of_vertices;
  }
  template <typename view_t = first_view_t>
  __host__ __device__ __forceinline__ edge_t get_number_of_edges() const {
    return first_view_t::number_of_edges;
  }
  __host__ __device__ __forceinline__ bool is_directed() const { return properties.directed; }
  __host__ __device__ __forceinline__ bool is_symmetric() const { return properties.symmetric; }
  __host__ __device__ __forceinline__ bool is_weighted() const { return properties.weighted; }

  template <typename view_t = first_view_t>
  __host__ __device__ __forceinline__ edge_t get_starting_edge(vertex_t const& v) const {
    return thread::load(&first_view_t::offsets[v]);
  }
  template <typename view_t = first_view_t>
  __host__ __device__ __forceinline__ edge_t get_number_of_neighbors(vertex_t const& v) const {
    return thread::load(&first_view_t::offsets[v + 1]) - thread::load(&first_view_t::offsets[v]);
  }
  template <typename view_t = first_view_t>
  __host__ __device__ __forceinline__ vertex_t get_destination_vertex(edge_t const& e) const {
    return thread::load(&first_view_t::indices[e]);
  }
  template <typename view_t = first_view_t>
  __host__ __device__ __forceinline__ weight_t get_edge_weight(edge_t const& e) const {
    return thread::load(&first_view_t::values[e]);
  }
  // Row owning edge e: largest v with offsets[v] <= e.
  template <typename view_t = first_view_t>
  __host__ __device__ __forceinline__ vertex_t get_source_vertex(edge_t const& e) const {
    vertex_t lo = 0, hi = first_view_t::number_of_vertices;
    while (hi - lo > 1) {
      const vertex_t mid = lo + (hi - lo) / 2;
      if (first_view_t::offsets[mid] <= e) lo = mid; else hi = mid;
    }
    return lo;
  }
  template <typename view_t = first_view_t>
  __host__ __device__ __forceinline__ auto get_row_offsets() const { return first_view_t::offsets; }
  template <typename view_t = first_view_t>
  __host__ __device__ __forceinline__ auto get_column_indices() const { return first_view_t::indices; }
  template <typename view_t = first_view_t>
  __host__ __device__ __forceinline__ auto get_nonzero_values() const { return first_view_t::values; }

  // ---- CSC view (graphs built with one): in-degree, in-edges and their sources ----------------------------------
  // (the reference spells these G.template get_number_of_neighbors<graph_csc_view_t>(v) etc.; its multi-view dispatch is
  // by template argument on every accessor -- here the in-edge accessors have names of their own and the whole reversed
  // graph is available as an ordinary CSR-shaped view, which is what the operators consume)
  __host__ __device__ __forceinline__ edge_t get_number_of_in_neighbors(vertex_t const& v) const {
    static_assert(has_csc_view, "this graph was built without a CSC view: graph::build(properties, csr, csc)");
    return thread::load(&graph_csc_view_t::c_offsets[v + 1]) - thread::load(&graph_csc_view_t::c_offsets[v]);
  }
  __host__ __device__ __forceinline__ edge_t get_starting_in_edge(vertex_t const& v) const {
    static_assert(has_csc_view, "this graph was built without a CSC view: graph::build(properties, csr, csc)");
    return thread::load(&graph_csc_view_t::c_offsets[v]);
  }
  // source of in-edge e (an index into the CSC arrays, NOT a CSR edge id)
  __host__ __device__ __forceinline__ vertex_t get_in_edge_source(edge_t const& e) const {
    static_assert(has_csc_view, "this graph was built without a CSC view: graph::build(properties, csr, csc)");
    return thread::load(&graph_csc_view_t::c_indices[e]);
  }
  __host__ __device__ __forceinline__ auto get_column_offsets() const { return graph_csc_view_t::c_offsets; }
  __host__ __device__ __forceinline__ auto get_row_indices() const { return graph_csc_view_t::c_indices; }
  // The reversed graph as a CSR-shaped view over the CSC arrays: row v lists the in-neighbours of v.  Non-owning, by value.
  __host__ __device__ auto reverse_view() const {
    static_assert(has_csc_view, "this graph was built without a CSC view: graph::build(properties, csr, csc)");
    graph_t<space, vertex_t, edge_t, weight_t, first_view_t> R;
    R.properties = properties;
    first_view_t& r = R;
    r.number_of_vertices = graph_csc_view_t::c_number_of_vertices;
    r.number_of_edges = graph_csc_view_t::c_number_of_edges;
    r.offsets = graph_csc_view_t::c_offsets;
    r.indices = graph_csc_view_t::c_indices;
    r.values = graph_csc_view_t::c_values;
    return R;
  }
};

}  // namespace graph
}  // namespace gunrock
