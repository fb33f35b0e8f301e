// driver_common.hxx -- shared plumbing of the example drivers: load the input
// (--market f.mtx, or a binary f.csr), build the device graph view.
#pragma once

#include <gunrock/algorithms/algorithms.hxx>
#include <gunrock/io/parameters.hxx>

namespace driver {

using vertex_t = int;
using edge_t = int;
using weight_t = float;
using csr_t = gunrock::format::csr_t<gunrock::memory_space_t::device, vertex_t, edge_t, weight_t>;

// Unlike upstream (which calls mm.load() even for .csr inputs and therefore
// cannot actually read them, examples/algorithms/bfs/bfs.cu:28-37), a .csr file
// is read directly; it carries no properties, so it is treated as directed+weighted.
inline gunrock::graph::graph_properties_t load(const gunrock::io::cli::parameters_t& args, csr_t& csr) {
  gunrock::graph::graph_properties_t properties;
  if (args.binary) {
    csr.read_binary(args.filename);
    properties.directed = true;
    properties.symmetric = false;
    properties.weighted = true;
  } else {
    gunrock::io::matrix_market_t<vertex_t, edge_t, weight_t> mm;
    auto loaded = mm.load(args.filename);
    properties = std::get<0>(loaded);
    csr.from_coo(std::get<1>(loaded));
  }
  return properties;
}

}  // namespace driver
