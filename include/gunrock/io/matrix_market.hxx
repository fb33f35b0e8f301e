// matrix_market.hxx -- Matrix Market (.mtx) reader -> host COO.
// API parity: include/gunrock/io/matrix_market.hxx:72-254 (reference):
// matrix_market_t<V,E,W>::load(filename) -> tuple<graph_properties_t, coo_t<host>>,
// public fields filename/dataset/format/data/scheme, enums of the same names.
// Semantics kept (they fix the edge ORDER): 1-based -> 0-based,
// pattern => weight 1.0, real/integer read as double then cast, symmetric =>
// every off-diagonal entry is followed by its mirror, dense arrays rejected,
// IO errors print to stderr and exit(1) like the reference.
// Implementation: the file is read once and tokenised in memory (the reference
// issues one fscanf per entry through NIST mmio).
#pragma once

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <string>
#include <tuple>
#include <vector>

#include <gunrock/error.hxx>
#include <gunrock/formats/formats.hxx>
#include <gunrock/graph/properties.hxx>
#include <gunrock/util/filepath.hxx>

namespace gunrock {
namespace io {

enum matrix_market_format_t { coordinate, array };
enum matrix_market_data_t { real, integer, complex, pattern };
enum matrix_market_storage_scheme_t { general, hermitian, symmetric, skew };

template <typename vertex_t, typename edge_t, typename weight_t>
struct matrix_market_t {
  std::string filename;
  std::string dataset;
  matrix_market_format_t format = coordinate;
  matrix_market_data_t data = real;
  matrix_market_storage_scheme_t scheme = general;

  matrix_market_t() {}

  std::tuple<graph::graph_properties_t, format::coo_t<memory_space_t::host, vertex_t, edge_t, weight_t>> load(
      std::string _filename) {
    filename = _filename;
    dataset = util::extract_dataset(util::extract_filename(filename));

    std::ifstream in(filename, std::ios::binary | std::ios::ate);
    if (!in) die("File could not be opened: " + filename);
    const std::streamsize bytes = in.tellg();
    in.seekg(0);
    std::vector<char> text((std::size_t)bytes + 1, 0);
    if (bytes > 0 && !in.read(text.data(), bytes)) die("File could not be opened: " + filename);
    const char* p = text.data();
    const char* end = p + bytes;

    // ---- banner -------------------------------------------------------------
    std::string banner = take_line(p, end);
    char head[64] = {0}, object[64] = {0}, fmt[64] = {0}, field[64] = {0}, symm[64] = {0};
    if (sscanf(banner.c_str(), "%63s %63s %63s %63s %63s", head, object, fmt, field, symm) != 5 ||
        std::strncmp(head, "%%MatrixMarket", 14) != 0 || lower(object) != "matrix")
      die("Could not process Matrix Market banner");
    const std::string f = lower(fmt), d = lower(field), s = lower(symm);
    if (f == "coordinate") format = coordinate; else if (f == "array") format = array;
    else die("Could not process Matrix Market banner");
    if (d == "real") data = real; else if (d == "integer") data = integer;
    else if (d == "pattern") data = pattern; else if (d == "complex") data = complex;
    else die("Could not process Matrix Market banner");
    if (s == "general") scheme = general; else if (s == "symmetric") scheme = symmetric;
    else if (s == "hermitian") scheme = hermitian; else if (s == "skew-symmetric") scheme = skew;
    else die("Could not process Matrix Market banner");
    if (format == array) die("File is not a sparse matrix");

    // ---- sizes ---------------------------------------------------------------
    while (p < end && *p == '%') take_line(p, end);
    unsigned long long rows = 0, cols = 0, nnz = 0;
    if (!take_uint(p, end, rows) || !take_uint(p, end, cols) || !take_uint(p, end, nnz))
      die("Could not read file info (M, N, NNZ)");
    error::throw_if_exception(rows >= (unsigned long long)std::numeric_limits<vertex_t>::max() ||
                                  cols >= (unsigned long long)std::numeric_limits<vertex_t>::max(),
                              "vertex_t overflow");
    error::throw_if_exception(nnz >= (unsigned long long)std::numeric_limits<edge_t>::max(), "edge_t overflow");
    if (data == complex) die("Unrecognized matrix market format type");

    graph::graph_properties_t properties;
    properties.weighted = data != pattern;
    const bool mirror = scheme == symmetric;
    properties.symmetric = mirror;
    properties.directed = !mirror;

    std::vector<vertex_t> I, J;
    std::vector<weight_t> X;
    I.reserve((std::size_t)nnz * (mirror ? 2 : 1));
    J.reserve(I.capacity());
    X.reserve(I.capacity());
    for (unsigned long long k = 0; k < nnz; ++k) {
      unsigned long long r = 0, c = 0;
      double value = 1.0;
      if (!take_uint(p, end, r) || !take_uint(p, end, c))
        error::throw_if_exception(true, "Could not read edge from market file");
      if (data != pattern && !take_real(p, end, value))
        error::throw_if_exception(true, "Could not read weighted edge from market file");
      error::throw_if_exception(r == 0 || c == 0, "Market file is zero-indexed");
      const vertex_t i = (vertex_t)(r - 1), j = (vertex_t)(c - 1);
      I.push_back(i); J.push_back(j); X.push_back((weight_t)value);
      if (mirror && i != j) { I.push_back(j); J.push_back(i); X.push_back((weight_t)value); }
    }

    format::coo_t<memory_space_t::host, vertex_t, edge_t, weight_t> coo((vertex_t)rows, (vertex_t)cols,
                                                                         (edge_t)I.size());
    for (std::size_t k = 0; k < I.size(); ++k) {
      coo.row_indices[k] = I[k];
      coo.column_indices[k] = J[k];
      coo.nonzero_values[k] = X[k];
    }
    return {properties, coo};
  }

 private:
  [[noreturn]] static void die(const std::string& why) {
    std::cerr << why << std::endl;
    std::exit(1);
  }
  static std::string lower(std::string s) {
    for (auto& ch : s) ch = (char)std::tolower((unsigned char)ch);
    return s;
  }
  static std::string take_line(const char*& p, const char* end) {
    const char* b = p;
    while (p < end && *p != '\n') ++p;
    std::string line(b, p);
    if (p < end) ++p;
    return line;
  }
  static void skip_space(const char*& p, const char* end) {
    while (p < end && std::isspace((unsigned char)*p)) ++p;
  }
  static bool take_uint(const char*& p, const char* end, unsigned long long& out) {
    skip_space(p, end);
    if (p >= end || !std::isdigit((unsigned char)*p)) return false;
    unsigned long long v = 0;
    while (p < end && std::isdigit((unsigned char)*p)) v = v * 10 + (unsigned long long)(*p++ - '0');
    out = v;
    return true;
  }
  static bool take_real(const char*& p, const char* end, double& out) {
    skip_space(p, end);
    if (p >= end) return false;
    char* stop = nullptr;
    out = std::strtod(p, &stop);  // buffer is NUL terminated
    if (stop == p) return false;
    p = stop;
    return true;
  }
};

}  // namespace io
}  // namespace gunrock
