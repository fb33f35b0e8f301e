// random.hxx -- fill a vector with uniform random numbers / draw one random value.
// Reference surface: gunrock::generate::random::{uniform_distribution, get_random}
// (include/gunrock/algorithms/generate/random.hxx:20-52 of the reference; used by the
// spmv driver to build its input vector).  Values are drawn on the host with a fixed seed
// (reproducible runs) and copied to wherever the vector lives.
#pragma once

#include <cstddef>
#include <random>
#include <vector>

namespace gunrock {
namespace generate {
namespace random {

template <typename vector_t, typename type_t = typename vector_t::value_type>
void uniform_distribution(vector_t& input, type_t begin = 0.0f, type_t end = 1.0f) {
  std::vector<type_t> host(input.size());
  std::mt19937 gen(42u);
  std::uniform_real_distribution<double> dis((double)begin, (double)end);
  for (auto& x : host) x = (type_t)dis(gen);
  input.assign(host.begin(), host.end());  // one bulk copy, host or device vector alike
}

template <typename rand_t = float>
rand_t get_random(rand_t begin = 0.0f, rand_t end = 1.0f) {
  std::random_device rd;
  std::mt19937 gen(rd());
  std::uniform_real_distribution<double> dis((double)begin, (double)end);
  return (rand_t)dis(gen);
}

}  // namespace random
}  // namespace generate
}  // namespace gunrock
