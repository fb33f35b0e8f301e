// print.hxx -- print::head, the driver's "GPU distances[:40] = ..." line.
// API parity: include/gunrock/util/print.hxx:17-53 (reference); output format is
// byte-compatible ("name[:k] = v0 v1 ... \n").
#pragma once

#include <hip/hip_runtime.h>
#include <thrust/copy.h>
#include <thrust/device_vector.h>
#include <thrust/host_vector.h>

#include <iostream>
#include <iterator>
#include <string>
#include <vector>

namespace gunrock {
namespace print {

template <typename vector_t>
void head(vector_t& x, int k, std::string name = "") {
  using type_t = typename vector_t::value_type;
  if ((int)x.size() < k) k = (int)x.size();
  if (!name.empty()) std::cout << name << "[:" << k << "] = ";
  thrust::host_vector<type_t> h(x.begin(), x.begin() + k);
  for (int i = 0; i < k; ++i) std::cout << h[i] << " ";
  std::cout << std::endl;
}

template <typename type_t>
void head(type_t* x, int k, int n, std::string name = "") {
  if (n < k) k = n;
  if (!name.empty()) std::cout << name << "[:" << k << "] = ";
  std::vector<type_t> h((size_t)(k > 0 ? k : 0));
  if (k > 0) (void)hipMemcpy(h.data(), x, (size_t)k * sizeof(type_t), hipMemcpyDefault);
  for (int i = 0; i < k; ++i) std::cout << h[i] << " ";
  std::cout << std::endl;
}

}  // namespace print
}  // namespace gunrock
