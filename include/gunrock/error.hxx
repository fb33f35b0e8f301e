// error.hxx -- exception type carried across the whole API.
// API parity: include/gunrock/error.hxx:15-45 (reference): error_t, exception_t
// (what() = "<hipGetErrorString>\t: <message>"), throw_if_exception(status|bool, msg).
#pragma once

#include <hip/hip_runtime.h>

#include <exception>
#include <string>

namespace gunrock {
namespace error {

typedef hipError_t error_t;

struct exception_t : std::exception {
  std::string report;
  exception_t(error_t status, std::string message = "")
      : report(std::string(hipGetErrorString(status)) + "\t: " + message) {}
  exception_t(std::string message = "") : report(message) {}
  const char* what() const noexcept override { return report.c_str(); }
};

inline void throw_if_exception(error_t status, std::string message = "") {
  if (status != hipSuccess) throw exception_t(status, message);
}

inline void throw_if_exception(bool is_exception, std::string message = "") {
  if (is_exception) throw exception_t(message);
}

}  // namespace error
}  // namespace gunrock
