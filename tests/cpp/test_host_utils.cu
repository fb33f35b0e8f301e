// Host-only checks of the header utilities that need no GPU: operators::batch::execute,
// generate::random and format::csc_t::from_csr (reference: framework/operators/batch/batch.hxx:70-94,
// algorithms/generate/random.hxx:20-52, formats/csc.hxx:62-101).  Built with hipcc, runs on the CPU.
#include <gunrock/algorithms/generate/random.hxx>
#include <gunrock/algorithms/search/binary_search.hxx>
#include <gunrock/formats/formats.hxx>
#include <gunrock/io/sample.hxx>
#include <gunrock/framework/operators/batch/batch.hxx>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <stdexcept>
#include <vector>

static int failures = 0;
#define CHECK(cond)                                               \
  do {                                                            \
    if (!(cond)) {                                                \
      std::printf("CHECK FAILED %s:%d %s\n", __FILE__, __LINE__, #cond); \
      ++failures;                                                 \
    } else {                                                      \
      std::printf("CHECK ok: %s\n", #cond);                       \
    }                                                             \
  } while (0)

int main() {
  using namespace gunrock;
  // every job runs exactly once, whatever the pool size
  {
    const std::size_t n = 1000;
    std::vector<std::atomic<int>> hits(n);
    for (auto& h : hits) h = 0;
    float total = -1.0f;
    operators::batch::execute([&](std::size_t j) -> float { ++hits[j]; return (float)j; }, n, &total);
    bool once = true;
    for (auto& h : hits) once = once && h.load() == 1;
    CHECK(once);
    CHECK(total >= 0.0f);
  }
  // zero jobs: returns, elapsed written
  {
    float total = -1.0f;
    operators::batch::execute([](std::size_t) -> float { return 1.0f; }, 0, &total);
    CHECK(total >= 0.0f);
  }
  // an exception in a job surfaces in the caller after the batch drained
  {
    float total = 0.0f;
    bool thrown = false;
    std::atomic<int> ran{0};
    try {
      operators::batch::execute(
          [&](std::size_t j) -> float {
            ++ran;
            if (j == 3) throw std::runtime_error("job 3");
            return 0.0f;
          },
          16, &total);
    } catch (const std::runtime_error&) {
      thrown = true;
    }
    CHECK(thrown);
    CHECK(ran.load() == 16);
  }
  // uniform_distribution fills the whole vector inside [begin, end) and is reproducible
  {
    std::vector<float> a(4096), b(4096);
    generate::random::uniform_distribution(a, 2.0f, 3.0f);
    generate::random::uniform_distribution(b, 2.0f, 3.0f);
    bool in_range = true;
    for (float x : a) in_range = in_range && x >= 2.0f && x < 3.0f;
    CHECK(in_range);
    CHECK(a == b);
    const float r = generate::random::get_random<float>(5.0f, 6.0f);
    CHECK(r >= 5.0f && r <= 6.0f);
  }
  // csc_t::from_csr: the transpose, entries of a column in row order (duplicates and self loops kept), values carried along
  {
    using csr_h = format::csr_t<memory_space_t::host, int, int, float>;
    using coo_h = format::coo_t<memory_space_t::host, int, int, float>;
    const int I[] = {0, 0, 2, 2, 2, 3, 1, 0}, J[] = {1, 3, 0, 3, 3, 3, 1, 1};  // file order; (2,3) twice, loops (3,3), (1,1)
    coo_h coo(4, 4, 8);
    for (int k = 0; k < 8; ++k) { coo.row_indices[k] = I[k]; coo.column_indices[k] = J[k]; coo.nonzero_values[k] = 10.0f * I[k] + J[k]; }
    csr_h csr;
    csr.from_coo(coo);
    format::csc_t<memory_space_t::host, int, int, float> csc;
    csc.from_csr(csr);
    const std::vector<int> want_off = {0, 1, 4, 4, 8};
    const std::vector<int> want_rows = {2, 0, 0, 1, 0, 2, 2, 3};  // column 1: rows 0 (file order: two entries), 1; column 3: 0, 2, 2, 3
    bool ok = csc.number_of_rows == 4 && csc.number_of_columns == 4 && csc.number_of_nonzeros == 8;
    for (int c = 0; c <= 4 && ok; ++c) ok = csc.column_offsets[c] == want_off[c];
    for (int k = 0; k < 8 && ok; ++k) ok = csc.row_indices[k] == want_rows[k];
    for (int c = 0; c < 4 && ok; ++c)
      for (int k = want_off[c]; k < want_off[c + 1] && ok; ++k) ok = csc.nonzero_values[k] == 10.0f * csc.row_indices[k] + c;
    CHECK(ok);
  }
  // io::sample::csr: the reference's 4 x 4 unit-test matrix (io/sample.hxx:47-78; SURVEY 8c golden vector 2)
  {
    auto m = io::sample::csr<memory_space_t::host>();
    const int ro[5] = {0, 0, 2, 3, 4}, ci[4] = {0, 1, 2, 1};
    const float w[4] = {5, 8, 3, 6};
    bool ok = m.number_of_rows == 4 && m.number_of_columns == 4 && m.number_of_nonzeros == 4;
    for (int i = 0; i < 5 && ok; ++i) ok = m.row_offsets[i] == ro[i];
    for (int i = 0; i < 4 && ok; ++i) ok = m.column_indices[i] == ci[i] && m.nonzero_values[i] == w[i];
    CHECK(ok);
  }
  // search::binary::execute: upper bound by default, lower bound on request (algorithms/search/binary_search.hxx:41-60)
  {
    const std::vector<int> keys = {0, 0, 2, 3, 3, 3, 9};
    const int* k = keys.data();
    bool ok = true;
    for (int key = -1; key <= 10; ++key) {
      const int up = (int)(std::upper_bound(keys.begin(), keys.end(), key) - keys.begin());
      const int lo = (int)(std::lower_bound(keys.begin(), keys.end(), key) - keys.begin());
      ok = ok && search::binary::execute(k, key, 0, (int)keys.size()) == up;
      ok = ok && search::binary::execute(k, key, 0, (int)keys.size(), search::bound_t::lower) == lo;
    }
    ok = ok && search::binary::execute(k, 3, 4, 4) == 4;  // empty range
    CHECK(ok);
  }
  std::printf(failures ? "FAILED\n" : "ALL CHECKS PASSED\n");
  return failures ? 1 : 0;
}
