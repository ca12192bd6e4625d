// Minimal stand-in for the Boost.Test surface the GraphBLAST unit tests touch
// (test/g*.cu: BOOST_AUTO_TEST_SUITE / BOOST_FIXTURE_TEST_CASE / BOOST_AUTO_TEST_CASE /
// BOOST_ASSERT, with BOOST_TEST_MAIN supplying main).  Part of the drop-in boundary check of
// graphblast_amd (tools/build_reference_tests.sh compiles the reference's tests UNCHANGED against
// this backend's frontend header); never used to build reference code as an oracle.
#ifndef GRAPHBLAST_AMD_BOOST_TEST_SHIM_HPP_
#define GRAPHBLAST_AMD_BOOST_TEST_SHIM_HPP_

#pragma push_macro("private")
#undef private
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>
#pragma pop_macro("private")

#undef BOOST_ASSERT
#define BOOST_ASSERT(expr)                                                                  \
  do {                                                                                      \
    if (!(expr)) {                                                                          \
      std::fprintf(stderr, "%s:%d: assertion failed: %s\n", __FILE__, __LINE__, #expr);     \
      std::fflush(stderr);                                                                  \
      std::abort();                                                                         \
    }                                                                                       \
  } while (0)

namespace grb_boost_test_shim {
struct nil {};
struct test_case {
  std::string name;
  std::function<void()> run;
};
inline std::vector<test_case>& registry() {
  static std::vector<test_case> r;
  return r;
}
struct registrar {
  registrar(const char* name, std::function<void()> fn) { registry().push_back(test_case{name, fn}); }
};
inline int run_all() {
  for (const test_case& t : registry()) {
    std::printf("[ RUN  ] %s\n", t.name.c_str());
    std::fflush(stdout);
    t.run();
    std::printf("[  OK  ] %s\n", t.name.c_str());
  }
  std::printf("\n*** No errors detected (%d test cases)\n", static_cast<int>(registry().size()));
  return 0;
}
}  // namespace grb_boost_test_shim

#define BOOST_AUTO_TEST_SUITE(suite_name) namespace suite_name {
#define BOOST_AUTO_TEST_SUITE_END() }
#define BOOST_FIXTURE_TEST_CASE(case_name, fixture)                                              \
  struct case_name : public fixture {                                                            \
    void test_method();                                                                          \
  };                                                                                             \
  static grb_boost_test_shim::registrar case_name##_registrar(#case_name, []() {                 \
    case_name t;                                                                                 \
    t.test_method();                                                                             \
  });                                                                                            \
  void case_name::test_method()
#define BOOST_AUTO_TEST_CASE(case_name) BOOST_FIXTURE_TEST_CASE(case_name, grb_boost_test_shim::nil)

#ifdef BOOST_TEST_MAIN
int main(int, char**) { return grb_boost_test_shim::run_all(); }
#endif

#endif  // GRAPHBLAST_AMD_BOOST_TEST_SHIM_HPP_
