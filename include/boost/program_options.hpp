// Minimal stand-in for the one Boost.ProgramOptions surface the GraphBLAST examples
// touch: `po::variables_map vm; parseArgs(argc, argv, &vm); vm["name"].as<T>()`.
// The option table itself lives in graphblas/graphblas.hpp (parseArgs).  Part of the
// drop-in boundary of graphblast_amd; not used by the oracle and never used to build
// reference code for testing.
#ifndef GRAPHBLAST_AMD_BOOST_PROGRAM_OPTIONS_SHIM_HPP_
#define GRAPHBLAST_AMD_BOOST_PROGRAM_OPTIONS_SHIM_HPP_

// applications do `#define private public` before including this header; libstdc++ does
// not survive that, so standard headers are pulled in with the macro suspended
#pragma push_macro("private")
#undef private
#include <cassert>
#include <cstdlib>
#include <map>
#include <sstream>
#include <string>
#pragma pop_macro("private")

#ifndef BOOST_ASSERT
#define BOOST_ASSERT(expr) assert(expr)
#endif

namespace boost {
namespace program_options {

class variable_value {
 public:
  variable_value() {}
  explicit variable_value(const std::string& text) : text_(text) {}
  template <typename T>
  T as() const {
    T out = T();
    std::istringstream ss(text_);
    ss >> out;
    return out;
  }
  const std::string& text() const { return text_; }

 private:
  std::string text_;
};

template <>
inline bool variable_value::as<bool>() const {
  return !(text_.empty() || text_ == "0" || text_ == "false");
}
template <>
inline std::string variable_value::as<std::string>() const {
  return text_;
}

class variables_map {
 public:
  const variable_value& operator[](const std::string& key) const {
    static const variable_value empty;
    std::map<std::string, variable_value>::const_iterator it = kv_.find(key);
    return it == kv_.end() ? empty : it->second;
  }
  size_t count(const std::string& key) const { return kv_.count(key); }
  void set(const std::string& key, const std::string& text) { kv_[key] = variable_value(text); }

 private:
  std::map<std::string, variable_value> kv_;
};

}  // namespace program_options
}  // namespace boost

#endif  // GRAPHBLAST_AMD_BOOST_PROGRAM_OPTIONS_SHIM_HPP_
