// fused_detail.hpp -- shared by the algorithm shadows in this directory (bfs.hpp, sssp.hpp,
// pr.hpp).  An application that includes "graphblas/algorithm/<x>.hpp" with include/ ahead of
// the reference tree on its include path gets these files; each one
//   * compiles the reference's own header of that name, found further down the include path
//     (#include_next), under another function name -- that text stays the op-by-op path, and
//   * defines the driver with the reference's signature: the library's one-launch / fused
//     driver (grb_bfs_fused, grb_sssp, grb_pr) when the call is one those drivers reproduce
//     exactly, the reference's text otherwise (--debug, GRB_FRONTEND_FUSED=0, or the library
//     declining the call).
// Same labels / distances / ranks, same return value meaning (the "tight" time), the same
// per-iteration lines under --timing 1 (printed from the drivers' per-iteration records).
#ifndef GRB_HIP_ALGORITHM_FUSED_DETAIL_HPP_
#define GRB_HIP_ALGORITHM_FUSED_DETAIL_HPP_

#include <cstdlib>
#include <iostream>
#include <string>
#include <vector>

namespace graphblas {
namespace algorithm {
namespace detail {

inline bool fused_enabled() {
  static const bool on = [] { const char* e = getenv("GRB_FRONTEND_FUSED"); return !e || atoi(e) != 0; }();
  return on;
}

// The applications read and write the mirrored members directly (#define private public);
// the library's descriptor must see what they set before a driver runs.
inline void push_mirror(backend::Descriptor* d) {
  grb_descriptor_set_arg(d->h_, "max_niter", static_cast<double>(d->max_niter_));
  grb_descriptor_set_arg(d->h_, "timing", static_cast<double>(d->timing_));
}

inline const char* mode_name(int lastmxv) { return lastmxv == GRB_PUSHONLY ? "push" : "pull"; }

inline std::vector<grb_algo_iter> iter_log(backend::Descriptor* d) {
  int count = 0;
  grb_descriptor_iter_log(d->h_, NULL, 0, &count);
  std::vector<grb_algo_iter> log(static_cast<size_t>(count > 0 ? count : 0));
  if (count > 0) grb_descriptor_iter_log(d->h_, log.data(), count, &count);
  return log;
}

}  // namespace detail
}  // namespace algorithm
}  // namespace graphblas

#endif  // GRB_HIP_ALGORITHM_FUSED_DETAIL_HPP_
