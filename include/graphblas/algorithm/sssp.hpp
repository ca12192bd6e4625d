// Shadow of graphblas/algorithm/sssp.hpp (see fused_detail.hpp).  algorithm::sssp runs the
// reference's synchronous Bellman-Ford rounds inside the library (grb_sssp: one persistent
// launch while the frontier is sparse, sssp_persist.hip): the same distances after every
// round, the same iteration count under any --max_niter.
#ifndef GRB_HIP_ALGORITHM_SSSP_SHADOW_HPP_
#define GRB_HIP_ALGORITHM_SSSP_SHADOW_HPP_

#include <limits>
#include <vector>
#include <string>
#include <queue>
#include <utility>
#include <functional>
#include "graphblas/algorithm/test_sssp.hpp"
#include "graphblas/backend/cuda/util.hpp"
#include "graphblas/algorithm/fused_detail.hpp"

#define sssp sssp_call_sequence
#include_next "graphblas/algorithm/sssp.hpp"
#undef sssp

namespace graphblas {
namespace algorithm {

inline float sssp(Vector<float>* v, const Matrix<float>* A, Index s, Descriptor* desc) {
  backend::Descriptor* d = &desc->descriptor_;
  if (!detail::fused_enabled() || d->debug()) return sssp_call_sequence(v, A, s, desc);
  detail::push_mirror(d);
  grb_algo_result r;
  const grb_info info = grb_sssp(v->handle(), A->handle(), s, d->h_, &r);
  if (info != GRB_SUCCESS) return sssp_call_sequence(v, A, s, desc);
  d->sync();
  if (d->timing_ == 1) {
    Index A_nrows;
    A->nrows(&A_nrows);
    const std::vector<grb_algo_iter> log = detail::iter_log(d);
    // sssp.hpp:55-62 prints round k at the top of iteration k + 1, :92-96 the last one after the loop
    for (size_t k = 0; k < log.size(); ++k) {
      const bool last = k + 1 == log.size();
      const int label = last ? r.iterations : log[k].iteration;
      std::cout << label << ", " << static_cast<Index>(log[k].value) << "/" << A_nrows << ", "
                << detail::mode_name(log[k].direction) << ", " << log[k].ms << "\n";
    }
  }
  return r.tight_ms;
}

}  // namespace algorithm
}  // namespace graphblas

#endif  // GRB_HIP_ALGORITHM_SSSP_SHADOW_HPP_
