// Shadow of graphblas/algorithm/tc.hpp (see fused_detail.hpp).  algorithm::tc goes to the library's grb_tc: on the
// matrix the reference's driver builds (a strictly lower triangle of ones) the count on the degree-ordered orientation
// of the same edges (csrc/tc_count.hip), B -- "buffer matrix" in the reference's signature -- left as it was; on any
// other matrix the reference's two calls (mxm into B, reduce B) inside the library.  The same number, the same
// return value (the "tight" time), the same line under --timing 1 / 2.
#ifndef GRB_HIP_ALGORITHM_TC_SHADOW_HPP_
#define GRB_HIP_ALGORITHM_TC_SHADOW_HPP_

#include <limits>
#include <vector>
#include <string>
#include "graphblas/algorithm/test_tc.hpp"
#include "graphblas/backend/cuda/util.hpp"
#include "graphblas/algorithm/fused_detail.hpp"

#define tc tc_call_sequence
#include_next "graphblas/algorithm/tc.hpp"
#undef tc

namespace graphblas {
namespace algorithm {

inline float tc(int* ntris, const Matrix<int>* A, Matrix<int>* B, Descriptor* desc) {
  backend::Descriptor* d = &desc->descriptor_;
  if (!detail::fused_enabled() || d->debug()) return tc_call_sequence(ntris, A, B, desc);
  detail::push_mirror(d);
  grb_algo_result r;
  int64_t wide = 0;
  // (grb_tc toggles GrB_INP1 itself and leaves it toggled, as tc.hpp:23 does)
  const grb_info info = grb_tc(&wide, A->handle(), B->handle(), d->h_, &r);
  if (info != GRB_SUCCESS) {
    grb_descriptor_toggle(d->h_, GrB_INP1);             // the call sequence toggles it again
    return tc_call_sequence(ntris, A, B, desc);
  }
  d->sync();
  *ntris = static_cast<int>(wide);                      // reduce<int, int> of the reference: the count in an int
  if (d->timing_ > 0) {
    Index A_nrows;
    A->nrows(&A_nrows);
    // tc.hpp:46-50: iter - 1 = 0, error = 1
    std::cout << 0 << ", " << 1.f << "/" << A_nrows << ", " << detail::mode_name(d->lastmxv_) << ", " << r.tight_ms << "\n";
  }
  return r.tight_ms;
}

}  // namespace algorithm
}  // namespace graphblas

#endif  // GRB_HIP_ALGORITHM_TC_SHADOW_HPP_
