// Shadow of graphblas/algorithm/cc.hpp (see fused_detail.hpp).  algorithm::cc runs FastSV inside the
// library (grb_cc): per iteration the MinimumSelectSecond product plus ONE launch for the element-wise
// tail (three eWiseAdd, assignScatter, extractGather, eWiseMult, reduce, masked assign and two dup of
// cc.hpp:77-119; csrc/algorithms.hip: cc_tail_kernel).  Same parent vector at convergence, the same
// stopping rule, the per-iteration lines of --timing 2 printed from the driver's records.
#ifndef GRB_HIP_ALGORITHM_CC_SHADOW_HPP_
#define GRB_HIP_ALGORITHM_CC_SHADOW_HPP_

#include <limits>
#include <vector>
#include <string>
#include "graphblas/algorithm/test_cc.hpp"
#include "graphblas/backend/cuda/util.hpp"
#include "graphblas/algorithm/fused_detail.hpp"

#define cc cc_call_sequence
#include_next "graphblas/algorithm/cc.hpp"
#undef cc

namespace graphblas {
namespace algorithm {

inline float cc(Vector<int>* v, const Matrix<int>* A, int seed, Descriptor* desc) {
  backend::Descriptor* d = &desc->descriptor_;
  if (!detail::fused_enabled() || d->debug()) return cc_call_sequence(v, A, seed, desc);
  detail::push_mirror(d);
  grb_algo_result r;
  const grb_info info = grb_cc(v->handle(), A->handle(), seed, d->h_, &r);
  if (info != GRB_SUCCESS) return cc_call_sequence(v, A, seed, desc);
  d->sync();
  if (d->timing_ == 2) {
    Index A_nrows;
    A->nrows(&A_nrows);
    const std::vector<grb_algo_iter> log = detail::iter_log(d);
    // cc.hpp:62-73 prints iteration k at the top of iteration k + 1, :124-130 the last one after the loop
    for (size_t k = 0; k < log.size(); ++k) {
      const bool last = k + 1 == log.size();
      const int label = last ? r.iterations : log[k].iteration;
      std::cout << label << ", " << static_cast<int>(log[k].value) << "/" << A_nrows << ", "
                << detail::mode_name(log[k].direction) << ", " << log[k].ms << "\n";
    }
  }
  return d->timing_ > 0 ? r.tight_ms : 0.f;
}

}  // namespace algorithm
}  // namespace graphblas

#endif  // GRB_HIP_ALGORITHM_CC_SHADOW_HPP_
