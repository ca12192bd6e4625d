// Shadow of graphblas/algorithm/pr.hpp (see fused_detail.hpp).  algorithm::pr runs the power
// iteration inside the library (grb_pr: the SpMV kernel + one pass for the element-wise tail
// and the residual): per element the reference's arithmetic, the same stopping rule.
#ifndef GRB_HIP_ALGORITHM_PR_SHADOW_HPP_
#define GRB_HIP_ALGORITHM_PR_SHADOW_HPP_

#include <limits>
#include <vector>
#include <string>
#include <queue>
#include <utility>
#include <functional>
#include <cmath>
#include "graphblas/algorithm/test_pr.hpp"
#include "graphblas/backend/cuda/util.hpp"
#include "graphblas/algorithm/fused_detail.hpp"

#define pr pr_call_sequence
#include_next "graphblas/algorithm/pr.hpp"
#undef pr

namespace graphblas {
namespace algorithm {

inline float pr(Vector<float>* p, const Matrix<float>* A, float alpha, float eps, Descriptor* desc) {
  backend::Descriptor* d = &desc->descriptor_;
  if (!detail::fused_enabled() || d->debug()) return pr_call_sequence(p, A, alpha, eps, desc);
  detail::push_mirror(d);
  grb_algo_result r;
  const grb_info info = grb_pr(p->handle(), A->handle(), alpha, eps, d->h_, &r);
  if (info != GRB_SUCCESS) return pr_call_sequence(p, A, alpha, eps, desc);
  d->sync();
  if (d->timing_ > 0) {
    Index A_nrows;
    A->nrows(&A_nrows);
    const std::vector<grb_algo_iter> log = detail::iter_log(d);
    // pr.hpp:53-63: iteration k is printed at the top of iteration k + 1 (--timing 1 only), after
    // `unvisited -= (int)error` of the iterations before it; :85-89 the last line under --timing 1 or 2
    Index unvisited = A_nrows - 1;                      // error starts at 1.f
    for (size_t k = 0; k < log.size(); ++k) {
      const bool last = k + 1 == log.size();
      if (last)
        std::cout << r.iterations + 1 << ", " << static_cast<float>(log[k].value) << "/" << A_nrows << ", " << unvisited
                  << ", " << detail::mode_name(log[k].direction) << ", " << log[k].ms << "\n";
      else if (d->timing_ == 1)
        std::cout << log[k].iteration << ", " << static_cast<float>(log[k].value) << "/" << A_nrows << ", " << unvisited
                  << ", " << detail::mode_name(log[k].direction) << ", " << log[k].ms << "\n";
      unvisited -= static_cast<int>(static_cast<float>(log[k].value));
    }
  }
  return r.tight_ms;
}

}  // namespace algorithm
}  // namespace graphblas

#endif  // GRB_HIP_ALGORITHM_PR_SHADOW_HPP_
