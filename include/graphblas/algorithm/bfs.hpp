// Shadow of graphblas/algorithm/bfs.hpp (see fused_detail.hpp).  algorithm::bfs runs the whole
// direction-optimised traversal as one launch (grb_bfs_fused -> bfs_persist.hip): the same
// depth labels, the same direction per level, the same lastmxv_ afterwards.
#ifndef GRB_HIP_ALGORITHM_BFS_SHADOW_HPP_
#define GRB_HIP_ALGORITHM_BFS_SHADOW_HPP_

#include <string>
#include <vector>
#include <deque>
#include "graphblas/algorithm/test_bfs.hpp"
#include "graphblas/backend/cuda/util.hpp"
#include "graphblas/algorithm/fused_detail.hpp"

// the reference's own text = the call-sequence path (also provides bfsCpu)
#define bfs bfs_call_sequence
#include_next "graphblas/algorithm/bfs.hpp"
#undef bfs

namespace graphblas {
namespace algorithm {

inline float bfs(Vector<float>* v, const Matrix<float>* A, Index s, Descriptor* desc) {
  backend::Descriptor* d = &desc->descriptor_;
  if (!detail::fused_enabled() || d->debug()) return bfs_call_sequence(v, A, s, desc);
  detail::push_mirror(d);
  const bool lines = d->timing_ == 1;
  std::vector<grb_bfs_level> lv(lines ? 1 << 15 : 0);
  grb_bfs_result r;
  const grb_info info = grb_bfs_fused(v->handle(), A->handle(), s, d->h_, &r, lines ? lv.data() : NULL,
                                      static_cast<int>(lv.size()), 0);
  if (info != GRB_SUCCESS) return bfs_call_sequence(v, A, s, desc);
  d->sync();
  if (lines && r.levels > 0 && r.levels <= static_cast<int>(lv.size())) {
    Index A_nrows;
    A->nrows(&A_nrows);
    // bfs.hpp:52-60 prints level k at the top of iteration k + 1, :81-86 the last one after the loop
    Index unvisited = A_nrows;
    for (int k = 1; k <= r.levels; ++k) {
      const grb_bfs_level& L = lv[k - 1];
      const bool last = k == r.levels;
      const int label = (last && L.discovered != 0) ? k + 1 : k;      // loop left by the max_niter cap: iter = cap + 1
      std::cout << label << ", " << static_cast<float>(L.discovered) << "/" << A_nrows << ", " << unvisited << ", "
                << detail::mode_name(L.direction ? GRB_PULLONLY : GRB_PUSHONLY) << ", " << L.ms << "\n";
      unvisited -= L.discovered;
    }
  }
  return r.tight_ms;
}

}  // namespace algorithm
}  // namespace graphblas

#endif  // GRB_HIP_ALGORITHM_BFS_SHADOW_HPP_
