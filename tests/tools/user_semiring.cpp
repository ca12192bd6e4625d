// Test driver (host only): an application registers its own monoid and semirings with the reference's macros
// (graphblas/stddef.hpp:140-191) against the drop-in frontend, and prints the C-ABI ids they resolve to.
#define GRB_USE_CUDA
#include <cstdio>
#include <limits>
#include "graphblas/graphblas.hpp"

REGISTER_MONOID(MyFloorMonoid, maximum, -1000)
namespace graphblas {
REGISTER_SEMIRING(MaxPlusSemiring, MaximumMonoid, plus)                   // tropical (max, +): not among the 17
REGISTER_SEMIRING(FloorTimesSemiring, MyFloorMonoid, multiplies)         // a user monoid with its own identity
REGISTER_SEMIRING(MyPlusTimesSemiring, PlusMonoid, multiplies)           // the same composition as PlusMultiplies
}  // namespace graphblas

int main() {
  using namespace graphblas;
  MaxPlusSemiring<float> a;
  FloorTimesSemiring<float> b;
  printf("%d %d %d %d\n", (int)detail::sr_id<PlusMultipliesSemiring<float> >(), (int)detail::sr_id<MaxPlusSemiring<float> >(),
         (int)detail::sr_id<FloorTimesSemiring<float> >(), (int)detail::sr_id<MyPlusTimesSemiring<float> >());
  printf("%g %g %g %g %g\n", a.identity(), a.add_op(2.f, 5.f), a.mul_op(2.f, 5.f), b.identity(), b.mul_op(3.f, 4.f));
  printf("%d\n", (int)detail::sr_id<MaxPlusSemiring<float> >());            // registered once
  return 0;
}
