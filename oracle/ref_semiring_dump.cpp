// Driver compiled against the REFERENCE's own graphblas/stddef.hpp (never copied into
// this repo) to print its semiring table as JSON.  Built only when /root/reference is
// present:  make -C oracle ref   ->  oracle/_ref/semiring_ref.
// The header needs no stand-ins: GRB_HOST_DEVICE / __host__ / __device__ are emptied on
// the command line and the unqualified min/max it calls are std::min/std::max.
#include <algorithm>
#include <cstdio>
#include <limits>
using std::max;
using std::min;
#include "graphblas/stddef.hpp"

using namespace graphblas;

static const double kProbe[] = {0, 1, 3, 5, -2, 7};
static bool first_entry = true;

template <typename SR, typename T>
void dump(const char* name, const char* dtype) {
  SR sr;
  printf("%s\n  {\"semiring\": \"%s\", \"dtype\": \"%s\", \"identity\": %.9g, \"add\": [", first_entry ? "" : ",",
         name, dtype, (double)sr.identity());
  first_entry = false;
  bool f = true;
  for (double a : kProbe)
    for (double b : kProbe) {
      printf("%s[%g, %g, %.9g]", f ? "" : ", ", a, b, (double)sr.add_op((T)a, (T)b));
      f = false;
    }
  printf("], \"mul\": [");
  f = true;
  for (double a : kProbe)
    for (double b : kProbe) {
      if (b == 0) continue;  // keep integer division defined
      printf("%s[%g, %g, %.9g]", f ? "" : ", ", a, b, (double)sr.mul_op((T)a, (T)b));
      f = false;
    }
  printf("]}");
}

#define DUMP(SR) dump<SR##Semiring<float>, float>(#SR, "f32"); dump<SR##Semiring<int>, int>(#SR, "i32");

int main() {
  printf("[");
  DUMP(LogicalOrAnd) DUMP(PlusMultiplies) DUMP(MinimumPlus) DUMP(MaximumMultiplies) DUMP(PlusDivides)
  DUMP(PlusGreater) DUMP(GreaterPlus) DUMP(PlusMinus) DUMP(PlusLess) DUMP(CustomLessPlus)
  DUMP(MinimumMultiplies) DUMP(MultipliesMultiplies) DUMP(NotEqualToPlus) DUMP(MinimumSelectSecond)
  DUMP(PlusNotEqualTo) DUMP(CustomLessLess) DUMP(MinimumNotEqualTo)
  printf("\n]\n");
  return 0;
}
