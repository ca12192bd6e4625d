// ORACLE -- TEST INFRASTRUCTURE ONLY.  Prints the numeric values of the enumerations of the reference's own
// graphblas/types.hpp (+ backend/cuda/types.hpp), compiled from where it lies with nothing but a -D for the
// backend root (oracle/Makefile, target `ref`).  tests/golden/make_golden.py stores the output as
// tests/golden/types_ref.json; tests/test_abi.py compares include/grb_hip.h and the Python mirror with it.
#include <cstdio>

#include "graphblas/types.hpp"

#define P(x) std::printf("  \"%s\": %d,\n", #x, static_cast<int>(graphblas::x))
int main() {
  std::printf("{\n");
  P(GrB_UNKNOWN); P(GrB_SPARSE); P(GrB_DENSE);
  P(GrB_SUCCESS); P(GrB_UNINITIALIZED_OBJECT); P(GrB_NULL_POINTER); P(GrB_INVALID_VALUE); P(GrB_INVALID_INDEX);
  P(GrB_DOMAIN_MISMATCH); P(GrB_DIMENSION_MISMATCH); P(GrB_OUTPUT_NOT_EMPTY); P(GrB_NO_VALUE); P(GrB_NOT_IMPLEMENTED);
  P(GrB_OUT_OF_MEMORY); P(GrB_INSUFFICIENT_SPACE); P(GrB_INVALID_OBJECT); P(GrB_INDEX_OUT_OF_BOUNDS); P(GrB_PANIC);
  P(GrB_MASK); P(GrB_OUTP); P(GrB_INP0); P(GrB_INP1); P(GrB_MODE); P(GrB_TA); P(GrB_TB); P(GrB_NT); P(GrB_MXVMODE);
  P(GrB_TOL); P(GrB_BACKEND); P(GrB_NDESCFIELD);
  P(GrB_SCMP); P(GrB_REPLACE); P(GrB_TRAN); P(GrB_DEFAULT); P(GrB_FIXEDROW); P(GrB_PUSHPULL); P(GrB_PUSHONLY);
  P(GrB_PULLONLY); P(GrB_SEQUENTIAL); P(GrB_CUDA); P(GrB_8); P(GrB_16); P(GrB_32); P(GrB_128);
  std::printf("  \"sizeof_Index\": %d,\n  \"sizeof_T\": %d\n}\n", (int)sizeof(graphblas::Index), (int)sizeof(graphblas::T));
  return 0;
}
