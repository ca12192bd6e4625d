// Test driver: graphblas::apply through the drop-in frontend.  A unary operator the device knows (the header's
// unary_* / bind_first / bind_second functors) runs as one kernel whatever GrB_BACKEND says; any other functor -- here
// a stateful counter like the reference's set_random (algorithm/common.hpp:8-20) -- keeps the reference's host loop in
// index order under GrB_SEQUENTIAL (backend/cuda/apply.hpp:34-42,102-111).  Prints the results for the test to check.
#define GRB_USE_CUDA
#include <cstdio>
#include <vector>
#include "graphblas/graphblas.hpp"

template <typename T>
struct counting {                      // stateful: the k-th call returns x + k
  int k;
  counting() : k(0) {}
  inline T operator()(T x) { return x + static_cast<T>(k++); }
};

int main() {
  using namespace graphblas;
  const Index n = 10;
  Descriptor desc;
  std::vector<float> vals(n);
  for (Index i = 0; i < n; ++i) vals[i] = static_cast<float>(i) - 4.f;
  Vector<float> u(n), w(n);
  u.build(&vals, n);
  // device operators; GrB_BACKEND stays at its default (not GrB_SEQUENTIAL): the reference would print
  // "DeVec apply GPU / not implemented" here
  std::vector<float> out(n);
  Index m = n;
  apply<float, float, float>(&w, GrB_NULL, GrB_NULL, bind_second<multiplies<float>, float>(2.5f), &u, &desc);
  w.extractTuples(&out, &m);
  for (Index i = 0; i < n; ++i) printf("%g ", out[i]);
  printf("\n");
  apply<float, float, float>(&w, GrB_NULL, GrB_NULL, unary_abs<float>(), &u, &desc);
  w.extractTuples(&out, &m);
  for (Index i = 0; i < n; ++i) printf("%g ", out[i]);
  printf("\n");
  apply<float, float, float>(&u, GrB_NULL, GrB_NULL, bind_first<minus<float>, float>(10.f), &u, &desc);   // in place
  u.extractTuples(&out, &m);
  for (Index i = 0; i < n; ++i) printf("%g ", out[i]);
  printf("\n");
  // a host functor: the reference's host loop, in index order
  desc.set(GrB_BACKEND, GrB_SEQUENTIAL);
  apply<float, float, float>(&w, GrB_NULL, GrB_NULL, counting<float>(), &u, &desc);
  desc.set(GrB_BACKEND, GrB_CUDA);
  w.extractTuples(&out, &m);
  for (Index i = 0; i < n; ++i) printf("%g ", out[i]);
  printf("\n");
  // a matrix, in place on the device, then a product that must see the new values
  std::vector<Index> ri, ci;
  std::vector<float> av;
  for (Index i = 0; i < n; ++i) { ri.push_back(i); ci.push_back((i + 1) % n); av.push_back(static_cast<float>(i + 1)); }
  Matrix<float> A(n, n);
  A.build(&ri, &ci, &av, n, GrB_NULL);
  apply<float, float, float>(&A, GrB_NULL, GrB_NULL, bind_second<plus<float>, float>(100.f), &A, &desc);
  std::vector<float> ones(n, 1.f);
  Vector<float> x(n), y(n);
  x.build(&ones, n);
  mxv<float, float, float, float>(&y, GrB_NULL, GrB_NULL, PlusMultipliesSemiring<float>(), &A, &x, &desc);
  y.extractTuples(&out, &m);
  for (Index i = 0; i < n; ++i) printf("%g ", out[i]);
  printf("\n");
  return 0;
}
