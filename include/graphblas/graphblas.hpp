// graphblas/graphblas.hpp -- the GraphBLAST C++ frontend over libgrb_hip.so.
//
// This single header is the "reference-side binding" of the drop-in boundary: it
// provides the `graphblas::` names a GraphBLAST application compiles against
// (graphblas/{types,stddef,descriptor,vector,matrix,operations,util}.hpp of the
// reference) and forwards every call to the C ABI of include/grb_hip.h, where the
// storage/direction dispatch and the gfx950 kernels live.  With
//     -I<graphblast_amd>/include -I<reference root>
// the reference's own example/gbfs.cu (+ its graphblas/algorithm/*.hpp and
// test/test.hpp) builds unchanged with g++ or hipcc and links against libgrb_hip.so.
//
// Same template parameter orders, argument orders, Info return codes and (because the
// reference's applications do `#define private public` and reach inside) the same
// member names: Descriptor::descriptor_.{max_niter_, timing_, lastmxv_, debug()},
// Matrix::matrix_.{nrows_, sparse_.h_csrRowPtr_ ...}, backend::GpuTimer.
// Element types: float and int (bool vectors are stored as int), as the reference
// instantiates them.  Semirings/monoids are the 17 + 9 of graphblas/stddef.hpp.
#ifndef GRAPHBLAST_AMD_GRAPHBLAS_HPP_
#define GRAPHBLAST_AMD_GRAPHBLAS_HPP_

#pragma push_macro("private")
#undef private
#include <sys/time.h>

#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <limits>
#include <numeric>
#include <string>
#include <type_traits>
#include <vector>
#pragma pop_macro("private")

#include <boost/program_options.hpp>

#include "grb_hip.h"

#define GrB_NULL NULL
#define GrB_ALL NULL
#ifndef GRB_HOST_DEVICE
#define GRB_HOST_DEVICE            /* user functors (algorithm/common.hpp) run on the host here */
#endif

namespace po = boost::program_options;

namespace graphblas {

typedef int Index;
typedef float T;

enum Storage { GrB_UNKNOWN, GrB_SPARSE, GrB_DENSE };
enum Major { GrB_ROWMAJOR, GrB_COLMAJOR };
enum Info {
  GrB_SUCCESS, GrB_UNINITIALIZED_OBJECT, GrB_NULL_POINTER, GrB_INVALID_VALUE, GrB_INVALID_INDEX,
  GrB_DOMAIN_MISMATCH, GrB_DIMENSION_MISMATCH, GrB_OUTPUT_NOT_EMPTY, GrB_NO_VALUE, GrB_NOT_IMPLEMENTED,
  GrB_OUT_OF_MEMORY, GrB_INSUFFICIENT_SPACE, GrB_INVALID_OBJECT, GrB_INDEX_OUT_OF_BOUNDS, GrB_PANIC
};
enum Desc_field {
  GrB_MASK, GrB_OUTP, GrB_INP0, GrB_INP1, GrB_MODE, GrB_TA, GrB_TB, GrB_NT, GrB_MXVMODE, GrB_TOL, GrB_BACKEND,
  GrB_NDESCFIELD
};
enum Desc_value {
  GrB_SCMP = 0, GrB_REPLACE = 1, GrB_TRAN = 2, GrB_DEFAULT = 3, GrB_CUSPARSE = 4, GrB_CUSPARSE2 = 5,
  GrB_FIXEDROW = 6, GrB_FIXEDCOL = 7, GrB_MERGEPATH = 9, GrB_PUSHPULL = 10, GrB_PUSHONLY = 11, GrB_PULLONLY = 12,
  GrB_SEQUENTIAL = 13, GrB_CUDA = 14, GrB_8 = 8, GrB_16 = 16, GrB_32 = 32, GrB_64 = 64, GrB_128 = 128,
  GrB_256 = 256, GrB_512 = 512, GrB_1024 = 1024
};

inline Info to_info(grb_info i) { return static_cast<Info>(i); }

}  // namespace graphblas

#define CHECK(x)                                                                                   \
  do {                                                                                             \
    graphblas::Info err = x;                                                                       \
    if (err != graphblas::GrB_SUCCESS) {                                                           \
      fprintf(stderr, "Runtime error: %s returned %d at %s:%d\n", #x, err, __FILE__, __LINE__);    \
      return err;                                                                                  \
    }                                                                                              \
  } while (0)

#define CHECKVOID(x)                                                                               \
  do {                                                                                             \
    graphblas::Info err = x;                                                                       \
    if (err != graphblas::GrB_SUCCESS) {                                                           \
      fprintf(stderr, "Runtime error: %s returned %d at %s:%d\n", #x, err, __FILE__, __LINE__);    \
      return;                                                                                      \
    }                                                                                              \
  } while (0)

// ------------------------------------------------------------------------------------
// Operators, monoids, semirings: type tags carrying the C-ABI id, plus host evaluators
// (identity / add_op / mul_op) with the semantics of the reference's functors.
namespace graphblas {
namespace detail {
enum { kLor, kLand, kEq, kNe, kGt, kLt, kFirst, kSecond, kMin, kMax, kPlus, kMinus, kTimes, kDiv };
template <int OP, typename X>
inline X apply_op(X a, X b) {
  switch (OP) {
    case kLor: return static_cast<X>(a || b);
    case kLand: return static_cast<X>(a && b);
    case kEq: return static_cast<X>(a == b);
    case kNe: return static_cast<X>(a != b);
    case kGt: return static_cast<X>(a > b);
    case kLt: return static_cast<X>(a < b);
    case kFirst: return a;
    case kSecond: return b;
    case kMin: return std::min(a, b);
    case kMax: return std::max(a, b);
    case kPlus: return a + b;
    case kMinus: return a - b;
    case kTimes: return a * b;
    default: return a / b;
  }
}
// detail::k* -> grb_binary_op (include/grb_hip.h), for semirings composed with REGISTER_SEMIRING
template <int OP> struct op_code;
template <> struct op_code<kLor> { static const int value = GRB_OP_LOGICAL_OR; };
template <> struct op_code<kLand> { static const int value = GRB_OP_LOGICAL_AND; };
template <> struct op_code<kEq> { static const int value = GRB_OP_EQUAL; };
template <> struct op_code<kNe> { static const int value = GRB_OP_NOT_EQUAL_TO; };
template <> struct op_code<kGt> { static const int value = GRB_OP_GREATER; };
template <> struct op_code<kLt> { static const int value = GRB_OP_LESS; };
template <> struct op_code<kFirst> { static const int value = GRB_OP_FIRST; };
template <> struct op_code<kSecond> { static const int value = GRB_OP_SECOND; };
template <> struct op_code<kMin> { static const int value = GRB_OP_MINIMUM; };
template <> struct op_code<kMax> { static const int value = GRB_OP_MAXIMUM; };
template <> struct op_code<kPlus> { static const int value = GRB_OP_PLUS; };
template <> struct op_code<kMinus> { static const int value = GRB_OP_MINUS; };
template <> struct op_code<kTimes> { static const int value = GRB_OP_MULTIPLIES; };
template <> struct op_code<kDiv> { static const int value = GRB_OP_DIVIDES; };
template <typename X> inline X ident_zero() { return static_cast<X>(0); }
template <typename X> inline X ident_one() { return static_cast<X>(1); }
template <typename X> inline X ident_max() { return std::numeric_limits<X>::max(); }
template <typename X> inline X ident_min() { return std::numeric_limits<X>::min(); }
template <typename X> struct dtype_of;
template <> struct dtype_of<float> { static const grb_dtype value = GRB_F32; typedef float storage; };
template <> struct dtype_of<int> { static const grb_dtype value = GRB_I32; typedef int storage; };
template <> struct dtype_of<bool> { static const grb_dtype value = GRB_I32; typedef int storage; };
template <typename A> struct accum_present {
  static const bool value = !std::is_integral<A>::value && !std::is_pointer<A>::value &&
                            !std::is_same<A, std::nullptr_t>::value;
};
template <typename A> inline grb_accum accum_of(const A&) {
  return accum_present<A>::value ? GRB_ACCUM_PRESENT : GRB_ACCUM_NULL;
}
}  // namespace detail

#define GRB_MONOID(NAME, ID, OP, IDENT)                                                            \
  template <typename T_out>                                                                        \
  struct NAME {                                                                                    \
    static const int grb_id = ID;                                                                  \
    static const int grb_op = detail::op_code<detail::OP>::value;                                  \
    inline T_out identity() const { return detail::IDENT<T_out>(); }                               \
    inline T_out operator()(T_out lhs, T_out rhs) const { return detail::apply_op<detail::OP, T_out>(lhs, rhs); } \
  };
GRB_MONOID(PlusMonoid, GRB_PLUS_MONOID, kPlus, ident_zero)
GRB_MONOID(MultipliesMonoid, GRB_MULTIPLIES_MONOID, kTimes, ident_one)
GRB_MONOID(MinimumMonoid, GRB_MINIMUM_MONOID, kMin, ident_max)
GRB_MONOID(MaximumMonoid, GRB_MAXIMUM_MONOID, kMax, ident_zero)
GRB_MONOID(LogicalOrMonoid, GRB_LOGICAL_OR_MONOID, kLor, ident_zero)
GRB_MONOID(LogicalAndMonoid, GRB_LOGICAL_AND_MONOID, kLand, ident_zero)
GRB_MONOID(GreaterMonoid, GRB_GREATER_MONOID, kGt, ident_min)
GRB_MONOID(CustomLessMonoid, GRB_CUSTOM_LESS_MONOID, kLt, ident_max)
GRB_MONOID(NotEqualToMonoid, GRB_NOT_EQUAL_TO_MONOID, kNe, ident_max)
#undef GRB_MONOID

#define GRB_SEMIRING(NAME, ID, MONOID, MUL)                                                        \
  template <typename T_in1, typename T_in2 = T_in1, typename T_out = T_in1>                        \
  struct NAME {                                                                                    \
    typedef T_out result_type;                                                                     \
    typedef T_out T_out_type;                                                                      \
    static const int grb_id = ID;                                                                  \
    static int grb_registered_id() { return ID; }                                                  \
    inline T_out identity() const { return MONOID<T_out>().identity(); }                           \
    inline T_out add_op(T_out lhs, T_out rhs) const { return MONOID<T_out>()(lhs, rhs); }          \
    inline T_out mul_op(T_in1 lhs, T_in2 rhs) const {                                              \
      return detail::apply_op<detail::MUL, T_out>(static_cast<T_out>(lhs), static_cast<T_out>(rhs)); \
    }                                                                                              \
  };
GRB_SEMIRING(LogicalOrAndSemiring, GRB_LOGICAL_OR_AND, LogicalOrMonoid, kLand)
GRB_SEMIRING(PlusMultipliesSemiring, GRB_PLUS_MULTIPLIES, PlusMonoid, kTimes)
GRB_SEMIRING(MinimumPlusSemiring, GRB_MINIMUM_PLUS, MinimumMonoid, kPlus)
GRB_SEMIRING(MaximumMultipliesSemiring, GRB_MAXIMUM_MULTIPLIES, MaximumMonoid, kTimes)
GRB_SEMIRING(PlusDividesSemiring, GRB_PLUS_DIVIDES, PlusMonoid, kDiv)
GRB_SEMIRING(PlusGreaterSemiring, GRB_PLUS_GREATER, PlusMonoid, kGt)
GRB_SEMIRING(GreaterPlusSemiring, GRB_GREATER_PLUS, GreaterMonoid, kPlus)
GRB_SEMIRING(PlusMinusSemiring, GRB_PLUS_MINUS, PlusMonoid, kMinus)
GRB_SEMIRING(PlusLessSemiring, GRB_PLUS_LESS, PlusMonoid, kLt)
GRB_SEMIRING(CustomLessPlusSemiring, GRB_CUSTOM_LESS_PLUS, CustomLessMonoid, kPlus)
GRB_SEMIRING(MinimumMultipliesSemiring, GRB_MINIMUM_MULTIPLIES, MinimumMonoid, kTimes)
GRB_SEMIRING(MultipliesMultipliesSemiring, GRB_MULTIPLIES_MULTIPLIES, MultipliesMonoid, kTimes)
GRB_SEMIRING(NotEqualToPlusSemiring, GRB_NOT_EQUAL_TO_PLUS, NotEqualToMonoid, kPlus)
GRB_SEMIRING(MinimumSelectSecondSemiring, GRB_MINIMUM_SELECT_SECOND, MinimumMonoid, kSecond)
GRB_SEMIRING(PlusNotEqualToSemiring, GRB_PLUS_NOT_EQUAL_TO, PlusMonoid, kNe)
GRB_SEMIRING(CustomLessLessSemiring, GRB_CUSTOM_LESS_LESS, CustomLessMonoid, kLt)
GRB_SEMIRING(MinimumNotEqualToSemiring, GRB_MINIMUM_NOT_EQUAL_TO, MinimumMonoid, kNe)
#undef GRB_SEMIRING

// The binary-operator functor templates of graphblas/stddef.hpp:14-138, each carrying its grb_binary_op code,
// and the two generator macros applications use to make their own monoids and semirings (:140-191).
#define GRB_BINARYOP(NAME, CODE, DEF1, DEF2, DEFO, EXPR)                                           \
  template <typename T_in1 DEF1, typename T_in2 DEF2, typename T_out DEFO>                         \
  struct NAME {                                                                                    \
    static const int grb_op = CODE;                                                                \
    inline T_out operator()(T_in1 lhs, T_in2 rhs) const { return static_cast<T_out>(EXPR); }       \
  };
#define GRB_C ,
GRB_BINARYOP(logical_or, GRB_OP_LOGICAL_OR, = bool, = bool, = bool, lhs || rhs)
GRB_BINARYOP(logical_and, GRB_OP_LOGICAL_AND, = bool, = bool, = bool, lhs && rhs)
GRB_BINARYOP(logical_xor, GRB_OP_LOGICAL_XOR, = bool, = bool, = bool, (lhs && !rhs) || (!lhs && rhs))
GRB_BINARYOP(equal, GRB_OP_EQUAL, , = T_in1, = T_in1, lhs == rhs)
GRB_BINARYOP(not_equal_to, GRB_OP_NOT_EQUAL_TO, , = T_in1, = T_in1, lhs != rhs)
GRB_BINARYOP(greater, GRB_OP_GREATER, , = T_in1, = bool, lhs > rhs)
GRB_BINARYOP(less, GRB_OP_LESS, , = T_in1, = bool, lhs < rhs)
GRB_BINARYOP(greater_equal, GRB_OP_GREATER_EQUAL, , = T_in1, = bool, lhs >= rhs)
GRB_BINARYOP(less_equal, GRB_OP_LESS_EQUAL, , = T_in1, = bool, lhs <= rhs)
GRB_BINARYOP(first, GRB_OP_FIRST, , = T_in1, = T_in1, lhs)
GRB_BINARYOP(second, GRB_OP_SECOND, , = T_in1, = T_in1, rhs)
GRB_BINARYOP(minimum, GRB_OP_MINIMUM, , = T_in1, = T_in1, lhs < rhs ? lhs : rhs)
GRB_BINARYOP(maximum, GRB_OP_MAXIMUM, , = T_in1, = T_in1, lhs < rhs ? rhs : lhs)
GRB_BINARYOP(plus, GRB_OP_PLUS, , = T_in1, = T_in1, lhs + rhs)
GRB_BINARYOP(minus, GRB_OP_MINUS, , = T_in1, = T_in1, lhs - rhs)
GRB_BINARYOP(multiplies, GRB_OP_MULTIPLIES, , = T_in1, = T_in1, lhs * rhs)
GRB_BINARYOP(divides, GRB_OP_DIVIDES, , = T_in1, = T_in1, lhs / rhs)
GRB_BINARYOP(select_second, GRB_OP_SECOND, , = T_in1, = T_in1, rhs)
#undef GRB_C
#undef GRB_BINARYOP

namespace detail {
// the C-ABI id of a semiring functor type: the enum for the 17 of stddef.hpp, a registered id for the rest
template <typename SemiringT>
inline grb_semiring sr_id() {
  return static_cast<grb_semiring>(SemiringT::grb_id >= 0 ? SemiringT::grb_id : SemiringT::grb_registered_id());
}
}  // namespace detail
}  // namespace graphblas

// REGISTER_MONOID / REGISTER_SEMIRING (graphblas/stddef.hpp:140-191), same argument meaning: the binary operator
// and multiply are functor templates from the list above, the identity any expression in T_out.  The composition is
// registered with the library on first use (grb_semiring_register); it then works in vxm / mxv / eWiseAdd / eWiseMult
// / mxm like the built-in ones.  (`reduce` takes the nine monoids of stddef.hpp only.)
#define REGISTER_MONOID(M_NAME, BINARYOP, IDENTITY)                                                \
  template <typename T_out>                                                                        \
  struct M_NAME {                                                                                  \
    static const int grb_id = -1;                                                                  \
    static const int grb_op = BINARYOP<T_out>::grb_op;                                             \
    inline T_out identity() const { return static_cast<T_out>(IDENTITY); }                         \
    inline T_out operator()(T_out lhs, T_out rhs) const { return BINARYOP<T_out>()(lhs, rhs); }    \
  };

#define REGISTER_SEMIRING(SR_NAME, ADD_MONOID, MULT_BINARYOP)                                      \
  template <typename T_in1, typename T_in2 = T_in1, typename T_out = T_in1>                        \
  struct SR_NAME {                                                                                 \
    typedef T_out result_type;                                                                     \
    typedef T_out T_out_type;                                                                      \
    static const int grb_id = -1;                                                                  \
    static int grb_registered_id() {                                                               \
      static int id = -1;                                                                          \
      if (id < 0)                                                                                  \
        grb_semiring_register(ADD_MONOID<T_out>::grb_op, static_cast<double>(ADD_MONOID<T_out>().identity()), \
                              MULT_BINARYOP<T_in1, T_in2, T_out>::grb_op, &id);                    \
      return id;                                                                                   \
    }                                                                                              \
    inline T_out identity() const { return ADD_MONOID<T_out>().identity(); }                       \
    inline T_out add_op(T_out lhs, T_out rhs) const { return ADD_MONOID<T_out>()(lhs, rhs); }      \
    inline T_out mul_op(T_in1 lhs, T_in2 rhs) const { return MULT_BINARYOP<T_in1, T_in2, T_out>()(lhs, rhs); } \
  };

// ------------------------------------------------------------------------------------
// Host utilities the applications use (graphblas/util.hpp of the reference)
template <typename X>
inline void printArray(const char* str, const X* array, int length = 40, bool limit = true) {
  if (limit && length > 40) length = 40;
  std::cout << str << ":\n";
  for (int i = 0; i < length; i++) std::cout << "[" << i << "]:" << array[i] << " ";
  std::cout << "\n";
}
template <typename X>
inline void printArray(const char* str, const std::vector<X>& array, int length = 40, bool limit = true) {
  printArray(str, array.data(), std::min<int>(length, static_cast<int>(array.size())), limit);
}

struct CpuTimer {
  timeval start, stop;
  void Start() { gettimeofday(&start, NULL); }
  void Stop() { gettimeofday(&stop, NULL); }
  float ElapsedMillis() {
    float sec = stop.tv_sec - start.tv_sec;
    float usec = stop.tv_usec - start.tv_usec;
    return (sec * 1000) + (usec / 1000);
  }
};

template <typename X>
inline X getEnv(const char* key, X default_val) {
  const char* val = std::getenv(key);
  return val == NULL ? default_val : static_cast<X>(atoi(val));
}
template <typename X>
inline void setEnv(const char* key, X default_val) {
  setenv(key, std::to_string(default_val).c_str(), 0);
}

// Command line -> variables_map with the reference's 35 flags and defaults.
inline void parseArgs(int argc, char** argv, po::variables_map* vm) {
  static const char* kDefaults[][2] = {
      {"ta", "32"}, {"tb", "32"}, {"mode", "fixedrow"}, {"split", "0"}, {"niter", "10"},
      {"max_niter", "10000"}, {"directed", "0"}, {"timing", "1"}, {"transpose", "0"}, {"mtxinfo", "1"},
      {"verbose", "1"}, {"skip_cpu_verify", "0"}, {"source", "0"}, {"source_start", "0"},
      {"source_end", "1"}, {"mxvmode", "1"}, {"switchpoint", "0.01"}, {"dirinfo", "0"},
      {"struconly", "0"}, {"opreuse", "0"}, {"memusage", "1.0"}, {"endbit", "1"}, {"sort", "1"},
      {"atomic", "0"}, {"earlyexit", "1"}, {"fusedmask", "1"}, {"maxcolors", "10000"}, {"gcalgo", "0"},
      {"ccalgo", "0"}, {"seed", "-1"}, {"nthread", "128"}, {"ndevice", "0"}, {"debug", "0"},
      {"memory", "0"}, {"edgeswitch", "0"}};
  for (size_t i = 0; i < sizeof(kDefaults) / sizeof(kDefaults[0]); ++i) vm->set(kDefaults[i][0], kDefaults[i][1]);
  for (int i = 1; i + 1 < argc; ++i) {
    if (argv[i][0] == '-' && argv[i][1] == '-') {
      std::string text(argv[i + 1]);
      if (text == "true") text = "1";
      if (text == "false") text = "0";
      vm->set(argv[i] + 2, text);
      ++i;
    }
  }
}

// MatrixMarket coordinate reader with the reference loader's semantics (util.hpp:197-329, 363-430):
// 1-based -> 0-based, pattern -> 1, symmetric (or directed == 2) adds the reverse of every
// off-diagonal entry, sorted by (row, col), self loops and duplicates dropped -- and, as there,
// the compaction after a removal moves the INDICES only: the value array stays in sorted order
// of all entries and is cut to the new length (util.hpp:311-323; invisible on pattern graphs).
// dat_name (when asked for) is the binary cache's path (util.hpp:340-357); when that file exists
// the lists come back empty and Matrix::build(..., dat_name) reads the cache (util.hpp:398-409).
#ifndef GRB_MAXLEN
#define GRB_MAXLEN 256
#endif
namespace graphblas {
namespace detail {
// Cache names handed out by readMtx / convert.  Matrix::build only acts on (and frees) a dat_name it finds here:
// the reference's own test/gtrace.cu:32-36 passes an UNINITIALISED char* to build(), which the reference
// dereferences (it works there when the stack word happens to be zero).
inline std::vector<char*>& issued_cache_names() {
  static std::vector<char*> v;
  return v;
}
inline bool take_cache_name(char* p) {
  std::vector<char*>& v = issued_cache_names();
  for (size_t i = 0; i < v.size(); ++i)
    if (v[i] == p) { v.erase(v.begin() + i); return true; }
  return false;
}
}  // namespace detail
}  // namespace graphblas

// util.hpp:340-357: the binary cache's path for a .mtx path (malloc'ed, freed by Matrix::build)
inline char* convert(const char* fname, bool is_undirected = true) {
  char* dat_name = reinterpret_cast<char*>(malloc(GRB_MAXLEN));
  if (grb_cache_name(fname, is_undirected, dat_name, GRB_MAXLEN) != 0) dat_name[0] = 0;
  graphblas::detail::issued_cache_names().push_back(dat_name);
  return dat_name;
}

template <typename X>
inline int readMtx(const char* fname, std::vector<graphblas::Index>* row_indices,
                   std::vector<graphblas::Index>* col_indices, std::vector<X>* values, graphblas::Index* nrows,
                   graphblas::Index* ncols, graphblas::Index* nvals, int directed, bool mtxinfo,
                   char** dat_name = NULL) {
  FILE* f = fopen(fname, "r");
  if (!f) {
    printf("File %s not found\n", fname);
    exit(1);
  }
  char line[1024], banner[64], mtx[64], crd[64], dtype[64], sym[64];
  if (!fgets(line, sizeof(line), f) ||
      sscanf(line, "%63s %63s %63s %63s %63s", banner, mtx, crd, dtype, sym) != 5 ||
      strcmp(banner, "%%MatrixMarket") != 0) {
    printf("Could not process Matrix Market banner.\n");
    exit(1);
  }
  for (char* p = dtype; *p; ++p) *p = tolower(*p);
  for (char* p = sym; *p; ++p) *p = tolower(*p);
  do {
    if (!fgets(line, sizeof(line), f)) exit(1);
  } while (line[0] == '%');
  int nr, nc, nz;
  if (sscanf(line, "%d %d %d", &nr, &nc, &nz) != 3) exit(1);
  *nrows = nr; *ncols = nc; *nvals = nz;
  const bool pattern = strcmp(dtype, "pattern") == 0;
  const bool integer = strcmp(dtype, "integer") == 0;
  const bool symmetric = strcmp(sym, "symmetric") == 0;
  bool undirected = (symmetric || directed == 2) && directed != 1;
  row_indices->clear(); col_indices->clear(); values->clear();
  if (dat_name) {
    *dat_name = convert(fname, undirected);                              // freed by Matrix::build, as in the reference
    FILE* c = (*dat_name)[0] ? fopen(*dat_name, "rb") : NULL;
    if (c) {                                   // empty lists tell Matrix::build that the cache exists
      fclose(c);
      fclose(f);
      return 0;
    }
  }
  struct Entry { graphblas::Index r, c; X v; };
  std::vector<Entry> e;
  e.reserve(static_cast<size_t>(nz) * (undirected ? 2 : 1));
  for (int i = 0; i < nz; ++i) {
    int r, c;
    if (fscanf(f, "%d", &r) == EOF) { std::cout << "Error: Not enough rows in mtx file!\n"; break; }
    if (fscanf(f, "%d", &c) != 1) c = 0;
    X v = static_cast<X>(1);
    if (integer) { int raw = 0; if (fscanf(f, "%d", &raw) != 1) raw = 0; v = static_cast<X>(raw); }
    else if (!pattern) { float raw = 0.f; if (fscanf(f, "%f", &raw) != 1) raw = 0.f; v = static_cast<X>(raw); }
    Entry x = {r - 1, c - 1, v};
    e.push_back(x);
  }
  fclose(f);
  if (undirected) {                            // reverse entries are appended after the originals (util.hpp:271-279)
    const size_t m = e.size();
    for (size_t i = 0; i < m; ++i)
      if (e[i].r != e[i].c) { Entry y = {e[i].c, e[i].r, e[i].v}; e.push_back(y); }
  }
  // customSort is std::sort on (row, col): the order of equal coordinates with different values is
  // libstdc++'s; stable order is used here (identical whenever duplicates carry equal values)
  std::stable_sort(e.begin(), e.end(), [](const Entry& a, const Entry& b) {
    return a.r != b.r ? a.r < b.r : a.c < b.c;
  });
  const char* keep_sl = getenv("GRB_UTIL_REMOVE_SELFLOOP");
  const bool remove_self_loops = !(keep_sl && atoi(keep_sl) == 0);
  for (size_t i = 0; i < e.size(); ++i) {
    if (remove_self_loops && e[i].r == e[i].c) continue;
    if (i > 0 && e[i - 1].r == e[i].r && e[i - 1].c == e[i].c) continue;
    row_indices->push_back(e[i].r);
    col_indices->push_back(e[i].c);
  }
  *nvals = static_cast<graphblas::Index>(row_indices->size());
  values->resize(row_indices->size());
  for (size_t i = 0; i < values->size(); ++i) (*values)[i] = e[i].v;   // values are NOT compacted (util.hpp:311-326)
  if (mtxinfo) printf("%s: %d x %d, %d stored entries (undirected: %d)\n", fname, nr, nc, *nvals, undirected);
  return 0;
}

// ------------------------------------------------------------------------------------
namespace graphblas {
namespace backend {

struct GpuTimer {                  // backend/cuda/util.hpp:92-120, on HIP events in the library
  float ms_;
  GpuTimer() : ms_(0.f) {}
  void Start() { grb_timer_start(); }
  void Stop() { grb_timer_stop(&ms_); }
  float ElapsedMillis() { return ms_; }
};

class Descriptor {
 public:
  Descriptor() : h_(NULL), max_niter_(0), niter_(0), timing_(0), directed_(0), debug_(false), memory_(false),
                 lastmxv_(GrB_PUSHONLY) { grb_descriptor_new(&h_); }
  ~Descriptor() { grb_descriptor_free(h_); }
  Info set(Desc_field field, Desc_value value) { return to_info(grb_descriptor_set(h_, field, value)); }
  Info get(Desc_field field, Desc_value* value) const {
    int v = 0;
    Info i = to_info(grb_descriptor_get(h_, field, &v));
    *value = static_cast<Desc_value>(v);
    return i;
  }
  Info toggle(Desc_field field) { return to_info(grb_descriptor_toggle(h_, field)); }
  Info loadArgs(const po::variables_map& vm) {
    grb_info i = grb_descriptor_load_defaults(h_);
    static const char* kArgs[] = {"mxvmode", "switchpoint", "struconly", "opreuse", "earlyexit", "fusedmask",
                                  "sort", "endbit", "memusage", "atomic", "dirinfo", "nthread", "max_niter",
                                  "niter", "timing", "debug", "directed", "transpose", "edgeswitch"};
    for (size_t k = 0; i == GRB_SUCCESS && k < sizeof(kArgs) / sizeof(kArgs[0]); ++k)
      if (vm.count(kArgs[k])) i = grb_descriptor_set_arg(h_, kArgs[k], vm[kArgs[k]].as<double>());
    max_niter_ = vm["max_niter"].as<int>();
    niter_ = vm["niter"].as<int>();
    timing_ = vm["timing"].as<int>();
    directed_ = vm["directed"].as<int>();
    debug_ = vm["debug"].as<bool>();
    memory_ = vm["memory"].as<bool>();
    return to_info(i);
  }
  inline bool debug() { return debug_; }
  inline bool memory() { return memory_; }
  void sync() {                    // refresh the mirrored fields the applications read directly
    int v = GRB_PUSHONLY;
    grb_descriptor_lastmxv(h_, &v);
    lastmxv_ = static_cast<Desc_value>(v);
  }
  grb_descriptor h_;
  int max_niter_, niter_, timing_, directed_;
  bool debug_, memory_;
  Desc_value lastmxv_;

 private:
  Descriptor(const Descriptor&);
  void operator=(const Descriptor&);
};

template <typename X>
class Vector {
 public:
  typedef typename detail::dtype_of<X>::storage S;
  Vector() : h_(NULL), nsize_(0) {}
  explicit Vector(Index nsize) : h_(NULL), nsize_(nsize) { grb_vector_new(&h_, detail::dtype_of<X>::value, nsize); }
  ~Vector() { grb_vector_free(h_); }
  Info nnew(Index nsize) {
    grb_vector_free(h_);
    nsize_ = nsize;
    return to_info(grb_vector_new(&h_, detail::dtype_of<X>::value, nsize));
  }
  // representation switches the reference's tests reach through `#define private public`
  // (backend/cuda/vector.hpp:291-425; test/gvxm.cu:73)
  Info sparse2dense(X identity, Descriptor* desc = NULL) {
    return to_info(grb_vector_sparse2dense(h_, static_cast<double>(identity), desc ? desc->h_ : static_cast<grb_descriptor>(NULL)));
  }
  Info dense2sparse(X identity, Descriptor* desc) {
    return desc ? to_info(grb_vector_dense2sparse(h_, static_cast<double>(identity), desc->h_)) : GrB_UNINITIALIZED_OBJECT;
  }
  Info convert(X identity, float switchpoint, Descriptor* desc) {
    return desc ? to_info(grb_vector_convert(h_, static_cast<double>(identity), switchpoint, desc->h_)) : GrB_UNINITIALIZED_OBJECT;
  }
  grb_vector h_;
  Index nsize_;

 private:
  Vector(const Vector&);
  void operator=(const Vector&);
};

template <typename X>
struct SparseMatrix {              // host mirrors the CPU oracles read (sparse_matrix.hpp:120-132)
  const Index* h_csrRowPtr_;
  const Index* h_csrColInd_;
  const X* h_csrVal_;
  const Index* h_cscColPtr_;
  const Index* h_cscRowInd_;
  const X* h_cscVal_;
  SparseMatrix() : h_csrRowPtr_(NULL), h_csrColInd_(NULL), h_csrVal_(NULL), h_cscColPtr_(NULL),
                   h_cscRowInd_(NULL), h_cscVal_(NULL) {}
};

template <typename X>
class Matrix {
 public:
  Matrix() : h_(NULL), nrows_(0), ncols_(0), nvals_(0) {}
  Matrix(Index nrows, Index ncols) : h_(NULL), nrows_(nrows), ncols_(ncols), nvals_(0) {
    grb_matrix_new(&h_, detail::dtype_of<X>::value, nrows, ncols);
  }
  ~Matrix() { grb_matrix_free(h_); }
  Info refresh_host() {
    const void *v1 = NULL, *v2 = NULL;
    grb_info i = grb_matrix_host_csr(h_, &sparse_.h_csrRowPtr_, &sparse_.h_csrColInd_, &v1);
    if (i == GRB_SUCCESS) i = grb_matrix_host_csc(h_, &sparse_.h_cscColPtr_, &sparse_.h_cscRowInd_, &v2);
    sparse_.h_csrVal_ = static_cast<const X*>(v1);
    sparse_.h_cscVal_ = static_cast<const X*>(v2);
    return to_info(i);
  }
  grb_matrix h_;
  Index nrows_, ncols_, nvals_;
  SparseMatrix<X> sparse_;

 private:
  Matrix(const Matrix&);
  void operator=(const Matrix&);
};

}  // namespace backend

// ---- frontend containers --------------------------------------------------------------
class Descriptor {
 public:
  Descriptor() : descriptor_() {}
  Info set(Desc_field field, Desc_value value) { return descriptor_.set(field, value); }
  Info get(Desc_field field, Desc_value* value) const { return descriptor_.get(field, value); }
  Info toggle(Desc_field field) { return descriptor_.toggle(field); }
  Info loadArgs(const po::variables_map& vm) { return descriptor_.loadArgs(vm); }
  grb_descriptor handle() const { return descriptor_.h_; }
  void sync() { descriptor_.sync(); }

 private:
  backend::Descriptor descriptor_;
};

template <typename X>
class Vector {
 public:
  typedef typename detail::dtype_of<X>::storage S;
  Vector() : vector_() {}
  explicit Vector(Index nsize) : vector_(nsize) {}
  Info nnew(Index nsize) { return vector_.nnew(nsize); }
  Info dup(const Vector* rhs) { return rhs ? to_info(grb_vector_dup(vector_.h_, rhs->vector_.h_)) : GrB_NULL_POINTER; }
  void operator=(const Vector& rhs) { grb_vector_dup(vector_.h_, rhs.vector_.h_); }
  Info clear() { return to_info(grb_vector_clear(vector_.h_)); }
  Info size(Index* nsize) const { return to_info(grb_vector_size(vector_.h_, nsize)); }
  Info nvals(Index* nvals) const { return to_info(grb_vector_nvals(vector_.h_, nvals)); }
  template <typename BinaryOpT>
  Info build(const std::vector<Index>* indices, const std::vector<X>* values, Index nvals, BinaryOpT) {
    if (!indices || !values) return GrB_NULL_POINTER;
    std::vector<S> tmp(values->begin(), values->end());
    return to_info(grb_vector_build_sparse(vector_.h_, indices->data(), tmp.data(), nvals));
  }
  Info build(const std::vector<X>* values, Index nvals) {
    if (!values) return GrB_NULL_POINTER;
    std::vector<S> tmp(values->begin(), values->end());
    return to_info(grb_vector_build_dense(vector_.h_, tmp.data(), nvals));
  }
  Info build(Index* d_indices, X* d_values, Index nvals) {
    return to_info(grb_vector_adopt_sparse(vector_.h_, d_indices, d_values, nvals));
  }
  Info build(X* d_values, Index nvals) { return to_info(grb_vector_adopt_dense(vector_.h_, d_values, nvals)); }
  Info setElement(X val, Index index) { return to_info(grb_vector_set_element(vector_.h_, static_cast<double>(val), index)); }
  Info extractElement(X* val, Index index) {
    double d = 0;
    Info i = to_info(grb_vector_extract_element(vector_.h_, &d, index));
    *val = static_cast<X>(d);
    return i;
  }
  Info extractTuples(std::vector<Index>* indices, std::vector<X>* values, Index* n) {
    std::vector<Index> ti(*n > 0 ? *n : 1);
    std::vector<S> tv(*n > 0 ? *n : 1);
    Info i = to_info(grb_vector_extract_tuples_sparse(vector_.h_, ti.data(), tv.data(), n));
    if (i != GrB_SUCCESS) return i;
    indices->assign(ti.begin(), ti.begin() + *n);
    values->assign(tv.begin(), tv.begin() + *n);
    return i;
  }
  Info extractTuples(std::vector<X>* values, Index* n) {
    std::vector<S> tv(*n > 0 ? *n : 1);
    Info i = to_info(grb_vector_extract_tuples_dense(vector_.h_, tv.data(), n));
    if (i != GrB_SUCCESS) return i;
    values->assign(tv.begin(), tv.begin() + *n);
    return i;
  }
  Info resize(Index nvals) { return to_info(grb_vector_resize(vector_.h_, nvals)); }
  Info fill(X val) { return to_info(grb_vector_fill(vector_.h_, static_cast<double>(val))); }
  Info fillAscending(Index nvals) { return to_info(grb_vector_fill_ascending(vector_.h_, nvals)); }
  Info print(bool force_update = false) {
    (void)force_update;
    Index n = 0;
    grb_vector_size(vector_.h_, &n);
    std::vector<X> v;
    Info i = extractTuples(&v, &n);
    if (i == GrB_SUCCESS) printArray("val", v, n);
    return i;
  }
  Info countUnique(Index* count) { (void)count; return GrB_SUCCESS; }
  Info setStorage(Storage s) { return to_info(grb_vector_set_storage(vector_.h_, s)); }
  Info getStorage(Storage* s) const {
    int v = 0;
    Info i = to_info(grb_vector_get_storage(vector_.h_, &v));
    *s = static_cast<Storage>(v);
    return i;
  }
  Info swap(Vector* rhs) { return rhs ? to_info(grb_vector_swap(vector_.h_, rhs->vector_.h_)) : GrB_NULL_POINTER; }
  grb_vector handle() const { return vector_.h_; }

 private:
  backend::Vector<X> vector_;
};

template <typename X>
class Matrix {
 public:
  typedef typename detail::dtype_of<X>::storage S;
  Matrix() : matrix_() {}
  Matrix(Index nrows, Index ncols) : matrix_(nrows, ncols) {}
  Info nrows(Index* n) const { *n = matrix_.nrows_; return GrB_SUCCESS; }
  Info ncols(Index* n) const { *n = matrix_.ncols_; return GrB_SUCCESS; }
  Info nvals(Index* n) const { return to_info(grb_matrix_nvals(matrix_.h_, n)); }
  // build(row_indices, col_indices, values, nvals, dup, dat_name) (graphblas/matrix.hpp:125-144):
  // empty lists + dat_name = read the binary cache (sparse_matrix.hpp:355-407); lists + dat_name =
  // build, then write the cache if the file does not exist yet (sparse_matrix.hpp:328-348)
  template <typename V, typename BinaryOpT>
  Info build(const std::vector<Index>* rows, const std::vector<Index>* cols, const std::vector<V>* values,
             Index nvals, BinaryOpT, char* dat_name = NULL) {
    if (!rows || !cols || !values) return GrB_NULL_POINTER;
    if (dat_name != NULL && !detail::take_cache_name(dat_name)) dat_name = NULL;   // not a name readMtx / convert issued
    if (rows->empty() && cols->empty() && values->empty() && dat_name == NULL) return GrB_NO_VALUE;
    Info i;
    if (dat_name == NULL || !rows->empty()) {
      std::vector<S> tmp(values->begin(), values->end());
      i = to_info(grb_matrix_build(matrix_.h_, rows->data(), cols->data(), tmp.data(), nvals));
      if (i == GrB_SUCCESS && dat_name != NULL && dat_name[0]) {
        FILE* c = fopen(dat_name, "rb");
        if (c) {
          fclose(c);
        } else {
          printf("Writing %s\n", dat_name);
          if (grb_matrix_write_cache(matrix_.h_, dat_name) != 0)
            std::cout << "Error: Unable to open file for writing!\n";
        }
      }
    } else {
      printf("Reading %s\n", dat_name);
      i = to_info(grb_matrix_build_cache(matrix_.h_, dat_name));
      if (i == GrB_NO_VALUE) {                               // sparse_matrix.hpp:403-406: prints, reports success
        std::cout << "Error: Unable to read file!\n";
        i = GrB_SUCCESS;
      } else if (i == GrB_SUCCESS) {
        grb_matrix_nrows(matrix_.h_, &matrix_.nrows_);
        grb_matrix_ncols(matrix_.h_, &matrix_.ncols_);
      }
    }
    if (dat_name) free(dat_name);
    if (i != GrB_SUCCESS) return i;
    grb_matrix_nvals(matrix_.h_, &matrix_.nvals_);
    return matrix_.refresh_host();
  }
  Info print(bool force_update = false) {
    (void)force_update;
    printArray("csrRowPtr", matrix_.sparse_.h_csrRowPtr_, std::min(matrix_.nrows_ + 1, 40));
    printArray("csrColInd", matrix_.sparse_.h_csrColInd_, std::min(matrix_.nvals_, 40));
    return GrB_SUCCESS;
  }
  grb_matrix handle() const { return matrix_.h_; }
  Info refresh_all() {
    grb_matrix_nvals(matrix_.h_, &matrix_.nvals_);
    return matrix_.refresh_host();
  }
  void refresh_csr_only() {          // result of mxm: CSR only (csc_initialized_ = false in the reference)
    const void* v = NULL;
    grb_matrix_nvals(matrix_.h_, &matrix_.nvals_);
    grb_matrix_host_csr(matrix_.h_, &matrix_.sparse_.h_csrRowPtr_, &matrix_.sparse_.h_csrColInd_, &v);
    matrix_.sparse_.h_csrVal_ = static_cast<const X*>(v);
  }
  // Set-up time rewrite of the stored values on the host (CSR order); CSC is re-derived.
  template <typename F>
  Info transform_values(F f) {
    Info i = matrix_.refresh_host();
    if (i != GrB_SUCCESS) return i;
    Index nv = 0;
    grb_matrix_nvals(matrix_.h_, &nv);
    std::vector<S> out(nv > 0 ? nv : 1);
    const Index* ptr = matrix_.sparse_.h_csrRowPtr_;
    for (Index r = 0; r < matrix_.nrows_; ++r)
      for (Index p = ptr[r]; p < ptr[r + 1]; ++p)
        out[p] = static_cast<S>(f(r, matrix_.sparse_.h_csrColInd_[p], matrix_.sparse_.h_csrVal_[p]));
    i = to_info(grb_matrix_set_values(matrix_.h_, out.data()));
    if (i != GrB_SUCCESS) return i;
    return matrix_.refresh_host();
  }

 private:
  backend::Matrix<X> matrix_;
};

// ---- operations (graphblas/operations.hpp): same template parameter and argument orders
namespace detail {
template <typename X> inline grb_vector hv(const Vector<X>* v) { return v ? v->handle() : static_cast<grb_vector>(NULL); }
}  // namespace detail
#define GRB_H(v) detail::hv(v)

template <typename W, typename M, typename U, typename a, typename BinaryOpT, typename SemiringT>
Info vxm(Vector<W>* w, const Vector<M>* mask, BinaryOpT accum, SemiringT op, const Vector<U>* u, const Matrix<a>* A,
         Descriptor* desc) {
  if (w == NULL || u == NULL || A == NULL || desc == NULL) return GrB_UNINITIALIZED_OBJECT;
  Descriptor* d = desc;
  Info i = to_info(grb_vxm(GRB_H(w), GRB_H(mask), detail::accum_of(accum), detail::sr_id<SemiringT>(),
                           GRB_H(u), A->handle(), d->handle()));
  d->sync();
  return i;
}

template <typename W, typename M, typename a, typename U, typename BinaryOpT, typename SemiringT>
Info mxv(Vector<W>* w, const Vector<M>* mask, BinaryOpT accum, SemiringT op, const Matrix<a>* A, const Vector<U>* u,
         Descriptor* desc) {
  if (w == NULL || u == NULL || A == NULL || desc == NULL) return GrB_UNINITIALIZED_OBJECT;
  Descriptor* d = desc;
  Info i = to_info(grb_mxv(GRB_H(w), GRB_H(mask), detail::accum_of(accum), detail::sr_id<SemiringT>(),
                           A->handle(), GRB_H(u), d->handle()));
  d->sync();
  return i;
}

template <typename W, typename M, typename U, typename V, typename BinaryOpT, typename SemiringT>
Info eWiseMult(Vector<W>* w, const Vector<M>* mask, BinaryOpT accum, SemiringT op, const Vector<U>* u,
               const Vector<V>* v, Descriptor* desc) {
  if (w == NULL || u == NULL || v == NULL || desc == NULL) return GrB_UNINITIALIZED_OBJECT;
  return to_info(grb_eWiseMult(GRB_H(w), GRB_H(mask), detail::accum_of(accum),
                               detail::sr_id<SemiringT>(), GRB_H(u), GRB_H(v),
                               desc->handle()));
}

template <typename W, typename M, typename U, typename V, typename BinaryOpT, typename SemiringT>
Info eWiseAdd(Vector<W>* w, const Vector<M>* mask, BinaryOpT accum, SemiringT op, const Vector<U>* u,
              const Vector<V>* v, Descriptor* desc) {
  if (w == NULL || u == NULL || v == NULL || desc == NULL) return GrB_UNINITIALIZED_OBJECT;
  return to_info(grb_eWiseAdd(GRB_H(w), GRB_H(mask), detail::accum_of(accum),
                              detail::sr_id<SemiringT>(), GRB_H(u), GRB_H(v),
                              desc->handle()));
}

template <typename W, typename M, typename U, typename V, typename BinaryOpT, typename SemiringT>
Info eWiseAdd(Vector<W>* w, const Vector<M>* mask, BinaryOpT accum, SemiringT op, const Vector<U>* u, V val,
              Descriptor* desc) {
  if (w == NULL || u == NULL || desc == NULL) return GrB_UNINITIALIZED_OBJECT;
  return to_info(grb_eWiseAdd_scalar(GRB_H(w), GRB_H(mask), detail::accum_of(accum),
                                     detail::sr_id<SemiringT>(), GRB_H(u),
                                     static_cast<double>(val), desc->handle()));
}

template <typename X, typename U, typename BinaryOpT, typename MonoidT>
Info reduce(X* val, BinaryOpT accum, MonoidT op, const Vector<U>* u, Descriptor* desc) {
  if (val == NULL || u == NULL || desc == NULL) return GrB_UNINITIALIZED_OBJECT;
  double d = 0;
  Info i = to_info(grb_reduce_vector(&d, detail::accum_of(accum), static_cast<grb_monoid>(MonoidT::grb_id), GRB_H(u),
                                     desc->handle()));
  *val = static_cast<X>(d);
  return i;
}

template <typename W, typename M, typename a, typename BinaryOpT, typename MonoidT>
Info reduce(Vector<W>* w, const Vector<M>* mask, BinaryOpT accum, MonoidT op, const Matrix<a>* A, Descriptor* desc) {
  if (w == NULL || A == NULL || desc == NULL) return GrB_UNINITIALIZED_OBJECT;
  return to_info(grb_reduce_matrix_rows(GRB_H(w), GRB_H(mask), detail::accum_of(accum),
                                        static_cast<grb_monoid>(MonoidT::grb_id),
                                        A->handle(),
                                        desc->handle()));
}

template <typename W, typename M, typename X, typename I, typename BinaryOpT>
Info assign(Vector<W>* w, Vector<M>* mask, BinaryOpT accum, X val, const Vector<I>* indices, Index nindices,
            Descriptor* desc) {
  (void)nindices;
  if (w == NULL || desc == NULL) return GrB_UNINITIALIZED_OBJECT;
  if (indices != NULL) return GrB_NOT_IMPLEMENTED;      // only GrB_ALL, as in the reference
  return to_info(grb_assign(GRB_H(w), GRB_H(mask), detail::accum_of(accum), static_cast<double>(val),
                            desc->handle()));
}

template <typename W, typename M, typename U, typename I, typename BinaryOpT>
Info assignScatter(Vector<W>* w, const Vector<M>* mask, BinaryOpT accum, const Vector<U>* u, const Vector<I>* indices,
                   Descriptor* desc) {
  if (w == NULL || u == NULL || indices == NULL || desc == NULL) return GrB_UNINITIALIZED_OBJECT;
  return to_info(grb_assignScatter(GRB_H(w), GRB_H(mask), detail::accum_of(accum), GRB_H(u), GRB_H(indices),
                                   desc->handle()));
}

template <typename W, typename M, typename U, typename I, typename BinaryOpT>
Info extractGather(Vector<W>* w, const Vector<M>* mask, BinaryOpT accum, const Vector<U>* u, const Vector<I>* indices,
                   Descriptor* desc) {
  if (w == NULL || u == NULL || indices == NULL || desc == NULL) return GrB_UNINITIALIZED_OBJECT;
  return to_info(grb_extractGather(GRB_H(w), GRB_H(mask), detail::accum_of(accum), GRB_H(u), GRB_H(indices),
                                   desc->handle()));
}

// scatter (extension, operations.hpp:748-761): w[(Index)indices[k]] = val
template <typename W, typename M, typename I, typename X>
Info scatter(Vector<W>* w, const Vector<M>* mask, const Vector<I>* indices, X val, Descriptor* desc) {
  if (indices == NULL || w == NULL) return GrB_UNINITIALIZED_OBJECT;
  return to_info(grb_scatter(GRB_H(w), GRB_H(mask), GRB_H(indices), static_cast<double>(val),
                             desc ? desc->handle() : static_cast<grb_descriptor>(NULL)));
}

// graphColor (operations.hpp:816-826; cuSPARSE csrcolor in the reference): colours from 0
template <typename W, typename a>
Info graphColor(Vector<W>* w, const Matrix<a>* A, Descriptor* desc) {
  if (A == NULL || w == NULL || desc == NULL) return GrB_UNINITIALIZED_OBJECT;
  return to_info(grb_graph_color(GRB_H(w), A->handle(), desc->handle(), NULL));
}

// ---- unary operators the DEVICE knows (not in the reference: its stddef.hpp has none, and its only apply() callers
// pass the stateful host generators of algorithm/common.hpp).  apply() with one of these runs on the GPU whatever
// GrB_BACKEND says; any other functor type is a host functor and keeps the reference's host loop in index order.
#define GRB_UNARYOP(NAME, CODE, EXPR)                                                              \
  template <typename T_in = float, typename T_out = T_in>                                          \
  struct NAME {                                                                                    \
    static const int grb_unary = CODE;                                                             \
    static const int grb_op = 0;                                                                   \
    inline double grb_scalar() const { return 0.0; }                                               \
    inline T_out operator()(T_in x) const { return static_cast<T_out>(EXPR); }                     \
  };
GRB_UNARYOP(unary_identity, GRB_UNARY_IDENTITY, x)
GRB_UNARYOP(unary_minus, GRB_UNARY_AINV, -x)
GRB_UNARYOP(unary_reciprocal, GRB_UNARY_MINV, static_cast<T_in>(1) / x)
GRB_UNARYOP(unary_abs, GRB_UNARY_ABS, x < static_cast<T_in>(0) ? -x : x)
GRB_UNARYOP(unary_logical_not, GRB_UNARY_LNOT, !x)
#undef GRB_UNARYOP
// x -> op(scalar, x) and x -> op(x, scalar) for any of the binary-operator functors above
template <typename BinaryOpT, typename T = float>
struct bind_first {
  static const int grb_unary = GRB_UNARY_BIND_FIRST;
  static const int grb_op = BinaryOpT::grb_op;
  T scalar;
  explicit bind_first(T s) : scalar(s) {}
  inline double grb_scalar() const { return static_cast<double>(scalar); }
  inline T operator()(T x) const { return static_cast<T>(BinaryOpT()(scalar, x)); }
};
template <typename BinaryOpT, typename T = float>
struct bind_second {
  static const int grb_unary = GRB_UNARY_BIND_SECOND;
  static const int grb_op = BinaryOpT::grb_op;
  T scalar;
  explicit bind_second(T s) : scalar(s) {}
  inline double grb_scalar() const { return static_cast<double>(scalar); }
  inline T operator()(T x) const { return static_cast<T>(BinaryOpT()(x, scalar)); }
};
namespace detail {
template <typename F> struct is_device_unary {
  template <typename G> static char test(decltype(G::grb_unary)*);
  template <typename G> static long test(...);
  static const bool value = sizeof(test<F>(0)) == sizeof(char);
};
}  // namespace detail

// apply on a vector (operations.hpp:559-579 -> backend :878-910, apply.hpp:10-62): implemented in the
// reference only for a dense u, no mask, under GrB_BACKEND = GrB_SEQUENTIAL -- a host loop in
// index order between a device->host and a host->device copy (so a stateful functor such as
// set_random, algorithm/common.hpp:8-20, sees the elements in order).  Every other case prints
// its "not implemented" line there and changes nothing but w's storage flag.
namespace detail {
template <typename W, typename M, typename U, typename BinaryOpT, typename UnaryOpT>
inline typename std::enable_if<is_device_unary<UnaryOpT>::value, Info>::type
apply_on_device(Vector<W>* w, const Vector<M>* mask, BinaryOpT accum, const UnaryOpT& op, const Vector<U>* u, Descriptor* desc) {
  const grb_info i = grb_vector_apply(GRB_H(w), mask ? GRB_H(const_cast<Vector<M>*>(mask)) : static_cast<grb_vector>(NULL),
                                      accum_of(accum), UnaryOpT::grb_unary, UnaryOpT::grb_op, op.grb_scalar(),
                                      GRB_H(const_cast<Vector<U>*>(u)), desc->handle());
  if (i == GRB_NOT_IMPLEMENTED) {
    std::cout << "Error: DeVec apply masked not implemented yet!\n";
    return GrB_SUCCESS;
  }
  return to_info(i);
}
template <typename W, typename M, typename U, typename BinaryOpT, typename UnaryOpT>
inline typename std::enable_if<!is_device_unary<UnaryOpT>::value, Info>::type
apply_on_device(Vector<W>*, const Vector<M>*, BinaryOpT, const UnaryOpT&, const Vector<U>*, Descriptor*) {
  return GrB_NOT_IMPLEMENTED;
}
template <typename c, typename UnaryOpT>
inline typename std::enable_if<is_device_unary<UnaryOpT>::value, Info>::type
apply_matrix_on_device(Matrix<c>* C, const UnaryOpT& op, Descriptor* desc) {
  const Info i = to_info(grb_matrix_apply(C->handle(), static_cast<grb_matrix>(NULL), GRB_ACCUM_NULL, UnaryOpT::grb_unary,
                                          UnaryOpT::grb_op, op.grb_scalar(), C->handle(), desc->handle()));
  if (i != GrB_SUCCESS) return i;
  return C->refresh_all();
}
template <typename c, typename UnaryOpT>
inline typename std::enable_if<!is_device_unary<UnaryOpT>::value, Info>::type
apply_matrix_on_device(Matrix<c>*, const UnaryOpT&, Descriptor*) {
  return GrB_NOT_IMPLEMENTED;
}
}  // namespace detail

template <typename W, typename M, typename U, typename BinaryOpT, typename UnaryOpT>
Info apply(Vector<W>* w, const Vector<M>* mask, BinaryOpT accum, UnaryOpT op, const Vector<U>* u, Descriptor* desc) {
  (void)accum;
  if (w == NULL || u == NULL) return GrB_UNINITIALIZED_OBJECT;
  if (desc == NULL) return GrB_UNINITIALIZED_OBJECT;
  Index un = 0, wn = 0, mn = 0;
  u->size(&un);
  w->size(&wn);
  if (un != wn) return GrB_DIMENSION_MISMATCH;
  if (mask != NULL) {
    mask->size(&mn);
    if (mn != wn) return GrB_DIMENSION_MISMATCH;
  }
  if (detail::is_device_unary<UnaryOpT>::value)           // a unary operator the device knows: one kernel, any backend flag
    return detail::apply_on_device(w, mask, accum, op, u, desc);
  Storage s;
  u->getStorage(&s);
  if (s == GrB_SPARSE) {
    w->setStorage(GrB_SPARSE);
    std::cout << "SpVec Apply\nError: Feature not implemented yet!\n";
    return GrB_SUCCESS;
  }
  if (s != GrB_DENSE) return GrB_UNINITIALIZED_OBJECT;
  w->setStorage(GrB_DENSE);
  Desc_value backend;
  desc->get(GrB_BACKEND, &backend);
  if (backend != GrB_SEQUENTIAL) {
    std::cout << "DeVec apply GPU\nError: Feature not implemented yet!\n";
    return GrB_SUCCESS;
  }
  if (mask != NULL) {
    std::cout << "Error: DeVec apply masked not implemented yet!\n";
    return GrB_SUCCESS;
  }
  std::vector<U> uv;
  Info i = const_cast<Vector<U>*>(u)->extractTuples(&uv, &un);
  if (i != GrB_SUCCESS) return i;
  std::vector<W> wv(un);
  for (Index k = 0; k < un; ++k) wv[k] = op(uv[k]);
  return w->build(&wv, un);
}

// mxm: masked SpGEMM only (operations.hpp:22-48; unmasked is a cuSPARSE call in the reference)
template <typename c, typename m, typename a, typename b, typename BinaryOpT, typename SemiringT>
Info mxm(Matrix<c>* C, const Matrix<m>* mask, BinaryOpT accum, SemiringT op, const Matrix<a>* A, const Matrix<b>* B,
         Descriptor* desc) {
  if (C == NULL || A == NULL || B == NULL || desc == NULL) return GrB_UNINITIALIZED_OBJECT;
  Info i = to_info(grb_mxm(C->handle(), mask ? mask->handle() : static_cast<grb_matrix>(NULL), detail::accum_of(accum),
                           detail::sr_id<SemiringT>(), A->handle(), B->handle(), desc->handle()));
  if (i == GrB_SUCCESS) C->refresh_csr_only();
  return i;
}

template <typename X, typename a, typename BinaryOpT, typename MonoidT>
Info reduce(X* val, BinaryOpT accum, MonoidT op, const Matrix<a>* A, Descriptor* desc) {
  if (val == NULL || A == NULL || desc == NULL) return GrB_UNINITIALIZED_OBJECT;
  double d = 0;
  Info i = to_info(grb_reduce_matrix_scalar(&d, detail::accum_of(accum), static_cast<grb_monoid>(MonoidT::grb_id),
                                            A->handle(), desc->handle()));
  *val = static_cast<X>(d);
  return i;
}

// traceMxmTranspose (extension, operations.hpp:698-711): *val = trace(A (+).(x) B^T)
template <typename X, typename a, typename b, typename SemiringT>
Info traceMxmTranspose(X* val, SemiringT op, const Matrix<a>* A, const Matrix<b>* B, Descriptor* desc) {
  (void)op;
  if (val == NULL || A == NULL || B == NULL) return GrB_UNINITIALIZED_OBJECT;
  double d = 0;
  Info i = to_info(grb_trace_mxm_transpose(&d, detail::sr_id<SemiringT>(), A->handle(), B->handle(),
                                           desc ? desc->handle() : static_cast<grb_descriptor>(NULL)));
  *val = static_cast<X>(d);
  return i;
}

template <typename c, typename a>
Info tril(Matrix<c>* C, Matrix<a>* A, Descriptor* desc) {
  if (C == NULL || A == NULL || desc == NULL) return GrB_UNINITIALIZED_OBJECT;
  Info i = to_info(grb_matrix_tril(C->handle(), A->handle(), desc->handle()));
  if (i == GrB_SUCCESS) i = C->refresh_all();
  return i;
}

// ---- set-up time matrix operations (host side, as apply() is in the reference:
// backend/cuda/apply.hpp:102-111 runs only with GrB_BACKEND = GrB_SEQUENTIAL) -----------
// apply: C = op(A) on the stored values in CSR order (example/gsssp.cu:79-86)
template <typename c, typename a, typename m, typename BinaryOpT, typename UnaryOpT>
Info apply(Matrix<c>* C, const Matrix<m>* mask, BinaryOpT accum, UnaryOpT op, const Matrix<a>* A, Descriptor* desc) {
  if (C == NULL || A == NULL || desc == NULL) return GrB_UNINITIALIZED_OBJECT;
  if (mask != NULL || static_cast<const void*>(C) != static_cast<const void*>(A)) return GrB_NOT_IMPLEMENTED;
  if (detail::is_device_unary<UnaryOpT>::value) {       // a unary operator the device knows: in place on the device
    const Info di = detail::apply_matrix_on_device(C, op, desc);
    if (di != GrB_NOT_IMPLEMENTED && di != GrB_INVALID_OBJECT) return di;   // adopted storage: the host path below
  }
  return C->transform_values([&](Index, Index, c v) { return op(v); });
}

// eWiseMult, matrix x broadcast scalar (operations.hpp:206-228; example/gpr.cu:82-84)
template <typename c, typename m, typename a, typename b, typename BinaryOpT, typename SemiringT>
Info eWiseMult(Matrix<c>* C, const Matrix<m>* mask, BinaryOpT accum, SemiringT op, const Matrix<a>* A, b val,
               Descriptor* desc) {
  if (C == NULL || A == NULL || desc == NULL) return GrB_UNINITIALIZED_OBJECT;
  if (mask != NULL || static_cast<const void*>(C) != static_cast<const void*>(A)) return GrB_NOT_IMPLEMENTED;
  (void)accum;
  const Info i = to_info(grb_matrix_eWiseMult_scalar(C->handle(), detail::sr_id<SemiringT>(), C->handle(),
                                                     static_cast<double>(val)));
  if (i != GrB_SUCCESS) return i;
  return C->refresh_all();
}

// eWiseMult, matrix x broadcast column vector: C(i,j) = A(i,j) (x) B(i); with GrB_INP1 =
// GrB_TRAN the vector is broadcast along rows instead: C(i,j) = A(i,j) (x) B(j)
// (operations.hpp:240-267, backend ewisemult.hpp:470-622; example/gpr.cu:86-88)
template <typename c, typename m, typename a, typename b, typename BinaryOpT, typename SemiringT>
Info eWiseMult(Matrix<c>* C, const Matrix<m>* mask, BinaryOpT accum, SemiringT op, const Matrix<a>* A,
               const Vector<b>* B, Descriptor* desc) {
  if (C == NULL || A == NULL || B == NULL || desc == NULL) return GrB_UNINITIALIZED_OBJECT;
  if (mask != NULL || static_cast<const void*>(C) != static_cast<const void*>(A)) return GrB_NOT_IMPLEMENTED;
  Desc_value inp0, inp1;
  desc->get(GrB_INP0, &inp0);
  desc->get(GrB_INP1, &inp1);
  if (inp0 != GrB_DEFAULT) return GrB_INVALID_VALUE;
  (void)accum;
  {                                  // device path: dense B (the case of example/gpr.cu); else host below
    const grb_info di = grb_matrix_eWiseMult_vector(C->handle(), detail::sr_id<SemiringT>(), C->handle(),
                                                    const_cast<Vector<b>*>(B)->handle(), desc->handle());
    if (di == GRB_SUCCESS) return C->refresh_all();
    if (di != GRB_NOT_IMPLEMENTED && di != GRB_INVALID_OBJECT) return to_info(di);
  }
  Index n = 0;
  B->size(&n);
  std::vector<b> bv;
  Info i = const_cast<Vector<b>*>(B)->extractTuples(&bv, &n);
  if (i != GrB_SUCCESS) return i;
  const bool by_col = (inp1 == GrB_TRAN);
  return C->transform_values([&](Index r, Index col, c v) { return op.mul_op(v, static_cast<c>(bv[by_col ? col : r])); });
}

#undef GRB_H
}  // namespace graphblas

using namespace graphblas;   // the reference's util.hpp ends with the same directive (util.hpp:499)

#endif  // GRAPHBLAST_AMD_GRAPHBLAS_HPP_
