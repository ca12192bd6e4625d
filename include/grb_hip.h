/*
 * grb_hip.h -- C ABI of libgrb_hip.so, the MI355X (gfx950) backend for the
 * GraphBLAST mxv/vxm hot path.
 *
 * The reference (gunrock/graphblast) has no FFI: its boundary is the C++ template
 * frontend `graphblas::{Vector,Matrix,Descriptor}` + the free functions of
 * graphblas/operations.hpp.  This header is that frontend flattened to C: one
 * opaque handle per frontend class, one entry point per frontend method/operation
 * on the path, enums in place of the functor template arguments.  Every entry point
 * cites the reference interface it replaces.  All functions return a grb_info
 * whose values are `graphblas::Info` (graphblas/types.hpp:28-42); nothing calls
 * exit().  Plain pointers and sizes only; no C++ or torch types.
 *
 * One process drives one GPU (hipSetDevice is the caller's business); all work is
 * enqueued on the stream set with grb_set_stream (default: the null stream).
 * Host-visible results (nvals, reduce, extractTuples) synchronise that stream, as
 * the reference's API does.
 */
#ifndef GRB_HIP_H_
#define GRB_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int32_t grb_index;                 /* graphblas::Index, types.hpp:18 */
typedef int grb_info;                      /* graphblas::Info,  types.hpp:28-42 */

enum {
  GRB_SUCCESS = 0, GRB_UNINITIALIZED_OBJECT, GRB_NULL_POINTER, GRB_INVALID_VALUE,
  GRB_INVALID_INDEX, GRB_DOMAIN_MISMATCH, GRB_DIMENSION_MISMATCH, GRB_OUTPUT_NOT_EMPTY,
  GRB_NO_VALUE, GRB_NOT_IMPLEMENTED, GRB_OUT_OF_MEMORY, GRB_INSUFFICIENT_SPACE,
  GRB_INVALID_OBJECT, GRB_INDEX_OUT_OF_BOUNDS, GRB_PANIC
};

/* graphblas::Storage, types.hpp:21-23 */
enum { GRB_UNKNOWN = 0, GRB_SPARSE = 1, GRB_DENSE = 2 };

/* graphblas::Desc_field, types.hpp:44-55 */
enum { GRB_MASK = 0, GRB_OUTP, GRB_INP0, GRB_INP1, GRB_MODE, GRB_TA, GRB_TB, GRB_NT,
       GRB_MXVMODE, GRB_TOL, GRB_BACKEND, GRB_NDESCFIELD };

/* graphblas::Desc_value, types.hpp:57-78 (GRB_HIP takes the slot of GrB_CUDA) */
enum { GRB_SCMP = 0, GRB_REPLACE = 1, GRB_TRAN = 2, GRB_DEFAULT = 3, GRB_FIXEDROW = 6,
       GRB_PUSHPULL = 10, GRB_PUSHONLY = 11, GRB_PULLONLY = 12, GRB_SEQUENTIAL = 13,
       GRB_HIP = 14 };

/* Element type of a Vector<T> / Matrix<T>.  The reference instantiates float
 * (BFS/SSSP/PR) and int (CC/TC). */
typedef enum { GRB_F32 = 0, GRB_I32 = 1 } grb_dtype;

/* Additive monoids, graphblas/stddef.hpp:159-172 (same order). */
typedef enum {
  GRB_PLUS_MONOID = 0, GRB_MULTIPLIES_MONOID, GRB_MINIMUM_MONOID, GRB_MAXIMUM_MONOID,
  GRB_LOGICAL_OR_MONOID, GRB_LOGICAL_AND_MONOID, GRB_GREATER_MONOID,
  GRB_CUSTOM_LESS_MONOID, GRB_NOT_EQUAL_TO_MONOID, GRB_N_MONOIDS
} grb_monoid;

/* Semirings, graphblas/stddef.hpp:195-213 (same order). */
typedef enum {
  GRB_LOGICAL_OR_AND = 0, GRB_PLUS_MULTIPLIES, GRB_MINIMUM_PLUS, GRB_MAXIMUM_MULTIPLIES,
  GRB_PLUS_DIVIDES, GRB_PLUS_GREATER, GRB_GREATER_PLUS, GRB_PLUS_MINUS, GRB_PLUS_LESS,
  GRB_CUSTOM_LESS_PLUS, GRB_MINIMUM_MULTIPLIES, GRB_MULTIPLIES_MULTIPLIES,
  GRB_NOT_EQUAL_TO_PLUS, GRB_MINIMUM_SELECT_SECOND, GRB_PLUS_NOT_EQUAL_TO,
  GRB_CUSTOM_LESS_LESS, GRB_MINIMUM_NOT_EQUAL_TO, GRB_N_SEMIRINGS
} grb_semiring;

/* Binary operators, in the order graphblas/stddef.hpp:14-138 defines its functor templates
 * (logical_or ... divides; select_second == second). */
typedef enum {
  GRB_OP_LOGICAL_OR = 0, GRB_OP_LOGICAL_AND, GRB_OP_LOGICAL_XOR, GRB_OP_EQUAL, GRB_OP_NOT_EQUAL_TO,
  GRB_OP_GREATER, GRB_OP_LESS, GRB_OP_GREATER_EQUAL, GRB_OP_LESS_EQUAL, GRB_OP_FIRST, GRB_OP_SECOND,
  GRB_OP_MINIMUM, GRB_OP_MAXIMUM, GRB_OP_PLUS, GRB_OP_MINUS, GRB_OP_MULTIPLIES, GRB_OP_DIVIDES, GRB_N_BINARY_OPS
} grb_binary_op;

/* Unary operators of the device-side apply (the reference declares apply and implements it as a HOST loop under
 * GrB_BACKEND = GrB_SEQUENTIAL only, backend/cuda/apply.hpp:34-42,102-111; its own callers pass stateful random
 * functors that only a host loop in index order can serve).  BIND_FIRST / BIND_SECOND make a unary operator of any
 * grb_binary_op and a scalar: x -> op(scalar, x) / x -> op(x, scalar). */
typedef enum {
  GRB_UNARY_IDENTITY = 0, GRB_UNARY_AINV /* -x */, GRB_UNARY_MINV /* 1 / x */, GRB_UNARY_ABS, GRB_UNARY_LNOT /* !x */,
  GRB_UNARY_BIND_FIRST, GRB_UNARY_BIND_SECOND, GRB_N_UNARY_OPS
} grb_unary_op;

/* `accum` argument: the reference only tests its presence
 * (typeid(accum).name().size() > 1, backend/cuda/spmv.hpp:34-40). */
typedef enum { GRB_ACCUM_NULL = 0, GRB_ACCUM_PRESENT = 1 } grb_accum;

/* Semirings an application registers itself -- REGISTER_SEMIRING(SR, ADD_MONOID, MULT_BINARYOP) over
 * REGISTER_MONOID(M, BINARYOP, IDENTITY), graphblas/stddef.hpp:140-191: *id (>= 64) is accepted wherever a
 * grb_semiring is.  A composition that is one of the 17 above runs their compiled kernels; any other runs the
 * same kernels with the operators selected at run time. */
grb_info grb_semiring_register(int add_op /* grb_binary_op */, double add_identity, int mul_op /* grb_binary_op */,
                               int* id);

typedef struct grb_vector_s*     grb_vector;      /* graphblas::Vector<T>   vector.hpp:12-66 */
typedef struct grb_matrix_s*     grb_matrix;      /* graphblas::Matrix<T>   matrix.hpp:13-84 */
typedef struct grb_descriptor_s* grb_descriptor;  /* graphblas::Descriptor  descriptor.hpp:17-39 */

/* ---- library ----------------------------------------------------------------- */
/* Stream for all subsequent work (a hipStream_t cast to void*; NULL = null stream). */
grb_info grb_set_stream(void* hip_stream);
/* "gfx950:<CU count>:<device name>" of the current device, for logs. */
grb_info grb_device_info(char* buf, size_t buflen);
const char* grb_version(void);
/* HIP-event stopwatch on the library stream (backend::GpuTimer, backend/cuda/util.hpp:92-120). */
grb_info grb_timer_start(void);
grb_info grb_timer_stop(float* elapsed_ms);

/* ---- Descriptor (graphblas/descriptor.hpp:17-39, backend/cuda/descriptor.hpp) -- */
grb_info grb_descriptor_new(grb_descriptor* desc);
grb_info grb_descriptor_free(grb_descriptor desc);
grb_info grb_descriptor_set(grb_descriptor desc, int field, int value);      /* Descriptor::set    */
grb_info grb_descriptor_get(grb_descriptor desc, int field, int* value);     /* Descriptor::get    */
grb_info grb_descriptor_toggle(grb_descriptor desc, int field);              /* Descriptor::toggle */
/* Descriptor::loadArgs (backend/cuda/descriptor.hpp:207-287): grb_descriptor_load_defaults
 * applies the parseArgs defaults of graphblas/util.hpp:39-132, grb_descriptor_set_arg one
 * named CLI flag: mxvmode switchpoint struconly opreuse earlyexit fusedmask sort endbit
 * memusage atomic dirinfo nthread max_niter niter timing debug directed transpose. */
grb_info grb_descriptor_load_defaults(grb_descriptor desc);
grb_info grb_descriptor_set_arg(grb_descriptor desc, const char* name, double value);
grb_info grb_descriptor_get_arg(grb_descriptor desc, const char* name, double* value);
/* desc->descriptor_.lastmxv_ : GRB_PUSHONLY or GRB_PULLONLY of the last mxv/vxm. */
grb_info grb_descriptor_lastmxv(grb_descriptor desc, int* value);

/* ---- Vector (graphblas/vector.hpp:12-66) ------------------------------------- */
grb_info grb_vector_new(grb_vector* v, grb_dtype dtype, grb_index nsize);    /* Vector(Index)      */
grb_info grb_vector_free(grb_vector v);
grb_info grb_vector_dup(grb_vector dst, grb_vector src);                     /* dup / operator=    */
grb_info grb_vector_clear(grb_vector v);
grb_info grb_vector_size(grb_vector v, grb_index* nsize);
/* Vector::resize (vector.hpp:230-237): keeps the first min(nsize, nvals) stored entries. */
grb_info grb_vector_resize(grb_vector v, grb_index nsize);
grb_info grb_vector_nvals(grb_vector v, grb_index* nvals);
/* build(indices, values, nvals, dup) -> sparse; host arrays; values are dtype-typed. */
grb_info grb_vector_build_sparse(grb_vector v, const grb_index* indices, const void* values,
                                 grb_index nvals);
/* build(values, nvals) -> dense; host array. */
grb_info grb_vector_build_dense(grb_vector v, const void* values, grb_index nvals);
/* build(T* values, nvals) / build(Index*, T*, nvals): adopt caller DEVICE pointers
 * without ownership (dense_vector.hpp:215-224, sparse_vector.hpp:164-175). */
grb_info grb_vector_adopt_dense(grb_vector v, void* d_values, grb_index nvals);
grb_info grb_vector_adopt_sparse(grb_vector v, grb_index* d_indices, void* d_values,
                                 grb_index nvals);
grb_info grb_vector_set_element(grb_vector v, double val, grb_index index);
grb_info grb_vector_extract_element(grb_vector v, double* val, grb_index index);
/* extractTuples(indices, values, n): sparse vector only; *n must equal nvals. */
grb_info grb_vector_extract_tuples_sparse(grb_vector v, grb_index* indices, void* values,
                                          grb_index* n);
/* extractTuples(values, n): densifies a sparse vector with fill 0 (vector.hpp:208-217);
 * *n must equal size. */
grb_info grb_vector_extract_tuples_dense(grb_vector v, void* values, grb_index* n);
grb_info grb_vector_fill(grb_vector v, double val);
grb_info grb_vector_fill_ascending(grb_vector v, grb_index nvals);
grb_info grb_vector_get_storage(grb_vector v, int* storage);
grb_info grb_vector_set_storage(grb_vector v, int storage);
grb_info grb_vector_swap(grb_vector a, grb_vector b);
/* backend::Vector::{convert,sparse2dense,dense2sparse} (backend/cuda/vector.hpp:291-425). */
grb_info grb_vector_convert(grb_vector v, double identity, float switchpoint, grb_descriptor desc);
grb_info grb_vector_sparse2dense(grb_vector v, double identity, grb_descriptor desc /*nullable*/);
grb_info grb_vector_dense2sparse(grb_vector v, double identity, grb_descriptor desc);
/* Raw device views (sparse_.d_ind_, sparse_.d_val_, dense_.d_val_) for zero-copy
 * interop (RCCL buffers, torch tensors); valid until the vector is freed or resized. */
grb_info grb_vector_device_ptrs(grb_vector v, grb_index** d_sparse_ind, void** d_sparse_val,
                                void** d_dense_val);

/* ---- Matrix (graphblas/matrix.hpp:13-84) ------------------------------------- */
grb_info grb_matrix_new(grb_matrix* A, grb_dtype dtype, grb_index nrows, grb_index ncols);
grb_info grb_matrix_free(grb_matrix A);
/* build(row_indices, col_indices, values, nvals, dup): host COO, sorted internally
 * (coo2csr + coo2csc, backend/cuda/sparse_matrix.hpp:289-351). */
grb_info grb_matrix_build(grb_matrix A, const grb_index* row_indices, const grb_index* col_indices,
                          const void* values, grb_index nvals);
/* Ingest on the device: the loader's semantics of readMtx (util.hpp:197-329,
 * 363-430) applied to a DEVICE coordinate list, then the same sort + compress as build().
 * flags: 1 add the reverse of every off-diagonal entry (symmetric / --directed 2), 2 drop self
 * loops, 4 drop duplicates (first occurrence wins).  d_values NULL = pattern (every value 1). */
grb_info grb_matrix_ingest_device(grb_matrix A, const grb_index* d_rows, const grb_index* d_cols,
                                  const void* d_values, grb_index nvals, int flags);
/* Host CSR in, CSC derived (csr2csc); `csc_*` may be given to skip the transpose. */
grb_info grb_matrix_build_csr(grb_matrix A, const grb_index* csr_row_ptr, const grb_index* csr_col_ind,
                              const void* csr_val, grb_index nvals, const grb_index* csc_col_ptr,
                              const grb_index* csc_row_ind, const void* csc_val);
/* build(row_ptr, col_ind, values, nvals) adopting DEVICE CSR (+ optional CSC) pointers
 * without ownership (sparse_matrix.hpp:417-435). */
grb_info grb_matrix_adopt_device_csr(grb_matrix A, grb_index* d_csr_row_ptr, grb_index* d_csr_col_ind,
                                     void* d_csr_val, grb_index nvals, grb_index* d_csc_col_ptr,
                                     grb_index* d_csc_row_ind, void* d_csc_val);
grb_info grb_matrix_nrows(grb_matrix A, grb_index* nrows);
grb_info grb_matrix_ncols(grb_matrix A, grb_index* ncols);
grb_info grb_matrix_nvals(grb_matrix A, grb_index* nvals);
/* Host mirrors h_csrRowPtr_/h_csrColInd_/h_csrVal_ and h_cscColPtr_/h_cscRowInd_/h_cscVal_
 * (sparse_matrix.hpp:120-132) that the CPU oracles read (algorithm/bfs.hpp:100-107). */
grb_info grb_matrix_host_csr(grb_matrix A, const grb_index** row_ptr, const grb_index** col_ind,
                             const void** val);
grb_info grb_matrix_host_csc(grb_matrix A, const grb_index** col_ptr, const grb_index** row_ind,
                             const void** val);
/* Overwrite the stored values (CSR order; CSC is re-derived): what gsssp.cu:79-86 does
 * through apply() + syncCpu, and gpr.cu:82-90 through the matrix eWiseMult variants. */
grb_info grb_matrix_set_values(grb_matrix A, const void* csr_val);

/* Binary CSR cache, the reference's interchange format (Appendix B of SURVEY.md):
 * `<dir>/.<file>.<ud|d>.<nosl|sl>.bin` = int32 nrows, int32 nvals, int32 rowptr[nrows+1],
 * int32 colind[nvals]; square matrices, values implied 1.
 *   grb_cache_name          util.hpp:340-357 (convert): the cache path for a .mtx path
 *   grb_matrix_write_cache  backend/cuda/sparse_matrix.hpp:328-348 (end of build(..., dat_name))
 *   grb_matrix_build_cache  backend/cuda/sparse_matrix.hpp:355-407 (build(char* dat_name)):
 *                           GrB_NO_VALUE when the file cannot be opened */
grb_info grb_cache_name(const char* mtx_path, int is_undirected, char* out, size_t cap);
grb_info grb_matrix_write_cache(grb_matrix A, const char* path);
grb_info grb_matrix_build_cache(grb_matrix A, const char* path);

/* ---- Operations (graphblas/operations.hpp) ----------------------------------- */
/* vxm  operations.hpp:59-87   -> backend/cuda/operations.hpp:80-209  */
grb_info grb_vxm(grb_vector w, grb_vector mask, grb_accum accum, grb_semiring op, grb_vector u,
                 grb_matrix A, grb_descriptor desc);
/* mxv  operations.hpp:97-127  -> backend/cuda/operations.hpp:215-327 */
grb_info grb_mxv(grb_vector w, grb_vector mask, grb_accum accum, grb_semiring op, grb_matrix A,
                 grb_vector u, grb_descriptor desc);
/* eWiseMult (vector x vector)  operations.hpp:137-158 -> backend :331-410 */
grb_info grb_eWiseMult(grb_vector w, grb_vector mask, grb_accum accum, grb_semiring op, grb_vector u,
                       grb_vector v, grb_descriptor desc);
/* eWiseAdd (vector + vector)   operations.hpp:277-298 -> backend :567-627 */
grb_info grb_eWiseAdd(grb_vector w, grb_vector mask, grb_accum accum, grb_semiring op, grb_vector u,
                      grb_vector v, grb_descriptor desc);
/* eWiseAdd (vector + scalar)   operations.hpp:333-352 -> backend :649-699 */
grb_info grb_eWiseAdd_scalar(grb_vector w, grb_vector mask, grb_accum accum, grb_semiring op,
                             grb_vector u, double val, grb_descriptor desc);
/* reduce (vector -> scalar)    operations.hpp:642-660 -> backend :1004-1030; result on host. */
grb_info grb_reduce_vector(double* val, grb_accum accum, grb_monoid op, grb_vector u,
                           grb_descriptor desc);
/* reduce (matrix -> vector, row-wise)  operations.hpp:620-640 -> backend :953-986 */
grb_info grb_reduce_matrix_rows(grb_vector w, grb_vector mask, grb_accum accum, grb_monoid op,
                                grb_matrix A, grb_descriptor desc);
/* assign (constant under mask, GrB_ALL)  operations.hpp:509-530 -> backend :822-860 */
grb_info grb_assign(grb_vector w, grb_vector mask, grb_accum accum, double val, grb_descriptor desc);

/* assignScatter  w[indices[k]] = u[k]   operations.hpp:771-790 -> backend :1170-1210 (scatter.hpp:85-123) */
grb_info grb_assignScatter(grb_vector w, grb_vector mask, grb_accum accum, grb_vector u, grb_vector indices,
                           grb_descriptor desc);
/* extractGather  w[k] = u[indices[k]]   operations.hpp:800-815 -> backend :1212-1253 (gather.hpp:11-50) */
grb_info grb_extractGather(grb_vector w, grb_vector mask, grb_accum accum, grb_vector u, grb_vector indices,
                           grb_descriptor desc);

/* The queue of element-wise calls (csrc/lazy.hip).  eWiseAdd / eWiseMult / dup on dense, library-owned
 * vectors without a mask are not run when they are called: up to six of them wait and run as ONE kernel -- every vector
 * read once, every result written once -- as soon as ANY other entry point of this header is called (each one starts by
 * flushing), so nothing can observe the difference except the launch count.  Vectors over adopted (caller-owned)
 * storage never take part.  grb_set_lazy(0) / GRB_LAZY=0: every call runs when it is called; on < 0 only queries;
 * returns the previous setting.  grb_lazy_pending(): steps waiting (does not flush). */
int grb_set_lazy(int on);
int grb_lazy_pending(void);
/* reductions that ran inside a chain's launch so far (grb_reduce_vector on a pending result); does not flush */
int grb_lazy_fused_reductions(void);

/* apply on the device   operations.hpp:559-579 / :581-601 -> backend :878-957 (apply.hpp: host loops there).
 * Vector: w = f(u) on every stored element (dense: all of them; sparse: the nvals stored ones, indices copied), w takes
 * u's storage; w == u is allowed.  A mask is GrB_NOT_IMPLEMENTED (the reference prints "apply masked not implemented").
 * Matrix: in place on the stored values of both orientations (C == A, owned storage), as the reference's
 * C->h_csrVal_[i] = op(A->h_csrVal_[i]) + syncCpu.  binop is a grb_binary_op for the two BIND kinds, ignored otherwise. */
grb_info grb_vector_apply(grb_vector w, grb_vector mask, grb_accum accum, int unary, int binop, double scalar, grb_vector u,
                          grb_descriptor desc);
grb_info grb_matrix_apply(grb_matrix C, grb_matrix mask, grb_accum accum, int unary, int binop, double scalar, grb_matrix A,
                          grb_descriptor desc);

/* mxm, masked SpGEMM only   operations.hpp:22-48 -> backend :18-78 (spgemm.hpp:22-110) */
grb_info grb_mxm(grb_matrix C, grb_matrix mask, grb_accum accum, grb_semiring op, grb_matrix A, grb_matrix B,
                 grb_descriptor desc);
/* eWiseMult, matrix (x) broadcast scalar / vector, in place (C == A)   operations.hpp:206-267 ->
 * backend ewisemult.hpp:275-341,470-622 + kernels/ewisemult.hpp:160-237.  Vector form:
 * C(i,j) = A(i,j) (x) B(i), or B(j) with GrB_INP1 = GrB_TRAN; B must be dense. */
grb_info grb_matrix_eWiseMult_scalar(grb_matrix C, grb_semiring op, grb_matrix A, double val);
grb_info grb_matrix_eWiseMult_vector(grb_matrix C, grb_semiring op, grb_matrix A, grb_vector B,
                                     grb_descriptor desc);
/* reduce (matrix -> scalar)   operations.hpp:662-680 -> backend :1032-1059 (reduce.hpp:81-91) */
grb_info grb_reduce_matrix_scalar(double* val, grb_accum accum, grb_monoid op, grb_matrix A, grb_descriptor desc);
/* traceMxmTranspose (extension)   operations.hpp:698-711 -> backend :1076-1108 (trace.hpp:10-52):
 * *val = sum_i (+)_k A(i,k) (x) B(i,k), the trace of A (+).(x) B^T; A and B of one element type. */
grb_info grb_trace_mxm_transpose(double* val, grb_semiring op, grb_matrix A, grb_matrix B, grb_descriptor desc);
/* Extension: readMtx (graphblas/util.hpp:363-430) + Matrix::build in one call with the MatrixMarket text
 * parsed on the device (coordinate pattern / integer / real, general / symmetric).  *A is created here;
 * directed as readMtx (0 banner decides, 1 directed, 2 undirected); dims_out (nullable) = {nrows, ncols,
 * nvals after the loader}.  Where the loader drops entries of a valued file each survivor keeps its own
 * value (the reference leaves the values array unshifted there, util.hpp:311-323). */
grb_info grb_matrix_load_mtx(grb_matrix* A, const char* path, grb_dtype dtype, int directed, grb_index* dims_out);
/* tril   operations.hpp:872-886 -> tri.hpp:10-53 (host side, as in the reference) */
grb_info grb_matrix_tril(grb_matrix C, grb_matrix A, grb_descriptor desc);

/* ---- Algorithms: the drivers of graphblas/algorithm/{bfs,...}.hpp built on the ops above.
 * *_fused variants run the same level loop on the device-resident representation
 * (bitmap frontier, no per-op host round trips) and must return identical results. */
typedef struct {
  int      levels;            /* iterations executed                                  */
  float    tight_ms;          /* HIP-event time of the level loop ("tight", bfs.hpp:42-88) */
  int64_t  edges_traversed;   /* sum of out-degree over reached vertices (TEPS numerator) */
  int32_t  reached;           /* vertices with a nonzero label                         */
} grb_bfs_result;

/* Per-level record, filled when `levels_out` is non-NULL (capacity max_levels). */
typedef struct {
  int32_t direction;          /* 0 push (SpMSpV), 1 pull (SpMV): desc lastmxv_          */
  int32_t frontier;           /* nf: vertices in the input frontier                    */
  int64_t frontier_edges;     /* push: mf = sum of frontier out-degrees; pull: inspected edges (profile bit 1) */
  int32_t discovered;         /* vertices labelled by this level                       */
  float   ms;                 /* HIP-event time of this level (only in profiling mode) */
} grb_bfs_level;

/* algorithm::bfs (algorithm/bfs.hpp:14-89) op by op through grb_assign/grb_vxm/grb_reduce. */
grb_info grb_bfs(grb_vector v, grb_matrix A, grb_index source, grb_descriptor desc,
                 grb_bfs_result* result);
/* Same result, fused device-resident level loop.  profile bit 0: HIP events around every
 * level's expansion kernel(s) -> grb_bfs_level.ms (cheap).  profile bit 1: pull levels also
 * count the edges they inspect (slower kernel variant) -> grb_bfs_level.frontier_edges. */
grb_info grb_bfs_fused(grb_vector v, grb_matrix A, grb_index source, grb_descriptor desc,
                       grb_bfs_result* result, grb_bfs_level* levels_out, int max_levels,
                       int profile);
/* The same traversal, queued without waiting: the launch is put on the library's stream and the call returns; the
 * result block is read by grb_bfs_wait.  K traversals into K vectors run back to back on the device with no host
 * round trip between them (the reference's loop returns to the host several times per LEVEL, algorithm/bfs.hpp:42-88:
 * `reduce` -> succ, `nvals`, the timing branches).  Traversals execute in the order they were queued; v, A and desc must
 * stay alive and untouched by the caller until the ticket has been waited for (v's contents are undefined until then;
 * any other entry point may be called meanwhile -- it is ordered behind the queued traversals on the stream).  At most
 * 256 tickets may be outstanding (GRB_INSUFFICIENT_SPACE).  A traversal the one-launch kernel does not serve (road-network
 * queues, GRB_SPARSE_MATRIX_FORMAT=1) runs to its end inside the enqueue call; its ticket waits like any other.
 * grb_bfs_wait returns what grb_bfs_fused would have returned (a launch that could not finish is re-run through the
 * host-driven level loop there); a ticket can be waited for once (GRB_INVALID_VALUE afterwards).
 * Ordering of the labels: grb_bfs_wait returns when the traversal's RECORD has arrived; the library's own stream is
 * ordered behind the launch, so every entry point called afterwards sees the complete depth vector.  A caller that reads
 * v's storage itself (grb_vector_device_ptrs) on a stream of its own gets the same guarantee a different way: for a
 * vector whose storage has been handed out, grb_bfs_wait waits for the launch itself
 * (tests/test_gpu_algorithms.py::test_bfs_wait_on_a_vector_read_through_its_device_pointer). */
typedef int64_t grb_bfs_ticket;
grb_info grb_bfs_fused_enqueue(grb_vector v, grb_matrix A, grb_index source, grb_descriptor desc,
                               grb_bfs_ticket* ticket);
grb_info grb_bfs_wait(grb_bfs_ticket ticket, grb_bfs_result* result);
/* Traversals in flight at once (1 .. 8; default 1).  With n > 1 the traversals queued by grb_bfs_fused_enqueue go round n
 * lanes -- a stream each, launches of (CUs / n) workgroups -- so that n of them are resident together: a traversal spends
 * two thirds of its time in barriers and latency chains that more CUs do not shorten, so two on half the device each
 * finish in 1.4 x the time of one on all of it (RMAT-22), four in 2 x.  Per-traversal results are unchanged; the
 * blocking grb_bfs_fused keeps the whole device (and, issued while lanes are busy, waits for their grids to drain).
 * Changing the number waits for everything queued.  n < 1 only queries.  Returns the previous value. */
int grb_bfs_set_lanes(int n);
/* Traversals side by side in one LAUNCH (1 .. 12; default 1).  With k > 1 the traversals queued by grb_bfs_fused_enqueue
 * share launches: a launch is k sub-grids of one workgroup per CU each (512 threads for two, 256 up to four, 128 up to
 * twelve -- that instance is built for six waves per SIMD), every sub-grid runs one traversal at a time on state,
 * barrier counters and bitmaps of its own, and when it has finished one it draws the next of the launch's traversals (up to 48 per launch) from a counter.  A traversal is
 * barriers and dependent-load chains for half of its time; the CU's wave scheduler fills them with the other
 * traversals' work, and -- unlike lanes -- nothing depends on how the runtime maps streams to hardware queues: it is one
 * launch on the library's stream.  A ticket is issued at once; the launch goes out when one of the gathered tickets is
 * waited for, when any other entry point is called (so the ordering rules of grb_bfs_fused_enqueue hold unchanged),
 * when a traversal of another matrix or descriptor is queued, or when 48 have gathered.  Per-traversal labels and
 * result blocks are those of grb_bfs_fused; a traversal's record is written after a barrier of its sub-grid, i.e. when
 * every label store has completed.  A traversal runs under the descriptor fields (mxvmode, switchpoint, edgeswitch,
 * max_niter) as they stood when it was QUEUED -- the descriptor's setters launch nothing, and one launch serves one set of
 * rules (a traversal queued under other rules starts a new launch).  Ignored while grb_bfs_set_lanes is above 1.  Changing the number launches and waits
 * for everything queued.  k < 1 only queries.  Returns the previous value.  (No counterpart in the reference, whose loop
 * is one traversal with several host round trips per level: algorithm/bfs.hpp:42-88.) */
int grb_bfs_set_coschedule(int k);
/* Measurement: HIP events on the library's stream around every launch of several traversals.  on != 0 starts collecting
 * (what has gathered is launched first); on == 0 stops, waits for the launches and reports their summed duration, their
 * number and the traversals they ran (bench.py's roofline block of the co-scheduled sibling). */
grb_info grb_bfs_coschedule_profile(int on, double* launch_ms_total, int* launches, int* traversals);
/* Host time (microseconds, summed since the last reset) inside the one-launch traversal's two halves -- queueing the
 * launches / waiting for and unpacking the record -- and the number of traversals; any pointer may be NULL. */
grb_info grb_bfs_host_times(double* enqueue_us, double* wait_us, long long* calls, int reset);

/* ---- 1-D vertex-partitioned BFS level steps (one process per GPU).
 * The reference has no multi-GPU code (backend/cuda/descriptor.hpp:242,283-284 parse
 * --ndevice and ignore it); these follow the single-GPU level loop.  The rank owns
 * vertices [lo, lo + nrows(A)), lo % 64 == 0; A_out / A_in are nrows x n_global matrices
 * whose CSR rows are the owned vertices' out- / in-neighbour lists (global ids).  Bitmaps
 * are 2*ceil(n_global/64) 32-bit words on the device.  The "new bits" bitmaps of all ranks
 * are OR-combined by the caller (RCCL) between push/pull and apply. */
grb_info grb_bfs_part_pull(grb_matrix A_in, grb_index lo, grb_index n_global, const uint32_t* d_vis,
                           uint32_t* d_new /* fully written: zeroed, then the owned word range */,
                           float* d_label_local, float new_label);
grb_info grb_bfs_part_push(grb_matrix A_out, grb_index lo, grb_index n_global, const uint32_t* d_frontier,
                           const uint32_t* d_vis, uint32_t* d_work /* scratch bitmap */,
                           uint32_t* d_new /* fully written */, int64_t* expanded_edges_out /* nullable */);
grb_info grb_bfs_part_apply(const uint32_t* d_new_global, uint32_t* d_vis, grb_index lo, grb_index n_local,
                            grb_index n_global, float* d_label_local, float new_label,
                            int32_t* discovered_out);
grb_info grb_bfs_part_tally(grb_matrix A_out, const float* d_label_local, int64_t* edges_out,
                            int32_t* reached_out);
/* Leaner variants for the host-driven level loop (one launch and one host wake-up each):
 * apply2 = apply + the out-degree sum of the OWNED newly discovered vertices (A_out nullable) and, given the
 * out-degree of every vertex, of ALL of them (the same number on every rank: edge-aware direction switch);
 * push_small = push for a frontier with few out-edges on this rank (d_new fully written);
 * seed = zero vis / new / labels and plant the source; or_parts = OR of `world` all-gathered bitmaps. */
grb_info grb_bfs_part_apply2(const uint32_t* d_new_global, uint32_t* d_vis, grb_index lo, grb_index n_local,
                             grb_index n_global, grb_matrix A_out, const int32_t* d_deg_full /* nullable */,
                             float* d_label_local, float new_label, int32_t* discovered_out,
                             int64_t* local_frontier_edges_out, int64_t* frontier_edges_out /* -1 without d_deg_full */);
grb_info grb_bfs_part_push_small(grb_matrix A_out, grb_index lo, grb_index n_global, const uint32_t* d_frontier,
                                 const uint32_t* d_vis, uint32_t* d_new);
grb_info grb_bfs_part_seed(uint32_t* d_vis, uint32_t* d_new_global, float* d_label_local, grb_index lo,
                           grb_index n_local, grb_index n_global, grb_index source);
grb_info grb_bitmap_or_parts(const uint32_t* d_parts, int world, grb_index nwords, uint32_t* d_out);
/* label == value -> 0 on the owned labels (the host-driven loop after a max_niter cut-off, bfs.hpp:48-66) */
grb_info grb_bfs_part_unlabel(float* d_label_local, grb_index n_local, float value);

/* ---- The same traversal with the level loop on the DEVICE (csrc/bfs_part_run.hip).  A rank context holds the
 * shards (A_out / A_in: the owned vertices' out- / in-neighbour lists, nrows x n_global, global column ids;
 * A_in NULL = the graph is symmetric and one shard serves both), the replicated n-bit visited bitmap, the
 * frontier bitmaps of its owned vertices and the scalars of the level loop.  d_deg_full: out-degree of every
 * vertex (device, int32, replicated; must outlive the context).  grb_bfs_part_run enqueues ONE co-resident
 * launch per level (apply the gathered new bits -> grid barrier -> decide -> expand) followed by ONE all-gather
 * of the n/8-byte new-bits bitmaps on the library communicator (grb_comm_*; skipped when world == 1) and reads
 * nothing back until the traversal has ended: launch k + 1 is enqueued when launch k - 1 has reported through
 * one pinned word -- a rule that uses only values identical on every rank, so all ranks issue the same number
 * of collectives.  levels_per_launch > 1 (world == 1 only) runs that many levels inside one launch.
 * grb_bfs_part_run_group drives `nranks` contexts of a world of `nranks` in lock-step on ONE device with device
 * copies as the all-gather: the test stand-in for a multi-GPU run.  Labels are those of algorithm::bfs
 * (algorithm/bfs.hpp:14-89); the reference has no multi-GPU code (backend/cuda/descriptor.hpp:242). */
typedef struct grb_part_s* grb_part;
typedef struct {
  int32_t levels;             /* levels expanded                                        */
  int32_t launches;           /* level launches enqueued (levels + 2 when one level per launch) */
  int32_t hit_cap;            /* max_niter ended the loop with a non-empty frontier     */
  int64_t edges_traversed;    /* sum of out-degree over reached vertices, whole graph   */
  int64_t reached;            /* vertices with a nonzero label, whole graph             */
  float   ms;                 /* HIP-event time: first launch .. label pass             */
} grb_part_bfs_result;
grb_info grb_part_new(grb_part* part, int rank, int world, grb_index n_global, grb_index lo, grb_matrix A_out,
                      grb_matrix A_in /* nullable */, const int32_t* d_deg_full, int64_t nnz_global);
grb_info grb_part_free(grb_part part);
grb_info grb_bfs_part_run(grb_part part, grb_index source, int mxvmode, float switchpoint, float edgeswitch,
                          int max_niter, int levels_per_launch, float* d_label_local, grb_part_bfs_result* result,
                          grb_bfs_level* levels_out /* nullable */, int max_levels);
grb_info grb_bfs_part_run_group(grb_part* parts, int nranks, grb_index source, int mxvmode, float switchpoint,
                                float edgeswitch, int max_niter, float* const* d_labels,
                                grb_part_bfs_result* results /* nranks */, grb_bfs_level* levels_out, int max_levels);

/* ---- algorithm::sssp (algorithm/sssp.hpp:15-103) on the same 1-D partition, in its frontier form with the round
 * loop on the device (csrc/sssp_part_run.hip).  A_out: the owned vertices' out-edges (nrows x n_global, global
 * column ids, f32 weights >= 0).  Per launch: the (vertex, candidate) pairs the ranks exchanged are applied, a
 * round that every rank has finished expanding is closed (synchronous rounds: the distances after every round and
 * the loop counter are the reference's), the next queue is expanded -- owned targets at once, the others into an
 * outbox of `outbox_pairs` pairs; one all-gather of the outboxes follows every launch (grb_comm_*).  A full outbox
 * only postpones the rest of the round to the next launch.  Nothing is read back until the loop has ended (launch
 * rule as grb_bfs_part_run).  d_dist_local receives the owned distances (FLT_MAX = unreached).
 * grb_sssp_part_run_group: every rank of a world of `nranks` on one device (tests). */
typedef struct grb_part_sssp_s* grb_part_sssp;
typedef struct {
  int32_t iterations;         /* the reference's loop counter at exit (max_niter + 1 when the cap ended it) */
  int32_t rounds;             /* rounds closed                                          */
  int32_t launches;
  int32_t hit_cap;
  float   ms;                 /* HIP-event time, first launch .. distances copied       */
} grb_part_sssp_result;
grb_info grb_part_sssp_new(grb_part_sssp* part, int rank, int world, grb_index n_global, grb_index lo, grb_matrix A_out,
                           int outbox_pairs);
grb_info grb_part_sssp_free(grb_part_sssp part);
grb_info grb_sssp_part_run(grb_part_sssp part, grb_index source, int max_niter, int rounds_per_launch,
                           float* d_dist_local, grb_part_sssp_result* result);
grb_info grb_sssp_part_run_group(grb_part_sssp* parts, int nranks, grb_index source, int max_niter,
                                 float* const* d_dist_local, grb_part_sssp_result* results /* nranks */);

typedef struct {
  int    iterations;          /* loop iterations executed                              */
  float  tight_ms;            /* HIP-event time of the loop                            */
  double last_value;          /* sssp: distances improved by the last round (the part of the
                               * reference's `succ` that decides termination; the op-by-op
                               * driver reports the reference's reduce(m) itself);
                               * pr: last residual `error`                              */
} grb_algo_result;

/* mxm with a dense right-hand side (declared and left a stub by the reference:
 * backend/cuda/operations.hpp:52-70 "SpMM and GEMM not implemented yet", spmm.hpp:15-27):
 *   C = A (+).(x) B   (tran = 0)      C = A^T (+).(x) B   (tran = 1)
 * B [ncols(A or A^T) x k] and C [nrows x k] dense row-major DEVICE arrays of A's element type, any of
 * the 17 semirings; no mask / accum (NULL in the reference's signature).  Row sums in the stored order of a row's
 * entries.  (The Boolean multi-frontier product -- many traversals at once -- is grb_bfs_batch, not this.) */
grb_info grb_spmm(grb_semiring op, grb_matrix A, int tran, const void* d_B, void* d_C, grb_index k,
                  grb_descriptor desc);

/* How the generic SpMV (grb_k_spmv, the pull half of mxv / vxm) reads the input vector for this orientation:
 * `nhot` leading values of the rank-packed vector are staged in LDS; with `bands` > 1 the matrix is also split
 * by column rank into that many bands with an LDS prefix each (`band_nnz` entries in `pieces` (row, band) runs
 * live outside the first prefix's part).  warm != 0 prepares the plan first (otherwise the first product
 * does).  No reference counterpart: mgpu::SpmvCsrBinary (backend/cuda/spmv.hpp:188-190) has no plan. */
grb_info grb_spmv_plan_info(grb_matrix A, int tran, int warm, int* bands, int64_t* band_nnz, int64_t* pieces, int* nhot);
/* How many LDS prefixes (column bands) SpMV plans prepared from now on may use: 1..8, 1 = the one-prefix kernel
 * (the default; the environment variable GRB_SPMV_BANDS sets the initial value).  k <= 0 only queries.  Returns
 * the value in force.  Plans already prepared keep their layout. */
int grb_spmv_set_bands(int k);
/* Which matrix format the generic SpMV multiplies from (csrc/spmv_cband.hpp; GRB_SPMV_FORMAT=csr|auto|cband sets the
 * initial value): 0 the CSR arrays only; 1 (default) a second, column-sorted copy -- row bands of <= 32 Ki rows
 * whose entries are sorted by column rank and coded in 16 + 16 bits, values dropped when all equal -- on
 * orientations whose columns are skewed enough for the hub packing, for the commutative monoids (plus, times,
 * min, max, or, and); 2 that format wherever the monoid allows it.  fmt < 0 only queries.  Returns the value in
 * force.  grb_spmv_format_info reports the prepared copy of an orientation: groups of 64 coded entries (padding
 * included), bands, work items, rows in the hub band, whether the values are stored (iso = 0) and the bytes one
 * launch moves by design.  Results: identical for integer data and the idempotent monoids; float sums are formed
 * in column-rank order by atomics (within rounding of the CSR kernel's, not bit-reproducible run to run).
 * Replaces mgpu::SpmvCsrBinary (backend/cuda/spmv.hpp:178-220) as the CSR kernel does. */
int grb_spmv_set_format(int fmt);
/* Under `auto` an orientation takes the column-sorted format once it has run this many products through the CSR
 * kernel (default 48, GRB_SPMV_CBAND_AFTER; 0 = at its first product): preparing the second copy costs about
 * 70 launches' worth of what it saves per launch, so a matrix multiplied a few dozen times never pays for it.
 * launches < 0 only queries.  Returns the previous value. */
int grb_spmv_set_reuse_threshold(int launches);
grb_info grb_spmv_format_info(grb_matrix A, int tran, int* in_use, int64_t* groups, int* bands, int* items,
                              int* hub_rows, int* iso, int64_t* bytes_per_launch);

/* Which order grb_sssp (and algorithm::sssp through the shadow header) relaxes in on eligible matrices:
 *   -1 (default)  the work-efficient near / far order (csrc/sssp_nearfar.hip) for graphs with fewer than 8 stored
 *                 entries per row whose weights are small non-negative integers -- distances AND the reported round
 *                 count are the reference's (graphblas/algorithm/sssp.hpp:53-90); everything else runs the
 *                 reference's synchronous rounds (csrc/sssp_persist.hip);
 *    0            always the synchronous rounds;
 *    1            near / far whenever the matrix is f32 with non-negative weights (distances identical; with
 *                 non-integer weights the reported round count may differ from the reference's).
 * A run whose round count would exceed the descriptor's max_niter, or that asks for per-round records
 * (--timing), always takes the synchronous rounds.  mode < -1 only queries.  Returns the mode in force;
 * the environment variable GRB_SSSP_NEARFAR sets the initial value. */
int grb_sssp_set_nearfar(int mode);
/* 0 if the last grb_sssp of this process produced its result with the synchronous rounds, else the number of
 * passes the near / far order took. */
int grb_sssp_last_order(void);
/* After a grb_sssp that ran the near / far order: {vertices expanded, out-edges relaxed, vertices marked dirty} summed
 * over its passes -- what that kernel's algorithmic bytes are priced on (bench.py, config 3). */
void grb_sssp_last_work(int64_t* out3);

/* Batched traversals = the multi-frontier product (extension; the reference leaves sparse x dense
 * mxm a stub, backend/cuda/operations.hpp:52-70, spmm.hpp:15-27): 1 <= k <= 64 sources traversed at
 * once, one 64-bit word per vertex (bit s = source s), levels as word-wide OR.  v[s] (k dense f32
 * vectors of size nrows) receive exactly the labels grb_bfs / grb_bfs_fused give for sources[s]
 * under the same descriptor (mxvmode forces a direction; max_niter caps the levels).
 * result->edges_traversed / reached are summed over the k traversals. */
grb_info grb_bfs_batch(grb_vector* v, int k, grb_matrix A, const grb_index* sources, grb_descriptor desc,
                       grb_bfs_result* result);
/* Levels in which every live source is pushed and at most `edges` out-edges leave the frontiers run inside one
 * co-resident launch, one grid barrier per level, instead of four kernels and a host round trip each (the first level,
 * the tail, every level of a high-diameter graph).  0: never; < 0: only query.  Returns the previous limit
 * (default 1 Mi; GRB_BATCH_TAIL=0 / GRB_BATCH_TAIL_EDGES set it from the environment).  Same labels either way. */
long long grb_bfs_batch_set_tail(long long edges);

/* One record per iteration of the last grb_sssp / grb_pr / grb_cc call on a descriptor: what the
 * reference's drivers print per iteration under --timing 1 (sssp.hpp:55-62, pr.hpp:53-62) or 2
 * (cc.hpp:61-71).  Kept when the descriptor's `timing` argument is non-zero. */
typedef struct {
  int32_t iteration;          /* 1-based                                                        */
  int32_t direction;          /* desc lastmxv_ after the iteration: GRB_PUSHONLY / GRB_PULLONLY */
  double  value;              /* sssp: f1.nvals (vertices improved); pr: error; cc: succ        */
  float   ms;                 /* time of the iteration ("gpu_tight")                            */
  int32_t reserved;
} grb_algo_iter;
/* Copies up to `cap` records into out (nullable); *count = records held (may exceed cap). */
grb_info grb_descriptor_iter_log(grb_descriptor desc, grb_algo_iter* out, int cap, int* count);

/* algorithm::sssp (algorithm/sssp.hpp:15-103): v = distances, FLT_MAX when unreachable.
 * Non-negative f32 weights run the same synchronous rounds in one launch (sssp_persist.hip);
 * anything else, or GRB_SSSP_FUSED=0, runs the reference's call sequence op by op. */
grb_info grb_sssp(grb_vector v, grb_matrix A, grb_index source, grb_descriptor desc,
                  grb_algo_result* result);
/* algorithm::pr (algorithm/pr.hpp:15-94): A must already be the scaled column-stochastic
 * matrix the driver prepares (example/gpr.cu:67-90). */
grb_info grb_pr(grb_vector p, grb_matrix A, float alpha, float eps, grb_descriptor desc,
                grb_algo_result* result);

/* algorithm::cc (algorithm/cc.hpp:17-136): FastSV; v and A are int; v = parent labels.  Per iteration the
 * MinimumSelectSecond product and ONE launch for the element-wise tail (cc.hpp:77-119: three eWiseAdd,
 * assignScatter, extractGather, eWiseMult, reduce, masked assign, two dup); grb_cc_set_fused(0) (or GRB_CC_FUSED=0
 * in the environment) runs the tail as the reference's call sequence instead; on < 0 only queries. */
grb_info grb_cc(grb_vector v, grb_matrix A, int seed, grb_descriptor desc, grb_algo_result* result);
int grb_cc_set_fused(int on);

/* algorithm::tc (algorithm/tc.hpp:15-54): A = lower triangle (int), B = buffer matrix.
 * The reference forms B<A> = A (+.x) A^T and reduces it (tc.hpp:38-43); B is its "buffer matrix" (tc.hpp:17) and is not
 * read again.  When A is a STRICTLY lower (or strictly upper) triangle whose stored values are all 1 -- the matrix the
 * reference's driver builds (example/gtc.cu) -- that sum is the number of triangles, which does not depend on which way the edges point:
 * the library then counts on the degree-ordered orientation of the same edges (csrc/tc_count.hip: every vertex keeps
 * the neighbours of higher degree, the lists are short, one end of every edge is looked up in an LDS bitmap of the other)
 * and LEAVES B AS IT WAS.  The orientation is prepared by the first count on a matrix and kept with it.  Any other
 * matrix or descriptor (INP0 transposed, ...), and every matrix after grb_tc_set_product(1) (or GRB_TC_PRODUCT=1 in the
 * environment), runs the reference's two calls with the product in B.  ntris is the same number either way. */
grb_info grb_tc(int64_t* ntris, grb_matrix A, grb_matrix B, grb_descriptor desc, grb_algo_result* result);
/* 0 (default): the count without the product where it is a count AND pays -- the first count asked of a matrix without
 * long rows (squared row lengths summing to at most 256 per entry: road networks, uniform random graphs, whose product is
 * cheap) goes through the product, the second prepares the orientation; 1: always the product in B; 2: the count wherever
 * it is a count; < 0: query.  Returns the previous setting.  GRB_TC_PRODUCT=<0|1|2> in the environment sets the start. */
int grb_tc_set_product(int on);
/* The orientation a matrix keeps after its first count (about 34 bytes per stored entry: the lists, a 16-byte descriptor
 * per edge end, the task lists) is released by grb_matrix_free and by whatever rewrites the matrix; this releases it
 * at once -- the next count prepares it again. */
grb_info grb_tc_release(grb_matrix A);
typedef struct {
  int32_t path;           /* of the last grb_tc: 0 = product + reduce, 1 = counted on the oriented lists            */
  float   prep_ms;        /* the orientation, when that call had to build it (0 when the matrix brought it along)    */
  float   count_ms;       /* the counting kernels                                                                  */
  int32_t longest_list;   /* the longest list of the orientation                                                   */
  int32_t tasks[3];       /* workgroups launched: a wave + hash table, 512 threads + bitmap, 512 threads + hash table */
} grb_tc_info;
grb_info grb_tc_last(grb_tc_info* out);

/* The DENSE CORE of that product (csrc/mxm_core.hip): the k_want longest rows of the lower triangle L (all rows of a
 * length or none) as K x K bit rows H, and C_H(i, j) = sum_k H[i][k] H[j][k] for the entries (i, j) of L between core
 * rows -- the part of C<L> = L (+.x) L^T (algorithm/tc.hpp:15-54, kernels/spgemm.hpp:17-79) in which the lists are
 * dense enough to be bit rows.  method 0: a lane per mask entry, AND + popcount over 32 columns at a time; method 1:
 * v_mfma_i32_16x16x64_i8 on 0/1 bytes expanded from the bit rows, every pair of a 128 x 128 tile computed and the mask
 * applied to the finished tile; method 2: MFMA for the tiles with at least dense_from mask entries (of 16 384), popcount
 * for the others.  count = sum of C_H over those entries; checksum = position-weighted sum of the per-entry results (equal
 * across methods only if every entry is).  No counterpart in the reference (SURVEY.md 8(f)2 sketches it). */
typedef struct {
  int32_t  core_rows;          /* K */
  int32_t  min_row_length;     /* rows of at least this many entries are core rows */
  int64_t  core_entries;       /* entries of L between core rows = results */
  int64_t  count;
  uint64_t checksum;
  float    build_ms;           /* HIP events: ranks, bit rows, prefix counts, tile list */
  float    product_ms;         /* HIP events: the product kernels alone */
  int32_t  tiles;              /* 128 x 128 tiles with at least one entry */
  int32_t  tiles_mfma;         /* ... of which the MFMA kernel took */
  int32_t  tiles_by_density[10];   /* those tiles by tenths of 16 384 entries */
} grb_tc_core_result;
grb_info grb_tc_dense_core(grb_matrix L, int k_want, int method, int dense_from, grb_tc_core_result* res);

/* ---- The remaining drivers of graphblas/algorithm/  and the two extension
 * operations only they use. */
/* scatter   operations.hpp:748-761 -> backend :1110-1142 (scatter.hpp:10-82): w[(Index)u[k]] = val
 * for every stored u[k] with 0 < u[k] < size(w); masked variants are no-ops as in the reference. */
grb_info grb_scatter(grb_vector w, grb_vector mask, grb_vector u, double val, grb_descriptor desc);
/* graphColor   operations.hpp:816-826 -> backend/cuda/color.hpp:18-88 (cuSPARSE csrcolor there):
 * w = a proper colouring of A's graph, colours from 0; *ncolors (nullable) = colours used. */
grb_info grb_graph_color(grb_vector w, grb_matrix A, grb_descriptor desc, int* ncolors);
/* algorithm::mis (algorithm/mis.hpp:22-141): v = 1 on a maximal independent set chosen by Luby
 * rounds over the int weight vector `weights` (NULL: srand(seed) + rand() per vertex on the host,
 * as apply(set_random) does there, algorithm/common.hpp:8-20).  v, A, weights are int. */
grb_info grb_mis(grb_vector v, grb_matrix A, int seed, grb_vector weights, grb_descriptor desc,
                 grb_algo_result* result);
/* algorithm::gcJP / gcMIS / gcIS (algorithm/gc.hpp:258-421, :152-255, :43-149); algo = --gcalgo
 * (0 JP, 1 MIS, 2 IS).  v = colours from 1. */
grb_info grb_gc(grb_vector v, grb_matrix A, int seed, grb_vector weights, int max_colors, int algo,
                grb_descriptor desc, grb_algo_result* result);
/* algorithm::lgc (algorithm/lgc.hpp:14-176): p = approximate personalised PageRank from s. */
grb_info grb_lgc(grb_vector p, grb_matrix A, grb_index s, double alpha, double eps, grb_descriptor desc,
                 grb_algo_result* result);
/* algorithm::diameter (algorithm/diameter.hpp:14-59): largest BFS eccentricity over the sources
 * s_start..s_end-1 and the last source attaining it; v = the last traversal's depth labels. */
grb_info grb_diameter(grb_vector v, grb_matrix A, grb_index s_start, grb_index s_end, grb_descriptor desc,
                      int* diameter_max, int* diameter_ind);

/* ---- The library's RCCL communicator (the reference has no multi-GPU path:
 * backend/cuda/descriptor.hpp:242,283-284 leave --ndevice unused).  One process per GPU; collectives
 * are enqueued from C++ on a second HIP stream, fenced with events against the stream of
 * grb_set_stream: a collective starts when the work enqueued before it has finished, and the
 * compute stream continues until grb_comm_wait() makes it wait (on the device) for the last one.
 * RCCL is bound with dlopen; GrB_NOT_IMPLEMENTED when it is absent. */
/* Host-staged transport instead of RCCL: the same collectives, each one a callback over pinned HOST buffers (the
 * library waits for its compute stream, stages the data, calls fn, copies the result back).  For ranks that cannot form
 * an RCCL communicator -- several processes on one GPU (the tests drive grb_bfs_part_run / grb_sssp_part_run from two
 * processes this way), a gloo group.  op 0: all-gather of `bytes` per rank, send -> recv[world * bytes]; op 1: in-place
 * all-gather of the byte ranges offsets[r] .. + counts[r] of recv (bytes = the buffer's extent); op 2: sum all-reduce
 * of bytes / 8 doubles in place.  fn returns 0 on success.  fn == NULL switches it off. */
typedef int (*grb_comm_host_fn)(void* user, int op, const void* send, void* recv, long long bytes,
                                const long long* offsets, const long long* counts);
grb_info grb_comm_set_host_transport(int rank, int world, grb_comm_host_fn fn, void* user);
grb_info grb_comm_unique_id(void* out128);                  /* ncclGetUniqueId: rank 0 calls it, every rank gets the bytes */
grb_info grb_comm_init(const void* id128, int rank, int world);
grb_info grb_comm_destroy(void);
grb_info grb_comm_info(int* rank, int* world);               /* world = 0: not initialised */
grb_info grb_comm_wait(void);
grb_info grb_comm_allgather(const void* d_send, void* d_recv, size_t bytes_per_rank);
/* rank r's slice = d_buf[offsets[r] .. +counts[r]) bytes, in place; afterwards every rank holds all slices */
grb_info grb_comm_allgatherv_inplace(void* d_buf, const long long* offsets, const long long* counts);
grb_info grb_comm_allreduce_sum_f64(void* d_buf, size_t count);
grb_info grb_comm_timing(int on);                            /* HIP-event time of every collective (adds a host wait) */
grb_info grb_comm_stats(double* total_us, long long* calls, int reset);
/* element-wise tail of a PageRank iteration on a chunk of owned rows (algorithm/pr.hpp:70-80):
 * p_next = y + c, *d_acc (double) += sum (p_next - p_old)^2 */
grb_info grb_pr_part_update(const void* d_y, const void* d_p_old, float c, void* d_p_next, grb_index n, void* d_acc);
/* The whole iteration loop of algorithm::pr (algorithm/pr.hpp:52-90) on this rank's in-edge shard, cut into nchunks
 * row chunks: chunk c's slice of the next vector is all-gathered (library communicator, if one is up) while chunk
 * c + 1 is multiplied; the squared residual is all-reduced and read once per iteration -- the only host read.
 * chunks[c]: local rows [row_cut[c], row_cut[c+1]) x n, values alpha / outdeg; vertex_cut: [world][nchunks+1] global
 * vertex ids of every rank's chunk boundaries; lo: this rank's first vertex; c_add = (1 - alpha) / n.
 * d_p_cur (initial vector) / d_p_next: n floats; d_y: n_local floats; d_acc: one double.
 * errors (may be NULL): max_niter doubles; *result_in_next: 1 when the result is in d_p_next. */
grb_info grb_pr_part_run(int nchunks, const grb_matrix* chunks, const long long* row_cut, const long long* vertex_cut,
                         grb_index lo, float c_add, float eps, int max_niter, void* d_p_cur, void* d_p_next, void* d_y,
                         void* d_acc, int* iterations, double* errors, int* result_in_next);

/* ---- Raw kernels on plain device pointers (micro-benchmarks / multi-GPU shards) --- */
/* Generic semiring SpMV  w[i] = (+)_j A[i,j] (x) u[j] on this matrix's CSR (tran=0) or
 * CSC (tran=1) arrays: the kernel behind the pull branch (backend/cuda/spmv.hpp:178-220).
 * mask may be NULL. */
grb_info grb_k_spmv(grb_matrix A, int tran, grb_semiring op, const void* d_u, const void* d_mask,
                    int scmp, int accum, void* d_w);
/* Algorithmic bytes of one grb_k_spmv launch: 8*nnz + 12*n + 4. */
int64_t grb_k_spmv_bytes(grb_matrix A, int tran);

#ifdef __cplusplus
}
#endif
#endif  /* GRB_HIP_H_ */
