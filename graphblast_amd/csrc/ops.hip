// ops.hip -- the operation layer: graphblas::{vxm,mxv,eWiseAdd,eWiseMult,reduce,assign}
// (graphblas/operations.hpp) and the storage/direction dispatch underneath them
// (backend/cuda/operations.hpp), behind the C ABI.  The dispatch rules, the direction
// heuristic and the documented quirks are the reference's; the kernels they select are
// the gfx950 ones in spmv.hip / spmspv.hip / elementwise.hip.
#include "common.hpp"

using namespace grb;

static inline int mask_is_f32(grb_vector m) { return m && m->dtype == GRB_F32; }

// ---------------------------------------------------------------------------------
// pull: backend/cuda/spmv.hpp:20-236
static grb_info spmv_dispatch(grb_vector w, grb_vector mask, grb_accum accum, int op, grb_matrix A, grb_vector u,
                              grb_descriptor desc) {
  const bool use_mask = mask != nullptr;
  const bool use_accum = accum != GRB_ACCUM_NULL;
  const bool use_scmp = desc->desc[GRB_MASK] == GRB_SCMP;
  const bool use_tran = desc->desc[GRB_INP0] == GRB_TRAN || desc->desc[GRB_INP1] == GRB_TRAN;
  const CsrArrays& M = use_tran ? A->csc : A->csr;
  SpmvPlan& plan = use_tran ? A->plan_csc : A->plan_csr;
  if (!M.ptr) return GRB_INVALID_OBJECT;
  GRB_TRY(matrix_ensure_plan(A, use_tran));
  // "functor == 1": add_op(3, 5) == 1 selects the Boolean fused-mask kernel (spmv.hpp:84-96)
  const int functor = (int)semiring_add(op, w->dtype, 3, 5);
  if (use_mask && desc->fusedmask && functor == 1) {
    if (mask->vec_type == GRB_DENSE) {
      // the per-vertex hint of the one-launch traversal is reused when that path has built it
      // (it describes the CSC orientation; building it here would cost more than a few pulls save)
      const Index* hint = use_tran ? A->d_pull_hint : nullptr;
      return k_spmv_masked_or(w->dtype, M, u->d_val, semiring_identity(op, w->dtype), mask->d_val,
                              mask_is_f32(mask), use_scmp, desc->earlyexit, desc->opreuse, hint, w->d_val);
    }
    if (mask->vec_type == GRB_SPARSE) return GRB_SUCCESS;   // "not implemented": prints, no-op
    return GRB_UNINITIALIZED_OBJECT;
  }
  // the generic branch reads mask->dense_.d_val_ without looking at the mask's storage
  // (spmv.hpp:203-212): a sparse or cleared mask acts through whatever its dense buffer last held
  if (use_mask && !mask->d_val) return GRB_INVALID_OBJECT;
  w->d_nnz = u->d_nnz;
  const CsrArrays& other = use_tran ? A->csr : A->csc;
  return k_spmv(op, w->dtype, M, plan, u->d_val, use_mask ? mask->d_val : nullptr, mask_is_f32(mask), use_scmp,
                use_accum, w->d_val, (other.ptr && other.n == plan.nminor && !A->csc_alias) ? other.ptr : nullptr);
}

// push: backend/cuda/spmspv.hpp:15-257
static grb_info spmspv_dispatch(grb_vector w, grb_vector mask, grb_accum accum, int op, grb_matrix A, grb_vector u,
                                grb_descriptor desc) {
  (void)accum;
  const bool use_mask = mask != nullptr;
  const bool desc_scmp = desc->desc[GRB_MASK] == GRB_SCMP;
  const bool use_tran = desc->desc[GRB_INP0] == GRB_TRAN || desc->desc[GRB_INP1] == GRB_TRAN;
  // default is CSC here: the transposed product walks CSR rows (spmspv.hpp:52-55)
  const CsrArrays& M = use_tran ? A->csr : A->csc;
  if (!M.ptr) return GRB_INVALID_OBJECT;
  // mask storage (spmspv.hpp:147-164, :199-216): dense -> applied; sparse -> "not implemented" is
  // printed and the product goes on UNMASKED, still through the masked epilogue (key-value mode
  // prunes reduced values == 0); anything else -> GrB_UNINITIALIZED_OBJECT
  int mask_mode = 0;
  if (use_mask) {
    if (mask->vec_type == GRB_DENSE) mask_mode = 1;
    else if (mask->vec_type == GRB_SPARSE) mask_mode = 2;
    else return GRB_UNINITIALIZED_OBJECT;
  }
  const Index out_size = use_tran ? A->ncols : A->nrows;
  Index nv = 0;
  GRB_TRY(k_spmspv(op, w->dtype, M, out_size, desc->struconly, u->s_ind, u->s_val, u->s_nvals,
                   mask_mode == 1 ? mask->d_val : nullptr, mask_is_f32(mask), mask_mode, desc_scmp ? 1 : 0,
                   w->s_ind, w->s_val, &nv));
  w->s_nvals = nv;
  return GRB_SUCCESS;
}

// backend/cuda/operations.hpp:80-209 (vxm) and :215-327 (mxv)
static grb_info mxv_common(grb_vector w, grb_vector mask, grb_accum accum, int op, grb_matrix A, grb_vector u,
                           grb_descriptor desc, bool is_vxm) {
  if (is_vxm) {
    if (desc->desc[GRB_INP0] != GRB_DEFAULT) return GRB_INVALID_VALUE;
    grb_descriptor_toggle(desc, GRB_INP1);                 // vxm == mxv on the transpose
  } else if (desc->desc[GRB_INP1] != GRB_DEFAULT) {
    return GRB_INVALID_VALUE;
  }
  const double identity = semiring_identity(op, u->dtype);
  const int mode = desc->desc[GRB_MXVMODE];
  // GRB_LOAD_BALANCE_MODE is read on every call (operations.hpp:110, :242): 2 = merge (default) is the only
  // implemented sparse-input path; 0 prints and returns GrB_NOT_IMPLEMENTED *without undoing the INP1 toggle*
  // (:161-163), 1 prints, leaves w an empty sparse vector and reports success (:167-177)
  const char* lb_env = getenv("GRB_LOAD_BALANCE_MODE");
  const int lb_mode = lb_env ? atoi(lb_env) : 2;
  grb_info info = GRB_SUCCESS;
  // no CSC storage (GRB_SPARSE_MATRIX_FORMAT = 1; getSymmetry() reports false for every matrix,
  // sparse_matrix.hpp:577-582): the direction is forced whatever the mxvmode says (:131-133, :258-260)
  if (A->format == 1) {
    if (is_vxm && u->vec_type == GRB_DENSE) info = grb_vector_dense2sparse(u, identity, desc);
    else if (!is_vxm && u->vec_type == GRB_SPARSE) info = grb_vector_sparse2dense(u, identity, desc);
  } else
  if (mode == GRB_PUSHPULL) info = grb_vector_convert(u, identity, desc->switchpoint, desc);
  else if (mode == GRB_PUSHONLY && u->vec_type == GRB_DENSE) info = grb_vector_dense2sparse(u, identity, desc);
  else if (mode == GRB_PULLONLY && u->vec_type == GRB_SPARSE) info = grb_vector_sparse2dense(u, identity, desc);
  if (info == GRB_SUCCESS) {
    if (u->vec_type == GRB_SPARSE && lb_mode == 0) {
      (void)grb_vector_set_storage(w, GRB_DENSE);
      fprintf(stdout, "Simple SPMSPV not implemented yet!\n");
      return GRB_NOT_IMPLEMENTED;                          // INP1 stays toggled, as in the reference
    } else if (u->vec_type == GRB_SPARSE && lb_mode == 1) {
      fprintf(stdout, "Error: B40C load-balance algorithm not implemented yet!\n");
      info = grb_vector_set_storage(w, GRB_DENSE);
      if (info == GRB_SUCCESS) info = grb_vector_dense2sparse(w, identity, desc);
      desc->lastmxv = GRB_PUSHONLY;
    } else if (u->vec_type == GRB_SPARSE) {
      if (lb_mode != 2) fprintf(stdout, "Error: Invalid load-balance algorithm!\n");
      else {
        info = grb_vector_set_storage(w, GRB_SPARSE);
        if (info == GRB_SUCCESS) info = spmspv_dispatch(w, mask, accum, op, A, u, desc);
      }
      desc->lastmxv = GRB_PUSHONLY;
    } else {
      if (is_vxm) info = grb_vector_set_storage(w, GRB_DENSE);
      else info = grb_vector_sparse2dense(w, identity, desc);
      if (info == GRB_SUCCESS) info = spmv_dispatch(w, mask, accum, op, A, u, desc);
      desc->lastmxv = GRB_PULLONLY;
    }
  }
  if (is_vxm) grb_descriptor_toggle(desc, GRB_INP1);
  return info;
}

extern "C" {

grb_info grb_vxm(grb_vector w, grb_vector mask, grb_accum accum, grb_semiring op, grb_vector u, grb_matrix A,
                 grb_descriptor desc) { GRB_API_ENTER();
  if (!w || !u || !A || !desc) return GRB_UNINITIALIZED_OBJECT;        // operations.hpp:66-68
  grb_index u_nvals = 0;
  GRB_TRY(grb_vector_nvals(u, &u_nvals));
  if (u_nvals == 0) return GRB_UNINITIALIZED_OBJECT;                     // operations.hpp:71-74
  if (A->nrows != u->nsize) return GRB_DIMENSION_MISMATCH;               // checkDimRowSize
  if (A->ncols != w->nsize) return GRB_DIMENSION_MISMATCH;               // checkDimColSize
  if (mask && mask->nsize != w->nsize) return GRB_DIMENSION_MISMATCH;    // checkDimSizeSize
  if (!A->built) return GRB_UNINITIALIZED_OBJECT;
  return mxv_common(w, mask, accum, op, A, u, desc, true);
}

grb_info grb_mxv(grb_vector w, grb_vector mask, grb_accum accum, grb_semiring op, grb_matrix A, grb_vector u,
                 grb_descriptor desc) { GRB_API_ENTER();
  if (!w || !u || !A || !desc) return GRB_UNINITIALIZED_OBJECT;        // operations.hpp:106-108
  grb_index u_nvals = 0;
  GRB_TRY(grb_vector_nvals(u, &u_nvals));
  if (u_nvals == 0) return GRB_UNINITIALIZED_OBJECT;                     // operations.hpp:111-114
  if (A->ncols != u->nsize) return GRB_DIMENSION_MISMATCH;
  if (A->nrows != w->nsize) return GRB_DIMENSION_MISMATCH;
  if (mask && mask->nsize != w->nsize) return GRB_DIMENSION_MISMATCH;
  if (!A->built) return GRB_UNINITIALIZED_OBJECT;
  return mxv_common(w, mask, accum, op, A, u, desc, false);
}

// backend/cuda/operations.hpp:331-410 + ewisemult.hpp:32-270
grb_info grb_eWiseMult(grb_vector w, grb_vector mask, grb_accum accum, grb_semiring op, grb_vector u, grb_vector v,
                       grb_descriptor desc) { GRB_API_ENTER_QUEUE();
  (void)accum;
  if (!w || !u || !v || !desc) { GRB_TRY(lazy_flush()); return GRB_UNINITIALIZED_OBJECT; }
  if (u->nsize != v->nsize || u->nsize != w->nsize || (mask && mask->nsize != w->nsize)) {
    GRB_TRY(lazy_flush());
    return GRB_DIMENSION_MISMATCH;
  }
  const int dt = u->dtype;
  {
    // dense (x) dense, no mask: queued when every operand is library-owned (lazy.hip); anything else flushes first
    grb_info fi = GRB_SUCCESS;
    const bool plain = !mask && u->vec_type == GRB_DENSE && v->vec_type == GRB_DENSE && w->d_val && w->d_owned;
    if (plain) {
      const int before = w->vec_type;
      w->vec_type = GRB_DENSE;                              // what grb_vector_set_storage(w, GRB_DENSE) does below
      if (lazy_try(LZ_MULT_VV, op, w, u, v, 0.0, &fi)) return GRB_SUCCESS;
      w->vec_type = before;
    } else {
      fi = lazy_flush();
    }
    GRB_TRY(fi);
  }
  if (u->vec_type == GRB_SPARSE && v->vec_type == GRB_SPARSE)
    GRB_TRY(grb_vector_set_storage(v, GRB_DENSE));          // operations.hpp:361-367: flag only
  if (u->vec_type == GRB_DENSE && v->vec_type == GRB_DENSE) {
    if (mask && mask->vec_type == GRB_SPARSE) {
      GRB_TRY(grb_vector_set_storage(w, GRB_SPARSE));
      GRB_TRY(k_ewise_mult_dense_dense_spmask(op, dt, w->s_ind, w->s_val, mask->s_ind, mask->s_val,
                                              mask_is_f32(mask), mask->s_nvals, u->d_val, v->d_val));
      w->s_nvals = mask->s_nvals;
      return GRB_SUCCESS;
    }
    if (mask && mask->vec_type != GRB_DENSE) return GRB_INVALID_OBJECT;
    GRB_TRY(grb_vector_set_storage(w, GRB_DENSE));
    return k_ewise_mult_dense_dense(op, dt, w->d_val, mask ? mask->d_val : nullptr, mask_is_f32(mask), u->d_val,
                                    v->d_val, u->nsize);
  }
  grb_vector sp, de;
  int reverse;
  if (u->vec_type == GRB_SPARSE && v->vec_type == GRB_DENSE) { sp = u; de = v; reverse = 0; }
  else if (u->vec_type == GRB_DENSE && v->vec_type == GRB_SPARSE) { sp = v; de = u; reverse = 1; }
  else return GRB_INVALID_OBJECT;
  GRB_TRY(grb_vector_set_storage(w, GRB_SPARSE));
  if (mask && mask->vec_type == GRB_SPARSE) {
    GRB_TRY(k_ewise_mult_sparse_dense_spmask(op, dt, w->s_ind, w->s_val, mask->s_ind, mask->s_val,
                                             mask_is_f32(mask), mask->s_nvals, sp->s_ind, sp->s_val, sp->s_nvals,
                                             de->d_val, reverse));
    w->s_nvals = mask->s_nvals;
    return GRB_SUCCESS;
  }
  GRB_TRY(k_ewise_mult_sparse_dense(op, dt, w->s_ind, w->s_val, sp->s_ind, sp->s_val, sp->s_nvals, de->d_val,
                                    reverse));
  w->s_nvals = sp->s_nvals;
  if (mask && mask->vec_type == GRB_DENSE)
    GRB_TRY(k_zero_dense_identity(dt, mask->d_val, mask_is_f32(mask), semiring_identity(op, dt), w->s_ind, w->s_val,
                                  w->s_nvals));
  return GRB_SUCCESS;
}

// backend/cuda/operations.hpp:567-627 + ewiseadd.hpp
grb_info grb_eWiseAdd(grb_vector w, grb_vector mask, grb_accum accum, grb_semiring op, grb_vector u, grb_vector v,
                      grb_descriptor desc) { GRB_API_ENTER_QUEUE();
  (void)accum;
  if (!w || !u || !v || !desc) { GRB_TRY(lazy_flush()); return GRB_UNINITIALIZED_OBJECT; }
  if (u->nsize != v->nsize || u->nsize != w->nsize || (mask && mask->nsize != w->nsize)) {
    GRB_TRY(lazy_flush());
    return GRB_DIMENSION_MISMATCH;
  }
  const int dt = u->dtype;
  {
    grb_info fi = GRB_SUCCESS;
    const bool plain = !mask && u->vec_type == GRB_DENSE && v->vec_type == GRB_DENSE && w->d_val && w->d_owned;
    if (plain) {
      const int before = w->vec_type;
      w->vec_type = GRB_DENSE;
      if (lazy_try(LZ_ADD_VV, op, w, u, v, 0.0, &fi)) return GRB_SUCCESS;
      w->vec_type = before;
    } else {
      fi = lazy_flush();
    }
    GRB_TRY(fi);
  }
  const double identity = semiring_identity(op, dt);
  int ut = u->vec_type, vt = v->vec_type;
  if ((u == w && ut == GRB_SPARSE) || (v == w && vt == GRB_SPARSE)) {
    if (u == w) { GRB_TRY(grb_vector_sparse2dense(u, identity, desc)); ut = GRB_DENSE; }
    else { GRB_TRY(grb_vector_sparse2dense(v, identity, desc)); vt = GRB_DENSE; }
  }
  GRB_TRY(grb_vector_set_storage(w, GRB_DENSE));
  if (ut == GRB_SPARSE && vt == GRB_SPARSE) return GRB_SUCCESS;       // "not implemented", w untouched
  if (mask) return GRB_SUCCESS;                                       // masked variants: error print, no-op
  if (ut == GRB_DENSE && vt == GRB_DENSE)
    return k_ewise_add_dense_dense(op, dt, w->d_val, u->d_val, v->d_val, u->nsize);
  grb_vector sp, de;
  int reverse;
  if (ut == GRB_SPARSE && vt == GRB_DENSE) { sp = u; de = v; reverse = 0; }
  else if (ut == GRB_DENSE && vt == GRB_SPARSE) { sp = v; de = u; reverse = 1; }
  else return GRB_INVALID_OBJECT;
  // ewiseadd.hpp:93-156: w = dup(dense); w = op(w, identity) everywhere; overwrite at sparse indices
  if (de != w && de->d_val != w->d_val)
    GRB_TRY(k_copy(w->d_val, de->d_val, 4 * (size_t)w->nsize));
  GRB_TRY(k_ewise_add_const(op, dt, w->d_val, identity, reverse, w->nsize));
  return k_ewise_add_sparse_dense(op, dt, w->d_val, sp->s_ind, sp->s_val, de->d_val, sp->s_nvals);
}

// backend/cuda/operations.hpp:649-699 + ewiseadd.hpp:161-280
grb_info grb_eWiseAdd_scalar(grb_vector w, grb_vector mask, grb_accum accum, grb_semiring op, grb_vector u,
                             double val, grb_descriptor desc) { GRB_API_ENTER_QUEUE();
  (void)accum;
  if (!w || !u || !desc) { GRB_TRY(lazy_flush()); return GRB_UNINITIALIZED_OBJECT; }
  if (u->nsize != w->nsize) { GRB_TRY(lazy_flush()); return GRB_DIMENSION_MISMATCH; }
  if (mask) { GRB_TRY(lazy_flush()); return GRB_NOT_IMPLEMENTED; }
  const int dt = u->dtype;
  {
    grb_info fi = GRB_SUCCESS;
    if (u->vec_type == GRB_DENSE && w->d_val && w->d_owned) {
      const int before = w->vec_type;
      w->vec_type = GRB_DENSE;
      if (lazy_try(LZ_ADD_VS, op, w, u, nullptr, val, &fi)) return GRB_SUCCESS;
      w->vec_type = before;
    } else {
      fi = lazy_flush();
    }
    GRB_TRY(fi);
  }
  if (u->vec_type == GRB_DENSE) {
    GRB_TRY(grb_vector_set_storage(w, GRB_DENSE));
    if (u != w)
      GRB_TRY(k_copy(w->d_val, u->d_val, 4 * (size_t)w->nsize));
    return k_ewise_scalar(op, dt, 1, w->d_val, val, w->nsize);
  }
  if (u->vec_type == GRB_SPARSE) {
    GRB_TRY(grb_vector_set_storage(w, GRB_DENSE));
    const double fillv = semiring_add(op, dt, semiring_identity(op, dt), val);
    GRB_TRY(k_fill(dt, w->d_val, fillv, w->nsize));
    return k_ewise_add_sparse_dense(op, dt, w->d_val, u->s_ind, u->s_val, w->d_val, u->s_nvals);
  }
  return GRB_INVALID_OBJECT;
}

// backend/cuda/operations.hpp:1004-1030 + reduce.hpp:13-76
grb_info grb_reduce_vector(double* val, grb_accum accum, grb_monoid op, grb_vector u, grb_descriptor desc) { GRB_API_ENTER_QUEUE();
  (void)accum;
  if (!val || !u || !desc) { GRB_TRY(lazy_flush()); return GRB_UNINITIALIZED_OBJECT; }
  // the result of a pending chain of element-wise calls (pr.hpp:72-80: ..., eWiseAdd, reduce): folded by the chain's own launch
  if (u->vec_type == GRB_DENSE && grb::ApiScope::depth == 1) {
    bool done = false;
    GRB_TRY(lazy_flush_reduce(u, (int)op, val, &done));
    if (done) return GRB_SUCCESS;
  }
  GRB_TRY(lazy_flush());
  if (u->vec_type == GRB_SPARSE) {
    if (desc->struconly) { *val = (double)u->s_nvals; return GRB_SUCCESS; }   // reduce.hpp:71-72
    return k_reduce(op, u->dtype, u->s_val, u->s_nvals, val);
  }
  if (u->vec_type == GRB_DENSE) return k_reduce(op, u->dtype, u->d_val, u->nsize, val);
  return GRB_UNINITIALIZED_OBJECT;
}

// backend/cuda/operations.hpp:953-986 + reduce.hpp:109-145
grb_info grb_reduce_matrix_rows(grb_vector w, grb_vector mask, grb_accum accum, grb_monoid op, grb_matrix A,
                                grb_descriptor desc) { GRB_API_ENTER();
  (void)accum;
  if (!w || !A || !desc) return GRB_UNINITIALIZED_OBJECT;
  if (!A->built) return GRB_UNINITIALIZED_OBJECT;
  if (A->nrows != w->nsize) return GRB_DIMENSION_MISMATCH;
  GRB_TRY(grb_vector_set_storage(w, GRB_DENSE));
  if (mask) return GRB_NOT_IMPLEMENTED;
  if (desc->struconly) return GRB_SUCCESS;
  GRB_TRY(k_reduce_rows(op, A->dtype, A->csr.ptr, A->csr.val, A->nrows, w->d_val));
  w->d_nnz = A->nrows;
  return GRB_SUCCESS;
}

// backend/cuda/operations.hpp:822-860 + assign.hpp:14-241
grb_info grb_assign(grb_vector w, grb_vector mask, grb_accum accum, double val, grb_descriptor desc) { GRB_API_ENTER_QUEUE();
  (void)accum;
  if (!w || !desc) { GRB_TRY(lazy_flush()); return GRB_UNINITIALIZED_OBJECT; }
  if (mask && mask->nsize != w->nsize) { GRB_TRY(lazy_flush()); return GRB_DIMENSION_MISMATCH; }
  const int scmp = desc->desc[GRB_MASK] == GRB_SCMP;
  {
    // dense w under a dense mask of the same type: joins the queue of element-wise calls (lazy.hip)
    grb_info fi = GRB_SUCCESS;
    if (mask && mask != w && w->vec_type == GRB_DENSE && mask->vec_type == GRB_DENSE && mask->dtype == w->dtype) {
      if (lazy_try(LZ_ASSIGN, scmp, w, mask, nullptr, val, &fi)) return GRB_SUCCESS;
    } else {
      fi = lazy_flush();
    }
    GRB_TRY(fi);
  }
  if (w->vec_type == GRB_DENSE) {
    if (!mask) return GRB_SUCCESS;                          // unmasked: error print, no-op
    if (mask->vec_type == GRB_DENSE)
      return k_assign_dense_mask_dense(w->dtype, w->d_val, w->nsize, mask->d_val, mask_is_f32(mask), scmp, val);
    if (mask->vec_type == GRB_SPARSE) {
      if (scmp) return GRB_SUCCESS;                         // SCMP variant: error print, no-op
      return k_assign_dense_mask_sparse(w->dtype, w->d_val, mask->s_ind, mask->s_nvals, val);
    }
    return GRB_UNINITIALIZED_OBJECT;
  }
  if (w->vec_type == GRB_SPARSE) {
    if (!mask) return GRB_UNINITIALIZED_OBJECT;
    if (mask->vec_type == GRB_SPARSE) GRB_TRY(grb_vector_convert(mask, 0.0, 0.3f, desc));   // assign.hpp:147-150
    if (mask->vec_type == GRB_DENSE)
      GRB_TRY(k_assign_sparse_mask_dense(w->dtype, w->s_ind, w->s_val, w->s_nvals, mask->d_val, mask_is_f32(mask),
                                         scmp, val));
    else if (mask->vec_type != GRB_SPARSE)
      return GRB_UNINITIALIZED_OBJECT;
    Index nv = 0;
    GRB_TRY(k_sparse_prune(w->dtype, w->s_ind, w->s_val, w->s_nvals, val, &nv));             // assign.hpp:213-233
    w->s_nvals = nv;
    return GRB_SUCCESS;
  }
  return GRB_SUCCESS;                                       // unknown storage: nothing happens
}

// backend/cuda/operations.hpp:1170-1253 + scatter.hpp:85-123 / gather.hpp:11-50
static grb_info scatter_gather(grb_vector w, grb_vector mask, grb_vector u, grb_vector indices, bool gather) {
  if (!w || !u || !indices) return GRB_UNINITIALIZED_OBJECT;
  if (mask && mask->nsize != w->nsize) return GRB_DIMENSION_MISMATCH;
  if (indices->dtype != GRB_I32) return GRB_DOMAIN_MISMATCH;
  grb_index nindices = 0;
  GRB_TRY(grb_vector_nvals(indices, &nindices));
  const int ut = u->vec_type;
  if (indices->vec_type != ut) GRB_TRY(grb_vector_set_storage(indices, ut));
  if (w->vec_type != ut) GRB_TRY(grb_vector_set_storage(w, ut));
  if (mask) return GRB_SUCCESS;                          // masked variants: empty branch in the reference
  const void* uv;
  const int* iv;
  if (ut == GRB_DENSE) { uv = u->d_val; iv = (const int*)indices->d_val; }
  else if (ut == GRB_SPARSE) {
    // reference quirk: w is switched to SPARSE storage above but its DENSE buffer is written
    uv = u->s_val; iv = (const int*)indices->s_val;
  } else return GRB_SUCCESS;
  if (gather) return k_gather_indexed(w->dtype, w->d_val, w->nsize, iv, nindices, uv);
  return k_scatter_indexed(w->dtype, w->d_val, w->nsize, iv, nindices, uv);
}

grb_info grb_assignScatter(grb_vector w, grb_vector mask, grb_accum accum, grb_vector u, grb_vector indices,
                           grb_descriptor desc) { GRB_API_ENTER();
  (void)accum;
  if (!desc) return GRB_UNINITIALIZED_OBJECT;
  return scatter_gather(w, mask, u, indices, false);
}
grb_info grb_extractGather(grb_vector w, grb_vector mask, grb_accum accum, grb_vector u, grb_vector indices,
                           grb_descriptor desc) { GRB_API_ENTER();
  (void)accum;
  if (!desc) return GRB_UNINITIALIZED_OBJECT;
  return scatter_gather(w, mask, u, indices, true);
}

// ---- algorithm::bfs op by op (graphblas/algorithm/bfs.hpp:14-89) ------------------
grb_info grb_bfs(grb_vector v, grb_matrix A, grb_index source, grb_descriptor desc, grb_bfs_result* result) { GRB_API_ENTER();
  if (!v || !A || !desc) return GRB_UNINITIALIZED_OBJECT;
  if (source < 0 || source >= A->nrows) return GRB_INVALID_INDEX;
  const Index n = A->nrows;
  GRB_TRY(grb_vector_fill(v, 0.0));
  grb_vector f1 = nullptr, f2 = nullptr;
  GRB_TRY(grb_vector_new(&f1, GRB_F32, n));
  GRB_TRY(grb_vector_new(&f2, GRB_F32, n));
  grb_info info = GRB_SUCCESS;
  if (desc->desc[GRB_MXVMODE] == GRB_PULLONLY) {
    info = grb_vector_fill(f1, 0.0);
    if (info == GRB_SUCCESS) info = grb_vector_set_element(f1, 1.0, source);
  } else {
    float one = 1.f;
    info = grb_vector_build_sparse(f1, &source, &one, 1);
  }
  int iter = 1;
  double succ = 0;
  float ms = 0.f;
  if (info == GRB_SUCCESS) info = grb_timer_start();
  for (; info == GRB_SUCCESS && iter <= desc->max_niter; ++iter) {
    info = grb_assign(v, f1, GRB_ACCUM_NULL, (double)iter, desc);
    if (info != GRB_SUCCESS) break;
    grb_descriptor_toggle(desc, GRB_MASK);
    info = grb_vxm(f2, v, GRB_ACCUM_NULL, GRB_LOGICAL_OR_AND, f1, A, desc);
    grb_descriptor_toggle(desc, GRB_MASK);
    if (info != GRB_SUCCESS) break;
    info = grb_vector_swap(f2, f1);
    if (info != GRB_SUCCESS) break;
    info = grb_reduce_vector(&succ, GRB_ACCUM_NULL, GRB_PLUS_MONOID, f1, desc);
    if (info != GRB_SUCCESS || succ == 0) break;
  }
  if (info == GRB_SUCCESS) info = grb_timer_stop(&ms);
  if (result) {
    result->levels = iter;
    result->tight_ms = ms;
    result->edges_traversed = 0;
    result->reached = 0;
  }
  grb_vector_free(f1);
  grb_vector_free(f2);
  return info;
}

// ---- raw kernel entry points ------------------------------------------------------
grb_info grb_k_spmv(grb_matrix A, int tran, grb_semiring op, const void* d_u, const void* d_mask, int scmp,
                    int accum, void* d_w) { GRB_API_ENTER();
  if (!A || !A->built) return GRB_UNINITIALIZED_OBJECT;
  const CsrArrays& M = tran ? A->csc : A->csr;
  SpmvPlan& plan = tran ? A->plan_csc : A->plan_csr;
  if (!M.ptr) return GRB_INVALID_OBJECT;
  GRB_TRY(matrix_ensure_plan(A, tran != 0));
  const CsrArrays& other = tran ? A->csr : A->csc;
  return k_spmv(op, A->dtype, M, plan, d_u, d_mask, A->dtype == GRB_F32, scmp, accum, d_w,
                (other.ptr && other.n == plan.nminor && !A->csc_alias) ? other.ptr : nullptr);
}

int grb_spmv_set_bands(int k) { GRB_API_ENTER_NOINFO(); return spmv_bands_setting(k); }
int grb_spmv_set_format(int fmt) { GRB_API_ENTER_NOINFO(); return spmv_format_setting(fmt); }
int grb_spmv_set_reuse_threshold(int launches) { GRB_API_ENTER_NOINFO(); return spmv_reuse_threshold(launches); }

grb_info grb_spmv_format_info(grb_matrix A, int tran, int* in_use, int64_t* groups, int* bands, int* items, int* hub_rows,
                              int* iso, int64_t* bytes_per_launch) { GRB_API_ENTER();
  if (!A || !A->built) return GRB_UNINITIALIZED_OBJECT;
  GRB_TRY(matrix_ensure_plan(A, tran != 0));
  SpmvPlan& plan = tran ? A->plan_csc : A->plan_csr;
  long long g = 0, by = 0;
  int b = 0, it = 0, h = 0, is = 0;
  const int used = k_spmv_cband_info(plan, &g, &b, &it, &h, &is, &by);
  if (in_use) *in_use = used;
  if (groups) *groups = g;
  if (bands) *bands = b;
  if (items) *items = it;
  if (hub_rows) *hub_rows = h;
  if (iso) *iso = is;
  if (bytes_per_launch) *bytes_per_launch = by;
  return GRB_SUCCESS;
}
int grb_sssp_set_nearfar(int mode) { GRB_API_ENTER_NOINFO(); return sssp_nearfar_setting(mode, mode >= -1); }
int grb_sssp_last_order(void) { GRB_API_ENTER_NOINFO(); return sssp_last_order(-1); }
void grb_sssp_last_work(int64_t* out3) { GRB_API_ENTER_NOINFO(); long long w[3]; sssp_last_work(w); if (out3) for (int i = 0; i < 3; ++i) out3[i] = w[i]; }

grb_info grb_spmv_plan_info(grb_matrix A, int tran, int warm, int* bands, int64_t* band_nnz, int64_t* pieces,
                            int* nhot) { GRB_API_ENTER();
  if (!A || !A->built) return GRB_UNINITIALIZED_OBJECT;
  if (!bands || !band_nnz || !pieces || !nhot) return GRB_NULL_POINTER;
  const CsrArrays& M = tran ? A->csc : A->csr;
  SpmvPlan& plan = tran ? A->plan_csc : A->plan_csr;
  if (!M.ptr) return GRB_INVALID_OBJECT;
  GRB_TRY(matrix_ensure_plan(A, tran != 0));
  const CsrArrays& other = tran ? A->csr : A->csc;
  long long bn = 0, pc = 0;
  GRB_TRY(k_spmv_plan_info(M, plan, (other.ptr && other.n == plan.nminor && !A->csc_alias) ? other.ptr : nullptr, warm,
                           bands, &bn, &pc, nhot));
  *band_nnz = bn;
  *pieces = pc;
  return GRB_SUCCESS;
}

int64_t grb_k_spmv_bytes(grb_matrix A, int tran) { GRB_API_ENTER_NOINFO();
  if (!A) return 0;
  const int64_t n = tran ? A->ncols : A->nrows;
  return 8 * (int64_t)A->nvals + 12 * n + 4;
}

}  // extern "C"
