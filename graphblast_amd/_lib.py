"""ctypes binding of libgrb_hip.so (include/grb_hip.h).

There is deliberately no fallback: if the HIP library is missing or a call fails, this
module raises.  The Python layer is plumbing over the C ABI -- the same binding a
maintainer of the reference would write for any FFI host.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GRB_HIP_LIB") or os.path.join(_HERE, "libgrb_hip.so")

# graphblas::Info names, types.hpp:28-42
INFO_NAMES = ["GrB_SUCCESS", "GrB_UNINITIALIZED_OBJECT", "GrB_NULL_POINTER", "GrB_INVALID_VALUE",
              "GrB_INVALID_INDEX", "GrB_DOMAIN_MISMATCH", "GrB_DIMENSION_MISMATCH",
              "GrB_OUTPUT_NOT_EMPTY", "GrB_NO_VALUE", "GrB_NOT_IMPLEMENTED", "GrB_OUT_OF_MEMORY",
              "GrB_INSUFFICIENT_SPACE", "GrB_INVALID_OBJECT", "GrB_INDEX_OUT_OF_BOUNDS", "GrB_PANIC"]


class GrbError(RuntimeError):
    def __init__(self, info, what):
        self.info = info
        name = INFO_NAMES[info] if 0 <= info < len(INFO_NAMES) else str(info)
        super().__init__("%s returned %s" % (what, name))


class BfsResult(C.Structure):
    _fields_ = [("levels", C.c_int), ("tight_ms", C.c_float), ("edges_traversed", C.c_int64),
                ("reached", C.c_int32)]


class AlgoResult(C.Structure):
    _fields_ = [("iterations", C.c_int), ("tight_ms", C.c_float), ("last_value", C.c_double)]


class TcCoreResult(C.Structure):
    _fields_ = [("core_rows", C.c_int32), ("min_row_length", C.c_int32), ("core_entries", C.c_int64), ("count", C.c_int64),
                ("checksum", C.c_uint64), ("build_ms", C.c_float), ("product_ms", C.c_float), ("tiles", C.c_int32),
                ("tiles_mfma", C.c_int32), ("tiles_by_density", C.c_int32 * 10)]


class TcInfo(C.Structure):
    _fields_ = [("path", C.c_int32), ("prep_ms", C.c_float), ("count_ms", C.c_float), ("longest_list", C.c_int32),
                ("tasks", C.c_int32 * 3)]


class BfsLevel(C.Structure):
    _fields_ = [("direction", C.c_int32), ("frontier", C.c_int32), ("frontier_edges", C.c_int64),
                ("discovered", C.c_int32), ("ms", C.c_float)]


class PartSsspResult(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("rounds", C.c_int32), ("launches", C.c_int32), ("hit_cap", C.c_int32),
                ("ms", C.c_float)]


class PartBfsResult(C.Structure):
    _fields_ = [("levels", C.c_int32), ("launches", C.c_int32), ("hit_cap", C.c_int32),
                ("edges_traversed", C.c_int64), ("reached", C.c_int64), ("ms", C.c_float)]


_vp, _i, _d, _f = C.c_void_p, C.c_int, C.c_double, C.c_float
_ip = C.POINTER(C.c_int)
_SIGS = {
    "grb_set_stream": [_vp],
    "grb_device_info": [C.c_char_p, C.c_size_t],
    "grb_timer_start": [],
    "grb_timer_stop": [C.POINTER(_f)],
    "grb_descriptor_new": [C.POINTER(_vp)],
    "grb_descriptor_free": [_vp],
    "grb_descriptor_set": [_vp, _i, _i],
    "grb_descriptor_get": [_vp, _i, _ip],
    "grb_descriptor_toggle": [_vp, _i],
    "grb_descriptor_load_defaults": [_vp],
    "grb_descriptor_set_arg": [_vp, C.c_char_p, _d],
    "grb_descriptor_get_arg": [_vp, C.c_char_p, C.POINTER(_d)],
    "grb_descriptor_lastmxv": [_vp, _ip],
    "grb_vector_new": [C.POINTER(_vp), _i, _i],
    "grb_vector_free": [_vp],
    "grb_vector_dup": [_vp, _vp],
    "grb_vector_clear": [_vp],
    "grb_vector_size": [_vp, _ip],
    "grb_vector_nvals": [_vp, _ip],
    "grb_vector_build_sparse": [_vp, _vp, _vp, _i],
    "grb_vector_build_dense": [_vp, _vp, _i],
    "grb_vector_adopt_dense": [_vp, _vp, _i],
    "grb_vector_adopt_sparse": [_vp, _vp, _vp, _i],
    "grb_vector_set_element": [_vp, _d, _i],
    "grb_vector_extract_element": [_vp, C.POINTER(_d), _i],
    "grb_vector_extract_tuples_sparse": [_vp, _vp, _vp, _ip],
    "grb_vector_extract_tuples_dense": [_vp, _vp, _ip],
    "grb_vector_fill": [_vp, _d],
    "grb_vector_fill_ascending": [_vp, _i],
    "grb_vector_get_storage": [_vp, _ip],
    "grb_vector_set_storage": [_vp, _i],
    "grb_vector_swap": [_vp, _vp],
    "grb_vector_convert": [_vp, _d, _f, _vp],
    "grb_vector_sparse2dense": [_vp, _d, _vp],
    "grb_vector_dense2sparse": [_vp, _d, _vp],
    "grb_vector_device_ptrs": [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)],
    "grb_matrix_new": [C.POINTER(_vp), _i, _i, _i],
    "grb_matrix_free": [_vp],
    "grb_matrix_build": [_vp, _vp, _vp, _vp, _i],
    "grb_matrix_build_csr": [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp],
    "grb_matrix_adopt_device_csr": [_vp, _vp, _vp, _vp, _i, _vp, _vp, _vp],
    "grb_matrix_ingest_device": [_vp, _vp, _vp, _vp, _i, _i],
    "grb_matrix_eWiseMult_scalar": [_vp, _i, _vp, C.c_double],
    "grb_matrix_eWiseMult_vector": [_vp, _i, _vp, _vp, _vp],
    "grb_matrix_nrows": [_vp, _ip],
    "grb_matrix_ncols": [_vp, _ip],
    "grb_matrix_nvals": [_vp, _ip],
    "grb_matrix_host_csr": [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)],
    "grb_matrix_host_csc": [_vp, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)],
    "grb_matrix_set_values": [_vp, _vp],
    "grb_comm_unique_id": [_vp],
    "grb_comm_init": [_vp, _i, _i],
    "grb_comm_destroy": [],
    "grb_comm_info": [C.POINTER(_i), C.POINTER(_i)],
    "grb_comm_wait": [],
    "grb_comm_allgather": [_vp, _vp, C.c_size_t],
    "grb_comm_allgatherv_inplace": [_vp, _vp, _vp],
    "grb_comm_allreduce_sum_f64": [_vp, C.c_size_t],
    "grb_comm_timing": [_i],
    "grb_comm_stats": [C.POINTER(C.c_double), C.POINTER(C.c_longlong), _i],
    "grb_pr_part_update": [_vp, _vp, _f, _vp, _i, _vp],
    "grb_pr_part_run": [_i, _vp, _vp, _vp, _i, _f, _f, _i, _vp, _vp, _vp, _vp, C.POINTER(_i), C.POINTER(_d),
                        C.POINTER(_i)],
    "grb_semiring_register": [_i, _d, _i, C.POINTER(_i)],
    "grb_spmm": [_i, _vp, _i, _vp, _vp, _i, _vp],
    "grb_spmv_set_bands": [_i],
    "grb_spmv_set_format": [_i],
    "grb_spmv_set_reuse_threshold": [_i],
    "grb_set_lazy": [_i],
    "grb_lazy_pending": [],
    "grb_lazy_fused_reductions": [],
    "grb_vector_apply": [_vp, _vp, _i, _i, _i, C.c_double, _vp, _vp],
    "grb_matrix_apply": [_vp, _vp, _i, _i, _i, C.c_double, _vp, _vp],
    "grb_comm_set_host_transport": [_i, _i, _vp, _vp],
    "grb_cc_set_fused": [_i],
    "grb_spmv_format_info": [_vp, _i, C.POINTER(_i), C.POINTER(C.c_int64), C.POINTER(_i), C.POINTER(_i), C.POINTER(_i),
                             C.POINTER(_i), C.POINTER(C.c_int64)],
    "grb_sssp_set_nearfar": [_i],
    "grb_sssp_last_order": [],
    "grb_sssp_last_work": [_vp],
    "grb_spmv_plan_info": [_vp, _i, _i, C.POINTER(_i), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(_i)],
    "grb_bfs_batch": [_vp, _i, _vp, _vp, _vp, _vp],
    "grb_bfs_batch_set_tail": [C.c_longlong],
    "grb_descriptor_iter_log": [_vp, _vp, _i, C.POINTER(_i)],
    "grb_cache_name": [C.c_char_p, _i, C.c_char_p, C.c_size_t],
    "grb_matrix_write_cache": [_vp, C.c_char_p],
    "grb_matrix_build_cache": [_vp, C.c_char_p],
    "grb_vxm": [_vp, _vp, _i, _i, _vp, _vp, _vp],
    "grb_mxv": [_vp, _vp, _i, _i, _vp, _vp, _vp],
    "grb_eWiseMult": [_vp, _vp, _i, _i, _vp, _vp, _vp],
    "grb_eWiseAdd": [_vp, _vp, _i, _i, _vp, _vp, _vp],
    "grb_eWiseAdd_scalar": [_vp, _vp, _i, _i, _vp, _d, _vp],
    "grb_reduce_vector": [C.POINTER(_d), _i, _i, _vp, _vp],
    "grb_reduce_matrix_rows": [_vp, _vp, _i, _i, _vp, _vp],
    "grb_assign": [_vp, _vp, _i, _d, _vp],
    "grb_bfs": [_vp, _vp, _i, _vp, C.POINTER(BfsResult)],
    "grb_bfs_fused": [_vp, _vp, _i, _vp, C.POINTER(BfsResult), C.POINTER(BfsLevel), _i, _i],
    "grb_bfs_fused_enqueue": [_vp, _vp, _i, _vp, C.POINTER(C.c_int64)],
    "grb_bfs_wait": [C.c_int64, C.POINTER(BfsResult)],
    "grb_bfs_host_times": [C.POINTER(_d), C.POINTER(_d), C.POINTER(C.c_longlong), _i],
    "grb_bfs_set_lanes": [_i],
    "grb_bfs_set_coschedule": [_i],
    "grb_bfs_coschedule_profile": [_i, C.POINTER(C.c_double), C.POINTER(_i), C.POINTER(_i)],
    "grb_bfs_part_pull": [_vp, _i, _i, _vp, _vp, _vp, _f],
    "grb_bfs_part_push": [_vp, _i, _i, _vp, _vp, _vp, _vp, C.POINTER(C.c_int64)],
    "grb_bfs_part_apply": [_vp, _vp, _i, _i, _i, _vp, _f, C.POINTER(C.c_int32)],
    "grb_bfs_part_tally": [_vp, _vp, C.POINTER(C.c_int64), C.POINTER(C.c_int32)],
    "grb_assignScatter": [_vp, _vp, _i, _vp, _vp, _vp],
    "grb_extractGather": [_vp, _vp, _i, _vp, _vp, _vp],
    "grb_cc": [_vp, _vp, _i, _vp, C.POINTER(AlgoResult)],
    "grb_mxm": [_vp, _vp, _i, _i, _vp, _vp, _vp],
    "grb_reduce_matrix_scalar": [C.POINTER(_d), _i, _i, _vp, _vp],
    "grb_matrix_tril": [_vp, _vp, _vp],
    "grb_tc": [C.POINTER(C.c_int64), _vp, _vp, _vp, C.POINTER(AlgoResult)],
    "grb_tc_dense_core": [_vp, _i, _i, _i, C.POINTER(TcCoreResult)],
    "grb_tc_set_product": [_i],
    "grb_tc_release": [_vp],
    "grb_tc_last": [C.POINTER(TcInfo)],
    "grb_sssp": [_vp, _vp, _i, _vp, C.POINTER(AlgoResult)],
    "grb_pr": [_vp, _vp, _f, _f, _vp, C.POINTER(AlgoResult)],
    "grb_k_spmv": [_vp, _i, _i, _vp, _vp, _i, _i, _vp],
    "grb_bfs_part_apply2": [_vp, _vp, _i, _i, _i, _vp, _vp, _vp, _f, C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                            C.POINTER(C.c_int64)],
    "grb_bfs_part_push_small": [_vp, _i, _i, _vp, _vp, _vp],
    "grb_bfs_part_seed": [_vp, _vp, _vp, _i, _i, _i, _i],
    "grb_bitmap_or_parts": [_vp, _i, _i, _vp],
    "grb_bfs_part_unlabel": [_vp, _i, _f],
    "grb_part_new": [C.POINTER(_vp), _i, _i, _i, _i, _vp, _vp, _vp, C.c_int64],
    "grb_part_free": [_vp],
    "grb_part_sssp_new": [C.POINTER(_vp), _i, _i, _i, _i, _vp, _i],
    "grb_part_sssp_free": [_vp],
    "grb_sssp_part_run": [_vp, _i, _i, _i, _vp, C.POINTER(PartSsspResult)],
    "grb_sssp_part_run_group": [C.POINTER(_vp), _i, _i, _i, C.POINTER(_vp), C.POINTER(PartSsspResult)],
    "grb_bfs_part_run": [_vp, _i, _i, _f, _f, _i, _i, _vp, C.POINTER(PartBfsResult), C.POINTER(BfsLevel), _i],
    "grb_bfs_part_run_group": [C.POINTER(_vp), _i, _i, _i, _f, _f, _i, C.POINTER(_vp), C.POINTER(PartBfsResult),
                               C.POINTER(BfsLevel), _i],
    "grb_matrix_load_mtx": [_vp, C.c_char_p, _i, _i, _vp],
    "grb_scatter": [_vp, _vp, _vp, _d, _vp],
    "grb_vector_resize": [_vp, _i],
    "grb_trace_mxm_transpose": [C.POINTER(_d), _i, _vp, _vp, _vp],
    "grb_graph_color": [_vp, _vp, _vp, C.POINTER(_i)],
    "grb_mis": [_vp, _vp, _i, _vp, _vp, C.POINTER(AlgoResult)],
    "grb_gc": [_vp, _vp, _i, _vp, _i, _i, _vp, C.POINTER(AlgoResult)],
    "grb_lgc": [_vp, _vp, _i, _d, _d, _vp, C.POINTER(AlgoResult)],
    "grb_diameter": [_vp, _vp, _i, _i, _vp, C.POINTER(_i), C.POINTER(_i)],
}

_lib = None


def _share_torch_hip_runtime():
    """PyTorch wheels bundle their own libamdhip64.so (same SONAME as /opt/rocm's).  A process
    must run ONE HIP runtime for torch tensors, RCCL buffers and this library to share device
    pointers, and whichever copy is loaded first wins -- so when torch is installed its copy is
    loaded here first (without importing torch); libgrb_hip.so then binds to it."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load():
    """dlopen the library; never silently substitutes anything."""
    global _lib
    if _lib is not None:
        return _lib
    _share_torch_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise ImportError("graphblast_amd: %s not found -- build it with `python -c 'import "
                          "__graft_entry__ as g; g.build()'` (hipcc, gfx950). There is no CPU "
                          "fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, args in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.grb_version.restype = C.c_char_p
    lib.grb_version.argtypes = []
    lib.grb_k_spmv_bytes.restype = C.c_int64
    lib.grb_bfs_batch_set_tail.restype = C.c_longlong
    lib.grb_k_spmv_bytes.argtypes = [_vp, _i]
    _lib = lib
    return lib


def check(info, what):
    if info != 0:
        raise GrbError(info, what)


def call(name, *args):
    info = getattr(load(), name)(*args)
    if info != 0:
        raise GrbError(info, name)
