"""ctypes binding of libamgcore_hip.so (include/amgcore_hip.h).

The HIP library is the product: there is no CPU fallback.  If the shared
object is missing, import fails loudly; if no GPU is present every compute
call raises ``AmgDeviceError``.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AMGCORE_HIP_LIB") or os.path.join(HERE, "lib", "libamgcore_hip.so")   # env: A/B builds

c_int_p = C.POINTER(C.c_int)
c_dbl_p = C.POINTER(C.c_double)

AMG_OK, AMG_EINVAL, AMG_ENODEV, AMG_ENOMEM, AMG_ESTATE, AMG_ENOTIMPL = 0, -1, -2, -3, -4, -5


class AmgError(RuntimeError):
    pass


class AmgDeviceError(AmgError):
    pass


RELAX_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p)
COARSE_CALLBACK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, c_dbl_p, c_dbl_p)


class SmootherDesc(C.Structure):
    _fields_ = [("kind", C.c_int), ("iterations", C.c_int), ("sweep", C.c_int),
                ("omega", C.c_double), ("ncoef", C.c_int), ("coef", c_dbl_p),
                ("blocksize", C.c_int), ("Dinv", c_dbl_p), ("indices", c_int_p),
                ("nindices", C.c_int),
                ("Sj", c_int_p), ("Sp", c_int_p), ("Tp", c_int_p), ("Tx", c_dbl_p), ("nsdomains", C.c_int)]


def build_library():
    """Compile the HIP sources in-tree (hipcc cross-compiles gfx950 without a GPU)."""
    import subprocess
    subprocess.run(["make", "-s", "-C", os.path.join(HERE, "csrc")], check=True)


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "pyamg_amd: %s is missing -- build it with `make -C pyamg_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
    # When PyTorch is present, load it first: torch ships its own libamdhip64 and both libraries
    # must share ONE HIP runtime in the process (the multi-GPU driver hands torch tensors to the
    # kernels); loaded in the other order torch finds "No HIP GPUs".
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    I, D, V = C.c_int, C.c_double, C.c_void_p
    arr = [c_int_p, I, c_int_p, I, c_dbl_p, I]          # Ap, Aj, Ax with sizes
    xb = [c_dbl_p, I, c_dbl_p, I]                       # x, b with sizes
    sig = {
        "amgcore_gauss_seidel_f64": arr + xb + [I, I, I],
        "amgcore_bsr_gauss_seidel_f64": arr + xb + [I, I, I, I],
        "amgcore_jacobi_f64": arr + xb + [c_dbl_p, I, I, I, I, c_dbl_p, I],
        "amgcore_bsr_jacobi_f64": arr + xb + [c_dbl_p, I, I, I, I, I, c_dbl_p, I],
        "amgcore_gauss_seidel_indexed_f64": arr + xb + [c_int_p, I, I, I, I],
        "amgcore_jacobi_ne_f64": arr + xb + [c_dbl_p, I, c_dbl_p, I, I, I, I, c_dbl_p, I],
        "amgcore_overlapping_schwarz_csr_f64": arr + xb + [c_dbl_p, I, c_int_p, I, c_int_p, I, c_int_p, I,
                                                           I, I, I, I, I],
        "amgcore_gauss_seidel_ne_f64": arr + xb + [I, I, I, c_dbl_p, I, D],
        "amgcore_gauss_seidel_nr_f64": arr + xb + [I, I, I, c_dbl_p, I, D],
        "amgcore_block_jacobi_f64": arr + xb + [c_dbl_p, I, c_dbl_p, I, I, I, I, c_dbl_p, I, I],
        "amgcore_block_gauss_seidel_f64": arr + xb + [c_dbl_p, I, I, I, I, I],
        "amgcore_csr_matvec_f64": [I, I, c_int_p, c_int_p, c_dbl_p, c_dbl_p, c_dbl_p],
        "amgcore_bsr_matvec_f64": [I, I, I, I, c_int_p, c_int_p, c_dbl_p, c_dbl_p, c_dbl_p],
        "amgcore_norm2_f64": [c_dbl_p, C.c_long, c_dbl_p],
        "amg_hier_set_matrix": [V, I, I, I, I, I, I, I, V, V, V, I],
        "amg_hier_set_smoother": [V, I, I, C.POINTER(SmootherDesc)],
        "amg_hier_set_block_matrix": [V, I, I, I, I, c_int_p, c_int_p, c_dbl_p],
        "amg_hier_set_aux_matrix": [V, I, I, I, I, I, c_int_p, c_int_p, c_dbl_p],
        "amg_hier_set_coarse_dense": [V, c_dbl_p, I],
        "amg_hier_set_coarse_smoother": [V, C.POINTER(SmootherDesc)],
        "amg_hier_finalize": [V],
        "amg_hier_solve": [V, V, V, D, I, I, c_dbl_p, c_int_p, I],
        "amg_hier_cycle": [V, V, V, I, I],
        "amg_hier_pcg": [V, V, V, D, I, I, c_dbl_p, c_int_p, c_int_p, I],
        "amg_hier_relax": [V, I, I, c_dbl_p, c_dbl_p],
        "amg_hier_matvec": [V, I, I, c_dbl_p, c_dbl_p],
        "amg_hier_time_spmv": [V, I, I, I, I, c_dbl_p],
        "amg_hier_time_relax": [V, I, I, I, c_dbl_p],
        "amg_mat_apply": [V, I, V, V, V, V, V, D, D, V],
        "amg_mat_apply_rows": [V, I, I, I, V, V, V, V, V, D, D, V],
        "amg_dev_axpy_scaled": [V, V, D, C.c_long, V],
        "amg_mat_build_gs": [V, c_int_p, I],
        "amg_mat_gs_sweep": [V, V, V, I, I, V],
        "amg_mat_gs_sweeps": [V, V, V, V, I, I, V],
        "amg_dev_scale": [V, V, D, C.c_long, V],
        "amg_dev_axpy": [V, V, C.c_long, V],
        "amg_dev_norm2": [V, C.c_long, V, V, V],
        "amg_dev_dot": [V, V, C.c_long, V, V, V],
        "amg_dev_dense_apply": [V, V, V, I, V],
        "amg_dev_gather": [V, V, V, C.c_long, V],
        "amg_hier_set_callback_smoother": [V, I, I, RELAX_CALLBACK, V],
        "amg_hier_set_coarse_callback": [V, COARSE_CALLBACK, V],
        "amg_hier_apply": [V, I, I, V, V],
        "amg_hier_apply_aux": [V, I, I, I, V, V],
        "amg_dev_copy": [V, V, C.c_long, I, V],
        "amg_dev_fill": [V, D, C.c_long, V],
        "amg_dev_axmy": [V, V, D, C.c_long, V],
        "amg_dev_scale_add": [V, D, V, C.c_long, V],
        "amg_dev_sub": [V, V, V, C.c_long, V],
        "amg_dev_divide": [V, D, C.c_long, V],
        "amg_dev_dot_host": [V, V, C.c_long, V, C.POINTER(C.c_double), V],
        "amg_dev_norm_host": [V, C.c_long, V, C.POINTER(C.c_double), V],
        "amg_comm_add_channel": [V, c_int_p],
        "amg_comm_commit": [V, V],
        "amg_comm_connect": [V, V],
        "amg_comm_rccl_unique_id": [C.c_char_p, V],
        "amg_comm_rccl_init": [V, C.c_char_p, V],
        "amg_comm_exchange": [V, I, V, V, V, V],
        "amg_comm_allreduce_sqrt": [V, I, V, V, V],
        "amg_comm_check": [V],
        "amg_hier_set_comm": [V, V, I],
        "amg_hier_set_partition": [V, I, I, I, I, c_int_p, I, I],
        "amg_hier_set_gather": [V, I, I, I],
        "amg_hier_set_coarse_gather": [V, I, I],
        "amg_hier_comm_check": [V],
        "amg_arnoldi": [V, I, c_dbl_p, c_dbl_p, I, D, c_dbl_p, c_int_p, c_int_p],
        "amg_arnoldi_combine": [V, c_dbl_p, I, c_dbl_p],
        "amg_hier_galerkin": [V, I, I, V, V, V, V, V, V, V, C.POINTER(C.c_void_p)],
        "amg_galerkin_fetch": [V, V, V],
        "amg_csr_matmat_device": [I, I, I, V, V, V, V, V, V, V, C.POINTER(C.c_void_p)],
    }
    for name, args in sig.items():
        f = getattr(L, name)
        f.argtypes = args
        f.restype = I
    L.amg_last_error.restype = C.c_char_p
    L.amg_device_count.restype = I
    L.amg_device_name.argtypes = [I]
    L.amg_device_name.restype = C.c_char_p
    L.amg_hier_create.argtypes = [I, I]
    L.amg_hier_create.restype = V
    L.amg_hier_destroy.argtypes = [V]
    L.amg_hier_destroy.restype = None
    L.amg_hier_cycle_bytes.argtypes = [V, I]
    L.amg_hier_cycle_bytes.restype = D
    L.amg_hier_value_index.argtypes = [V, I, I]
    L.amg_hier_value_index.restype = I
    L.amg_set_value_index.argtypes = [I]
    L.amg_set_value_index.restype = None
    L.amg_value_index_enabled.argtypes = []
    L.amg_value_index_enabled.restype = I
    L.amg_hier_gs_natural.argtypes = [V, I, V, V, V, I]
    L.amg_hier_gs_natural.restype = I
    L.amg_hier_operator_form.argtypes = [V, I]
    L.amg_hier_operator_form.restype = I
    L.amg_hier_operator_bytes.argtypes = [V, I, I]
    L.amg_hier_operator_bytes.restype = D
    L.amg_hier_cycle_bytes_moved.argtypes = [V, I]
    L.amg_hier_cycle_bytes_moved.restype = D
    L.amg_hier_last_solve_ms.argtypes = [V]
    L.amg_hier_last_solve_ms.restype = D
    L.amg_hier_device_bytes.argtypes = [V]
    L.amg_hier_device_bytes.restype = C.c_long
    L.amg_hier_stream.argtypes = [V]
    L.amg_hier_stream.restype = V
    L.amg_hier_dev_x.argtypes = [V]
    L.amg_hier_dev_x.restype = V
    L.amg_hier_dev_b.argtypes = [V]
    L.amg_hier_dev_b.restype = V
    L.amg_dev_alloc.argtypes = [C.c_long]
    L.amg_dev_alloc.restype = V
    L.amg_dev_free.argtypes = [V]
    L.amg_dev_free.restype = None
    L.amg_hier_scratch.argtypes = [V]
    L.amg_hier_scratch.restype = V
    L.amg_comm_create.argtypes = [I, I, I, I]
    L.amg_comm_create.restype = V
    L.amg_comm_destroy.argtypes = [V]
    L.amg_comm_destroy.restype = None
    L.amg_hier_release_sources.argtypes = [V]
    L.amg_hier_release_sources.restype = C.c_long
    L.amg_mat_create.argtypes = [I, I, I, c_int_p, c_int_p, c_dbl_p]
    L.amg_mat_create.restype = V
    L.amg_mat_destroy.argtypes = [V]
    L.amg_mat_destroy.restype = None
    L.amg_mat_gs_levels.argtypes = [V]
    L.amg_mat_gs_levels.restype = I
    L.amg_mat_form.argtypes = [V]
    L.amg_mat_form.restype = I
    L.amg_mat_nnz.argtypes = [V]
    L.amg_mat_nnz.restype = C.c_long
    L.amg_arnoldi_free.argtypes = [V]
    L.amg_arnoldi_free.restype = None
    L.amg_set_stream_variant.argtypes = [I]
    L.amg_set_stream_variant.restype = None
    L.amg_set_stream_pipe.argtypes = [I]
    L.amg_set_stream_pipe.restype = None
    L.amg_hier_use_graphs.argtypes = [V, I]
    L.amg_hier_use_graphs.restype = None
    L.amg_hier_keep_residual.argtypes = [V, I]
    L.amg_hier_keep_residual.restype = None
    L.amg_set_tile_target.argtypes = [I]
    L.amg_set_tile_target.restype = None
    L.amg_set_index16.argtypes = [I]
    L.amg_set_index16.restype = None
    L.amg_set_bsr_spmv.argtypes = [I]
    L.amg_set_bsr_spmv.restype = None
    L.amg_set_gs_chain.argtypes = [I]
    L.amg_set_gs_chain.restype = None
    L.amg_set_gs_level_hint.argtypes = [I]
    L.amg_set_gs_level_hint.restype = None
    L.amg_set_gs_flow.argtypes = [I]
    L.amg_set_gs_flow.restype = None
    L.amg_set_gs_flow_lookahead.argtypes = [I]
    L.amg_set_gs_flow_lookahead.restype = None
    L.amg_gs_flow_status.argtypes = []
    L.amg_gs_flow_status.restype = I
    L.amg_set_stencil_form.argtypes = [I]
    L.amg_set_stencil_form.restype = None
    L.amg_set_stencil_pairs.argtypes = [I]
    L.amg_set_stencil_pairs.restype = None
    L.amg_set_sell_form.argtypes = [I]
    L.amg_set_sell_form.restype = None
    L.amg_set_sell_index16.argtypes = [I]
    L.amg_set_sell_index16.restype = None
    L.amg_set_xcd_period.argtypes = [I]
    L.amg_set_xcd_period.restype = None
    L.amg_set_xcd_chunk.argtypes = [I]
    L.amg_set_xcd_chunk.restype = None
    _lib = L
    return L


def check(rc):
    if rc == 0:
        return
    msg = lib().amg_last_error().decode("utf-8", "replace")
    if rc == AMG_ENODEV:
        raise AmgDeviceError(msg)
    if rc == AMG_ENOMEM:
        raise MemoryError(msg)
    if rc == AMG_ENOTIMPL:
        raise NotImplementedError(msg)
    if rc == AMG_EINVAL:
        raise ValueError(msg)
    raise AmgError(msg)


def ip(a):
    return a.ctypes.data_as(c_int_p)


def dp(a):
    return a.ctypes.data_as(c_dbl_p)


def device_count():
    return lib().amg_device_count()
