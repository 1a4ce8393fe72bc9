"""ctypes binding of the C ABI in include/mispmm.h (libmispmm.so, HIP / gfx950).

There is no CPU fallback: if the shared library is missing or a call fails, this
module raises.  Build the library with `make -C cuda-optimization-for-spmm_amd`
or `python -c "import __graft_entry__ as g; g.build()"`.
"""
import ctypes
import os

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("MISPMM_LIB") or os.path.join(PKG_DIR, "libmispmm.so")   # MISPMM_LIB: A/B builds

OK = 0
ERR_INVALID_ARG, ERR_UNSUPPORTED, ERR_HIP, ERR_NO_DEVICE, ERR_ALLOC = -1, -2, -3, -4, -5
ACC_REFERENCE, ACC_FAST = 0, 1
ACC_MODES = {"reference": ACC_REFERENCE, "fast": ACC_FAST}
H2H, H2D, D2H, D2D = 0, 1, 2, 3
GATHER_NONE, GATHER_TO_FIRST, GATHER_ALL_PEER, GATHER_ALL_RCCL, GATHER_ALL_RCCL_EQUAL = 0, 1, 2, 3, 4

_c = ctypes
_vp, _u32, _i, _sz = _c.c_void_p, _c.c_uint32, _c.c_int, _c.c_size_t
_pvp = _c.POINTER(_c.c_void_p)

# name -> (restype, argtypes); every symbol include/mispmm.h declares
SIGNATURES = {
    "mispmm_version": (_i, []),
    "mispmm_status_string": (_c.c_char_p, [_i]),
    "mispmm_last_error": (_c.c_char_p, []),
    "mispmm_last_kernel": (_c.c_char_p, []),
    "mispmm_device_count": (_i, [_c.POINTER(_i)]),
    "mispmm_set_device": (_i, [_i]),
    "mispmm_get_device": (_i, [_c.POINTER(_i)]),
    "mispmm_device_info": (_i, [_i, _c.c_char_p, _c.POINTER(_i), _c.POINTER(_sz)]),
    "mispmm_device_bus_id": (_i, [_i, _c.c_char_p, _i]),
    "mispmm_malloc": (_i, [_pvp, _sz]),
    "mispmm_free": (_i, [_vp]),
    "mispmm_host_alloc": (_i, [_pvp, _sz]),
    "mispmm_host_free": (_i, [_vp]),
    "mispmm_memcpy": (_i, [_vp, _vp, _sz, _i]),
    "mispmm_memcpy_async": (_i, [_vp, _vp, _sz, _i, _vp]),
    "mispmm_memset_async": (_i, [_vp, _i, _sz, _vp]),
    "mispmm_stream_create": (_i, [_pvp]),
    "mispmm_stream_destroy": (_i, [_vp]),
    "mispmm_stream_sync": (_i, [_vp]),
    "mispmm_device_sync": (_i, []),
    "mispmm_event_create": (_i, [_pvp]),
    "mispmm_event_destroy": (_i, [_vp]),
    "mispmm_event_record": (_i, [_vp, _vp]),
    "mispmm_event_sync": (_i, [_vp]),
    "mispmm_event_elapsed_ms": (_i, [_vp, _vp, _c.POINTER(_c.c_float)]),
    "mispmm_graph_begin": (_i, [_vp]),
    "mispmm_graph_end": (_i, [_vp, _pvp]),
    "mispmm_graph_launch": (_i, [_vp, _vp]),
    "mispmm_graph_destroy": (_i, [_vp]),
    "mispmm_csr_f32": (_i, [_vp, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _u32, _u32, _vp, _u32, _i, _i]),
    "mispmm_csr_uniform_f32": (_i, [_vp, _u32, _u32, _u32, _vp, _vp, _vp, _u32, _u32, _vp, _u32, _i]),
    "mispmm_csr_batch_f32": (_i, [_vp, _u32, _u32, _u32, _vp, _vp, _vp, _u32, _u32, _pvp, _u32, _u32, _pvp, _u32, _i]),
    "mispmm_csr_autotune_plan_f32": (_i, [_vp, _u32, _u32, _u32, _vp, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _u32, _i, _u32, _c.POINTER(_i),
                                          _c.POINTER(_c.c_float)]),
    "mispmm_autotune_pick": (_i, [_c.POINTER(_c.c_float), _u32, _c.c_float]),
    "mispmm_csr_tiles_host": (_i, [_u32, _u32, _vp, _vp, _u32, _u32, _c.POINTER(_u32), _c.POINTER(_u32), _vp, _vp, _vp, _vp, _vp]),
    "mispmm_csr_lds_tile_f32": (_i, [_vp, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _vp, _u32, _i]),
    "mispmm_csr_split_f32": (_i, [_vp, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _u32, _vp, _u32, _u32, _vp, _u32, _i]),
    "mispmm_csr_spans_by_length_host": (_i, [_u32, _vp, _u32, _c.POINTER(_u32), _vp]),
    "mispmm_csr_hybrid_f32": (_i, [_vp, _u32, _u32, _u32, _vp, _vp, _vp, _u32, _u32, _vp, _u32, _u32, _vp, _u32, _i]),
    "mispmm_csr_spans_long_count_host": (_i, [_u32, _vp, _u32, _c.POINTER(_u32)]),
    "mispmm_rows_hybrid_f32": (_i, [_vp, _u32, _u32, _u32, _vp, _vp, _vp, _u32, _u32, _vp, _u32, _u32, _vp, _u32, _i]),
    "mispmm_csr_cluster_rows_host": (_i, [_u32, _u32, _vp, _vp, _u32, _vp, _c.POINTER(_c.c_uint64), _c.POINTER(_c.c_uint64)]),
    "mispmm_csr_permute_rows_host": (_i, [_u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "mispmm_csr_plan_f32": (_i, [_vp, _u32, _u32, _u32, _vp, _vp, _vp, _u32, _vp, _u32, _pvp, _u32, _u32, _pvp, _u32, _i]),
    "mispmm_ell_f32": (_i, [_vp, _u32, _u32, _u32, _vp, _vp, _vp, _u32, _u32, _vp, _u32, _i, _i]),
    "mispmm_ell_colmajor_to_rowmajor_host": (_i, [_u32, _u32, _u32, _vp, _vp, _c.POINTER(_u32), _vp, _vp]),
    "mispmm_bsr_f32": (_i, [_vp, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _u32, _u32, _vp, _u32, _i, _i]),
    "mispmm_bsr_nonzeros_host": (_i, [_u32, _u32, _u32, _u32, _vp, _vp, _vp, _c.POINTER(_u32), _vp, _vp, _vp]),
    "mispmm_bsr_nonzeros_f32": (_i, [_vp, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _u32, _u32, _vp, _u32, _i]),
    "mispmm_rows_split_f32": (_i, [_vp, _u32, _u32, _u32, _vp, _vp, _vp, _u32, _vp, _u32, _u32, _vp, _u32, _i]),
    "mispmm_ell_compact_host": (_i, [_u32, _u32, _vp, _vp, _c.POINTER(_u32), _vp, _vp, _vp]),
    "mispmm_ell_compact_f32": (_i, [_vp, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _u32, _u32, _vp, _u32, _i]),
    "mispmm_bsr_bf16": (_i, [_vp, _u32, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _u32, _u32, _vp, _u32, _i]),
    "mispmm_bsr_compact_bf16_host": (_i, [_u32, _u32, _u32, _u32, _vp, _vp, _vp, _c.POINTER(_u32), _vp, _vp, _vp]),
    "mispmm_bsrc_bf16": (_i, [_vp, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _u32, _u32, _vp, _u32, _i]),
    "mispmm_bsr_compact_slots_bf16_host": (_i, [_u32, _u32, _u32, _u32, _vp, _vp, _vp, _c.POINTER(_u32), _c.POINTER(_u32), _vp, _vp, _vp]),
    "mispmm_bsrc_slots_bf16": (_i, [_vp, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _u32, _u32, _vp, _u32, _i]),
    "mispmm_coo_f32": (_i, [_vp, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _u32, _u32, _vp, _u32, _vp, _i, _i]),
    "mispmm_coo_row_bounds": (_i, [_vp, _u32, _u32, _vp, _vp]),
    "mispmm_coo_sort_by_row_host": (_i, [_u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _c.POINTER(_i)]),
    "mispmm_vendor_spmm_f32": (_i, [_vp, _i, _u32, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _u32, _u32, _vp, _u32,
                                    _c.POINTER(_c.c_double), _c.POINTER(_c.c_double), _c.POINTER(_c.c_double)]),
    "mispmm_dense_transpose_f32": (_i, [_vp, _u32, _u32, _vp, _vp]),
    "mispmm_f32_to_bf16": (_i, [_vp, _sz, _vp, _vp]),
    "mispmm_bf16_to_f32": (_i, [_vp, _sz, _vp, _vp]),
    "mispmm_shard_rows_by_nnz_host": (_i, [_u32, _vp, _u32, _vp]),
    "mispmm_enable_peer_access": (_i, [_u32, _c.POINTER(_i)]),
    "mispmm_comm_create": (_i, [_pvp, _u32, _c.POINTER(_i)]),
    "mispmm_comm_destroy": (_i, [_vp]),
    "mispmm_multi_csr_f32": (_i, [_u32, _c.POINTER(_i), _pvp, _c.POINTER(_u32), _u32, _pvp, _pvp, _pvp, _c.POINTER(_u32),
                                  _c.POINTER(_u32), _pvp, _u32, _u32, _pvp, _u32, _i, _i, _i, _vp]),
    "mispmm_multi_ell_f32": (_i, [_u32, _c.POINTER(_i), _pvp, _c.POINTER(_u32), _u32, _u32, _pvp, _pvp, _pvp, _u32, _u32, _pvp, _u32,
                                  _i, _i, _i, _vp]),
    "mispmm_multi_bsrc_slots_bf16": (_i, [_u32, _c.POINTER(_i), _pvp, _c.POINTER(_u32), _u32, _c.POINTER(_u32), _pvp, _pvp, _pvp, _pvp,
                                          _u32, _u32, _pvp, _u32, _i, _i, _vp]),
    "mispmm_slab_scatter": (_i, [_vp, _vp, _sz, _pvp, _u32]),
}

_lib = None


class MispmmError(RuntimeError):
    def __init__(self, status, detail):
        super().__init__(f"libmispmm: status {status}: {detail}")
        self.status = status


def lib():
    """Load libmispmm.so (once) and attach the signatures.  Raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise FileNotFoundError(
                f"{LIB_PATH} is not built -- run `make -C {PKG_DIR}`; mispmm has no CPU fallback")
        # The PyTorch wheel ships a HIP runtime of its own (torch/lib/libamdhip64.so); libmispmm.so links the system one
        # (/opt/rocm/lib/libamdhip64.so.7).  Same soname: whichever is loaded first serves both.  Loaded in THIS order --
        # torch, then the library -- both use torch's runtime and share its device context; the other way round torch finds
        # "no ROCm-capable device" (seen with `python __graft_entry__.py smoke`, whose build() loads the library first).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        handle = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(handle, name)          # AttributeError if the export is missing
            fn.restype, fn.argtypes = restype, argtypes
        _lib = handle
    return _lib


def check(status):
    if status != OK:
        l = lib()
        detail = l.mispmm_last_error().decode() or l.mispmm_status_string(status).decode()
        raise MispmmError(status, detail)


def device_count():
    n = _i(0)
    check(lib().mispmm_device_count(ctypes.byref(n)))
    return n.value


def device_info(ordinal=0):
    name = ctypes.create_string_buffer(256)
    cus, mem = _i(0), _sz(0)
    check(lib().mispmm_device_info(ordinal, name, ctypes.byref(cus), ctypes.byref(mem)))
    return {"name": name.value.decode(), "cu_count": cus.value, "hbm_bytes": mem.value}


def device_bus_id(ordinal=0):
    buf = ctypes.create_string_buffer(64)
    check(lib().mispmm_device_bus_id(ordinal, buf, 64))
    return buf.value.decode()


def last_kernel():
    """Tag of the device kernel the calling thread's last compute call enqueued."""
    return lib().mispmm_last_kernel().decode()
