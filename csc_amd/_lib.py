"""ctypes binding of libcsc_hip.so (include/csc_hip.h).  There is no CPU fallback: every compute
entry point of csc_amd goes through this library and raises if it is missing."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CSC_HIP_LIB") or os.path.join(_HERE, "csrc", "libcsc_hip.so")   # env: dev builds only

TILE_COLS = 144
TILE_ROWS = 256
MAX_SPARSITY = 16
MAX_CONNECTIONS = 8

_lib = None


class CscHipError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make -C csc_amd/csrc` "
                "(or `python -c 'import __graft_entry__ as g; g.build()'`). csc_amd has no CPU fallback.")
        # torch first: it carries its own HIP runtime (torch/lib/libamdhip64.so); loading ours before it
        # would bind libcsc_hip.so to a second runtime under /opt/rocm that shares no device state
        # (streams, allocations) with the tensors we are handed -> "no ROCm-capable device".
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        vp, i32, sz, dbl = C.c_void_p, C.c_int, C.c_size_t, C.c_double
        sig = {
            "csc_abi_version": (i32, []),
            "csc_last_error": (C.c_char_p, []),
            "csc_bank32_elems": (sz, [i32, i32]),
            "csc_sim_words": (i32, [i32, i32]),
            "csc_bank_pack": (i32, [vp, i32, i32, vp, vp]),
            "csc_similarity_table": (i32, [vp, i32, i32, i32, dbl, vp, vp]),
            "csc_similarity_table_packed": (i32, [vp, vp, i32, i32, i32, dbl, vp, vp]),
            "csc_lowrank_rank": (i32, []),
            "csc_bank_lowrank": (i32, [vp, vp, i32, i32, vp, i32, vp]),
            "csc_bank_simatoms": (i32, [vp, i32, i32, vp, i32, vp]),
            "csc_correlate": (i32, [vp, i32, i32, vp, i32, i32, vp, vp]),
            "csc_correlate64": (i32, [vp, i32, i32, vp, i32, i32, vp, vp]),
            "csc_tree_nodes": (i32, [i32, i32, i32, C.POINTER(i32)]),
            "csc_mmpdf_workspace_bytes": (sz, [i32, i32, i32, i32, i32]),
            "csc_mmpdf_batch": (i32, [vp, vp, vp, i32, i32, vp, i32, i32, i32, i32, i32, i32, vp, sz,
                                      vp, vp, vp, vp, vp, vp, vp, vp, vp]),
            "csc_mmpdf_batch_static": (i32, [vp, vp, vp, i32, i32, vp, i32, i32, i32, i32, i32, i32,
                                             vp, vp, vp, vp, vp, vp, vp, vp, vp]),
            "csc_static_scratch_bytes": (sz, []),
            "csc_filter_product": (i32, [vp, i32, i32, vp, i32, i32, i32, vp, vp, vp, vp]),
            "csc_mp_batch": (i32, [vp, vp, i32, i32, vp, i32, i32, i32, vp, sz, vp, vp, vp, vp, vp, vp]),
            "csc_match_batch": (i32, [vp, i32, i32, vp, i32, vp, i32, i32, dbl, dbl, vp, vp]),
            "csc_timer_create": (vp, [i32]),
            "csc_timer_destroy": (None, [vp]),
            "csc_timer_attach": (None, [vp]),
            "csc_timer_reset": (None, [vp]),
            "csc_timer_count": (i32, [vp]),
            "csc_timer_read": (i32, [vp, C.POINTER(C.c_float), C.POINTER(i32), i32]),
            "csc_generate_signals": (i32, [vp, vp, i32, i32, i32, i32, vp, vp, i32, dbl, dbl, C.c_uint64, C.c_uint64,
                                           vp, vp, vp, vp]),
            "csc_sparvar_json_write": (i32, [C.c_char_p, i32, vp, C.c_int64, i32, i32, vp, vp, vp, dbl, vp, vp, vp, i32,
                                             i32, i32]),
            "csc_format_double": (i32, [dbl, C.c_char_p, i32]),
            "csc_json_last_error": (C.c_char_p, []),
            "csc_project_batch": (i32, [vp, i32, i32, vp, i32, i32, vp, vp, i32, i32, vp, vp, vp, vp, vp]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


EXPORTS = ("csc_abi_version", "csc_last_error", "csc_bank32_elems", "csc_sim_words", "csc_bank_pack",
           "csc_similarity_table", "csc_similarity_table_packed", "csc_lowrank_rank", "csc_bank_lowrank", "csc_bank_simatoms", "csc_correlate", "csc_correlate64", "csc_tree_nodes", "csc_mmpdf_workspace_bytes",
           "csc_mmpdf_batch", "csc_mmpdf_batch_static", "csc_static_scratch_bytes", "csc_filter_product",
           "csc_mp_batch", "csc_project_batch", "csc_match_batch", "csc_timer_create", "csc_timer_destroy", "csc_timer_attach",
           "csc_timer_reset", "csc_timer_count", "csc_timer_read", "csc_sparvar_json_write", "csc_format_double", "csc_generate_signals",
           "csc_json_last_error")


def check(rc, what):
    if rc != 0:
        msg = lib().csc_last_error().decode()
        raise CscHipError(f"{what} failed (rc={rc}): {msg}")


def tree_shape(max_sparsity, connections, branches):
    """(level sizes F_1..F_K, total nodes) of runMMPDF -- see csc_tree_nodes in include/csc_hip.h."""
    arr = (C.c_int * max(1, max_sparsity))()
    nn = lib().csc_tree_nodes(max_sparsity, connections, branches, arr)
    if nn < 0:
        check(nn, "csc_tree_nodes")
    return list(arr[:max_sparsity]), nn


class LevelTimer:
    """HIP-event spans around every level launch of csc_mmpdf_batch (include/csc_hip.h, csc_timer)."""

    def __init__(self, capacity):
        self._h = lib().csc_timer_create(int(capacity))
        if not self._h:
            raise CscHipError("csc_timer_create failed")
        self.capacity = int(capacity)

    def __enter__(self):
        lib().csc_timer_reset(self._h)
        lib().csc_timer_attach(self._h)
        return self

    def __exit__(self, *exc):
        lib().csc_timer_attach(None)

    def read(self):
        """[(level, milliseconds)] -- call after the stream has been synchronised."""
        n = lib().csc_timer_count(self._h)
        ms, tag = (C.c_float * max(1, n))(), (C.c_int * max(1, n))()
        got = lib().csc_timer_read(self._h, ms, tag, n)
        if got < 0 or got != n:
            check(got if got else -1, "csc_timer_read")
        return [(int(tag[i]), float(ms[i])) for i in range(n)]

    def close(self):
        if self._h:
            lib().csc_timer_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
