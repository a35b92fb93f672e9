"""ctypes binding of libcsc_hip.so (include/csc_hip.h).  There is no CPU fallback: every compute
entry point of csc_amd goes through this library and raises if it is missing."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libcsc_hip.so")

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
        L = C.CDLL(LIB_PATH)
        vp, i32, sz, dbl = C.c_void_p, C.c_int, C.c_size_t, C.c_double
        sig = {
            "csc_abi_version": (i32, []),
            "csc_last_error": (C.c_char_p, []),
            "csc_bank32_elems": (sz, [i32, i32]),
            "csc_sim_words": (i32, [i32, i32]),
            "csc_bank_pack": (i32, [vp, i32, i32, vp, vp]),
            "csc_similarity_table": (i32, [vp, i32, i32, i32, dbl, vp, vp]),
            "csc_correlate": (i32, [vp, i32, i32, vp, i32, i32, vp, vp]),
            "csc_tree_nodes": (i32, [i32, i32, i32, C.POINTER(i32)]),
            "csc_mmpdf_workspace_bytes": (sz, [i32, i32, i32, i32, i32]),
            "csc_mmpdf_batch": (i32, [vp, vp, vp, i32, i32, vp, i32, i32, i32, i32, i32, vp, sz,
                                      vp, vp, vp, vp, vp, vp, vp, vp, vp]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


EXPORTS = ("csc_abi_version", "csc_last_error", "csc_bank32_elems", "csc_sim_words", "csc_bank_pack",
           "csc_similarity_table", "csc_correlate", "csc_tree_nodes", "csc_mmpdf_workspace_bytes",
           "csc_mmpdf_batch")


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
