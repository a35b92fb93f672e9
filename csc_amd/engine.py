"""Batched MMP-DF / OMP engine: device residency + launches of libcsc_hip.so.

One ``DeviceBank`` per (dictionary, GPU): the atom bank is materialised ONCE in HBM (fp64 for the
exact re-scoring / least squares, packed fp32 column tiles for the matrix cores) together with the
atom-vs-atom similarity bit-table per (signal length, dissimilarity).  ``mmpdf_batch`` runs the
level-synchronous frontier over all signals of a batch and returns plain NumPy node / record tables.

PyTorch is used for device memory and streams only; every kernel is in csc_amd/csrc/csc_hip.hip.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def path_digits(j, depth, connections):
    """Frontier index j of level `depth` -> 1-based path digits, root first.

    Candidate n = j + 1 of reference src/csc/mmp.py:303-313 (base-`connections` digits of n - 1,
    least significant digit first = root-level child rank first)."""
    out = []
    for _ in range(depth):
        out.append(j % connections + 1)
        j //= connections
    return out


@dataclass
class BatchResult:
    """Host copy of one csc_mmpdf_batch call (layouts: include/csc_hip.h)."""
    max_sparsity: int
    connections: int
    branches: int
    level_sizes: list       # F_1..F_K
    level_offsets: list     # node-id offset of level d (index d-1)
    node_pos: np.ndarray    # [S, NN] int32
    node_atom: np.ndarray   # [S, NN] int32
    node_mse: np.ndarray    # [S, NN] f64
    node_coef: np.ndarray   # [S, NN, K] f64 or None
    rec_mse: np.ndarray     # [S, K] f64
    rec_leaf: np.ndarray    # [S, K, 2] int32 (last-min, first-min)
    rec_sel: np.ndarray     # [S, K, K, 2] int32 (pos, atom)
    stats: np.ndarray       # [4] int32

    def node_id(self, depth, j):
        return self.level_offsets[depth - 1] + j

    def branch(self, s, depth, j):
        """[(pos, atom)] root -> node (depth, j)."""
        out = []
        for e in range(1, depth + 1):
            nid = self.node_id(e, j % (self.connections ** e))
            out.append((int(self.node_pos[s, nid]), int(self.node_atom[s, nid])))
        return out

    def path_str(self, depth, j):
        return "-".join(str(p) for p in path_digits(j, depth, self.connections))


class DeviceBank:
    """Atom bank resident in HBM (reference: the per-call np.array([atom() ...]) rebuilds of
    src/csc/dictionary.py:390,404 and the per-node atom correlations of src/csc/mmp.py:92-93)."""

    def __init__(self, bank64: np.ndarray, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("csc_amd needs a ROCm GPU (MI355X); there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.lib = _lib.lib()
        bank64 = np.ascontiguousarray(bank64, dtype=np.float64)
        self.K, self.L = bank64.shape
        with torch.cuda.device(self.device):
            self.bank64 = torch.from_numpy(bank64).to(self.device)
            n32 = self.lib.csc_bank32_elems(self.K, self.L)
            self.bank32t = torch.empty(n32, dtype=torch.float32, device=self.device)
            _lib.check(self.lib.csc_bank_pack(_ptr(self.bank64), self.K, self.L, _ptr(self.bank32t),
                                              _stream(self.device)), "csc_bank_pack")
        self._sim = {}

    def similarity_table(self, N: int, threshold: float):
        key = (int(N), float(threshold))
        if key not in self._sim:
            W = self.lib.csc_sim_words(N, self.L)
            if W <= 0:
                raise ValueError(f"signal length {N} shorter than atom length {self.L}")
            with torch.cuda.device(self.device):
                bits = torch.empty(self.K * self.K * W, dtype=torch.int32, device=self.device)
                _lib.check(self.lib.csc_similarity_table(_ptr(self.bank64), self.K, self.L, N,
                                                         float(threshold), _ptr(bits),
                                                         _stream(self.device)), "csc_similarity_table")
            self._sim[key] = bits
        return self._sim[key]

    # ---- batched correlation (computeCorrelations) ------------------------------------------------
    def correlate(self, signals) -> torch.Tensor:
        """signals [B, N] (host or device f64) -> device f32 [B, P, K]."""
        sig = self._to_device(signals)
        B, N = sig.shape
        P = N - self.L + 1
        if P <= 0:
            raise ValueError(f"signal length {N} shorter than atom length {self.L}")
        with torch.cuda.device(self.device):
            out = torch.empty((B, P, self.K), dtype=torch.float32, device=self.device)
            _lib.check(self.lib.csc_correlate(_ptr(self.bank32t), self.K, self.L, _ptr(sig), B, N,
                                              _ptr(out), _stream(self.device)), "csc_correlate")
        return out

    def _to_device(self, signals):
        if isinstance(signals, torch.Tensor):
            sig = signals.to(device=self.device, dtype=torch.float64)
        else:
            sig = torch.from_numpy(np.ascontiguousarray(signals, dtype=np.float64)).to(self.device)
        if sig.dim() == 1:
            sig = sig[None]
        return sig.contiguous()

    # ---- batched MMP-DF ------------------------------------------------------------------------------
    def mmpdf_launch(self, sig: torch.Tensor, max_sparsity, connections, branches, dissimilarity,
                     want_coef=True, out=None):
        """Enqueue one csc_mmpdf_batch on the current stream; returns the dict of device tensors."""
        S, N = sig.shape
        sizes, NN = _lib.tree_shape(max_sparsity, connections, branches)
        dev = self.device
        sim = self.similarity_table(N, dissimilarity) if connections > 1 else None
        with torch.cuda.device(dev):
            if out is None:
                ws_bytes = self.lib.csc_mmpdf_workspace_bytes(S, N, max_sparsity, connections, branches)
                out = {
                    "ws": torch.empty(max(ws_bytes, 8) // 8, dtype=torch.float64, device=dev),
                    "node_pos": torch.empty((S, NN), dtype=torch.int32, device=dev),
                    "node_atom": torch.empty((S, NN), dtype=torch.int32, device=dev),
                    "node_mse": torch.empty((S, NN), dtype=torch.float64, device=dev),
                    "node_coef": torch.empty((S, NN, max_sparsity), dtype=torch.float64, device=dev) if want_coef else None,
                    "rec_mse": torch.empty((S, max_sparsity), dtype=torch.float64, device=dev),
                    "rec_leaf": torch.empty((S, max_sparsity, 2), dtype=torch.int32, device=dev),
                    "rec_sel": torch.empty((S, max_sparsity, max_sparsity, 2), dtype=torch.int32, device=dev),
                    "stats": torch.zeros(4, dtype=torch.int32, device=dev),
                }
            rc = self.lib.csc_mmpdf_batch(
                _ptr(self.bank32t), _ptr(self.bank64), _ptr(sim), self.K, self.L, _ptr(sig), S, N,
                max_sparsity, connections, branches, _ptr(out["ws"]), out["ws"].numel() * 8,
                _ptr(out["node_pos"]), _ptr(out["node_atom"]), _ptr(out["node_mse"]), _ptr(out["node_coef"]),
                _ptr(out["rec_mse"]), _ptr(out["rec_leaf"]), _ptr(out["rec_sel"]), _ptr(out["stats"]),
                _stream(dev))
            _lib.check(rc, "csc_mmpdf_batch")
        return out

    def mmpdf_batch(self, signals, max_sparsity, connections, branches, dissimilarity,
                    want_coef=True) -> BatchResult:
        sig = self._to_device(signals)
        out = self.mmpdf_launch(sig, max_sparsity, connections, branches, dissimilarity, want_coef)
        torch.cuda.synchronize(self.device)
        sizes, NN = _lib.tree_shape(max_sparsity, connections, branches)
        offs = [int(x) for x in np.concatenate([[0], np.cumsum(sizes)[:-1]])]
        cpu = {k: (v.cpu().numpy() if v is not None else None) for k, v in out.items() if k != "ws"}
        return BatchResult(max_sparsity, connections, branches, sizes, offs, cpu["node_pos"],
                           cpu["node_atom"], cpu["node_mse"], cpu["node_coef"], cpu["rec_mse"],
                           cpu["rec_leaf"], cpu["rec_sel"], cpu["stats"])
