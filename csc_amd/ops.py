"""PyTorch custom ops over the C ABI (``torch.ops.csc_amd.*``).

BASELINE.json's north star asks that the Python host drive the HIP kernels "through PyTorch-ROCm custom ops":
the ops below are thin ``torch.library`` registrations around libcsc_hip.so's entry points (include/csc_hip.h), so
that the batched solver can sit inside a torch program like any other operator (device tensors in and out, the
current stream, shape inference through the fake implementations).  ``csc_amd.engine.DeviceBank`` calls the same
entry points directly through ctypes; both routes run the same kernels.

    node_pos, node_atom, node_mse, records = torch.ops.csc_amd.mmpdf_batch(
        bank32t, bank64, sim_bits, signals, max_sparsity, connections, branches, flags)

``records`` are the contiguous per-signal sparVar records of CSC_FLAG_RECORDS_AOS ([S, (16 + 8 K) K] uint8: K f64
mse | K x 2 i32 leaf | K x K x 2 i32 (pos, atom)), i.e. what replaces the list returned by the reference's
joblib fan-out (src/csc/dictionary.py:1147-1148).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _tree_nodes(max_sparsity, connections, branches):
    return _lib.tree_shape(max_sparsity, connections, branches)[1]


def _require(t, name, dtype, ndim, device=None, numel=None):
    """The C ABI takes raw pointers: a tensor of the wrong dtype, device or layout would be read as something
    else (a float32 ``signals`` as [S][N] float64 runs past its allocation), so the ops refuse it here."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if t.dim() != ndim:
        raise ValueError(f"{name}: expected {ndim} dimension(s), got shape {tuple(t.shape)}")
    if t.device.type != "cuda":
        raise ValueError(f"{name}: expected a CUDA (ROCm) tensor, got device {t.device}")
    if device is not None and t.device != device:
        raise ValueError(f"{name}: on {t.device}, but the signals are on {device}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    if numel is not None and t.numel() != numel:
        raise ValueError(f"{name}: expected {numel} elements, got {t.numel()}")


@torch.library.custom_op("csc_amd::bank_pack", mutates_args=(), device_types="cuda")
def bank_pack(bank64: torch.Tensor) -> torch.Tensor:
    """[K, L] float64 atom bank -> the packed matrix-core image (csc_bank_pack)."""
    lib = _lib.lib()
    _require(bank64, "bank64", torch.float64, 2)
    K, L = bank64.shape
    b = bank64
    out = torch.empty(lib.csc_bank32_elems(K, L), dtype=torch.float32, device=b.device)
    with torch.cuda.device(b.device):
        _lib.check(lib.csc_bank_pack(_ptr(b), K, L, _ptr(out),
                                     C.c_void_p(torch.cuda.current_stream(b.device).cuda_stream)), "csc_bank_pack")
    return out


@bank_pack.register_fake
def _(bank64):
    K, L = bank64.shape
    return bank64.new_empty(_lib.lib().csc_bank32_elems(K, L), dtype=torch.float32)


@torch.library.custom_op("csc_amd::similarity_table", mutates_args=(), device_types="cuda")
def similarity_table(bank64: torch.Tensor, signal_length: int, threshold: float) -> torch.Tensor:
    """Atom-vs-atom similarity bit table for signals of length N (csc_similarity_table)."""
    lib = _lib.lib()
    _require(bank64, "bank64", torch.float64, 2)
    K, L = bank64.shape
    W = lib.csc_sim_words(signal_length, L)
    if W <= 0:
        raise ValueError(f"signal length {signal_length} shorter than atom length {L}")
    b = bank64
    out = torch.empty(K * K * W, dtype=torch.int32, device=b.device)
    with torch.cuda.device(b.device):
        _lib.check(lib.csc_similarity_table(_ptr(b), K, L, signal_length, float(threshold), _ptr(out),
                                            C.c_void_p(torch.cuda.current_stream(b.device).cuda_stream)),
                   "csc_similarity_table")
    return out


@similarity_table.register_fake
def _(bank64, signal_length, threshold):
    K, L = bank64.shape
    return bank64.new_empty(K * K * max(_lib.lib().csc_sim_words(signal_length, L), 0), dtype=torch.int32)


@torch.library.custom_op("csc_amd::mmpdf_batch", mutates_args=(), device_types="cuda")
def mmpdf_batch(bank32t: torch.Tensor, bank64: torch.Tensor, sim_bits: Optional[torch.Tensor],
                signals: torch.Tensor, max_sparsity: int, connections: int, branches: int,
                flags: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """Batched MMP-DF / OMP on [S, N] float64 signals (csc_mmpdf_batch) -> (node_pos [S, NN] i32,
    node_atom [S, NN] i32, node_mse [S, NN] f64, records [S, (16 + 8 K) K] u8)."""
    lib = _lib.lib()
    if isinstance(signals, torch.Tensor) and signals.dim() == 2 and not signals.is_contiguous():
        signals = signals.contiguous()
    _require(signals, "signals", torch.float64, 2)
    sig = signals
    S, N = sig.shape
    dev = sig.device
    _require(bank64, "bank64", torch.float64, 2, dev)
    K, L = bank64.shape
    _require(bank32t, "bank32t", torch.float32, 1, dev, lib.csc_bank32_elems(K, L))
    if N < L:
        raise ValueError(f"signal length {N} shorter than atom length {L}")
    if sim_bits is not None:
        _require(sim_bits, "sim_bits", torch.int32, 1, dev, K * K * lib.csc_sim_words(N, L))
    NN = _tree_nodes(max_sparsity, connections, branches)
    Kx = max_sparsity
    with torch.cuda.device(dev):
        ws = torch.empty(max(lib.csc_mmpdf_workspace_bytes(max(S, 1), N, Kx, connections, branches), 8) // 8,
                         dtype=torch.float64, device=dev)
        pos = torch.empty((S, NN), dtype=torch.int32, device=dev)
        atom = torch.empty((S, NN), dtype=torch.int32, device=dev)
        mse = torch.empty((S, NN), dtype=torch.float64, device=dev)
        rec = torch.empty((S, Kx * (16 + 8 * Kx)), dtype=torch.uint8, device=dev)
        rc = lib.csc_mmpdf_batch(_ptr(bank32t), _ptr(bank64), _ptr(sim_bits), K, L, _ptr(sig), S, N, Kx, connections,
                                 branches, int(flags) | 4, _ptr(ws), ws.numel() * 8, _ptr(pos), _ptr(atom), _ptr(mse),
                                 None, _ptr(rec), None, None, None,
                                 C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        _lib.check(rc, "csc_mmpdf_batch")
    return pos, atom, mse, rec


@mmpdf_batch.register_fake
def _(bank32t, bank64, sim_bits, signals, max_sparsity, connections, branches, flags):
    S = signals.shape[0]
    NN = _tree_nodes(max_sparsity, connections, branches)
    return (signals.new_empty((S, NN), dtype=torch.int32), signals.new_empty((S, NN), dtype=torch.int32),
            signals.new_empty((S, NN), dtype=torch.float64),
            signals.new_empty((S, max_sparsity * (16 + 8 * max_sparsity)), dtype=torch.uint8))
