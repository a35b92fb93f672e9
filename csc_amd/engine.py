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

    _FIELDS = ("node_pos", "node_atom", "node_mse", "node_coef", "rec_mse", "rec_leaf", "rec_sel")

    def __len__(self):
        return self.rec_mse.shape[0]

    @staticmethod
    def field_specs(max_sparsity, connections, branches, want_coef, fields="all"):
        """(name, dtype, per-signal shape) of the fixed-size record of one signal.  ``fields="rec"``: the
        per-sparsity winners only (what the sparVar path needs and what its gather sends: (16 + 8 K) K bytes,
        280 B at K = 5); ``"all"``: the node tables in front of them."""
        sizes, NN = _lib.tree_shape(max_sparsity, connections, branches)
        K = max_sparsity
        specs = []
        if fields == "all":
            specs += [("node_pos", np.int32, (NN,)), ("node_atom", np.int32, (NN,)), ("node_mse", np.float64, (NN,))]
            if want_coef:
                specs.append(("node_coef", np.float64, (NN, K)))
        elif fields != "rec":
            raise ValueError(f"fields must be 'all' or 'rec', not {fields!r}")
        specs += [("rec_mse", np.float64, (K,)), ("rec_leaf", np.int32, (K, 2)), ("rec_sel", np.int32, (K, K, 2))]
        return specs

    @classmethod
    def record_width(cls, max_sparsity, connections, branches, want_coef, fields="all"):
        return sum(int(np.prod(shp)) * np.dtype(dt).itemsize
                   for _, dt, shp in cls.field_specs(max_sparsity, connections, branches, want_coef, fields))

    def to_records(self, fields="all") -> np.ndarray:
        """[S, rec_bytes] uint8: the per-signal fixed-size records that travel in the result gather
        (csc_amd.distributed.gather_rows)."""
        S = len(self)
        specs = self.field_specs(self.max_sparsity, self.connections, self.branches, self.node_coef is not None, fields)
        if S == 0:
            return np.zeros((0, sum(int(np.prod(shp)) * np.dtype(dt).itemsize for _, dt, shp in specs)), np.uint8)
        return np.concatenate([np.ascontiguousarray(getattr(self, n)).reshape(S, -1).view(np.uint8)
                               for n, _, _ in specs], axis=1)

    @classmethod
    def from_records(cls, raw, max_sparsity, connections, branches, want_coef, stats=None, fields="all"):
        sizes, NN = _lib.tree_shape(max_sparsity, connections, branches)
        offs = [int(x) for x in np.concatenate([[0], np.cumsum(sizes)[:-1]])]
        S = raw.shape[0]
        f, at = {"node_pos": None, "node_atom": None, "node_mse": None, "node_coef": None}, 0
        for n, dt, shp in cls.field_specs(max_sparsity, connections, branches, want_coef, fields):
            nb = int(np.prod(shp)) * np.dtype(dt).itemsize
            f[n] = np.ascontiguousarray(raw[:, at:at + nb]).view(dt).reshape((S,) + shp)
            at += nb
        assert at == raw.shape[1], (at, raw.shape)
        return cls(max_sparsity, connections, branches, sizes, offs, f["node_pos"], f["node_atom"],
                   f["node_mse"], f["node_coef"], f["rec_mse"], f["rec_leaf"], f["rec_sel"],
                   np.zeros(4, np.int32) if stats is None else stats)

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
        from . import distributed as D
        self.device = D.local_device() if device is None else torch.device(device)
        self.lib = _lib.lib()
        bank64 = np.ascontiguousarray(bank64, dtype=np.float64)
        self.K, self.L = bank64.shape
        with torch.cuda.device(self.device):
            self.bank64 = torch.from_numpy(bank64).to(self.device)
            n32 = self.lib.csc_bank32_elems(self.K, self.L)
            self.bank32t = torch.empty(n32, dtype=torch.float32, device=self.device)
            _lib.check(self.lib.csc_bank_pack(_ptr(self.bank64), self.K, self.L, _ptr(self.bank32t),
                                              _stream(self.device)), "csc_bank_pack")
            # dictionaries of more than one bank tile: the low-rank section of the image (csc_bank_lowrank) from the leading
            # eigenvectors of the bank's Gram matrix -- any orthonormal basis is valid, the library computes its bounds
            # from the basis it is given; this one is the best of its size.  (Gram product on the device, the small
            # symmetric eigenproblem on the host.)
            self.lowrank = False
            if self.K > _lib.TILE_COLS:
                r = self.lib.csc_lowrank_rank()
                if self.L >= r:
                    gram = (self.bank64.T @ self.bank64).cpu().numpy()
                    _, vec = np.linalg.eigh(gram)                       # ascending eigenvalues
                    basis = np.ascontiguousarray(vec[:, ::-1][:, :r].T)  # [r][L], orthonormal rows
                    self._basis = torch.from_numpy(basis).to(self.device)
                    _lib.check(self.lib.csc_bank_lowrank(_ptr(self.bank32t), _ptr(self.bank64), self.K, self.L,
                                                         _ptr(self._basis), r, _stream(self.device)), "csc_bank_lowrank")
                    self.lowrank = True                              # set to False to screen in full (same results)
        self._sim = {}
        self._simatoms_for = None        # the similarity table whose atom summary sits in the bank image (csc_bank_simatoms)
        self.simatoms = True             # set to False to look up every column's similarity rows in the later siblings' screen (same results)
        # correlations on the fp16 matrix cores as a filter in front of the float64 re-score (same results);
        # set to True to correlate in fp32 only (CSC_FLAG_FP32_ONLY)
        self.fp32_only = False
        # single-child tree levels on the chain ("lean") kernel, four nodes per CU (same results);
        # set to False to keep them on the register-resident kernel (CSC_FLAG_NO_LEAN)
        self.lean = True

    def similarity_table(self, N: int, threshold: float):
        key = (int(N), float(threshold))
        if key not in self._sim:
            W = self.lib.csc_sim_words(N, self.L)
            if W <= 0:
                raise ValueError(f"signal length {N} shorter than atom length {self.L}")
            with torch.cuda.device(self.device):
                bits = torch.empty(self.K * self.K * W, dtype=torch.int32, device=self.device)
                # (the packed-bank entry: fp32 matrix cores + float64 where the fp32 value is too close to the
                # threshold -- the same bits as csc_similarity_table, an order of magnitude sooner on large dictionaries)
                _lib.check(self.lib.csc_similarity_table_packed(_ptr(self.bank32t), _ptr(self.bank64), self.K, self.L, N,
                                                                float(threshold), _ptr(bits),
                                                                _stream(self.device)), "csc_similarity_table_packed")
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

    def correlate64(self, signals) -> torch.Tensor:
        """signals [B, N] -> device f64 [B, P, K]: float64 arithmetic throughout (csc_correlate64)."""
        sig = self._to_device(signals)
        B, N = sig.shape
        P = N - self.L + 1
        if P <= 0:
            raise ValueError(f"signal length {N} shorter than atom length {self.L}")
        with torch.cuda.device(self.device):
            out = torch.empty((B, P, self.K), dtype=torch.float64, device=self.device)
            _lib.check(self.lib.csc_correlate64(_ptr(self.bank64), self.K, self.L, _ptr(sig), B, N, _ptr(out),
                                                _stream(self.device)), "csc_correlate64")
        return out

    def _to_device(self, signals):
        if isinstance(signals, torch.Tensor):
            sig = signals.to(device=self.device, dtype=torch.float64)
        else:
            sig = torch.from_numpy(np.ascontiguousarray(signals, dtype=np.float64)).to(self.device)
        if sig.dim() == 1:
            sig = sig[None]
        return sig.contiguous()

    # ---- batched projection on given supports (getSignalProjectionFromAtoms) -------------------------
    def project_batch(self, signals, supports, sig_index=None, want_approx=False):
        """signals [S, N]; supports [M, kmax, 2] int32 (pos, atom), -1 padded; sig_index [M] or None.
        Returns host (coef [M, kmax], mse [M], approx [M, N] | None, n_duplicate_columns)."""
        sig = self._to_device(signals)
        S, N = sig.shape
        sup = np.ascontiguousarray(supports, dtype=np.int32)
        M, kmax = sup.shape[0], sup.shape[1]
        if sig_index is None and M > S:
            raise ValueError("more problems than signals and no sig_index")
        if sig_index is not None:
            sig_index = np.ascontiguousarray(sig_index, dtype=np.int32)
            if sig_index.shape != (M,) or (M and (sig_index.min() < 0 or sig_index.max() >= S)):
                raise ValueError("sig_index out of range")
        with torch.cuda.device(self.device):
            d_sup = torch.from_numpy(sup).to(self.device)
            d_idx = torch.from_numpy(sig_index).to(self.device) if sig_index is not None else None
            coef = torch.empty((M, kmax), dtype=torch.float64, device=self.device)
            mse = torch.empty((M,), dtype=torch.float64, device=self.device)
            approx = torch.empty((M, N), dtype=torch.float64, device=self.device) if want_approx else None
            stats = torch.zeros(4, dtype=torch.int32, device=self.device)
            _lib.check(self.lib.csc_project_batch(_ptr(self.bank64), self.K, self.L, _ptr(sig), S, N,
                                                  _ptr(d_idx), _ptr(d_sup), M, kmax, _ptr(coef), _ptr(mse),
                                                  _ptr(approx), _ptr(stats), _stream(self.device)),
                       "csc_project_batch")
        return (coef.cpu().numpy(), mse.cpu().numpy(), approx.cpu().numpy() if want_approx else None,
                int(stats[1].item()))

    # ---- batched atom matching (utils.computeMaxTruePositives) -------------------------------------------
    def match_batch(self, truth, approx, pos_err_threshold, corr_err_threshold) -> np.ndarray:
        """truth [M, kt, 2], approx [M, ka, 2] int32 (pos, atom), -1 padded -> true-positive counts [M]."""
        t = np.ascontiguousarray(truth, dtype=np.int32)
        a = np.ascontiguousarray(approx, dtype=np.int32)
        M = t.shape[0]
        if a.shape[0] != M or t.ndim != 3 or a.ndim != 3 or t.shape[2] != 2 or a.shape[2] != 2:
            raise ValueError("truth / approx must be [M, k, 2] with the same M")
        with torch.cuda.device(self.device):
            d_t, d_a = torch.from_numpy(t).to(self.device), torch.from_numpy(a).to(self.device)
            tp = torch.zeros(M, dtype=torch.int32, device=self.device)
            _lib.check(self.lib.csc_match_batch(_ptr(self.bank64), self.K, self.L, _ptr(d_t), t.shape[1], _ptr(d_a),
                                                a.shape[1], M, float(pos_err_threshold), float(corr_err_threshold),
                                                _ptr(tp), _stream(self.device)), "csc_match_batch")
        return tp.cpu().numpy()

    # ---- batched matching pursuit (mp / mpFromDict / mpSparVarFromDict) -----------------------------------
    def mp_batch(self, signals, iterations):
        """signals [S, N] -> host (pos [S, I], atom [S, I], mse [S, I], mask [S, I, I]); row i = state after
        iteration i + 1, mask[s, i, q] = accumulated mask value of the q-th selection (csc_mp_batch)."""
        sig = self._to_device(signals)
        S, N = sig.shape
        if N - self.L + 1 <= 0:
            raise ValueError(f"signal length {N} shorter than atom length {self.L}")
        I = int(iterations)
        dev = self.device
        with torch.cuda.device(dev):
            ws_bytes = self.lib.csc_mmpdf_workspace_bytes(max(S, 1), N, I, 1, 1)
            ws = torch.empty(max(ws_bytes, 8) // 8, dtype=torch.float64, device=dev)
            pos = torch.zeros((S, I), dtype=torch.int32, device=dev)
            atom = torch.zeros((S, I), dtype=torch.int32, device=dev)
            mse = torch.zeros((S, I), dtype=torch.float64, device=dev)
            mask = torch.zeros((S, I, I), dtype=torch.float64, device=dev)
            stats = torch.zeros(4, dtype=torch.int32, device=dev)
            _lib.check(self.lib.csc_mp_batch(_ptr(self.bank32t), _ptr(self.bank64), self.K, self.L, _ptr(sig), S, N,
                                             I, _ptr(ws), ws.numel() * 8, _ptr(pos), _ptr(atom), _ptr(mse),
                                             _ptr(mask), _ptr(stats), _stream(dev)), "csc_mp_batch")
        return pos.cpu().numpy(), atom.cpu().numpy(), mse.cpu().numpy(), mask.cpu().numpy()

    # ---- on-device constrained signal generator (generateConstrainedSignalsDB) ---------------------------------
    def generate(self, sparsities, snrs, signal_length, pos_err_threshold, corr_err_threshold, seed=0, first_index=0):
        """-> (signals [S, N] float64 ON THE DEVICE, truth [S, kmax, 2] int32 host array of (pos, atom), -1 padded).
        csc_generate_signals: Philox-keyed by (seed, first_index + s); raises if a signal could not be completed."""
        sp = np.ascontiguousarray(sparsities, dtype=np.int32)
        sn = np.ascontiguousarray(snrs, dtype=np.float64)
        S, N = len(sp), int(signal_length)
        if sn.shape != (S,):
            raise ValueError("one snr per signal")
        if N - self.L + 1 <= 0:
            raise ValueError(f"signal length {N} shorter than atom length {self.L}")
        kmax = max(1, int(sp.max()) if S else 1)
        if kmax > _lib.MAX_SPARSITY or (S and sp.min() < 0):
            raise ValueError(f"sparsity outside [0, {_lib.MAX_SPARSITY}]")
        dev = self.device
        with torch.cuda.device(dev):
            d_sp, d_sn = torch.from_numpy(sp).to(dev), torch.from_numpy(sn).to(dev)
            sig = torch.empty((S, N), dtype=torch.float64, device=dev)
            truth = torch.empty((S, kmax, 2), dtype=torch.int32, device=dev)
            fail = torch.zeros(max(S, 1), dtype=torch.int32, device=dev)
            _lib.check(self.lib.csc_generate_signals(_ptr(self.bank32t), _ptr(self.bank64), self.K, self.L, N, S,
                                                     _ptr(d_sp), _ptr(d_sn), kmax, float(pos_err_threshold),
                                                     float(corr_err_threshold), int(seed), int(first_index), _ptr(sig),
                                                     _ptr(truth), _ptr(fail), _stream(dev)), "csc_generate_signals")
            if S and int(fail.sum().item()):
                raise RuntimeError("csc_generate_signals: the placement constraints could not be met for "
                                   f"{int(fail.sum().item())} signal(s)")
        return sig, truth.cpu().numpy()

    # ---- batched MMP-DF ------------------------------------------------------------------------------
    def mmpdf_launch(self, sig: torch.Tensor, max_sparsity, connections, branches, dissimilarity,
                     want_coef=True, out=None, static_scratch=False):
        """Enqueue one csc_mmpdf_batch on the current stream; returns the dict of device tensors.
        ``static_scratch``: go through csc_mmpdf_batch_static (no workspace, nothing allocated by the call
        when ``out`` is passed; batches up to the library's static device array)."""
        S, N = sig.shape
        sizes, NN = _lib.tree_shape(max_sparsity, connections, branches)
        dev = self.device
        sim = self.similarity_table(N, dissimilarity) if connections > 1 else None
        with torch.cuda.device(dev):
            if out is None:
                ws_bytes = 0 if static_scratch else self.lib.csc_mmpdf_workspace_bytes(S, N, max_sparsity, connections, branches)
                K = max_sparsity
                # the per-sparsity records: one contiguous record per signal (CSC_FLAG_RECORDS_AOS), exactly the
                # rows the multi-GPU gather sends: K f64 mse | K x 2 i32 leaf | K x K x 2 i32 (pos, atom)
                rec = torch.empty((S, K * (16 + 8 * K)), dtype=torch.uint8, device=dev)
                out = {
                    "ws": torch.empty(max(ws_bytes, 8) // 8, dtype=torch.float64, device=dev),
                    "node_pos": torch.empty((S, NN), dtype=torch.int32, device=dev),
                    "node_atom": torch.empty((S, NN), dtype=torch.int32, device=dev),
                    "node_mse": torch.empty((S, NN), dtype=torch.float64, device=dev),
                    "node_coef": torch.empty((S, NN, K), dtype=torch.float64, device=dev) if want_coef else None,
                    "rec_buf": rec,
                    "rec_mse": rec[:, :8 * K].view(torch.float64),
                    "rec_leaf": rec[:, 8 * K:16 * K].view(torch.int32).unflatten(1, (K, 2)),
                    "rec_sel": rec[:, 16 * K:].view(torch.int32).unflatten(1, (K, K, 2)),
                    "stats": torch.zeros(4, dtype=torch.int32, device=dev),
                }
            flags = ((1 if self.fp32_only else 0) | (0 if self.lean else 2) | 4        # 4 = CSC_FLAG_RECORDS_AOS
                     | (8 if self.lowrank else 0))                                 # 8 = CSC_FLAG_LOWRANK
            if self.lowrank and self.simatoms and sim is not None:
                # the atom summary of THIS table in the bank image (once per table; 16 = CSC_FLAG_SIMATOMS)
                if self._simatoms_for is not sim:
                    _lib.check(self.lib.csc_bank_simatoms(_ptr(self.bank32t), self.K, self.L, _ptr(sim),
                                                          self.lib.csc_sim_words(N, self.L), _stream(dev)), "csc_bank_simatoms")
                    self._simatoms_for = sim
                flags |= 16
            tail = (_ptr(out["node_pos"]), _ptr(out["node_atom"]), _ptr(out["node_mse"]), _ptr(out["node_coef"]),
                    _ptr(out["rec_buf"]), None, None, _ptr(out["stats"]), _stream(dev))
            head = (_ptr(self.bank32t), _ptr(self.bank64), _ptr(sim), self.K, self.L, _ptr(sig), S, N,
                    max_sparsity, connections, branches, flags)
            if static_scratch:
                _lib.check(self.lib.csc_mmpdf_batch_static(*head, *tail), "csc_mmpdf_batch_static")
            else:
                _lib.check(self.lib.csc_mmpdf_batch(*head, _ptr(out["ws"]), out["ws"].numel() * 8, *tail),
                           "csc_mmpdf_batch")
        return out

    @staticmethod
    def pack_records(out, fields="rec", want_coef=False):
        """Device tensors of mmpdf_launch -> [S, width] uint8 per-signal records on the device, in the layout of
        BatchResult.field_specs.  "rec" is the buffer the kernel wrote, as it is; "all" puts the node tables in
        front of it (one device concatenation).  Nothing goes through the host."""
        if fields == "rec":
            return out["rec_buf"]
        S = out["rec_buf"].shape[0]
        names = ["node_pos", "node_atom", "node_mse"] + (["node_coef"] if want_coef else [])
        return torch.cat([out[n].contiguous().view(S, -1).view(torch.uint8) for n in names] + [out["rec_buf"]], dim=1)

    def mmpdf_records(self, signals, max_sparsity, connections, branches, dissimilarity, want_coef=False,
                      fields="rec"):
        """signals [S, N] (host array or device tensor) -> ([S, width] uint8 records ON THE DEVICE, stats
        tensor [4] int32 on the device).  What a rank hands to the result gather."""
        sig = self._to_device(signals)
        out = self.mmpdf_launch(sig, max_sparsity, connections, branches, dissimilarity,
                                want_coef=want_coef and fields == "all")
        with torch.cuda.device(self.device):
            return self.pack_records(out, fields, want_coef), out["stats"]

    def filter_product(self, signals, variant=0):
        """Test aid (csc_filter_product): the fp16 filter product of every signal, [B, P, K] f32 in units of
        ||signal||, the bound eps [B] the kernels assume for it and ||signal|| [B]; host arrays."""
        sig = self._to_device(signals)
        B, N = sig.shape
        P = N - self.L + 1
        with torch.cuda.device(self.device):
            prod = torch.full((B, max(P, 0), self.K), float("nan"), dtype=torch.float32, device=self.device)
            eps = torch.zeros(B, dtype=torch.float32, device=self.device)
            nrm = torch.zeros(B, dtype=torch.float64, device=self.device)
            _lib.check(self.lib.csc_filter_product(_ptr(self.bank32t), self.K, self.L, _ptr(sig), B, N, int(variant),
                                                   _ptr(prod), _ptr(eps), _ptr(nrm), _stream(self.device)),
                       "csc_filter_product")
        return prod.cpu().numpy(), eps.cpu().numpy(), nrm.cpu().numpy()

    def mmpdf_batch(self, signals, max_sparsity, connections, branches, dissimilarity,
                    want_coef=True) -> BatchResult:
        sig = self._to_device(signals)
        out = self.mmpdf_launch(sig, max_sparsity, connections, branches, dissimilarity, want_coef)
        torch.cuda.synchronize(self.device)
        sizes, NN = _lib.tree_shape(max_sparsity, connections, branches)
        offs = [int(x) for x in np.concatenate([[0], np.cumsum(sizes)[:-1]])]
        cpu = {k: (v.cpu().numpy() if v is not None else None) for k, v in out.items()
               if k not in ("ws", "rec_buf")}
        return BatchResult(max_sparsity, connections, branches, sizes, offs, cpu["node_pos"],
                           cpu["node_atom"], cpu["node_mse"], cpu["node_coef"], cpu["rec_mse"],
                           cpu["rec_leaf"], cpu["rec_sel"], cpu["stats"])
