"""``CSCDictionary`` / ``ZSDictionary``: the reference's dictionary API on the MI355X engine.

Same class name(s), method names, argument names / defaults and result dictionaries as the
reference ``src/csc/dictionary.py`` (cited per method), so a caller of the reference's OMP / MMP-DF
path can switch packages.  What changed underneath:

* the atom bank is materialised once in HBM (``DeviceBank``), not rebuilt per call;
* ``omp*`` / ``mmpdf*`` run on the GPU through ``libcsc_hip.so`` -- there is no CPU solver in this
  package (a missing library or GPU raises);
* the ``*PipelineFromDB`` methods replace the ``joblib.Parallel`` fan-out by ONE batched device call
  (``nb_cores`` is accepted and ignored) and, under ``torch.distributed``, shard the signals over
  ranks and gather the records once (csc_amd.distributed).

``'delay'`` fields are wall-clock in the reference; here every record of a batched call carries the
amortised per-signal time of that call (documented in DESIGN.md, excluded from parity).
"""
from __future__ import annotations

import itertools
import json
import time
from functools import partial
from typing import Dict, List, Tuple

import numpy as np

from .atoms import CSCAtom
from .utils import get_today_date_str, handle_non_serializable


def time_decorator(func):
    """Optional per-method wall-clock log (reference src/csc/dictionary.py:25-45)."""
    def wrapper(self, *args, **kwargs):
        if not self.timing_enabled:
            return func(self, *args, **kwargs)
        t0 = time.time()
        out = func(self, *args, **kwargs)
        dt = time.time() - t0
        self.performance_log.setdefault(func.__name__, []).append(dt)
        if kwargs.get("verbose", False):
            print(f"Temps d'exécution de {func.__name__}: {dt:.6f} secondes.")
        return out
    wrapper.__name__ = func.__name__
    return wrapper


class CSCDictionary:
    # maximum signals per device call of the pipelines (bounds host/device staging memory)
    PIPELINE_CHUNK = 32768

    def __init__(self, atoms_list: List[CSCAtom]):
        # pad every atom to the longest support (reference :49-56)
        self.atoms = atoms_list
        self.atoms_length = max(len(a) for a in self.atoms)
        for a in self.atoms:
            a.padBothSides(self.atoms_length)
        assert all(len(a) == self.atoms_length for a in self.atoms), "All atoms must have the same length"
        self.timing_enabled = False
        self.performance_log = {}
        self._device_bank = None
        self._by_shape = None

    @classmethod
    def from_values(cls, b_values, y_values, s_values) -> "CSCDictionary":
        """One atom per (b, y, s) in itertools.product order, b-major (reference :62-80)."""
        return cls([CSCAtom(b, y, s) for b, y, s in itertools.product(b_values, y_values, s_values)])

    def __str__(self) -> str:
        return "CSCMeasurementMatrix object with {} different atoms".format(len(self))

    def __len__(self) -> int:
        return len(self.atoms)

    def __contains__(self, item) -> bool:
        if isinstance(item, tuple):
            return any(a.params["b"] == item[0] and a.params["y"] == item[1] and a.params["sigma"] == item[2]
                       for a in self.atoms)
        if isinstance(item, CSCAtom):
            return (item.params["b"], item.params["y"], item.params["sigma"]) in self
        raise TypeError("Unsupported type for 'in' operation")

    def __getstate__(self):
        state = self.__dict__.copy()
        state["_device_bank"] = None          # device handles do not pickle (joblib callers)
        state["_by_shape"] = None
        return state

    def enable_timing(self):
        self.timing_enabled = True

    def disable_timing(self):
        self.timing_enabled = False

    def getAtom(self, idx: int) -> CSCAtom:
        return self.atoms[idx]

    def getAtoms(self) -> List[CSCAtom]:
        return self.atoms

    def getLocalDictionary(self) -> np.ndarray:
        return np.array([a() for a in self.atoms])

    def getAtomsLength(self) -> int:
        return self.atoms_length

    def getAtomFromParams(self, b, y, s, tolerance=1e-5):
        for a in self.atoms:
            if np.isclose(a.b, b, atol=tolerance) and np.isclose(a.y, y, atol=tolerance) and np.isclose(a.sigma, s, atol=tolerance):
                return a
        return None

    # ------------------------------------------------------------------------------------------
    # device residency
    # ------------------------------------------------------------------------------------------
    def device_bank(self):
        """The HBM-resident atom bank (built on first use on this process's GPU: the current device, or
        cuda:LOCAL_RANK under an initialised multi-rank process group -- csc_amd.distributed.local_device)."""
        import torch
        from . import distributed as D
        from .engine import DeviceBank
        dev = D.local_device() if torch.cuda.is_available() else None
        if self._device_bank is None or self._device_bank.device != dev:
            self._device_bank = DeviceBank(self.getLocalDictionary(), device=dev)
        return self._device_bank

    def _info(self, pos, atom_idx) -> dict:
        a = self.atoms[atom_idx]
        return {"x": int(pos), "b": a.params["b"], "y": a.params["y"], "s": a.params["sigma"]}

    # ------------------------------------------------------------------------------------------
    # convolutional operator (reference :380-484)
    # ------------------------------------------------------------------------------------------
    def computeCorrelations(self, signal, fast: bool = False):
        """(P, K) valid cross-correlations of ``signal`` with every atom (reference :395-407).  Float64
        arithmetic on the GPU by default (the reference's accuracy: callers such as the rmatvec of
        getMaskedConvOperator rely on it); ``fast=True`` takes them on the fp32 matrix cores instead
        (csc_correlate: fp32 accuracy, returned as float64)."""
        sig = np.asarray(signal, dtype=np.float64)[None]
        if fast:
            return self.device_bank().correlate(sig)[0].double().cpu().numpy()
        return self.device_bank().correlate64(sig)[0].cpu().numpy()

    def computeConvolutions(self, activations):
        """(N,) synthesis from a (P*K,) activation vector (reference :380-393).  Activations are
        sparse on this path (<= max sparsity non-zeros), so only the active columns are placed."""
        act = np.reshape(np.asarray(activations, dtype=np.float64), (-1, len(self.atoms)))
        P = act.shape[0]
        out = np.zeros(P + self.atoms_length - 1)
        for p, k in zip(*np.nonzero(act)):
            out[p:p + self.atoms_length] += act[p, k] * self.atoms[k]()
        return out

    def getMaskedConvOperator(self, activation_mask):
        """conv o diag(mask) as a scipy LinearOperator (reference :409-425), kept for callers that
        build their own solvers; the engine itself never forms it."""
        from scipy.sparse.linalg import LinearOperator
        A = activation_mask.shape[0]
        P = A // len(self.atoms)
        N = P + self.atoms_length - 1
        masked = lambda x: x * activation_mask
        mask_op = LinearOperator(shape=(A, A), matvec=masked, rmatvec=masked)
        conv_op = LinearOperator(shape=(N, A), matvec=partial(self.computeConvolutions),
                                 rmatvec=lambda r: self.computeCorrelations(r))
        return conv_op @ mask_op

    def getActivationIdxFromAtom(self, atom: CSCAtom, offset: int) -> int:
        return offset * len(self.atoms) + self.atoms.index(atom)

    def getActivationIdxFromParams(self, params_dict: dict) -> int:
        return self.getActivationIdxFromAtom(CSCAtom.from_dict(params_dict), params_dict["x"])

    def _atom_index(self, params_dict: dict) -> int:
        """Index of the first atom with the same (b, y) -- what ``self.atoms.index(atom)`` finds with
        CSCAtom.__eq__ (reference :436, atoms.py:79-85) -- without rebuilding the atom."""
        if getattr(self, "_by_shape", None) is None or len(self._by_shape_src) != len(self.atoms):
            self._by_shape = {}
            for i, a in enumerate(self.atoms):
                self._by_shape.setdefault((a.params["b"], a.params["y"]), i)
            self._by_shape_src = list(self.atoms)
        key = (params_dict["b"], params_dict["y"])
        if key not in self._by_shape:
            raise ValueError(f"atom b={key[0]}, y={key[1]} is not in the dictionary")
        return self._by_shape[key]

    def _supports_array(self, atoms_lists, P):
        kmax = max(1, max((len(a) for a in atoms_lists), default=1))
        sup = np.full((len(atoms_lists), kmax, 2), -1, dtype=np.int32)
        for m, infos in enumerate(atoms_lists):
            for q, info in enumerate(infos):
                x = int(info["x"])
                if not 0 <= x < P:
                    raise IndexError(f"atom position {x} outside [0, {P})")
                sup[m, q] = (x, self._atom_index(info))
        return sup

    def getSignalProjectionFromAtoms(self, signal: np.ndarray, atoms_dicts: List[dict]) -> Tuple[np.ndarray, np.ndarray]:
        """(approx, activations): least squares of ``signal`` on the placed atoms (reference :449-473;
        lsqr on the masked operator there, fp64 Gram / Cholesky on the GPU here).  ``activations`` is
        the dense (P*K,) vector of the reference with the coefficients at pos*K + atom."""
        signal = np.asarray(signal, dtype=np.float64)
        P = len(signal) - self.atoms_length + 1
        acts = np.zeros(P * len(self.atoms))
        if len(atoms_dicts) == 0:
            return np.zeros(len(signal)), acts
        sup = self._supports_array([atoms_dicts], P)
        coef, _, approx, _ = self.device_bank().project_batch(signal[None], sup, want_approx=True)
        for (x, k), c in zip(sup[0], coef[0]):
            acts[x * len(self.atoms) + k] += c
        return approx[0], acts

    def projectionMSEBatch(self, signal: np.ndarray, atoms_lists: List[List[dict]]) -> List[float]:
        """mean((signal - projection)^2) for several candidate supports of ONE signal in one device
        call (the loop of reference mmp.py:649-667)."""
        signal = np.asarray(signal, dtype=np.float64)
        if len(atoms_lists) == 0:
            return []
        P = len(signal) - self.atoms_length + 1
        sup = self._supports_array(atoms_lists, P)
        _, mse, _, _ = self.device_bank().project_batch(signal[None], sup,
                                                        sig_index=np.zeros(len(atoms_lists), np.int32))
        return [float(m) for m in mse]

    def maxTruePositivesBatch(self, true_atoms_lists: List[List[dict]], approx_atoms_lists: List[List[dict]],
                              pos_err_threshold: float, corr_err_threshold: float) -> np.ndarray:
        """Max number of true positives of every (truth, recovered) pair -- the first return value of
        utils.computeMaxTruePositives (reference utils.py:23-69) -- for a whole result set in one
        device call."""
        if len(true_atoms_lists) != len(approx_atoms_lists):
            raise ValueError("one recovered list per truth list")
        if len(true_atoms_lists) == 0:
            return np.zeros(0, np.int32)

        def pack(lists):
            k = max(1, max(len(x) for x in lists))
            arr = np.full((len(lists), k, 2), -1, dtype=np.int32)
            for m, infos in enumerate(lists):
                for q, info in enumerate(infos):
                    arr[m, q] = (int(info["x"]), self._atom_index(info))
            return arr
        return self.device_bank().match_batch(pack(true_atoms_lists), pack(approx_atoms_lists),
                                              pos_err_threshold, corr_err_threshold)

    def getAtomFromActivationIdx(self, activation_idx: int) -> Tuple[CSCAtom, int]:
        return self.atoms[activation_idx % len(self.atoms)], activation_idx // len(self.atoms)

    def correlationFromDicts(self, atom1_dict: dict, atom2_dict: dict, signal_length: int) -> float:
        """Inner product of two placed atoms given by parameter dicts (reference :166-185)."""
        sigs = []
        for d in (atom1_dict, atom2_dict):
            a = CSCAtom.from_dict(d)
            a.padBothSides(self.atoms_length)
            sigs.append(a.getAtomInSignal(signal_length, offset=d["x"]))
        (c,) = np.correlate(sigs[0], sigs[1], mode="valid")
        return c

    # ------------------------------------------------------------------------------------------
    # synthetic signals (reference :187-365) -- host side, global NumPy RNG like the reference
    # ------------------------------------------------------------------------------------------
    def _noisy(self, clean, snr_level):
        active = clean[np.argwhere(np.abs(clean) > CSCAtom.SUPPORT_THRESHOLD)]
        noise_var = np.var(active) / (10 ** (snr_level / 10))
        return clean + np.random.normal(0, np.sqrt(noise_var), len(clean))

    def generateTestSignal(self, signal_length, sparsity_level, snr_level):
        """Random unit-amplitude atoms + white noise at ``snr_level`` dB (reference :187-222)."""
        assert sparsity_level <= len(self.atoms), "The sparsity level must be less than or equal to the number of atoms in the dictionary"
        assert signal_length >= self.atoms_length, "The signal length must be greater than or equal to the maximum atom length"
        ks = np.random.choice(len(self.atoms), sparsity_level)
        offs = np.random.randint(0, signal_length - self.atoms_length + 1, sparsity_level)
        clean, infos = np.zeros(signal_length), []
        for k, off in zip(ks, offs):
            clean += self.atoms[k].getAtomInSignal(signal_length, off)
            infos.append({"x": off, "b": self.atoms[k].params["b"], "y": self.atoms[k].params["y"],
                          "s": self.atoms[k].params["sigma"]})
        return self._noisy(clean, snr_level), infos

    def generateConstrainedTestSignal(self, signal_length: int, sparsity_level: int, snr_level: int,
                                      pos_err_threshold: int, corr_err_threshold: float):
        """As above with rejection of atoms closer than ``pos_err_threshold`` samples to, or more
        correlated than ``corr_err_threshold`` with, an atom already placed (reference :271-324)."""
        assert sparsity_level <= len(self.atoms), "The sparsity level must be less than or equal to the number of atoms in the dictionary"
        assert signal_length >= self.atoms_length, "The signal length must be greater than or equal to the maximum atom length"
        clean = np.zeros(signal_length)
        kept, infos = [], []                       # kept: (position, placed signal)
        while len(infos) < sparsity_level:
            k = np.random.choice(len(self.atoms))
            pos = np.random.randint(0, signal_length - self.atoms_length + 1)
            cand = self.atoms[k].getAtomInSignal(signal_length, pos)
            rejected = False
            for opos, osig in kept:
                ncorr = np.correlate(cand, osig, mode="valid") / (np.linalg.norm(cand) * np.linalg.norm(osig))
                if np.abs(pos - opos) < pos_err_threshold or np.max(ncorr) > corr_err_threshold:
                    rejected = True
                    break
            if rejected:
                continue
            clean += cand
            kept.append((pos, cand))
            infos.append({"x": pos, "b": self.atoms[k].params["b"], "y": self.atoms[k].params["y"],
                          "s": self.atoms[k].params["sigma"]})
        return self._noisy(clean, snr_level), infos

    def _signals_db(self, make, batch_size, signal_length, sparsity_levels, snr_levels, extra, output_filename):
        db = {"date": get_today_date_str(), "batchSize": batch_size, "snrLevels": snr_levels,
              "signalLength": signal_length, "sparsityLevels": sparsity_levels}
        db.update(extra)
        db["dictionary"] = str(self)
        signals, idx = [], 0
        for snr, sparsity in itertools.product(snr_levels, sparsity_levels):
            for _ in range(batch_size):
                sig, infos = make(signal_length, sparsity, snr)
                signals.append({"id": idx, "snr": snr, "sparsity": sparsity, "signal": sig.tolist(), "atoms": infos})
                idx += 1
        db["signals"] = signals
        json.dump(db, open(output_filename, "w"), indent=4, default=handle_non_serializable)
        print(f"Signals database saved in {output_filename}")

    def generateSignalsDB(self, batch_size: int, signal_length: int, sparsity_levels: List[int],
                          snr_levels: List[float], output_filename: str) -> None:
        """reference :224-261"""
        self._signals_db(self.generateTestSignal, batch_size, signal_length, sparsity_levels, snr_levels, {},
                         output_filename)

    def generateConstrainedSignalsDB(self, batch_size: int, signal_length: int, sparsity_levels: List[int],
                                     snr_levels: List[float], pos_err_threshold: float,
                                     corr_err_threshold: float, output_filename: str) -> None:
        """reference :326-365"""
        make = lambda n, sp, snr: self.generateConstrainedTestSignal(n, sp, snr, pos_err_threshold, corr_err_threshold)
        self._signals_db(make, batch_size, signal_length, sparsity_levels, snr_levels,
                         {"posErrThreshold": pos_err_threshold, "corrErrThreshold": corr_err_threshold},
                         output_filename)

    def generateConstrainedSignalsDevice(self, batch_size: int, signal_length: int, sparsity_levels: List[int],
                                         snr_levels: List[float], pos_err_threshold: float,
                                         corr_err_threshold: float, seed: int = 0, first_index: int = 0):
        """generateConstrainedSignalsDB's batch (reference :326-365: ``batch_size`` signals per (snr, sparsity) in
        itertools.product(snr_levels, sparsity_levels) order, ids consecutive) generated ON THE GPU and left
        there: returns {'signals': float64 tensor [S, N] on the device, 'ids', 'snr', 'sparsity': arrays [S],
        'truth': int32 [S, kmax, 2] (pos, atom) in placement order, -1 padded}.  Same placement constraints and
        noise rule as generateConstrainedTestSignal (:271-324); the random stream is Philox (seed, signal id), not
        NumPy's global generator, so signals are not the ones the host generator would draw."""
        snr, sp = [], []
        for v, k in itertools.product(snr_levels, sparsity_levels):
            snr += [v] * batch_size
            sp += [k] * batch_size
        sig, truth = self.device_bank().generate(sp, snr, signal_length, pos_err_threshold, corr_err_threshold,
                                                 seed=seed, first_index=first_index)
        return {"signals": sig, "ids": np.arange(first_index, first_index + len(sp)), "snr": np.asarray(snr),
                "sparsity": np.asarray(sp, dtype=np.int32), "truth": truth}

    def truthInfos(self, truth_row) -> List[dict]:
        """[(pos, atom)] of the device generator -> the reference's atom dicts ({'x','b','y','s'})."""
        return [self._info(p, k) for p, k in truth_row if p >= 0]

    def atomsDictPositionMatchingErrors(self, atoms_info: List[dict], recov_atoms_info: List[dict]) -> List[int]:
        """reference :367-378"""
        err = [[np.abs(t["x"] - r["x"]) for r in recov_atoms_info] for t in atoms_info]
        return [recov_atoms_info[np.argmin(err[i])]["x"] - atoms_info[i]["x"] for i in range(len(atoms_info))]

    # ------------------------------------------------------------------------------------------
    # batched engine front-ends
    # ------------------------------------------------------------------------------------------
    def _records(self, signals, max_sparsity, connections, branches, dissimilarity, want_coef=False,
                 fields="rec", sharded=False, presharded=False, shard_sizes=None, recv=None, async_gather=False):
        """The device-level core of every batched front-end: signals [S, N] (host array or device tensor) ->
        ([S, width] uint8 per-signal records ON THE DEVICE, stats tensor), layout BatchResult.field_specs.

        ``sharded=True`` (the pipelines): under torch.distributed every rank solves its contiguous slice of
        the batch on its own GPU (chunks of PIPELINE_CHUNK signals) and the records are gathered ONCE,
        device to device, on rank 0; the other ranks get (None, stats).  Replaces joblib.Parallel
        (reference :1147).  ``presharded=True``: ``signals`` already is this rank's shard (``shard_sizes``:
        every rank's shard length; found with one all_gather when not given).  ``async_gather=True``: the
        gather is only enqueued and a ``distributed.PendingRows`` comes back in place of the rows (``wait()``)."""
        import torch
        from . import distributed as D
        from .engine import BatchResult
        assert branches > 0, "branches_number must be greater than 0"     # reference mmp.py:409
        bank = self.device_bank()
        if not isinstance(signals, torch.Tensor):
            signals = np.ascontiguousarray(signals, dtype=np.float64)
        if signals.ndim == 1:
            signals = signals[None]
        n = len(signals)
        dist_on = sharded and D.is_distributed()
        if dist_on and presharded:
            lo, hi = 0, n
            if shard_sizes is None:
                shard_sizes = [None] * D.world_size()
                torch.distributed.all_gather_object(shard_sizes, n)
        elif dist_on:
            lo, hi = D.shard_range(n, D.rank(), D.world_size())
            shard_sizes = D.shard_sizes(n, D.world_size())
        else:
            lo, hi = 0, n
        parts, stats = [], None
        for c0 in range(lo, hi, self.PIPELINE_CHUNK):
            rows, st = bank.mmpdf_records(signals[c0:min(hi, c0 + self.PIPELINE_CHUNK)], max_sparsity, connections,
                                          branches, dissimilarity, want_coef=want_coef, fields=fields)
            parts.append(rows)
            stats = st.clone() if stats is None else stats + st
        if not parts:                                    # an empty shard (fewer signals than ranks) still joins the gather
            width = BatchResult.record_width(max_sparsity, connections, branches, want_coef, fields)
            parts = [torch.zeros((0, width), dtype=torch.uint8, device=getattr(bank, "device", "cpu"))]
            stats = torch.zeros(4, dtype=torch.int32)
        local = parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)
        if not dist_on:
            return (D.PendingRows(None, [local], None, [local.shape[0]], True) if async_gather else local), stats
        if torch.distributed.get_backend() != "nccl":
            local = local.cpu()
        return D.gather_rows(local, shard_sizes, out=recv, async_op=async_gather), stats

    def _batch(self, signals, max_sparsity, connections, branches, dissimilarity, want_coef=False,
               sharded=False, fields="all"):
        """signals [S, N] -> (BatchResult on the host, amortised seconds per signal); (None, dt) on ranks != 0
        of a sharded call.  ``fields="rec"`` keeps the node tables on the device and brings back only the
        per-sparsity winners (the sparVar path)."""
        from . import distributed as D
        from .engine import BatchResult
        if not hasattr(signals, "device"):
            signals = np.ascontiguousarray(signals, dtype=np.float64)
        if signals.ndim == 1:
            signals = signals[None]
        assert branches > 0, "branches_number must be greater than 0"     # reference mmp.py:409
        S = len(signals)
        t0 = time.time()
        if not (sharded and D.is_distributed()) and S <= self.PIPELINE_CHUNK and fields == "all":
            res = self.device_bank().mmpdf_batch(signals, max_sparsity, connections, branches, dissimilarity,
                                                 want_coef=want_coef)
            return res, (time.time() - t0) / max(1, S)
        raw, stats = self._records(signals, max_sparsity, connections, branches, dissimilarity, want_coef, fields,
                                   sharded)
        dt = (time.time() - t0) / max(1, S)
        if raw is None:
            return None, dt
        return BatchResult.from_records(raw.cpu().numpy(), max_sparsity, connections, branches, want_coef,
                                        stats.cpu().numpy(), fields), dt

    def _sparvar_records(self, res, s, delay):
        """[{'mse','path','atoms','delay'}] for k = 1..max_sparsity (reference mmp.py:577-606)."""
        out = []
        for k in range(1, res.max_sparsity + 1):
            leaf = int(res.rec_leaf[s, k - 1, 0])
            sel = res.rec_sel[s, k - 1, :k]
            out.append({"mse": float(res.rec_mse[s, k - 1]), "path": res.path_str(k, leaf),
                        "atoms": [self._info(p, a) for p, a in sel], "delay": delay})
        return out

    def _omp_sorted_infos(self, support, coef):
        """Atoms by DESCENDING coefficient, zeros dropped: np.where on the flattened activations,
        then argsort()[::-1] (reference :534-550)."""
        K = len(self.atoms)
        flat = np.array([p * K + k for p, k in support], dtype=np.int64)
        o = np.argsort(flat)
        flat, vals = flat[o], np.asarray(coef, dtype=np.float64)[o]
        nz = vals != 0
        flat, vals = flat[nz], vals[nz]
        order = np.argsort(vals)[::-1]
        return [self._info(i // K, i % K) for i in flat[order]]

    def _approx(self, support, coef, N):
        out = np.zeros(N)
        for (p, k), c in zip(support, coef):
            out[p:p + self.atoms_length] += c * self.atoms[k]()
        return out

    # ------------------------------------------------------------------------------------------
    # OMP (reference :492-756) = MMP-DF with one connection and one branch
    # ------------------------------------------------------------------------------------------
    @time_decorator
    def omp(self, signal: np.ndarray, sparsity_level: int, verbose: bool = False) -> Tuple[np.ndarray, List[dict]]:
        """(approx, infos) of convolutional OMP (reference :492-552)."""
        signal = np.asarray(signal, dtype=np.float64)
        res, _ = self._batch(signal, sparsity_level, 1, 1, 1.0, want_coef=True)
        support = res.branch(0, sparsity_level, 0)
        coef = res.node_coef[0, res.node_id(sparsity_level, 0), :sparsity_level]
        return self._approx(support, coef, len(signal)), self._omp_sorted_infos(support, coef)

    def _omp_result(self, signal_dict, res, s, N):
        k = len(signal_dict["atoms"])
        support = res.branch(s, k, 0)
        coef = res.node_coef[s, res.node_id(k, 0), :k]
        infos = self._omp_sorted_infos(support, coef)
        return {"id": signal_dict["id"], "mse": float(res.node_mse[s, res.node_id(k, 0)]),
                "sparsity": len(infos), "approx": self._approx(support, coef, N).tolist(), "atoms": infos}

    def ompFromDict(self, signal_dict: dict, verbose: bool = False) -> dict:
        """{'id','mse','sparsity','approx','atoms'}, sparsity = number of true atoms (reference :554-572)."""
        signal = np.asarray(signal_dict["signal"], dtype=np.float64)
        res, _ = self._batch(signal, len(signal_dict["atoms"]), 1, 1, 1.0, want_coef=True)
        return self._omp_result(signal_dict, res, 0, len(signal))

    def _omp_sparvar_result(self, signal_dict, res, s, delay):
        out = []
        for k in range(1, res.max_sparsity + 1):
            nid = res.node_id(k, 0)
            support = res.branch(s, k, 0)
            out.append({"mse": float(res.node_mse[s, nid]),
                        "atoms": self._omp_sorted_infos(support, res.node_coef[s, nid, :k]),
                        "delay": delay * k / res.max_sparsity})
        return {"id": signal_dict["id"], "snr": signal_dict["snr"], "results": out}

    def ompSparVarFromDict(self, signal_dict: dict, max_sparsity: int, verbose: bool = False) -> dict:
        """One record per OMP iteration 1..max_sparsity (reference :574-649)."""
        res, dt = self._batch(signal_dict["signal"], max_sparsity, 1, 1, 1.0, want_coef=True)
        return self._omp_sparvar_result(signal_dict, res, 0, dt)

    # ------------------------------------------------------------------------------------------
    # plain matching pursuit (reference :764-996)
    # ------------------------------------------------------------------------------------------
    def _mp_batch(self, signals, iterations, sharded=False):
        """-> (pos, atom, mse, mask) host arrays of csc_mp_batch, or None on ranks != 0 when sharded."""
        from . import distributed as D
        signals = np.ascontiguousarray(signals, dtype=np.float64)
        if signals.ndim == 1:
            signals = signals[None]
        S, I = len(signals), int(iterations)
        bank = self.device_bank()
        lo, hi = D.shard_range(S, D.rank(), D.world_size()) if sharded else (0, S)
        parts = [bank.mp_batch(signals[c0:min(hi, c0 + self.PIPELINE_CHUNK)], I)
                 for c0 in range(lo, hi, self.PIPELINE_CHUNK)]
        shapes, dts = [(I,), (I,), (I,), (I, I)], [np.int32, np.int32, np.float64, np.float64]
        arrs = [np.concatenate([p[f] for p in parts]) if parts else np.zeros((0,) + shp, dt)
                for f, (shp, dt) in enumerate(zip(shapes, dts))]
        if not (sharded and D.is_distributed()):
            return arrs
        # explicit per-field widths: reshape(0, -1) of an empty shard (fewer signals than ranks) is an error
        raw = np.concatenate([np.ascontiguousarray(a).view(np.uint8).reshape(hi - lo, int(np.prod(shp)) * np.dtype(dt).itemsize)
                              for a, shp, dt in zip(arrs, shapes, dts)], axis=1)
        raw = D.gather_records(raw, S)
        if raw is None:
            return None
        out, at = [], 0
        for shp, dt in zip(shapes, dts):
            nb = int(np.prod(shp)) * np.dtype(dt).itemsize
            out.append(np.ascontiguousarray(raw[:, at:at + nb]).view(dt).reshape((S,) + shp))
            at += nb
        return out

    def _mp_state(self, pos, atom, mask, N):
        """(approx, infos) from one iteration's state: approx = sum mask^2 atoms (the reference applies
        the mask to itself, :803); atoms by descending mask value over np.where order (:808-823)."""
        K = len(self.atoms)
        flat = pos.astype(np.int64) * K + atom
        acts = {}
        for f, m in zip(flat, mask):
            if m != 0:
                acts[int(f)] = acts.get(int(f), 0.0) + float(m)
        idx = np.array(sorted(acts), dtype=np.int64)
        vals = np.array([acts[int(i)] for i in idx])
        approx = np.zeros(N)
        for i, v in zip(idx, vals):
            approx[i // K:i // K + self.atoms_length] += v * v * self.atoms[i % K]()
        order = np.argsort(vals)[::-1]
        return approx, [self._info(i // K, i % K) for i in idx[order]]

    @time_decorator
    def mp(self, signal: np.ndarray, sparsity_level: int, verbose: bool = False) -> Tuple[np.ndarray, List[dict]]:
        """(approx, infos) of convolutional matching pursuit (reference :764-825)."""
        signal = np.asarray(signal, dtype=np.float64)
        pos, atom, _, mask = self._mp_batch(signal, sparsity_level)
        k = sparsity_level
        return self._mp_state(pos[0, :k], atom[0, :k], mask[0, k - 1, :k], len(signal))

    def _mp_result(self, sd, arrs, row, N):
        pos, atom, mse, mask = arrs
        k = len(sd["atoms"])
        approx, infos = self._mp_state(pos[row, :k], atom[row, :k], mask[row, k - 1, :k], N)
        return {"id": sd["id"], "mse": float(mse[row, k - 1]), "sparsity": len(infos), "approx": approx.tolist(),
                "atoms": infos}

    def _mp_sparvar_result(self, sd, arrs, row, N, delay):
        pos, atom, mse, mask = arrs
        I = pos.shape[1]
        out = [{"mse": float(mse[row, k - 1]),
                "atoms": self._mp_state(pos[row, :k], atom[row, :k], mask[row, k - 1, :k], N)[1],
                "delay": delay * k / I} for k in range(1, I + 1)]
        return {"id": sd["id"], "snr": sd["snr"], "results": out}

    def mpFromDict(self, signal_dict: dict, verbose: bool = False) -> dict:
        """{'id','mse','sparsity','approx','atoms'} (reference :827-845)."""
        signal = np.asarray(signal_dict["signal"], dtype=np.float64)
        return self._mp_result(signal_dict, self._mp_batch(signal, len(signal_dict["atoms"])), 0, len(signal))

    def mpSparVarFromDict(self, signal_dict: dict, max_sparsity: int, verbose: bool = False) -> dict:
        """One record per MP iteration 1..max_sparsity (reference :847-921)."""
        signal = np.asarray(signal_dict["signal"], dtype=np.float64)
        t0 = time.time()
        arrs = self._mp_batch(signal, max_sparsity)
        return self._mp_sparvar_result(signal_dict, arrs, 0, len(signal), time.time() - t0)

    def _mp_pipeline(self, input_filename, output_filename, key, iterations, build, extra, verbose):
        data = self._load_db(input_filename)
        signals = data["signals"]
        groups = {}
        for i, sd in enumerate(signals):
            groups.setdefault(key(sd), []).append(i)
        out = [None] * len(signals)
        for k, idxs in groups.items():
            t0 = time.time()
            arrs = self._mp_batch(np.array([signals[i]["signal"] for i in idxs], dtype=np.float64), iterations(k),
                                  sharded=True)
            if arrs is None:
                out = None
                continue
            dt = (time.time() - t0) / max(1, len(idxs))
            for row, i in enumerate(idxs):
                out[i] = build(signals[i], arrs, row, k, dt)
        if out is None:
            return
        results = {"source": input_filename, "date": get_today_date_str(), "algorithm": "Convolutional MP",
                   "batchSize": data["batchSize"], "snrLevels": data["snrLevels"]}
        results.update(extra)
        results.update({"signalLength": data["signalLength"], "sparsityLevels": data["sparsityLevels"],
                        "dictionary": str(self), "mp": out})
        self._dump(results, output_filename)
        if verbose:
            print(f"MP Pipeline results saved in {output_filename}")

    def mpPipelineFromDB(self, input_filename: str, output_filename: str, nb_cores: int, verbose=False):
        """reference :923-957"""
        self._mp_pipeline(input_filename, output_filename, lambda sd: (len(sd["signal"]), len(sd["atoms"])),
                          lambda k: k[1], lambda sd, arrs, row, k, dt: self._mp_result(sd, arrs, row, k[0]), {},
                          verbose)

    def mpSparVarPipelineFromDB(self, input_filename: str, output_filename: str, nb_cores: int, max_sparsity: int,
                                verbose=False):
        """reference :959-996"""
        self._mp_pipeline(input_filename, output_filename, lambda sd: len(sd["signal"]), lambda k: max_sparsity,
                          lambda sd, arrs, row, k, dt: self._mp_sparvar_result(sd, arrs, row, k, dt),
                          {"maxSparsityLevel": max_sparsity}, verbose)

    # ------------------------------------------------------------------------------------------
    # MMP-DF (reference :1012-1152)
    # ------------------------------------------------------------------------------------------
    def mmpdf(self, signal: np.ndarray, sparsity_level: int, connections_level: int, dissimilarity: float,
              nb_branches: int, verbose: bool = False) -> Tuple[np.ndarray, List[dict]]:
        """(approx, infos) of the FIRST minimum-MSE leaf; approx is the sum of the UNIT atoms of the
        branch, LS amplitudes are not applied (reference :1012-1024, mmp.py:248-255, 458-467)."""
        signal = np.asarray(signal, dtype=np.float64)
        res, _ = self._batch(signal, sparsity_level, connections_level, nb_branches, dissimilarity)
        leaf = int(res.rec_leaf[0, sparsity_level - 1, 1])
        support = res.branch(0, sparsity_level, leaf)
        return self._approx(support, np.ones(len(support)), len(signal)), [self._info(p, k) for p, k in support]

    def _tree_dict(self, res, s, depth, delay):
        out = {}
        for j in range(res.level_sizes[depth - 1]):
            out[res.path_str(depth, j)] = {"mse": float(res.node_mse[s, res.node_id(depth, j)]),
                                           "atoms": [self._info(p, k) for p, k in res.branch(s, depth, j)],
                                           "delay": delay}
        return out

    def mmpdfTreeFromDict(self, signal_dict: dict, connections_level: int, dissimilarity: float,
                          nb_branches: int, verbose: bool = False) -> dict:
        """{'id','sparsity','mmp-tree': {path: {'mse','atoms','delay'}}} (reference :1026-1045)."""
        k = len(signal_dict["atoms"])
        res, dt = self._batch(signal_dict["signal"], k, connections_level, nb_branches, dissimilarity)
        return {"id": signal_dict["id"], "sparsity": k, "mmp-tree": self._tree_dict(res, 0, k, dt)}

    def mmpdfSparVarFromDict(self, signal_dict: dict, connections_level: int, dissimilarity: float,
                             nb_branches: int, max_sparsity: int, compute_time_type: str = "local",
                             verbose: bool = False) -> dict:
        """{'id','snr','results':[record for sparsity 1..max_sparsity]} (reference :1047-1071).  The
        reference rebuilds a tree per sparsity; every such tree is the depth-k truncation of the
        max_sparsity tree, so one device tree yields all levels."""
        res, dt = self._batch(signal_dict["signal"], max_sparsity, connections_level, nb_branches, dissimilarity)
        return {"id": signal_dict["id"], "snr": signal_dict["snr"], "results": self._sparvar_records(res, 0, dt)}

    def mmpdfSparVarBatch(self, signals, connections_level: int = 3, dissimilarity: float = 0.8,
                          nb_branches: int = 10, max_sparsity: int = 10, sharded: bool = False):
        """Array-in / array-out twin of mmpdfSparVarPipelineFromDB for batches that do not fit JSON
        (BASELINE config 5: 10^6 signals).  ``signals`` is an [S, N] float array (or anything
        np.asarray accepts); returns a ``SparVarBatch`` (None on ranks != 0 when ``sharded`` under
        torch.distributed).  Same numbers as the per-signal records of mmpdfSparVarFromDict
        (reference :1047-1071)."""
        res, dt = self._batch(signals, max_sparsity, connections_level, nb_branches, dissimilarity, sharded=sharded,
                              fields="rec")
        if res is None:
            return None
        return SparVarBatch(self, res, dt)

    def mmpdfSparVarRecords(self, signals, connections_level: int = 3, dissimilarity: float = 0.8,
                            nb_branches: int = 10, max_sparsity: int = 10, sharded: bool = False,
                            presharded: bool = False, shard_sizes=None, recv=None, async_gather: bool = False):
        """Device-in / device-out form of mmpdfSparVarBatch: ``signals`` may be a float64 tensor already in
        HBM; returns the [S, (16 + 8 K) K] uint8 sparVar records (K f64 mse | K x 2 i32 leaf | K x K x 2 i32
        (pos, atom) per signal, BatchResult.field_specs(fields="rec")) as a tensor on this process's GPU --
        after the one gather on rank 0, None on the other ranks, when ``sharded`` under torch.distributed.
        Nothing passes through the host.  ``SparVarBatch.from_records`` turns the tensor into results.
        ``async_gather=True`` returns a ``distributed.PendingRows`` at once (the gather enqueued, the next batch's
        solve free to run under it); its ``wait()`` gives the rows."""
        rows, _ = self._records(signals, max_sparsity, connections_level, nb_branches, dissimilarity, False, "rec",
                                sharded, presharded, shard_sizes, recv, async_gather)
        return rows

    # ------------------------------------------------------------------------------------------
    # pipelines: JSON DB -> one batched device call -> JSON (reference :683-756, 1073-1152)
    # ------------------------------------------------------------------------------------------
    @staticmethod
    def _load_db(input_filename):
        with open(input_filename, "r") as f:
            data = json.load(f)
        if data is None:
            raise ValueError("The input file is empty or does not contain any data.")
        return data

    def _grouped(self, signals, key, solve, build):
        """Group the DB's signal dicts by ``key(sd)`` (signals of one device call share length and
        depth), run ``solve(key, stacked signals) -> (BatchResult | None, dt)`` per group and let
        rank 0 ``build(sd, res, row, dt, key)`` each result, in input order.  Returns None on ranks != 0."""
        groups = {}
        for i, sd in enumerate(signals):
            groups.setdefault(key(sd), []).append(i)
        out = [None] * len(signals)
        for k, idxs in groups.items():
            res, dt = solve(k, np.array([signals[i]["signal"] for i in idxs], dtype=np.float64))
            if res is None:
                out = None
                continue
            for row, i in enumerate(idxs):
                out[i] = build(signals[i], res, row, dt, k)
        return out

    @staticmethod
    def _dump(results, output_filename):
        with open(output_filename, "w") as f:
            json.dump(results, f, indent=4, default=handle_non_serializable)

    def _dump_sparvar_stream(self, header: dict, list_key: str, res, ids, snrs, delay, output_filename) -> bool:
        """json.dump(dict(header, **{list_key: [sparVar entries]}), indent=4, default=handle_non_serializable) --
        the SAME BYTES (reference :1148-1150) -- with the list written by the library's streaming writer straight
        from the packed records (csc_sparvar_json_write, include/csc_hip.h) instead of through ~30 Python objects
        per signal.  Returns False (nothing written) when an id / snr is not a plain int / float, or an atom
        parameter is not a float (json.dump writes an int parameter as `1`, the writer formats doubles: `1.0`),
        so that the caller can fall back to json.dump."""
        import ctypes as C
        from . import _lib
        if not all(type(i) is int for i in ids) or not all(type(v) in (int, float) for v in snrs):
            return False
        if not all(isinstance(a.params[k], float) for a in self.atoms for k in ("b", "y", "sigma")):
            return False                           # (np.float64 is a float; a Python / NumPy int is not)
        if any(abs(i) >= 2 ** 63 for i in ids) or any(type(v) is int and abs(v) >= 2 ** 53 for v in snrs):
            return False
        S, K = len(res), res.max_sparsity
        raw = np.ascontiguousarray(res.to_records("rec"))
        ids_a = np.asarray(ids, dtype=np.int64)
        snr_a = np.asarray(snrs, dtype=np.float64)
        is_int = np.array([type(v) is int for v in snrs], dtype=np.uint8)
        pb, py, ps = (np.array([a.params[k] for a in self.atoms], dtype=np.float64) for k in ("b", "y", "sigma"))
        head = json.dumps(header, indent=4, default=handle_non_serializable)
        assert head.endswith("\n}") and header
        with open(output_filename, "w") as f:
            f.write(head[:-2] + ",\n    " + json.dumps(list_key) + ": ")
        ptr = lambda a: a.ctypes.data_as(C.c_void_p)
        rc = _lib.lib().csc_sparvar_json_write(output_filename.encode(), 1, ptr(raw), S, K, res.connections,
                                               ptr(ids_a), ptr(snr_a), ptr(is_int), float(delay), ptr(pb), ptr(py),
                                               ptr(ps), len(self.atoms), 1, 1)
        if rc != 0:
            raise _lib.CscHipError("csc_sparvar_json_write failed: " + _lib.lib().csc_json_last_error().decode())
        return True

    def ompPipelineFromDB(self, input_filename: str, output_filename: str, nb_cores: int, verbose=False):
        """reference :683-717 (``nb_cores`` is accepted and ignored: the batch runs on the GPU(s))"""
        data = self._load_db(input_filename)
        if verbose:
            print(f"OMP Pipeline from {input_filename} with {len(data['signals'])} signals")
        omp_results = self._grouped(
            data["signals"], lambda sd: (len(sd["signal"]), len(sd["atoms"])),
            lambda k, sig: self._batch(sig, k[1], 1, 1, 1.0, want_coef=True, sharded=True),
            lambda sd, res, row, dt, k: self._omp_result(sd, res, row, k[0]))
        if omp_results is None:
            return
        results = {"source": input_filename, "date": get_today_date_str(), "algorithm": "Convolutional OMP",
                   "batchSize": data["batchSize"], "snrLevels": data["snrLevels"],
                   "signalLength": data["signalLength"], "sparsityLevels": data["sparsityLevels"],
                   "dictionary": str(self), "omp": omp_results}
        self._dump(results, output_filename)
        if verbose:
            print(f"OMP Pipeline results saved in {output_filename}")

    def ompSparVarPipelineFromDB(self, input_filename: str, output_filename: str, nb_cores: int,
                                 max_sparsity: int, verbose=False):
        """reference :719-756"""
        data = self._load_db(input_filename)
        if verbose:
            print(f"OMP SparVar Pipeline from {input_filename} with {len(data['signals'])} signals")
        omp_results = self._grouped(
            data["signals"], lambda sd: len(sd["signal"]),
            lambda k, sig: self._batch(sig, max_sparsity, 1, 1, 1.0, want_coef=True, sharded=True),
            lambda sd, res, row, dt, k: self._omp_sparvar_result(sd, res, row, dt))
        if omp_results is None:
            return
        results = {"source": input_filename, "date": get_today_date_str(), "algorithm": "Convolutional OMP",
                   "batchSize": data["batchSize"], "snrLevels": data["snrLevels"],
                   "maxSparsityLevel": max_sparsity, "signalLength": data["signalLength"],
                   "sparsityLevels": data["sparsityLevels"], "dictionary": str(self), "omp": omp_results}
        self._dump(results, output_filename)
        if verbose:
            print(f"OMP Pipeline results saved in {output_filename}")

    def mmpdfPipelineFromDB(self, input_filename: str, output_filename: str, nb_cores: int, connections: int = 3,
                            dissimilarity: float = 0.8, branches: int = 10, verbose=False):
        """reference :1073-1111"""
        data = self._load_db(input_filename)
        if verbose:
            print(f"MMP-DF Pipeline from {input_filename} with {len(data['signals'])} signals")
        mmpdf_results = self._grouped(
            data["signals"], lambda sd: (len(sd["signal"]), len(sd["atoms"])),
            lambda k, sig: self._batch(sig, k[1], connections, branches, dissimilarity, sharded=True),
            lambda sd, res, row, dt, k: {"id": sd["id"], "sparsity": k[1],
                                         "mmp-tree": self._tree_dict(res, row, k[1], dt)})
        if mmpdf_results is None:
            return
        results = {"source": input_filename, "date": get_today_date_str(), "algorithm": "Convolutional MMP-DF",
                   "nbBranches": branches, "connections": connections, "dissimilarity": dissimilarity,
                   "batchSize": data["batchSize"], "snrLevels": data["snrLevels"],
                   "signalLength": data["signalLength"], "sparsityLevels": data["sparsityLevels"],
                   "dictionary": str(self), "mmp": mmpdf_results}
        self._dump(results, output_filename)
        if verbose:
            print(f"MMP-DF Pipeline results saved in {output_filename}")

    def mmpdfSparVarPipelineFromDB(self, input_filename: str, output_filename: str, nb_cores: int,
                                   connections: int = 3, dissimilarity: float = 0.8, branches: int = 10,
                                   max_sparsity: int = 10, compute_time_type: str = "local", verbose=False):
        """The north-star path (reference :1113-1152): sparVar JSON for every signal of the DB."""
        data = self._load_db(input_filename)
        if verbose:
            print(f"MMP-DF Pipeline from {input_filename} with {len(data['signals'])} signals")
        header = {"source": input_filename, "date": get_today_date_str(), "algorithm": "Convolutional MMP-DF",
                  "nbBranches": branches, "connections": connections, "dissimilarity": dissimilarity,
                  "maxSparsityLevel": max_sparsity, "batchSize": data["batchSize"],
                  "snrLevels": data["snrLevels"], "signalLength": data["signalLength"],
                  "sparsityLevels": data["sparsityLevels"], "dictionary": str(self)}
        sigs = data["signals"]
        if len(sigs) and len({len(sd["signal"]) for sd in sigs}) == 1:
            # one signal length (every DB the reference's generators write): one batched call, records -> JSON
            # by the streaming writer; same bytes as the json.dump below
            res, dt = self._batch(np.array([sd["signal"] for sd in sigs], dtype=np.float64), max_sparsity,
                                  connections, branches, dissimilarity, sharded=True, fields="rec")
            if res is None:
                return
            if self._dump_sparvar_stream(header, "mmp", res, [sd["id"] for sd in sigs], [sd["snr"] for sd in sigs],
                                         dt, output_filename):
                if verbose:
                    print(f"MMP-DF SparVar Pipeline results saved in {output_filename}")
                return
            mmpdf_results = [{"id": sd["id"], "snr": sd["snr"], "results": self._sparvar_records(res, row, dt)}
                             for row, sd in enumerate(sigs)]
            self._dump(dict(header, mmp=mmpdf_results), output_filename)
            return
        mmpdf_results = self._grouped(
            data["signals"], lambda sd: len(sd["signal"]),
            lambda k, sig: self._batch(sig, max_sparsity, connections, branches, dissimilarity, sharded=True,
                                       fields="rec"),
            lambda sd, res, row, dt, k: {"id": sd["id"], "snr": sd["snr"],
                                         "results": self._sparvar_records(res, row, dt)})
        if mmpdf_results is None:
            return
        results = {"source": input_filename, "date": get_today_date_str(), "algorithm": "Convolutional MMP-DF",
                   "nbBranches": branches, "connections": connections, "dissimilarity": dissimilarity,
                   "maxSparsityLevel": max_sparsity, "batchSize": data["batchSize"],
                   "snrLevels": data["snrLevels"], "signalLength": data["signalLength"],
                   "sparsityLevels": data["sparsityLevels"], "dictionary": str(self), "mmp": mmpdf_results}
        self._dump(results, output_filename)
        if verbose:
            print(f"MMP-DF SparVar Pipeline results saved in {output_filename}")


class SparVarBatch:
    """Per-sparsity winners of a whole batch as arrays (what one device call + one gather returns):
    ``mse`` [S, K]; ``leaf`` [S, K] index of the winning leaf ('<=' rule, reference mmp.py:586);
    ``pos`` / ``atom`` [S, K, K] root -> leaf selections of the sparsity-k winner, -1 padded."""

    def __init__(self, dictionary, res, seconds_per_signal):
        self.dictionary, self.connections, self.max_sparsity = dictionary, res.connections, res.max_sparsity
        self.mse = res.rec_mse
        self.leaf = res.rec_leaf[:, :, 0]
        self.pos, self.atom = res.rec_sel[..., 0], res.rec_sel[..., 1]
        self.seconds_per_signal = seconds_per_signal
        self._res = res

    def __len__(self):
        return len(self.mse)

    @classmethod
    def from_records(cls, dictionary, rows, connections, branches, max_sparsity, seconds_per_signal=0.0):
        """From the [S, (16 + 8 K) K] uint8 records of mmpdfSparVarRecords (tensor or array)."""
        from .engine import BatchResult
        raw = rows.cpu().numpy() if hasattr(rows, "cpu") else np.asarray(rows)
        return cls(dictionary, BatchResult.from_records(raw, max_sparsity, connections, branches, False,
                                                        fields="rec"), seconds_per_signal)

    def path(self, s, k) -> str:
        return self._res.path_str(k, int(self.leaf[s, k - 1]))

    def record(self, s, signal_id=None, snr=None) -> dict:
        """The reference's sparVar dict of signal ``s`` ({'id','snr','results':[...]})."""
        return {"id": s if signal_id is None else signal_id, "snr": snr,
                "results": self.dictionary._sparvar_records(self._res, s, self.seconds_per_signal)}

    def true_positives(self, true_atoms_lists, pos_err_threshold, corr_err_threshold) -> np.ndarray:
        """[S, K] max true positives of the sparsity-k winner of every signal against its ground-truth
        atoms (what the workbench's sparVar metrics start from), one device call."""
        d, S, K = self.dictionary, len(self), self.max_sparsity
        kt = max(1, max(len(t) for t in true_atoms_lists))
        truth = np.full((S, kt, 2), -1, np.int32)
        for s, infos in enumerate(true_atoms_lists):
            for q, info in enumerate(infos):
                truth[s, q] = (int(info["x"]), d._atom_index(info))
        approx = np.stack([self.pos, self.atom], axis=-1).astype(np.int32)          # [S, K, K, 2]
        tp = d.device_bank().match_batch(np.repeat(truth, K, axis=0), approx.reshape(S * K, K, 2),
                                         pos_err_threshold, corr_err_threshold)
        return tp.reshape(S, K)

    def save_json(self, path, ids=None, snrs=None, header=None, list_key="mmp"):
        """The reference's sparVar result DB ({...header, 'mmp': [{'id','snr','results':[...]}]}, reference
        dictionary.py:1132-1150) written by the streaming writer: ~1.5 KB of JSON per signal at millions of
        per-sparsity records per second.  ``ids`` default to 0..S-1, ``snrs`` to 0."""
        S = len(self)
        ids = list(range(S)) if ids is None else [int(i) for i in ids]
        snrs = [0] * S if snrs is None else list(snrs)
        header = {"algorithm": "Convolutional MMP-DF", "dictionary": str(self.dictionary)} if header is None else header
        if not self.dictionary._dump_sparvar_stream(header, list_key, self._res, ids, snrs, self.seconds_per_signal, path):
            raise TypeError("ids must be ints and snrs ints or floats")

    def save_records(self, path):
        """Uncompressed binary form: ``path`` (.npy) holds the [S, (16 + 8 K) K] uint8 records exactly as the
        device wrote them (K f64 mse | K x 2 i32 leaf | K x K x 2 i32 (pos, atom) per signal), ``path + '.json'``
        the few numbers needed to read them back (SparVarBatch.load_records)."""
        d = self.dictionary
        np.save(path, self._res.to_records("rec"), allow_pickle=False)
        with open(str(path) + ".json", "w") as f:
            json.dump({"connections": int(self.connections), "branches": int(self._res.branches),
                       "max_sparsity": int(self.max_sparsity), "seconds_per_signal": float(self.seconds_per_signal),
                       "atom_b": [a.b for a in d.atoms], "atom_y": [a.y for a in d.atoms],
                       "atom_sigma": [a.sigma for a in d.atoms]}, f)

    @classmethod
    def load_records(cls, dictionary, path):
        with open(str(path) + ".json") as f:
            meta = json.load(f)
        raw = np.load(path if str(path).endswith(".npy") else str(path) + ".npy", allow_pickle=False)
        return cls.from_records(dictionary, raw, meta["connections"], meta["branches"], meta["max_sparsity"],
                                meta["seconds_per_signal"])

    def save_npz(self, path, **extra):
        """Compact binary form (~(16 + 8K) K bytes per signal instead of ~1.5 KB of JSON)."""
        d = self.dictionary
        np.savez_compressed(path, mse=self.mse, leaf=self.leaf, pos=self.pos, atom=self.atom,
                            connections=self.connections, atom_b=np.array([a.b for a in d.atoms]),
                            atom_y=np.array([a.y for a in d.atoms]), atom_sigma=np.array([a.sigma for a in d.atoms]),
                            **extra)


ZSDictionary = CSCDictionary   # README / wheel spelling of the same class
