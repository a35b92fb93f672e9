"""CPU oracle for the convolutional OMP / MMP-DF hot path  --  TEST INFRASTRUCTURE ONLY.

This module is a float64 NumPy/SciPy restatement of the algorithm of the reference
(yanisgomes/csc @ 2024_10_08).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product package ``csc_amd`` never does.

Parity pin: the reference ships no tests and no golden vectors, so this
restatement is pinned by fixtures generated in the build container by importing the reference
itself (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz|json``) and checked in
``tests/test_oracle_golden.py``.

Two numerical back-ends share one tree walk:

* ``mode="faithful"``: the reference's own cost structure -- ``scipy.signal.oaconvolve`` over all
  K atoms for every matvec / rmatvec and ``scipy.sparse.linalg.lsqr`` with default tolerances
  (reference ``src/csc/dictionary.py:380-425``, ``src/csc/mmp.py:61-70``).  Used as the timed CPU
  baseline (kind "port").
* ``mode="direct"``: sliding-window dot products and a normal-equation solve on the selected
  columns; same results to ~1e-14, ~50x faster; used as the parity checker at larger sizes.

Every function cites the reference file:line it follows.
"""
from __future__ import annotations

import itertools
import time
from functools import partial

import numpy as np
from scipy.signal import oaconvolve
from scipy.sparse.linalg import LinearOperator, lsqr

STEP = 0.001               # reference src/csc/atoms.py:36
SUPPORT_THRESHOLD = 0.001  # reference src/csc/atoms.py:37


# --------------------------------------------------------------------------------------------
# atoms (reference src/csc/atoms.py:6-70, 130-143, 162-172)
# --------------------------------------------------------------------------------------------
def zs_dhr_dx(b, sigma, x, y):
    """d(Hr)/dx of the Zatsepin-Shcherbinin formula (reference src/csc/atoms.py:21-27)."""
    xp, xm = x + b, x - b
    denom = (xp ** 2 + y ** 2) * (xm ** 2 + y ** 2)
    der_denom = 2 * xp * (xm ** 2 + y ** 2) + (xp ** 2 + y ** 2) * 2 * xm
    num = -8 * sigma * b * x * y
    der_num = -8 * sigma * b * y
    return (der_num * denom - num * der_denom) / denom ** 2


def make_atom(b, y, sigma):
    """Trimmed, L2-normalised atom (reference src/csc/atoms.py:39-70).

    Returns (values, new_sigma).  The norm is taken BEFORE trimming (atoms.py:51-52), the trim
    keeps [first rising edge of (v > 1e-3), last falling edge) (atoms.py:61-65).
    """
    xmin, xmax = -b - 1, b + 1
    x = np.linspace(xmin, xmax, int((xmax - xmin) / STEP))
    v = zs_dhr_dx(b, sigma, x, y)
    nrm = np.linalg.norm(v)
    v = v / nrm
    edges = np.diff((v > SUPPORT_THRESHOLD).astype(int))
    start = np.argwhere(edges == 1).flatten()[0]
    end = np.argwhere(edges == -1).flatten()[-1]
    return v[start:end], sigma / nrm


def pad_both_sides(values, length):
    """Symmetric zero padding, left = total // 2 (reference src/csc/atoms.py:130-138)."""
    total = length - len(values)
    left = total // 2
    return np.pad(values, (left, total - left), mode="constant", constant_values=0)


class OracleDictionary:
    """Atom bank + convolutional operator (reference src/csc/dictionary.py:47-80, 380-484)."""

    def __init__(self, b_values, y_values, s_values):
        # itertools.product order, b-major (dictionary.py:77-79)
        self.params = []      # (b, y, normalised sigma)
        trimmed = []
        for b, y, s in itertools.product(b_values, y_values, s_values):
            v, ns = make_atom(b, y, s)
            trimmed.append(v)
            self.params.append((b, y, ns))
        self.support_widths = np.array([len(v) for v in trimmed])
        self.atoms_length = int(self.support_widths.max())          # dictionary.py:52
        self.bank = np.stack([pad_both_sides(v, self.atoms_length) for v in trimmed])  # (K, L)
        self.variances = np.array([np.var(v) for v in trimmed])      # atoms.py:68 (pre-padding)
        self.K, self.L = self.bank.shape

    def __len__(self):
        return self.K

    # ---- operator -------------------------------------------------------------------------
    def correlations(self, signal, mode="faithful"):
        """(P, K) valid cross-correlations (reference src/csc/dictionary.py:395-407)."""
        signal = np.asarray(signal, dtype=np.float64)
        if mode == "faithful":
            # the reference rebuilds the reversed atom matrix from the atom objects on every call
            # (dictionary.py:404); kept, because this mode is also the timed CPU baseline
            rev = np.array([row[::-1] for row in self.bank])
            stacked = np.stack([signal] * self.K, axis=1)
            return oaconvolve(stacked, rev.T, mode="valid", axes=0)
        win = np.lib.stride_tricks.sliding_window_view(signal, self.L)   # (P, L)
        return win @ self.bank.T

    def convolutions(self, activations):
        """(N,) synthesis (reference src/csc/dictionary.py:380-393)."""
        act = np.reshape(activations, (-1, self.K))
        atoms = np.array([row for row in self.bank])          # per-call rebuild, as dictionary.py:390
        return oaconvolve(act, atoms.T, mode="full", axes=0).sum(axis=1)

    def masked_conv_operator(self, mask):
        """conv o diag(mask) as a LinearOperator (reference src/csc/dictionary.py:409-425)."""
        A = mask.shape[0]
        P = A // self.K
        N = P + self.L - 1
        mfun = lambda x: x * mask
        mop = LinearOperator(shape=(A, A), matvec=mfun, rmatvec=mfun)
        cop = LinearOperator(shape=(N, A), matvec=partial(self.convolutions),
                             rmatvec=partial(self.correlations))
        return cop @ mop

    def place(self, atom_idx, pos, N):
        """Atom in a length-N zero signal (reference src/csc/atoms.py:170-172)."""
        s = np.zeros(N)
        s[pos:pos + self.L] = self.bank[atom_idx]
        return s

    def atom_info(self, pos, atom_idx):
        b, y, s = self.params[atom_idx]
        return {"x": int(pos), "b": b, "y": y, "s": s}

    # ---- LS projection on a support -------------------------------------------------------
    def project(self, target, support, N, mode="faithful"):
        """Least squares of ``target`` on the placed atoms ``support`` = [(pos, atom)].

        faithful: lsqr on the masked operator, defaults atol=btol=1e-6
        (reference src/csc/mmp.py:65-67, src/csc/dictionary.py:523-529).
        direct  : normal equations on the selected columns.
        Returns (approx, coefficients in ``support`` order).
        """
        target = np.asarray(target, dtype=np.float64)
        if len(support) == 0:
            return np.zeros(N), np.zeros(0)
        if mode == "faithful":
            P = N - self.L + 1
            mask = np.zeros(P * self.K)
            for pos, k in support:
                mask[pos * self.K + k] = 1.0
            op = self.masked_conv_operator(mask)
            x, *_ = lsqr(op, target)
            approx = op @ x
            return approx, np.array([x[pos * self.K + k] for pos, k in support])
        cols = np.stack([self.place(k, pos, N) for pos, k in support], axis=1)   # (N, k)
        G = cols.T @ cols
        coef = np.linalg.solve(G, cols.T @ target)
        return cols @ coef, coef

    # ---- similarity mask ------------------------------------------------------------------
    def similarity_mask(self, atom_idx, pos, N, threshold, mode="faithful"):
        """int 0/1 (P, K) mask: 1 where corr(placed atom)/||placed atom|| <= thr
        (reference src/csc/mmp.py:92-93, 196-201)."""
        s = self.place(atom_idx, pos, N)
        c = self.correlations(s, mode) / np.linalg.norm(s)
        return (c <= threshold).astype(int)


# --------------------------------------------------------------------------------------------
# MMP node / tree (reference src/csc/mmp.py:25-100, 210-231, 281-324, 388-424, 458-467, 564-606)
# --------------------------------------------------------------------------------------------
class OracleNode:
    def __init__(self, dic, signal, threshold, activation_idx=None, parent=None, mode="faithful",
                 t0=None):
        self.dic, self.parent, self.children, self.mode = dic, parent, [], mode
        self.signal = np.asarray(signal, dtype=np.float64)     # = parent's residual (mmp.py:228)
        N = len(self.signal)
        self.N = N
        self.P = N - dic.L + 1
        self.threshold = threshold
        self.t0 = time.time() if t0 is None else t0
        # support = parent's support + the new activation (mmp.py:50-56)
        self.support = [] if parent is None else list(parent.support)
        if activation_idx is not None:
            pos, k = divmod(int(activation_idx), dic.K)            # unravel (P, K), mmp.py:74
            self.pos, self.atom = pos, k
            if (pos, k) not in self.support:
                self.support.append((pos, k))
        else:
            self.pos = self.atom = None
        # "heart beat" (mmp.py:61-70): LS of the PARENT RESIDUAL on the masked columns
        approx, _ = dic.project(self.signal, self.support, N, mode)
        self.residual = self.signal - approx
        nrm = np.linalg.norm(self.residual)
        with np.errstate(divide="ignore", invalid="ignore"):
            self.corr = dic.correlations(self.residual, mode) / nrm
        if activation_idx is not None:
            self.sim_mask = dic.similarity_mask(self.atom, self.pos, N, threshold, mode)
        self.delay = time.time() - self.t0                         # mmp.py:100,122-126

    def mse(self):
        return float(np.mean(self.residual ** 2))                  # mmp.py:108-109

    def add_child(self):
        """Sibling selection with multiplicative suppression (mmp.py:210-231)."""
        m = self.corr.copy()
        for c in self.children:
            m *= c.sim_mask
        idx = int(m.argmax())
        child = OracleNode(self.dic, self.residual, self.threshold, idx, self, self.mode, self.t0)
        self.children.append(child)
        return child

    def branch(self):
        """[(pos, atom)] root -> this node in selection order (mmp.py:257-265)."""
        out, n = [], self
        while n.parent is not None:
            out.append((n.pos, n.atom))
            n = n.parent
        return out[::-1]


def candidate_path(sparsity, connections, n):
    """1-based candidate number -> path digits, least significant first (mmp.py:303-313)."""
    t, path = n - 1, []
    for _ in range(sparsity):
        path.append(t % connections + 1)
        t //= connections
    return tuple(path)


def candidate_number(sparsity, connections, path):
    """Inverse of candidate_path (mmp.py:315-324)."""
    assert len(path) == sparsity
    return 1 + sum((o - 1) * connections ** i for i, o in enumerate(path))


class OracleTree:
    def __init__(self, dic, signal, sparsity, connections, dissimilarity=0.8, mode="faithful"):
        self.dic, self.sparsity, self.connections = dic, sparsity, connections
        self.root = OracleNode(dic, signal, dissimilarity, mode=mode)
        self.leaves, self.paths = [], []

    def run(self, branches):
        """Depth-first candidate walk with node reuse (mmp.py:388-424)."""
        assert branches > 0
        nb = min(branches, self.connections ** self.sparsity)
        self.leaves, self.paths = [], []
        while len(self.paths) < nb:
            path = candidate_path(self.sparsity, self.connections, len(self.paths) + 1)
            node = self.root
            for order in path:
                if order > len(node.children):
                    node = node.add_child()
                else:
                    node = node.children[order - 1]
            self.paths.append(path)
            self.leaves.append(node)

    def result_record(self, dic=None):
        """Last-minimum leaf record (mmp.py:577-606, '<=' at :586)."""
        best, bi, m = None, None, np.inf
        for i, leaf in enumerate(self.leaves):
            if leaf.mse() <= m:
                best, bi, m = leaf, i, leaf.mse()
        return {
            "mse": best.mse(),
            "path": "-".join(str(p) for p in self.paths[bi]),
            "atoms": [self.dic.atom_info(p, k) for p, k in best.branch()],
            "delay": best.delay,
        }

    def first_min_leaf(self):
        """np.argmin (first minimum) leaf used by getResult (mmp.py:458-467)."""
        i = int(np.argmin([l.mse() for l in self.leaves]))
        return self.leaves[i], self.paths[i]

    def tree_dict(self):
        """{path: {'mse','atoms','delay'}} (mmp.py:564-575)."""
        out = {}
        for path, leaf in zip(self.paths, self.leaves):
            out["-".join(str(p) for p in path)] = {
                "mse": leaf.mse(),
                "atoms": [self.dic.atom_info(p, k) for p, k in leaf.branch()],
                "delay": leaf.delay,
            }
        return out

    def all_nodes(self):
        """{path-prefix string: (pos, atom, mse)} for every node created."""
        out = {}

        def rec(node, prefix):
            for i, c in enumerate(node.children):
                key = prefix + [str(i + 1)]
                out["-".join(key)] = (int(c.pos), int(c.atom), c.mse())
                rec(c, key)
        rec(self.root, [])
        return out


# --------------------------------------------------------------------------------------------
# drivers (reference src/csc/dictionary.py:492-649, 1012-1071)
# --------------------------------------------------------------------------------------------
def mmpdf_sparvar(dic, signal_dict, connections, dissimilarity, branches, max_sparsity,
                  mode="faithful"):
    """One fresh tree per sparsity 1..max (dictionary.py:1047-1071)."""
    results = []
    for k in range(1, max_sparsity + 1):
        tree = OracleTree(dic, signal_dict["signal"], k, connections, dissimilarity, mode)
        tree.run(branches)
        results.append(tree.result_record())
    return {"id": signal_dict["id"], "snr": signal_dict["snr"], "results": results}


def mmpdf(dic, signal, sparsity, connections, dissimilarity, branches, mode="faithful"):
    """(approx = sum of unit atoms, infos) -- dictionary.py:1012-1024, mmp.py:248-255,458-467."""
    tree = OracleTree(dic, signal, sparsity, connections, dissimilarity, mode)
    tree.run(branches)
    leaf, _ = tree.first_min_leaf()
    N = len(signal)
    approx = np.zeros(N)
    for p, k in leaf.branch():
        approx += dic.place(k, p, N)
    return approx, [dic.atom_info(p, k) for p, k in leaf.branch()]


def omp_steps(dic, signal, sparsity, mode="faithful"):
    """Greedy OMP loop (dictionary.py:512-531).  Yields per iteration
    (support [(pos, atom)] in selection order, coefficients vs the ORIGINAL signal, approx)."""
    signal = np.asarray(signal, dtype=np.float64)
    N = len(signal)
    support, residual = [], signal
    for _ in range(sparsity):
        c = dic.correlations(residual, mode)
        pos, k = divmod(int(c.argmax()), dic.K)
        if (pos, k) not in support:
            support.append((pos, k))
        approx, coef = dic.project(signal, support, N, mode)
        residual = signal - approx
        yield list(support), coef.copy(), approx


def _omp_atoms_sorted(dic, support, coef):
    """Atoms by DESCENDING coefficient, zero coefficients dropped (dictionary.py:534-550).
    The reference sorts the activations in flattened-index order with a non-stable argsort
    reversed; ties between equal coefficients do not occur on real data."""
    flat = np.array([p * dic.K + k for p, k in support])
    o = np.argsort(flat)
    flat, vals = flat[o], np.asarray(coef)[o]
    keep = vals != 0
    flat, vals = flat[keep], vals[keep]
    order = np.argsort(vals)[::-1]
    return [dic.atom_info(*divmod(int(i), dic.K)) for i in flat[order]]


def omp(dic, signal, sparsity, mode="faithful"):
    """(approx, infos) -- dictionary.py:492-552."""
    for support, coef, approx in omp_steps(dic, signal, sparsity, mode):
        pass
    return approx, _omp_atoms_sorted(dic, support, coef)


def omp_from_dict(dic, signal_dict, mode="faithful"):
    """dictionary.py:554-572."""
    signal = np.asarray(signal_dict["signal"], dtype=np.float64)
    approx, infos = omp(dic, signal, len(signal_dict["atoms"]), mode)
    return {"id": signal_dict["id"], "mse": float(np.mean((signal - approx) ** 2)),
            "sparsity": len(infos), "approx": approx.tolist(), "atoms": infos}


def omp_sparvar(dic, signal_dict, max_sparsity, mode="faithful"):
    """dictionary.py:574-649."""
    signal = np.asarray(signal_dict["signal"], dtype=np.float64)
    out, t0 = [], time.time()
    for support, coef, approx in omp_steps(dic, signal, max_sparsity, mode):
        out.append({"mse": float(np.mean((signal - approx) ** 2)),
                    "atoms": _omp_atoms_sorted(dic, support, coef),
                    "delay": time.time() - t0})
    return {"id": signal_dict["id"], "snr": signal_dict["snr"], "results": out}


# --------------------------------------------------------------------------------------------
# plain matching pursuit (reference src/csc/dictionary.py:764-921)
# --------------------------------------------------------------------------------------------
def mp_steps(dic, signal, sparsity, mode="faithful"):
    """Yields per iteration (mask {flat idx: value}, approx).  The reference accumulates
    activation_mask[idx] += corr[idx] and rebuilds approx = masked_conv_op @ activation_mask, i.e.
    conv(mask * mask): the mask multiplies itself (dictionary.py:796-804)."""
    signal = np.asarray(signal, dtype=np.float64)
    N = len(signal)
    mask, residual = {}, signal
    for _ in range(sparsity):
        c = dic.correlations(residual, mode)
        flat = int(c.argmax())
        mask[flat] = mask.get(flat, 0.0) + float(c.flat[flat])
        approx = np.zeros(N)
        for i, v in mask.items():
            pos, k = divmod(i, dic.K)
            approx += v * v * dic.place(k, pos, N)
        residual = signal - approx
        yield dict(mask), approx


def _mp_atoms_sorted(dic, mask):
    """np.where order (ascending flat index), then argsort()[::-1] on the mask values
    (dictionary.py:808-823)."""
    flat = np.array(sorted(i for i, v in mask.items() if v != 0), dtype=np.int64)
    vals = np.array([mask[int(i)] for i in flat])
    order = np.argsort(vals)[::-1]
    return [dic.atom_info(*divmod(int(i), dic.K)) for i in flat[order]]


def mp_from_dict(dic, signal_dict, mode="faithful"):
    """dictionary.py:827-845."""
    signal = np.asarray(signal_dict["signal"], dtype=np.float64)
    for mask, approx in mp_steps(dic, signal, len(signal_dict["atoms"]), mode):
        pass
    infos = _mp_atoms_sorted(dic, mask)
    return {"id": signal_dict["id"], "mse": float(np.mean((signal - approx) ** 2)), "sparsity": len(infos),
            "approx": approx.tolist(), "atoms": infos}


def mp_sparvar(dic, signal_dict, max_sparsity, mode="faithful"):
    """dictionary.py:847-921."""
    signal = np.asarray(signal_dict["signal"], dtype=np.float64)
    out, t0 = [], time.time()
    for mask, approx in mp_steps(dic, signal, max_sparsity, mode):
        out.append({"mse": float(np.mean((signal - approx) ** 2)), "atoms": _mp_atoms_sorted(dic, mask),
                    "delay": time.time() - t0})
    return {"id": signal_dict["id"], "snr": signal_dict["snr"], "results": out}


# --------------------------------------------------------------------------------------------
# synthetic constrained signals (reference src/csc/dictionary.py:271-365) -- benchmark input
# --------------------------------------------------------------------------------------------
def constrained_signal(dic, N, sparsity, snr, pos_err, corr_err):
    """Same draws from the GLOBAL numpy RNG, in the same order, as dictionary.py:292-324."""
    P = N - dic.L + 1
    signal = np.zeros(N)
    placed_sigs, placed_pos, infos = [], [], []
    while len(infos) < sparsity:
        k = np.random.choice(dic.K)
        pos = np.random.randint(0, P)
        full = dic.place(k, pos, N)
        clash = False
        for opos, osig in zip(placed_pos, placed_sigs):
            corr = np.correlate(full, osig, mode="valid")
            corr = corr / (np.linalg.norm(full) * np.linalg.norm(osig))
            if abs(pos - opos) < pos_err or np.max(corr) > corr_err:
                clash = True
                break
        if clash:
            continue
        signal += full
        placed_pos.append(pos)
        placed_sigs.append(full)
        infos.append(dic.atom_info(pos, k))
    nz = signal[np.argwhere(np.abs(signal) > SUPPORT_THRESHOLD)]
    noise_var = np.var(nz) / (10 ** (snr / 10))
    noise = np.random.normal(0, np.sqrt(noise_var), N)
    return signal + noise, infos


def constrained_signals_db(dic, batch, N, sparsity_levels, snr_levels, pos_err, corr_err):
    """List of signal dicts in product(snr, sparsity) x batch order (dictionary.py:350-361)."""
    out, idx = [], 0
    for snr, sp in itertools.product(snr_levels, sparsity_levels):
        for _ in range(batch):
            s, infos = constrained_signal(dic, N, sp, snr, pos_err, corr_err)
            out.append({"id": idx, "snr": snr, "sparsity": sp, "signal": s.tolist(),
                        "atoms": infos})
            idx += 1
    return out


def canonical_dictionary():
    """Dictionary of reference examples/conv-mmp.ipynb cell 0 (143 atoms, L = 492)."""
    return OracleDictionary(np.arange(0.005, 0.06, 0.005), np.arange(0.010, 0.036, 0.002), [1])
