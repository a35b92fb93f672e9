"""Test-only helpers.  ``OracleBank`` is a stand-in for csc_amd.engine.DeviceBank that answers
``mmpdf_batch`` / ``project_batch`` from the CPU oracle, so the HOST logic of the product (result
dictionaries, JSON pipelines, sharding + gather) can be exercised without a GPU.  It lives under
tests/ and is never imported by the package."""
import numpy as np

from csc_amd import _lib
from csc_amd.engine import BatchResult
from oracle import csc_oracle as O


class OracleBank:
    def __init__(self, oracle_dict):
        self.od = oracle_dict
        self.K, self.L = oracle_dict.K, oracle_dict.L
        self.calls = []

    def mmpdf_batch(self, signals, max_sparsity, connections, branches, dissimilarity, want_coef=True):
        signals = np.atleast_2d(np.asarray(signals, dtype=np.float64))
        S, N = signals.shape
        self.calls.append(S)
        sizes, NN = _lib.tree_shape(max_sparsity, connections, branches)
        offs = [int(x) for x in np.concatenate([[0], np.cumsum(sizes)[:-1]])]
        Kx = max_sparsity
        r = BatchResult(Kx, connections, branches, sizes, offs,
                        np.zeros((S, NN), np.int32), np.zeros((S, NN), np.int32), np.zeros((S, NN)),
                        np.zeros((S, NN, Kx)) if want_coef else None, np.zeros((S, Kx)),
                        np.zeros((S, Kx, 2), np.int32), np.full((S, Kx, Kx, 2), -1, np.int32),
                        np.zeros(4, np.int32))
        for s in range(S):
            t = O.OracleTree(self.od, signals[s], Kx, connections, dissimilarity, "direct")
            t.run(branches)
            nodes = {}

            def rec(node, digits):
                for i, c in enumerate(node.children):
                    nodes[tuple(digits + [i + 1])] = c
                    rec(c, digits + [i + 1])
            rec(t.root, [])
            for digits, node in nodes.items():
                d = len(digits)
                j = sum((o - 1) * connections ** i for i, o in enumerate(digits))
                nid = offs[d - 1] + j
                r.node_pos[s, nid], r.node_atom[s, nid], r.node_mse[s, nid] = node.pos, node.atom, node.mse()
                if want_coef:
                    _, coef = self.od.project(signals[s], node.branch(), N, "direct")
                    r.node_coef[s, nid, :d] = coef
            for k in range(1, Kx + 1):
                mse = r.node_mse[s, offs[k - 1]:offs[k - 1] + sizes[k - 1]]
                last = max(j for j in range(sizes[k - 1]) if mse[j] <= mse.min())
                r.rec_mse[s, k - 1] = mse[last]
                r.rec_leaf[s, k - 1] = (last, int(np.argmin(mse)))
                for e, (p, a) in enumerate(r.branch(s, k, last)):
                    r.rec_sel[s, k - 1, e] = (p, a)
        return r

    def mmpdf_records(self, signals, max_sparsity, connections, branches, dissimilarity, want_coef=False,
                      fields="rec"):
        import torch
        r = self.mmpdf_batch(signals, max_sparsity, connections, branches, dissimilarity,
                             want_coef=want_coef and fields == "all")
        return torch.from_numpy(r.to_records(fields)), torch.zeros(4, dtype=torch.int32)

    def mp_batch(self, signals, iterations):
        signals = np.atleast_2d(np.asarray(signals, dtype=np.float64))
        S, I = len(signals), iterations
        pos, atom = np.zeros((S, I), np.int32), np.zeros((S, I), np.int32)
        mse, mask = np.zeros((S, I)), np.zeros((S, I, I))
        for s in range(S):
            order, prev = [], {}
            for it, (m, approx) in enumerate(O.mp_steps(self.od, signals[s], I, "direct")):
                new = [f for f in m if m[f] != prev.get(f)][0]          # the activation touched this iteration
                order.append(new)
                prev = dict(m)
                pos[s, it], atom[s, it] = divmod(new, self.od.K)
                mse[s, it] = np.mean((signals[s] - approx) ** 2)
                seen = set()
                for q, f in enumerate(order):
                    if f not in seen:
                        mask[s, it, q] = m[f]
                        seen.add(f)
        return pos, atom, mse, mask

    def match_batch(self, truth, approx, pos_thr, corr_thr):
        from scipy import optimize
        out = np.zeros(len(truth), np.int32)
        N = 4 * self.L
        for m in range(len(truth)):
            T = [(int(p), int(k)) for p, k in truth[m] if p >= 0]
            A = [(int(p), int(k)) for p, k in approx[m] if p >= 0]
            if not T or not A:
                continue
            cost = np.ones((len(T), len(A)))
            for i, (pi, ki) in enumerate(T):
                for j, (pj, kj) in enumerate(A):
                    c = abs(np.dot(self.od.place(ki, pi, N), self.od.place(kj, pj, N)))
                    if abs(pi - pj) <= pos_thr and c >= corr_thr:
                        cost[i, j] = 0
            r, c = optimize.linear_sum_assignment(cost)
            out[m] = int(np.sum(cost[r, c] == 0))
        return out

    def project_batch(self, signals, supports, sig_index=None, want_approx=False):
        signals = np.atleast_2d(np.asarray(signals, dtype=np.float64))
        N = signals.shape[1]
        M, kmax = supports.shape[:2]
        coef, mse = np.zeros((M, kmax)), np.zeros(M)
        approx = np.zeros((M, N)) if want_approx else None
        for m in range(M):
            sig = signals[m if sig_index is None else sig_index[m]]
            sup = [(int(p), int(k)) for p, k in supports[m] if p >= 0]
            ap, c = self.od.project(sig, sup, N, "direct")
            coef[m, :len(sup)] = c
            mse[m] = np.mean((sig - ap) ** 2)
            if want_approx:
                approx[m] = ap
        return coef, mse, approx, 0


def strip_delay(obj):
    if isinstance(obj, dict):
        return {k: strip_delay(v) for k, v in obj.items() if k != "delay"}
    if isinstance(obj, list):
        return [strip_delay(v) for v in obj]
    return obj
