"""Generate the golden fixtures in this directory by IMPORTING THE REFERENCE (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--jobs 8]

The reference (``/root/reference/src/csc``) imports two packages that are absent here and that
the hot path never calls: ``numba`` (``njit``/``jit`` imported, never applied) and
``alphacsc.update_z`` (only used by ``alphaCSCResultFromDict``).  Empty in-process modules are
registered for both before the import.  Nothing of the reference is copied:
the fixtures hold inputs (seeded signals) and the reference's outputs only.

Outputs (all small, committed):
  bank.npz        F1  atom bank (K, L) f64 + per-atom params / support widths / variances
  signals.npz     F2  64 constrained signals (4 per (snr, sparsity)), ground-truth atoms
  trees.json      F3  whole-tree dumps: path -> (pos, atom, mse, top-2 gap) ; c=3, K=5, B in {3,10}
  sparvar.json    F4  mmpdfSparVarFromDict records (no 'delay')
  deep.json       F5  deep trees c=4, K=7, B in {4, 64}
  omp200.json     F6  config-1: 200 seeded signals -> omp atoms (output order), mse, coefficients
  paths.json      F7  getCandidatePath table
  rootcorr.npz    F8  root correlation matrices (P, K) of 2 signals
  simmask.json    F9  zero sets of similarity masks for a few (atom, pos, thr)
  misc.json           mmpdf()/getResult, conv_multipath_matching_pursuit, ompSparVar, tree dicts
"""
import argparse
import hashlib
import json
import os
import sys
import types
from multiprocessing import Pool

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def import_reference():
    nb = types.ModuleType("numba")
    nb.njit = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))
    nb.jit = nb.njit
    sys.modules.setdefault("numba", nb)
    ac = types.ModuleType("alphacsc")
    acz = types.ModuleType("alphacsc.update_z")
    acz.update_z = None
    ac.update_z = acz
    sys.modules.setdefault("alphacsc", ac)
    sys.modules.setdefault("alphacsc.update_z", acz)
    sys.path.insert(0, "/root/reference/src")
    import csc.dictionary as rd
    import csc.mmp as rm
    import csc.main as rmain
    return rd, rm, rmain


_REF = {}


def ref_dict():
    if "d" not in _REF:
        rd, rm, rmain = import_reference()
        _REF["rd"], _REF["rm"], _REF["rmain"] = rd, rm, rmain
        _REF["d"] = rd.CSCDictionary.from_values(np.arange(0.005, 0.06, 0.005),
                                                 np.arange(0.010, 0.036, 0.002), [1])
    return _REF["d"]


N_SIG = 738


def dump_tree(tree):
    """path-prefix -> [pos, atom, mse, rel top-2 gap of the masked correlations it was chosen from]"""
    out = {}

    def rec(node, prefix):
        m = node.correlations_with_residual.copy()
        for i, c in enumerate(node.children):
            flat = m.ravel()
            top = np.argpartition(flat, -2)[-2:]
            v = np.sort(flat[top])
            gap = float((v[1] - v[0]) / abs(v[1])) if v[1] != 0 else 0.0
            key = prefix + [str(i + 1)]
            out["-".join(key)] = [int(c.atom_pos), int(c.atom_idx), float(c.getMSE()), gap]
            m = m * c.atom_similarity_mask
            rec(c, key)
    rec(tree.root, [])
    return out


def job_tree(args):
    sig, c, K, B, thr = args
    d = ref_dict()
    t = _REF["rm"].MMPTree(d, np.asarray(sig), K, c, thr)
    t.runMMPDF(B)
    rec = t.buildMMPDFResultDict()
    rec.pop("delay")
    return {"nodes": dump_tree(t), "paths": ["-".join(map(str, p)) for p in t.leaves_paths],
            "record": rec}


def job_sparvar(args):
    sd, c, thr, B, Kmax = args
    r = ref_dict().mmpdfSparVarFromDict(sd, c, thr, B, Kmax)
    for x in r["results"]:
        x.pop("delay")
    return r


def job_omp(sd):
    d = ref_dict()
    r = d.ompFromDict(sd)
    _, act = d.getSignalProjectionFromAtoms(np.asarray(sd["signal"]), r["atoms"])
    nz = np.flatnonzero(act)
    r["coef"] = {str(int(i)): float(act[i]) for i in nz}
    r["approx_sha"] = hashlib.sha256(np.asarray(r.pop("approx")).round(9).tobytes()).hexdigest()
    return r


def jsonable(o):
    if isinstance(o, np.integer):
        return int(o)
    if isinstance(o, np.floating):
        return float(o)
    if isinstance(o, np.ndarray):
        return o.tolist()
    raise TypeError(type(o))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--jobs", type=int, default=8)
    a = ap.parse_args()
    d = ref_dict()
    rd, rm, rmain = _REF["rd"], _REF["rm"], _REF["rmain"]
    K, L = len(d), d.getAtomsLength()
    print("dictionary:", str(d), "K", K, "L", L)

    # ---- F1 bank -------------------------------------------------------------------------
    bank = d.getLocalDictionary()
    np.savez_compressed(
        os.path.join(HERE, "bank.npz"), bank=bank,
        b=np.array([at.b for at in d.getAtoms()]), y=np.array([at.y for at in d.getAtoms()]),
        sigma=np.array([at.sigma for at in d.getAtoms()]),
        support_width=np.array([at.support_width for at in d.getAtoms()]),
        var=np.array([at.var for at in d.getAtoms()]),
        sha256=np.array(hashlib.sha256(np.ascontiguousarray(bank).tobytes()).hexdigest()),
        dict_str=np.array(str(d)))

    # ---- F2 signals: 4 per (snr, sparsity), README.md:54-58, 73-75 ---------------------------
    np.random.seed(20241008)
    sigs = []
    idx = 0
    from itertools import product
    for snr, sp in product([0, 5, 10, 15], [2, 3, 4, 5]):
        for _ in range(4):
            s, infos = d.generateConstrainedTestSignal(N_SIG, sp, snr, 10, 0.75)
            sigs.append({"id": idx, "snr": snr, "sparsity": sp, "signal": s.tolist(), "atoms": infos})
            idx += 1
    np.savez_compressed(os.path.join(HERE, "signals.npz"),
                        signals=np.array([s["signal"] for s in sigs]),
                        snr=np.array([s["snr"] for s in sigs]),
                        sparsity=np.array([s["sparsity"] for s in sigs]),
                        atoms_json=np.array(json.dumps([s["atoms"] for s in sigs], default=jsonable)),
                        seed=np.array(20241008))

    pool = Pool(a.jobs)

    # ---- F3 trees ----------------------------------------------------------------------------
    trees = {}
    trees["c3_K5_B3_thr0.8"] = pool.map(job_tree, [(s["signal"], 3, 5, 3, 0.8) for s in sigs])
    trees["c3_K5_B10_thr0.8"] = pool.map(job_tree, [(s["signal"], 3, 5, 10, 0.8) for s in sigs[::4]])
    trees["c2_K4_B16_thr0.4"] = pool.map(job_tree, [(s["signal"], 2, 4, 16, 0.4) for s in sigs[1::8]])
    json.dump(trees, open(os.path.join(HERE, "trees.json"), "w"), default=jsonable)
    print("F3 done")

    # ---- F4 sparVar --------------------------------------------------------------------------
    sub = [{k: s[k] for k in ("id", "snr", "signal")} for s in sigs[::4]]
    sv = {"c3_B3_K5_thr0.8": pool.map(job_sparvar, [(s, 3, 0.8, 3, 5) for s in sub]),
          "c3_B10_K4_thr0.8": pool.map(job_sparvar, [(s, 3, 0.8, 10, 4) for s in sub[:8]])}
    json.dump(sv, open(os.path.join(HERE, "sparvar.json"), "w"), default=jsonable)
    print("F4 done")

    # ---- F5 deep trees -----------------------------------------------------------------------
    deep = {"c4_K7_B4_thr0.8": pool.map(job_tree, [(s["signal"], 4, 7, 4, 0.8) for s in sigs[3::8]]),
            "c4_K7_B64_thr0.8": pool.map(job_tree, [(s["signal"], 4, 7, 64, 0.8) for s in sigs[3::16]])}
    json.dump(deep, open(os.path.join(HERE, "deep.json"), "w"), default=jsonable)
    print("F5 done")

    # ---- F6 config 1: seed 0, 200 x (738, sparsity 3, snr 10, 10, 0.75) ----------------------------
    np.random.seed(0)
    c1 = []
    for i in range(200):
        s, infos = d.generateConstrainedTestSignal(N_SIG, 3, 10, 10, 0.75)
        c1.append({"id": i, "snr": 10, "sparsity": 3, "signal": s.tolist(), "atoms": infos})
    res = pool.map(job_omp, c1)
    c1sig = np.array([s["signal"] for s in c1])
    json.dump({"seed": 0, "signals_sha256": hashlib.sha256(c1sig.tobytes()).hexdigest(),
               "first_signal_head": c1sig[0, :8].tolist(),
               "truth": [s["atoms"] for s in c1], "omp": res},
              open(os.path.join(HERE, "omp200.json"), "w"), default=jsonable)
    print("F6 done")

    # ---- F7 candidate paths --------------------------------------------------------------------
    paths = {}
    for c in (1, 2, 3, 4):
        for K_ in range(1, 8):
            nb = min(c ** K_, 100)
            paths[f"{c}_{K_}"] = [list(rm.MMPTree.getCandidatePath(K_, c, n)) for n in range(1, nb + 1)]
    json.dump(paths, open(os.path.join(HERE, "paths.json"), "w"))

    # ---- F8 root correlations ------------------------------------------------------------------
    np.savez_compressed(os.path.join(HERE, "rootcorr.npz"),
                        sig_idx=np.array([0, 37]),
                        corr=np.stack([d.computeCorrelations(np.asarray(sigs[i]["signal"])) for i in (0, 37)]))

    # ---- F9 similarity masks ---------------------------------------------------------------------
    sm = []
    for (k, p, thr) in [(0, 0, 0.8), (71, 100, 0.8), (71, 130, 0.8), (142, 246, 0.8), (30, 5, 0.4),
                        (100, 200, 0.4), (12, 123, 0.95)]:
        node = rm.MMPNode(d, np.asarray(sigs[0]["signal"]), thr, p * K + k,
                          parent=rm.MMPNode(d, np.asarray(sigs[0]["signal"]), thr))
        zeros = np.flatnonzero(node.atom_similarity_mask.ravel() == 0)
        sm.append({"atom": k, "pos": p, "thr": thr, "zeros": zeros.tolist()})
    json.dump(sm, open(os.path.join(HERE, "simmask.json"), "w"))

    # ---- misc drivers ----------------------------------------------------------------------------
    misc = {"mmpdf": [], "cmmp": [], "omp_sparvar": [], "tree_dict": []}
    for s in sigs[5::16]:
        sg = np.asarray(s["signal"])
        ap_, infos = d.mmpdf(sg, 4, 3, 0.8, 5)
        misc["mmpdf"].append({"id": s["id"], "infos": infos, "approx_head": ap_[:4].tolist(),
                              "approx_sum": float(ap_.sum())})
        ap_, infos = rmain.conv_multipath_matching_pursuit(sg, d, 3, 4)
        misc["cmmp"].append({"id": s["id"], "infos": infos, "approx_sum": float(ap_.sum())})
        r = d.ompSparVarFromDict({"id": s["id"], "snr": s["snr"], "signal": s["signal"]}, 4)
        for x in r["results"]:
            x.pop("delay")
        misc["omp_sparvar"].append(r)
        r = d.mmpdfTreeFromDict(s, 3, 0.8, 4)
        for v in r["mmp-tree"].values():
            v.pop("delay")
        misc["tree_dict"].append(r)
    json.dump(misc, open(os.path.join(HERE, "misc.json"), "w"), default=jsonable)
    print("all done")


if __name__ == "__main__":
    main()
