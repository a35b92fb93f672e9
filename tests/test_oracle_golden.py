"""Pin the CPU oracle (oracle/csc_oracle.py) against the fixtures generated from the reference.

CPU only.  ``faithful`` mode (oaconvolve + lsqr, the reference's own numerics) is checked on a few
trees; ``direct`` mode (sliding-window GEMM + normal equations) on all of them.
"""
import hashlib
import json

import numpy as np
import pytest

from conftest import load_json
from oracle import csc_oracle as O


def test_bank_matches_reference(golden_bank, oracle_dict):
    d = oracle_dict
    assert (d.K, d.L) == (143, 492)
    assert np.array_equal(d.bank, golden_bank["bank"])
    assert hashlib.sha256(np.ascontiguousarray(d.bank).tobytes()).hexdigest() == str(golden_bank["sha256"])
    assert np.array_equal(np.array([p[0] for p in d.params]), golden_bank["b"])
    assert np.array_equal(np.array([p[1] for p in d.params]), golden_bank["y"])
    assert np.array_equal(np.array([p[2] for p in d.params]), golden_bank["sigma"])
    assert np.array_equal(d.support_widths, golden_bank["support_width"])
    assert np.allclose(d.variances, golden_bank["var"], rtol=0, atol=0)


def test_candidate_paths():
    table = load_json("paths.json")
    for key, paths in table.items():
        c, K = map(int, key.split("_"))
        for n, p in enumerate(paths, start=1):
            assert list(O.candidate_path(K, c, n)) == p
            assert O.candidate_number(K, c, p) == n


def test_root_correlations(golden_signals, oracle_dict):
    import os
    from conftest import GOLDEN
    z = np.load(os.path.join(GOLDEN, "rootcorr.npz"))
    for i, c_ref in zip(z["sig_idx"], z["corr"]):
        sig = golden_signals["signals"][i]
        assert np.array_equal(oracle_dict.correlations(sig, "faithful"), c_ref)
        assert np.abs(oracle_dict.correlations(sig, "direct") - c_ref).max() < 1e-12


def test_similarity_masks(golden_signals, oracle_dict):
    for rec in load_json("simmask.json"):
        for mode in ("faithful", "direct"):
            m = oracle_dict.similarity_mask(rec["atom"], rec["pos"], 738, rec["thr"], mode)
            assert np.flatnonzero(m.ravel() == 0).tolist() == rec["zeros"], (rec["atom"], mode)


def _check_tree(dic, sig, ref, c, K, B, thr, mode, mse_rtol):
    t = O.OracleTree(dic, sig, K, c, thr, mode)
    t.run(B)
    nodes = t.all_nodes()
    assert set(nodes) == set(ref["nodes"])
    for path, (pos, atom, mse, gap) in ref["nodes"].items():
        got = nodes[path]
        assert (got[0], got[1]) == (pos, atom), (path, got, pos, atom, gap)
        assert abs(got[2] - mse) <= mse_rtol * mse
    assert ["-".join(map(str, p)) for p in t.paths] == ref["paths"]
    return t


@pytest.mark.parametrize("key,c,K,B,thr", [("c3_K5_B3_thr0.8", 3, 5, 3, 0.8),
                                           ("c3_K5_B10_thr0.8", 3, 5, 10, 0.8),
                                           ("c2_K4_B16_thr0.4", 2, 4, 16, 0.4)])
def test_trees_direct(golden_signals, oracle_dict, key, c, K, B, thr):
    refs = load_json("trees.json")[key]
    stride = {"c3_K5_B3_thr0.8": (0, 1), "c3_K5_B10_thr0.8": (0, 4), "c2_K4_B16_thr0.4": (1, 8)}[key]
    sigs = golden_signals["signals"][stride[0]::stride[1]]
    assert len(sigs) == len(refs)
    for sig, ref in zip(sigs, refs):
        t = _check_tree(oracle_dict, sig, ref, c, K, B, thr, "direct", 1e-9)
        rec = t.result_record()
        # leaves with the same support have mathematically equal MSE; rounding picks the winner
        if rec["path"] != ref["record"]["path"]:
            assert abs(rec["mse"] - ref["record"]["mse"]) <= 1e-9 * rec["mse"]
        else:
            assert rec["atoms"] == ref["record"]["atoms"]


def test_trees_faithful(golden_signals, oracle_dict):
    refs = load_json("trees.json")["c3_K5_B3_thr0.8"]
    for i in (0, 21, 63):
        t = _check_tree(oracle_dict, golden_signals["signals"][i], refs[i], 3, 5, 3, 0.8, "faithful", 1e-13)
        rec = t.result_record()
        assert rec["path"] == refs[i]["record"]["path"]
        assert rec["mse"] == refs[i]["record"]["mse"]
        assert rec["atoms"] == refs[i]["record"]["atoms"]


def test_deep_trees_direct(golden_signals, oracle_dict):
    deep = load_json("deep.json")
    for sig, ref in zip(golden_signals["signals"][3::8], deep["c4_K7_B4_thr0.8"]):
        _check_tree(oracle_dict, sig, ref, 4, 7, 4, 0.8, "direct", 1e-9)
    sig, ref = golden_signals["signals"][3], deep["c4_K7_B64_thr0.8"][0]
    assert len(ref["nodes"]) == 4 + 16 + 64 * 5
    _check_tree(oracle_dict, sig, ref, 4, 7, 64, 0.8, "direct", 1e-9)


def test_sparvar_records(golden_signals, oracle_dict):
    sv = load_json("sparvar.json")
    sub = golden_signals["signals"][::4]
    for key, (c, B, K, thr) in {"c3_B3_K5_thr0.8": (3, 3, 5, 0.8), "c3_B10_K4_thr0.8": (3, 10, 4, 0.8)}.items():
        for i, ref in enumerate(sv[key]):
            sd = {"id": ref["id"], "snr": ref["snr"], "signal": sub[i]}
            got = O.mmpdf_sparvar(oracle_dict, sd, c, thr, B, K, mode="direct")
            assert (got["id"], got["snr"]) == (ref["id"], ref["snr"])
            for g, r in zip(got["results"], ref["results"]):
                assert abs(g["mse"] - r["mse"]) <= 1e-9 * r["mse"]
                if g["path"] == r["path"]:
                    assert g["atoms"] == r["atoms"]
    # reference-faithful numerics reproduce the record exactly (one signal: 50 nodes, ~4 s)
    ref = sv["c3_B3_K5_thr0.8"][2]
    got = O.mmpdf_sparvar(oracle_dict, {"id": ref["id"], "snr": ref["snr"], "signal": sub[2]}, 3, 0.8, 3, 5)
    for g in got["results"]:
        g.pop("delay")
    assert got == ref


def test_omp_config1(oracle_dict):
    ref = load_json("omp200.json")
    # regenerate the config-1 signals from the recorded seed with the oracle's generator
    np.random.seed(ref["seed"])
    sigs = [O.constrained_signal(oracle_dict, 738, 3, 10, 10, 0.75) for _ in range(200)]
    arr = np.array([s for s, _ in sigs])
    assert hashlib.sha256(arr.tobytes()).hexdigest() == ref["signals_sha256"]
    assert [a for _, a in sigs] == ref["truth"]
    for i in range(0, 200, 5):
        sd = {"id": i, "signal": arr[i], "atoms": sigs[i][1]}
        got = O.omp_from_dict(oracle_dict, sd, mode="direct")
        r = ref["omp"][i]
        assert got["atoms"] == r["atoms"]
        assert abs(got["mse"] - r["mse"]) <= 1e-9 * r["mse"]
        # coefficients of the final support
        for (sup, coef, _approx) in O.omp_steps(oracle_dict, arr[i], 3, "direct"):
            pass
        for (p, k), cval in zip(sup, coef):
            assert abs(cval - r["coef"][str(p * 143 + k)]) <= 1e-9 * abs(cval)
    got = O.omp_from_dict(oracle_dict, {"id": 7, "signal": arr[7], "atoms": sigs[7][1]})
    assert got["atoms"] == ref["omp"][7]["atoms"]
    assert abs(got["mse"] - ref["omp"][7]["mse"]) <= 1e-14 * got["mse"]


def test_misc_drivers(golden_signals, oracle_dict):
    misc = load_json("misc.json")
    sigs = golden_signals["signals"]
    for r in misc["mmpdf"]:
        approx, infos = O.mmpdf(oracle_dict, sigs[r["id"]], 4, 3, 0.8, 5, mode="direct")
        assert infos == r["infos"]
        assert abs(approx.sum() - r["approx_sum"]) < 1e-9
    for r in misc["cmmp"]:   # main.conv_multipath_matching_pursuit: dissimilarity 0.4, connections = branches
        approx, infos = O.mmpdf(oracle_dict, sigs[r["id"]], 3, 4, 0.4, 4, mode="direct")
        assert infos == r["infos"]
    for r in misc["omp_sparvar"]:
        got = O.omp_sparvar(oracle_dict, {"id": r["id"], "snr": r["snr"], "signal": sigs[r["id"]]}, 4, "direct")
        for g, x in zip(got["results"], r["results"]):
            assert g["atoms"] == x["atoms"]
            assert abs(g["mse"] - x["mse"]) <= 1e-9 * x["mse"]
    for r in misc["tree_dict"]:
        K = r["sparsity"]
        t = O.OracleTree(oracle_dict, sigs[r["id"]], K, 3, 0.8, "direct")
        t.run(4)
        td = t.tree_dict()
        assert list(td) == list(r["mmp-tree"])
        for p, v in r["mmp-tree"].items():
            assert td[p]["atoms"] == v["atoms"]
            assert abs(td[p]["mse"] - v["mse"]) <= 1e-9 * v["mse"]


def test_matching_pursuit(golden_signals, oracle_dict):
    ref = load_json("mp.json")
    sigs, atoms = golden_signals["signals"], golden_signals["atoms"]
    for n, r in enumerate(ref["mp"]):
        i = r["id"]
        mode = "faithful" if n % 7 == 0 else "direct"
        got = O.mp_from_dict(oracle_dict, {"id": i, "signal": sigs[i], "atoms": atoms[i]}, mode)
        assert got["atoms"] == r["atoms"] and got["sparsity"] == r["sparsity"]
        assert abs(got["mse"] - r["mse"]) <= (1e-13 if mode == "faithful" else 1e-9) * r["mse"]
    for r in ref["mp_sparvar"][:4]:
        i = r["id"]
        got = O.mp_sparvar(oracle_dict, {"id": i, "snr": r["snr"], "signal": sigs[i]}, 8, "direct")
        for g, x in zip(got["results"], r["results"]):
            assert g["atoms"] == x["atoms"] and abs(g["mse"] - x["mse"]) <= 1e-9 * x["mse"]
