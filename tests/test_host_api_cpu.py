"""CPU tests of the HOST side of the drop-in API (result dictionaries, JSON pipelines, MMPTree view,
sharding + the single gather with gloo at world_size 2).  The device is replaced by
tests/helpers.OracleBank (the CPU oracle behind the DeviceBank interface) -- the HIP path itself is
covered by the -m gpu tests, which run the same assertions through libcsc_hip.so."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, load_json
from helpers import OracleBank, strip_delay


@pytest.fixture(scope="module")
def dic(oracle_dict):
    import csc_amd
    d = csc_amd.ZSDictionary.from_values(np.arange(0.005, 0.06, 0.005), np.arange(0.010, 0.036, 0.002), [1])
    bank = OracleBank(oracle_dict)
    d.device_bank = lambda: bank
    return d


def _close(a, b, rtol=1e-9):
    return abs(a - b) <= rtol * abs(b)


def check_sparvar(got, ref):
    assert (got["id"], got["snr"]) == (ref["id"], ref["snr"])
    assert len(got["results"]) == len(ref["results"])
    for g, r in zip(got["results"], ref["results"]):
        assert list(g) == ["mse", "path", "atoms", "delay"]
        assert _close(g["mse"], r["mse"])
        if g["path"] == r["path"]:
            assert g["atoms"] == r["atoms"]
        else:   # equal-support leaves: identical MSE up to rounding decides the '<=' winner
            key = lambda a: (a["x"], float(a["b"]), float(a["y"]))
            assert sorted(map(key, g["atoms"])) == sorted(map(key, r["atoms"]))


def test_sparvar_from_dict(dic, golden_signals):
    sv = load_json("sparvar.json")
    sub = golden_signals["signals"][::4]
    for i in (0, 5, 11):
        ref = sv["c3_B3_K5_thr0.8"][i]
        got = dic.mmpdfSparVarFromDict({"id": ref["id"], "snr": ref["snr"], "signal": sub[i].tolist()}, 3, 0.8, 3, 5)
        check_sparvar(got, ref)
    ref = sv["c3_B10_K4_thr0.8"][3]
    got = dic.mmpdfSparVarFromDict({"id": ref["id"], "snr": ref["snr"], "signal": sub[3].tolist()}, 3, 0.8, 10, 4)
    check_sparvar(got, ref)
    json.dumps(got, default=__import__("csc_amd").handle_non_serializable)


def test_omp_from_dict(dic):
    ref = load_json("omp200.json")
    np.random.seed(ref["seed"])
    sigs = [dic.generateConstrainedTestSignal(738, 3, 10, 10, 0.75) for _ in range(12)]
    for i in (0, 7, 11):
        got = dic.ompFromDict({"id": i, "signal": sigs[i][0].tolist(), "atoms": sigs[i][1]})
        r = ref["omp"][i]
        assert list(got) == ["id", "mse", "sparsity", "approx", "atoms"]
        assert got["atoms"] == r["atoms"] and got["sparsity"] == r["sparsity"] and _close(got["mse"], r["mse"])
        approx, infos = dic.omp(sigs[i][0], 3)
        assert infos == r["atoms"] and _close(np.mean((sigs[i][0] - approx) ** 2), r["mse"])
        # amplitudes (getSignalProjectionFromAtoms) vs the reference's lsqr activations
        ap2, acts = dic.getSignalProjectionFromAtoms(sigs[i][0], r["atoms"])
        assert acts.shape == (247 * 143,) and np.count_nonzero(acts) == 3
        for idx, c in r["coef"].items():
            assert _close(acts[int(idx)], c, 1e-8)
        assert np.allclose(ap2, approx, rtol=0, atol=1e-12)


def test_misc_drivers(dic, golden_signals):
    import csc_amd
    misc = load_json("misc.json")
    sigs = golden_signals["signals"]
    for r in misc["mmpdf"]:
        approx, infos = dic.mmpdf(sigs[r["id"]], 4, 3, 0.8, 5)
        assert infos == r["infos"] and abs(approx.sum() - r["approx_sum"]) < 1e-9
        assert np.allclose(approx[:4], r["approx_head"], atol=1e-15)
    for r in misc["cmmp"]:
        approx, infos = csc_amd.conv_multipath_matching_pursuit(sigs[r["id"]], dic, 3, 4)
        assert infos == r["infos"] and abs(approx.sum() - r["approx_sum"]) < 1e-9
    for r in misc["omp_sparvar"]:
        got = dic.ompSparVarFromDict({"id": r["id"], "snr": r["snr"], "signal": sigs[r["id"]].tolist()}, 4)
        assert (got["id"], got["snr"]) == (r["id"], r["snr"])
        for g, x in zip(got["results"], r["results"]):
            assert list(g) == ["mse", "atoms", "delay"]
            assert g["atoms"] == x["atoms"] and _close(g["mse"], x["mse"])
    for r in misc["tree_dict"]:
        sd = {"id": r["id"], "signal": sigs[r["id"]].tolist(), "atoms": [None] * r["sparsity"]}
        got = dic.mmpdfTreeFromDict(sd, 3, 0.8, 4)
        assert got["id"] == r["id"] and got["sparsity"] == r["sparsity"]
        assert list(got["mmp-tree"]) == list(r["mmp-tree"])
        for p, v in r["mmp-tree"].items():
            assert got["mmp-tree"][p]["atoms"] == v["atoms"] and _close(got["mmp-tree"][p]["mse"], v["mse"])


def test_mmptree_view(dic, golden_signals):
    import csc_amd
    ref = load_json("trees.json")["c3_K5_B10_thr0.8"][2]
    sig = golden_signals["signals"][8]
    t = csc_amd.MMPTree(dic, sig, 5, 3, 0.8)
    with pytest.raises(AssertionError):
        t.runMMPDF(0)
    t.runMMPDF(10)
    assert ["-".join(map(str, p)) for p in t.leaves_paths] == ref["paths"]
    for path, leaf in zip(ref["paths"], t.leaves_nodes):
        pos, atom, mse, _ = ref["nodes"][path]
        assert (leaf.atom_pos, leaf.atom_idx) == (pos, atom) and _close(leaf.getMSE(), mse)
        assert leaf.getFullBranchPath() == path and leaf.isLeaf() and not leaf.isRoot()
        n = leaf
        while not n.parent.isRoot():
            n = n.parent
            p2 = n.getFullBranchPath()
            assert (n.atom_pos, n.atom_idx) == tuple(ref["nodes"][p2][:2])
    rec = t.buildMMPDFResultDict()
    assert rec["path"] == ref["record"]["path"] and rec["atoms"] == ref["record"]["atoms"]
    assert _close(rec["mse"], ref["record"]["mse"])
    assert len(t.root.children) == 3 and t.root.children[0].getChildrenIndex() == 1
    assert t.MMPDFBranchFromPath([1, 2, 1, 1, 1]) is t.leaves_nodes[3]
    with pytest.raises(IndexError):
        t.MMPDFBranchFromPath([3, 3, 3, 3, 3])
    # residual of a leaf is orthogonal to its atoms, and reproduces the MSE
    leaf = t.leaves_nodes[0]
    res = leaf.getResidual()
    assert _close(np.mean(res ** 2), leaf.getMSE(), 1e-9)
    td = t.buildMMPTreeDict()
    # offline helpers on stored tree dicts
    assert csc_amd.MMPTree.getTreeParamsFromMMPTreeDict(td) == (5, 3)
    sub = csc_amd.MMPTree.getSubTreeFromMMPTreeDict(td, dic, sig, 2)
    for p, v in sub.items():
        assert _close(v["mse"], ref["nodes"][p][2], 1e-8)
    atoms, mse = csc_amd.MMPTree.ompCandidateFromMMPTreeDict(td, dic, sig, 3)
    assert _close(mse, ref["nodes"]["1-1-1"][2], 1e-8) and len(atoms) == 3
    atoms, mse = csc_amd.MMPTree.mmpdfCandidateFromMMPTreeDict(td, dic, sig, 3)
    assert _close(mse, min(v[2] for p, v in ref["nodes"].items() if len(p) == 5), 1e-8)
    assert len(csc_amd.MMPTree.shrinkMMPTreeDict(dict(td), 4)) == 4


def test_sparvar_batch_arrays(dic, golden_signals, tmp_path):
    sub = golden_signals["signals"][::4][:5]
    ref = load_json("sparvar.json")["c3_B3_K5_thr0.8"]
    b = dic.mmpdfSparVarBatch(sub, 3, 0.8, 3, 5)
    assert len(b) == 5 and b.mse.shape == (5, 5) and b.pos.shape == (5, 5, 5)
    for s in range(5):
        check_sparvar(b.record(s, ref[s]["id"], ref[s]["snr"]), ref[s])
        assert (b.pos[s, 1, 2:] == -1).all() and b.path(s, 1) in ("1", "2", "3")
    b.save_npz(str(tmp_path / "r.npz"), ids=np.arange(5))
    z = np.load(tmp_path / "r.npz")
    assert np.array_equal(z["mse"], b.mse) and len(z["atom_b"]) == 143 and int(z["connections"]) == 3


def test_true_positive_counts(dic, golden_signals):
    """maxTruePositivesBatch / SparVarBatch.true_positives vs the reference's computeMaxTruePositives"""
    import csc_amd
    ref = load_json("metrics.json")
    truth = golden_signals["atoms"][::4]
    sv = load_json("sparvar.json")["c3_B3_K5_thr0.8"]
    for case in ref["cases"]:
        pt, ct = case["pos_err_threshold"], case["corr_err_threshold"]
        t_lists = [t for t in truth for _ in range(5)]
        a_lists = [r["atoms"] for rec in sv for r in rec["results"]]
        got = dic.maxTruePositivesBatch(t_lists, a_lists, pt, ct).reshape(16, 5)
        assert got.tolist() == case["tp"], (pt, ct)
    # the host-side single-pair function agrees too (scipy assignment, as the reference)
    case = ref["cases"][0]
    for s in (0, 7):
        for k in range(5):
            tp, _ = csc_amd.computeMaxTruePositives(dic, 738, truth[s], sv[s]["results"][k]["atoms"], 10, 0.75)
            assert tp == case["tp"][s][k]
    b = dic.mmpdfSparVarBatch(golden_signals["signals"][::4], 3, 0.8, 3, 5)
    got = b.true_positives(truth, 10, 0.75)
    want = np.array(case["tp"])
    same_path = np.array([[b.path(s, k + 1) == sv[s]["results"][k]["path"] for k in range(5)] for s in range(16)])
    assert (got == want)[same_path].all() and same_path.mean() > 0.9
    assert dic.maxTruePositivesBatch([], [], 1, 0.5).shape == (0,)


def test_matching_pursuit(dic, golden_signals):
    ref = load_json("mp.json")
    sigs, atoms = golden_signals["signals"], golden_signals["atoms"]
    for r in ref["mp"][::4]:
        i = r["id"]
        got = dic.mpFromDict({"id": i, "signal": sigs[i].tolist(), "atoms": atoms[i]})
        assert list(got) == ["id", "mse", "sparsity", "approx", "atoms"]
        assert got["atoms"] == r["atoms"] and got["sparsity"] == r["sparsity"] and _close(got["mse"], r["mse"])
        assert np.allclose(got["approx"][:6], r["approx_head"], rtol=1e-9, atol=1e-15)
        approx, infos = dic.mp(sigs[i], len(atoms[i]))
        assert infos == r["atoms"]
    for r in ref["mp_sparvar"]:       # 8 iterations: some runs re-select an activation (its mask accumulates)
        i = r["id"]
        got = dic.mpSparVarFromDict({"id": i, "snr": r["snr"], "signal": sigs[i].tolist()}, 8)
        assert (got["id"], got["snr"]) == (r["id"], r["snr"]) and len(got["results"]) == 8
        for g, x in zip(got["results"], r["results"]):
            assert list(g) == ["mse", "atoms", "delay"]
            assert g["atoms"] == x["atoms"] and _close(g["mse"], x["mse"])


def test_mp_pipelines_json(dic, golden_signals, tmp_path):
    ref = load_json("mp.json")
    idx = [r["id"] for r in ref["mp"][:3]]
    src = str(tmp_path / "db.json")
    _write_db(src, golden_signals, idx)
    dic.mpPipelineFromDB(src, str(tmp_path / "m.json"), 1)
    out = json.load(open(tmp_path / "m.json"))
    assert list(out) == ["source", "date", "algorithm", "batchSize", "snrLevels", "signalLength", "sparsityLevels",
                         "dictionary", "mp"] and out["algorithm"] == "Convolutional MP"
    for got, r in zip(out["mp"], ref["mp"][:3]):
        assert got["atoms"] == r["atoms"] and _close(got["mse"], r["mse"])
    dic.mpSparVarPipelineFromDB(src, str(tmp_path / "ms.json"), 1, 3)
    out = json.load(open(tmp_path / "ms.json"))
    assert list(out)[5] == "maxSparsityLevel" and len(out["mp"][0]["results"]) == 3


def _write_db(path, golden_signals, idx):
    sigs = [{"id": int(i), "snr": int(golden_signals["snr"][i]), "sparsity": int(golden_signals["sparsity"][i]),
             "signal": golden_signals["signals"][i].tolist(), "atoms": golden_signals["atoms"][i]} for i in idx]
    db = {"date": "241008", "batchSize": 1, "snrLevels": [0, 5, 10, 15], "signalLength": 738,
          "sparsityLevels": [2, 3, 4, 5], "dictionary": "x", "signals": sigs}
    with open(path, "w") as f:
        json.dump(db, f)


def test_sparvar_pipeline_json(dic, golden_signals, tmp_path):
    idx = list(range(0, 64, 4))[:6]
    src, dst = str(tmp_path / "db.json"), str(tmp_path / "out.json")
    _write_db(src, golden_signals, idx)
    dic.mmpdfSparVarPipelineFromDB(src, dst, nb_cores=4, connections=3, dissimilarity=0.8, branches=3, max_sparsity=5)
    out = json.load(open(dst))
    assert list(out) == ["source", "date", "algorithm", "nbBranches", "connections", "dissimilarity",
                         "maxSparsityLevel", "batchSize", "snrLevels", "signalLength", "sparsityLevels",
                         "dictionary", "mmp"]
    assert out["algorithm"] == "Convolutional MMP-DF" and out["dictionary"] == "CSCMeasurementMatrix object with 143 different atoms"
    assert open(dst).read().startswith('{\n    "source"')          # indent=4
    ref = load_json("sparvar.json")["c3_B3_K5_thr0.8"]
    for got, r in zip(out["mmp"], ref[:6]):
        check_sparvar(got, r)
    with pytest.raises(ValueError):
        open(src, "w").write("null")
        dic.mmpdfSparVarPipelineFromDB(src, dst, 1)


def test_streaming_writer_is_byte_identical_to_json_dump(dic, golden_signals, tmp_path):
    """csc_sparvar_json_write (the pipelines' writer) emits exactly the bytes of the reference's
    json.dump(results, indent=4, default=handle_non_serializable) (reference dictionary.py:1148-1150, utils.py:7-15):
    int and float snr values, ids out of order, awkward floats in the records, 0 / 1 / many signals, the
    SparVarBatch front-ends."""
    import csc_amd
    from csc_amd.utils import handle_non_serializable
    sigs = golden_signals["signals"][[3, 17, 40, 41, 63]]
    res, dt = dic._batch(sigs, 5, 3, 3, 0.8, fields="rec")
    res.rec_mse[1, 0], res.rec_mse[2, 3], res.rec_mse[3, 1] = 1e-5, 123456789012345680.0, 1.0   # exponent / integral forms
    ids, snrs = [7, 3, 10 ** 12, 0, -4], [10, 2.5, 0, 15.0, 1e-7]
    header = {"source": "db.json", "date": "241008", "algorithm": "Convolutional MMP-DF", "nbBranches": 3,
              "connections": 3, "dissimilarity": 0.8, "maxSparsityLevel": 5, "batchSize": 1, "snrLevels": [0, 5.5],
              "signalLength": 738, "sparsityLevels": [2, 3], "dictionary": str(dic)}
    for n in (5, 1, 0):
        sub, _ = dic._batch(sigs[:max(n, 1)], 5, 3, 3, 0.8, fields="rec")
        sub.rec_mse[:] = res.rec_mse[:max(n, 1)]
        if n == 0:
            sub = csc_amd.engine.BatchResult.from_records(sub.to_records("rec")[:0], 5, 3, 3, False, fields="rec")
        want = dict(header, mmp=[{"id": ids[i], "snr": snrs[i], "results": dic._sparvar_records(sub, i, 0.125)}
                                 for i in range(n)])
        a, b = str(tmp_path / f"a{n}.json"), str(tmp_path / f"b{n}.json")
        with open(a, "w") as f:
            json.dump(want, f, indent=4, default=handle_non_serializable)
        assert dic._dump_sparvar_stream(header, "mmp", sub, ids[:n], snrs[:n], 0.125, b)
        assert open(a, "rb").read() == open(b, "rb").read(), n
    # ids / snr values the fast writer does not take are left to json.dump
    assert not dic._dump_sparvar_stream(header, "mmp", res, ["a"] * 5, snrs, 0.1, str(tmp_path / "c.json"))
    # so is a dictionary with an integer atom parameter (json.dump writes `1`, a double formatter `1.0`)
    saved = dic.atoms[0].params["b"]
    try:
        dic.atoms[0].params["b"] = 1
        assert not dic._dump_sparvar_stream(header, "mmp", res, ids, snrs, 0.1, str(tmp_path / "c.json"))
    finally:
        dic.atoms[0].params["b"] = saved
    assert dic._dump_sparvar_stream(header, "mmp", res, ids, snrs, 0.1, str(tmp_path / "c.json"))
    # float formatting alone, against Python's repr
    import ctypes as C
    from csc_amd import _lib
    buf = C.create_string_buffer(64)
    rng = np.random.default_rng(0)
    vals = np.concatenate([rng.standard_normal(2000) * 10.0 ** rng.integers(-12, 20, 2000),
                           rng.integers(0, 2 ** 63, 2000).astype(np.uint64).view(np.float64),
                           [0.0, -0.0, 1e16, 1e-4, 9.999e-5, 5e-324, 1.7976931348623157e308, np.inf, -np.inf, np.nan]])
    for v in vals:
        assert _lib.lib().csc_format_double(float(v), buf, 64) > 0 and buf.value.decode() == json.dumps(float(v)), v
    # array front-ends: JSON and the uncompressed binary records round-trip
    batch = dic.mmpdfSparVarBatch(sigs, 3, 0.8, 3, 5)
    batch.save_json(str(tmp_path / "s.json"), ids=ids, snrs=snrs, header=header)
    out = json.load(open(tmp_path / "s.json"))
    assert [m["id"] for m in out["mmp"]] == ids and strip_delay(out["mmp"][4]) == strip_delay(batch.record(4, ids[4], snrs[4]))
    batch.save_records(str(tmp_path / "r.npy"))
    back = csc_amd.dictionary.SparVarBatch.load_records(dic, str(tmp_path / "r.npy"))
    assert np.array_equal(back.mse, batch.mse) and np.array_equal(back.pos, batch.pos) and back.path(2, 5) == batch.path(2, 5)


def test_other_pipelines_json(dic, golden_signals, tmp_path):
    idx = [5, 21, 37]
    src = str(tmp_path / "db.json")
    _write_db(src, golden_signals, idx)
    misc = load_json("misc.json")
    dic.mmpdfPipelineFromDB(src, str(tmp_path / "t.json"), 1, connections=3, dissimilarity=0.8, branches=4)
    out = json.load(open(tmp_path / "t.json"))
    assert list(out)[-1] == "mmp" and "maxSparsityLevel" not in out
    for got, r in zip(out["mmp"], misc["tree_dict"]):
        assert strip_delay(got)["mmp-tree"].keys() == r["mmp-tree"].keys() and got["id"] == r["id"]
        for p, v in r["mmp-tree"].items():
            assert got["mmp-tree"][p]["atoms"] == v["atoms"]
    dic.ompSparVarPipelineFromDB(src, str(tmp_path / "o.json"), 1, 4)
    out = json.load(open(tmp_path / "o.json"))
    assert out["algorithm"] == "Convolutional OMP" and out["maxSparsityLevel"] == 4
    for got, r in zip(out["omp"], misc["omp_sparvar"]):
        for g, x in zip(got["results"], r["results"]):
            assert g["atoms"] == x["atoms"] and _close(g["mse"], x["mse"])
    dic.ompPipelineFromDB(src, str(tmp_path / "p.json"), 1)
    out = json.load(open(tmp_path / "p.json"))
    assert [o["sparsity"] for o in out["omp"]] == [int(golden_signals["sparsity"][i]) for i in idx]
    assert len(out["omp"][0]["approx"]) == 738


def test_workbench_shell(dic, golden_signals, tmp_path):
    import csc_amd
    src = str(tmp_path / "db.json")
    _write_db(src, golden_signals, [0, 4])
    wb = csc_amd.CSCWorkbench(src)
    wb.set_dictionary(dic)
    assert wb.signalDictFromId(4)["id"] == 4 and wb.signalDictFromId(99) is None
    wb.runMMPDFSparVar(str(tmp_path / "o.json"), 3, 0.8, 3, 2)
    assert len(json.load(open(tmp_path / "o.json"))["mmp"]) == 2


def test_metrics_utils(dic, golden_signals):
    import csc_amd
    truth = golden_signals["atoms"][9]
    tp, matched = csc_amd.computeMaxTruePositives(dic, 738, truth, truth, 5, 0.9)
    assert tp == len(truth) and all(a == b for a, b in matched)
    shifted = [dict(a, x=(a["x"] + 60) % 247) for a in truth]
    tp, _ = csc_amd.computeMaxTruePositives(dic, 738, truth, shifted, 5, 0.9)
    assert tp < len(truth)
    assert len(csc_amd.getMatchedAtoms(dic, 738, truth, shifted, 5, 0.9)) == len(truth)


WORKER = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import numpy as np, torch.distributed as dist
rank, world, port, src, dst = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], sys.argv[6]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world))
dist.init_process_group("gloo", rank=rank, world_size=world)
import csc_amd
from csc_amd import distributed as D
from helpers import OracleBank
from oracle import csc_oracle as O
# 1. raw gather with uneven shards (7 records over 2 ranks -> 4 + 3)
lo, hi = D.shard_range(7, rank, world)
local = (np.arange(lo, hi, dtype=np.uint8)[:, None] * np.ones((1, 5), np.uint8))
got = D.gather_records(local, 7)
assert (got is None) == (rank != 0)
if rank == 0:
    assert got.shape == (7, 5) and (got[:, 0] == np.arange(7)).all()
# 2. the sparVar pipeline: every rank solves its slice, rank 0 writes the JSON
d = csc_amd.ZSDictionary.from_values(np.arange(0.005, 0.06, 0.005), np.arange(0.010, 0.036, 0.002), [1])
bank = OracleBank(O.canonical_dictionary())
d.device_bank = lambda: bank
d.mmpdfSparVarPipelineFromDB(src, dst, 1, 3, 0.8, 3, 5)
calls = list(bank.calls)
# 3. the gather on tensors (what the GPU path hands over: device rows in, device rows out), uneven shards
import torch
lo, hi = D.shard_range(5, rank, world)
rows = torch.arange(lo, hi, dtype=torch.uint8)[:, None].repeat(1, 3)
got = D.gather_rows(rows, D.shard_sizes(5, world))
assert (got is None) == (rank != 0)
if rank == 0:
    assert isinstance(got, torch.Tensor) and got.shape == (5, 3) and got[:, 0].tolist() == [0, 1, 2, 3, 4]
# 4. fewer signals than ranks: rank 1's shard is empty and still has to join the one gather (OMP / MMP-DF and MP cores)
sigs = np.array([sd["signal"] for sd in json.load(open(src))["signals"]])
res, _ = d._batch(sigs[:1], 2, 1, 1, 1.0, want_coef=True, sharded=True)
assert (res is None) == (rank != 0)
if rank == 0:
    assert len(res) == 1 and res.node_coef.shape == (1, 2, 2)
arrs = d._mp_batch(sigs[:1], 2, sharded=True)
assert (arrs is None) == (rank != 0)
if rank == 0:
    assert arrs[0].shape == (1, 2) and arrs[3].shape == (1, 2, 2)
# 5. pre-sharded device-level call (bench.py's config-5 leg): every rank passes its own rows
mine = sigs[:2] if rank == 0 else sigs[2:3]
rec = d.mmpdfSparVarRecords(mine, 3, 0.8, 3, 5, sharded=True, presharded=True)
assert (rec is None) == (rank != 0)
if rank == 0:
    b = csc_amd.dictionary.SparVarBatch.from_records(d, rec, 3, 3, 5)
    whole = d.mmpdfSparVarBatch(sigs[:3], 3, 0.8, 3, 5)
    assert len(b) == 3 and np.array_equal(b.mse, whole.mse) and np.array_equal(b.pos, whole.pos)
json.dump({"rank": rank, "calls": calls}, open(dst + f".rank{rank}", "w"))
dist.barrier()
dist.destroy_process_group()
'''


def test_pipeline_sharded_gloo_world2(golden_signals, tmp_path):
    """world_size 2 (gloo): shards [0,3) / [3,5), one gather, JSON written by rank 0 only and equal to
    the single-process result."""
    idx = list(range(0, 64, 4))[:5]
    src, dst = str(tmp_path / "db.json"), str(tmp_path / "out.json")
    _write_db(src, golden_signals, idx)
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = str(29500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), "2", port, src, dst],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    calls = [json.load(open(dst + f".rank{r}"))["calls"] for r in range(2)]
    assert calls == [[3], [2]]
    out = json.load(open(dst))
    assert [m["id"] for m in out["mmp"]] == idx
    ref = load_json("sparvar.json")["c3_B3_K5_thr0.8"]
    for got, r in zip(out["mmp"], ref[:5]):
        check_sparvar(got, r)
