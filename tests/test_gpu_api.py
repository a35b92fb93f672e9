"""GPU tests of the drop-in API through libcsc_hip.so: the same assertions as tests/test_host_api_cpu.py
(golden fixtures generated from the reference) with the REAL device bank, plus full-size property
tests at BASELINE.json's batch size and the C-ABI error paths."""
import itertools
import json
import os

import numpy as np
import pytest

import test_host_api_cpu as H
from conftest import load_json

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["fp16_filter", "fp32_only", "no_lean"])
def dic(request):
    """every API test runs three times: fp16 matrix-core filter + float64 re-score (default: single-child levels on
    the column-group kernel) / fp32 correlations only / the fp16 filter on the register-resident kernel at every level"""
    import csc_amd
    d = csc_amd.ZSDictionary.from_values(np.arange(0.005, 0.06, 0.005), np.arange(0.010, 0.036, 0.002), [1])
    d.device_bank().fp32_only = request.param == "fp32_only"
    d.device_bank().lean = request.param != "no_lean"
    return d


def test_native_library_is_loaded(dic):
    dic.device_bank()
    assert any("libcsc_hip.so" in line for line in open("/proc/self/maps"))


def test_sparvar_from_dict(dic, golden_signals):
    H.test_sparvar_from_dict(dic, golden_signals)


def test_sparvar_all_golden(dic, golden_signals):
    sv = load_json("sparvar.json")
    sub = golden_signals["signals"][::4]
    for key, (c, B, K, thr) in {"c3_B3_K5_thr0.8": (3, 3, 5, 0.8), "c3_B10_K4_thr0.8": (3, 10, 4, 0.8)}.items():
        for i, ref in enumerate(sv[key]):
            got = dic.mmpdfSparVarFromDict({"id": ref["id"], "snr": ref["snr"], "signal": sub[i].tolist()}, c, thr, B, K)
            H.check_sparvar(got, ref)


def test_omp_from_dict(dic):
    H.test_omp_from_dict(dic)


def test_omp_config1_all_200(dic):
    """BASELINE.json configs[0]: 200 constrained signals, sparsity 3, SNR 10 dB -- every atom list (in
    the reference's descending-coefficient order), MSE and amplitude."""
    ref = load_json("omp200.json")
    np.random.seed(ref["seed"])
    sigs = [dic.generateConstrainedTestSignal(738, 3, 10, 10, 0.75) for _ in range(200)]
    res, _ = dic._batch(np.array([s for s, _ in sigs]), 3, 1, 1, 1.0, want_coef=True)
    assert res.stats[0] == 0 and res.stats[1] == 0
    for i, r in enumerate(ref["omp"]):
        got = dic._omp_result({"id": i, "atoms": sigs[i][1]}, res, i, 738)
        assert got["atoms"] == r["atoms"], i
        assert abs(got["mse"] - r["mse"]) <= 1e-9 * r["mse"]
        sup = res.branch(i, 3, 0)
        for (p, k), c in zip(sup, res.node_coef[i, res.node_id(3, 0), :3]):
            assert abs(c - r["coef"][str(p * 143 + k)]) <= 1e-8 * abs(c)


def test_misc_drivers(dic, golden_signals):
    H.test_misc_drivers(dic, golden_signals)


def test_mmptree_view(dic, golden_signals):
    H.test_mmptree_view(dic, golden_signals)


def test_pipelines_json(dic, golden_signals, tmp_path):
    H.test_sparvar_pipeline_json(dic, golden_signals, tmp_path)
    H.test_other_pipelines_json(dic, golden_signals, tmp_path)


def test_matching_pursuit(dic, golden_signals, tmp_path):
    H.test_matching_pursuit(dic, golden_signals)
    H.test_mp_pipelines_json(dic, golden_signals, tmp_path)
    ref = load_json("mp.json")                       # all golden MP runs, batched in one call per depth
    sigs, atoms = golden_signals["signals"], golden_signals["atoms"]
    for r in ref["mp"]:
        got = dic.mpFromDict({"id": r["id"], "signal": sigs[r["id"]].tolist(), "atoms": atoms[r["id"]]})
        assert got["atoms"] == r["atoms"] and abs(got["mse"] - r["mse"]) <= 1e-9 * r["mse"]


def test_sparvar_batch_arrays(dic, golden_signals, tmp_path):
    H.test_sparvar_batch_arrays(dic, golden_signals, tmp_path)


def test_true_positive_counts(dic, golden_signals):
    H.test_true_positive_counts(dic, golden_signals)
    # adversarial matching instance: augmenting paths are needed (greedy would find 2 of 3)
    bank = dic.device_bank()
    truth = np.array([[[10, 5], [12, 5], [14, 5]]], np.int32)
    approx = np.array([[[11, 5], [13, 5], [30, 5]]], np.int32)
    assert bank.match_batch(truth, approx, 1, 0.5).tolist() == [2]
    approx = np.array([[[11, 5], [13, 5], [15, 5]]], np.int32)
    assert bank.match_batch(truth, approx, 1, 0.5).tolist() == [3]
    assert bank.match_batch(truth, approx, 1, 1.5).tolist() == [0]


def test_compute_correlations_api(dic, golden_signals):
    z = np.load(os.path.join(H.ROOT, "tests", "golden", "rootcorr.npz"))
    sig = golden_signals["signals"][int(z["sig_idx"][0])]
    # the public method is float64 arithmetic like the reference's (1e-12 against its FFT values); fast=True takes the
    # same numbers on the fp32 matrix cores (fp32 accumulation over L = 492: 3e-6 ||signal||)
    c = dic.computeCorrelations(sig)
    assert c.shape == (247, 143) and c.dtype == np.float64
    assert np.abs(c - z["corr"][0]).max() <= 1e-12 * np.linalg.norm(sig)
    cf = dic.computeCorrelations(sig, fast=True)
    assert cf.shape == (247, 143) and np.abs(cf - z["corr"][0]).max() <= 3e-6 * np.linalg.norm(sig)
    # the masked operator a caller may hand to its own solver (reference :409-425): lsqr on it reproduces the
    # projection of the engine, residual orthogonal to the selected atoms to float64 accuracy
    from scipy.sparse.linalg import lsqr
    mask = np.zeros(247 * 143)
    mask[[10 * 143 + 5, 120 * 143 + 70]] = 1
    op = dic.getMaskedConvOperator(mask)
    x = lsqr(op, sig)[0]
    approx, acts = dic.getSignalProjectionFromAtoms(sig, [dic._info(10, 5), dic._info(120, 70)])
    assert np.allclose(x[mask > 0], acts[mask > 0], rtol=1e-9) and np.abs(op.rmatvec(sig - op @ x)[mask > 0]).max() < 1e-10
    # synthesis of a sparse activation vector = the placed atoms
    act = np.zeros(247 * 143)
    act[10 * 143 + 5], act[200 * 143 + 99] = 2.0, -1.0
    syn = dic.computeConvolutions(act)
    ref = 2.0 * dic.getAtom(5).getAtomInSignal(738, 10) - dic.getAtom(99).getAtomInSignal(738, 200)
    assert np.array_equal(syn, ref)


def test_project_batch_duplicates_and_padding(dic, golden_signals):
    bank = dic.device_bank()
    sig = golden_signals["signals"][:3]
    sup = np.full((3, 4, 2), -1, np.int32)
    sup[0, :2] = [(10, 5), (100, 70)]
    sup[1, :3] = [(10, 5), (100, 70), (10, 5)]          # duplicate activation
    sup[2, :1] = [(246, 142)]
    coef, mse, approx, ndup = bank.project_batch(sig, sup, sig_index=np.array([0, 0, 2], np.int32), want_approx=True)
    assert ndup == 1
    assert np.allclose(coef[0, :2], coef[1, :2], rtol=1e-12) and coef[1, 2] == 0 and (coef[0, 2:] == 0).all()
    assert abs(mse[0] - mse[1]) <= 1e-15
    from oracle import csc_oracle as O
    od = O.canonical_dictionary()
    ap, c = od.project(sig[0], [(10, 5), (100, 70)], 738, "faithful")       # lsqr, as the reference
    assert np.allclose(coef[0, :2], c, rtol=1e-8) and np.allclose(approx[0], ap, atol=1e-10)
    ap, c = od.project(sig[2], [(246, 142)], 738, "direct")
    assert np.allclose(approx[2], ap, atol=1e-13) and abs(mse[2] - np.mean((sig[2] - ap) ** 2)) <= 1e-15
    with pytest.raises(ValueError):
        bank.project_batch(sig, sup, sig_index=np.array([0, 0, 3], np.int32))
    approx0, acts = dic.getSignalProjectionFromAtoms(sig[0], [])
    assert not approx0.any() and not acts.any()


def test_abi_error_paths(dic):
    from csc_amd import _lib
    bank = dic.device_bank()
    sig = np.zeros((2, 738))
    with pytest.raises(_lib.CscHipError, match="max_sparsity"):
        bank.mmpdf_batch(sig, 17, 3, 3, 0.8)
    with pytest.raises(_lib.CscHipError, match="branches"):
        bank.mmpdf_batch(sig, 3, 3, 0, 0.8)
    with pytest.raises((ValueError, _lib.CscHipError), match="shorter"):
        bank.mmpdf_batch(np.zeros((2, 400)), 3, 3, 3, 0.8)
    with pytest.raises(AssertionError):
        dic.mmpdf(np.zeros(738), 3, 3, 0.8, 0)
    # empty batch is a no-op
    res = bank.mmpdf_batch(np.zeros((0, 738)), 3, 3, 3, 0.8)
    assert res.node_pos.shape == (0, 9)
    # workspace: too small / not 16-byte aligned are argument errors, nothing is launched
    import torch
    d = torch.zeros((2, 738), dtype=torch.float64, device="cuda")
    out = bank.mmpdf_launch(d, 3, 3, 3, 0.8)
    good = out["ws"]
    out["ws"] = good[:good.numel() // 2]
    with pytest.raises(_lib.CscHipError, match="workspace too small"):
        bank.mmpdf_launch(d, 3, 3, 3, 0.8, out=out)
    big = torch.empty(good.numel() + 1, dtype=torch.float64, device="cuda")
    out["ws"] = big[1:]
    assert out["ws"].data_ptr() % 16 == 8
    with pytest.raises(_lib.CscHipError, match="16-byte aligned"):
        bank.mmpdf_launch(d, 3, 3, 3, 0.8, out=out)
    out["ws"] = good
    bank.mmpdf_launch(d, 3, 3, 3, 0.8, out=out)
    torch.cuda.synchronize()


def test_zero_signal_edge_case(dic):
    """all-zero residual: the reference gets NaN correlations and arg-max 0; with zero
    correlations the first index wins here as well."""
    res = dic.device_bank().mmpdf_batch(np.zeros((1, 738)), 2, 2, 2, 0.8)
    assert res.node_pos[0, 0] == 0 and res.node_atom[0, 0] == 0
    assert res.node_mse[0, 0] == 0.0


def test_level_timer(dic, golden_signals):
    import torch
    from csc_amd import _lib
    bank = dic.device_bank()
    d = torch.from_numpy(golden_signals["signals"]).cuda()
    t = _lib.LevelTimer(32)
    with t:
        bank.mmpdf_launch(d, 5, 3, 3, 0.8)
        bank.mmpdf_launch(d, 5, 3, 3, 0.8)
        torch.cuda.synchronize()
    spans = t.read()
    # one span per launch group, tagged with its first level: with the lean kernel level 0 and the chains below
    # (levels 1-4 in one launch), otherwise one per level
    want = [0, 1] if (bank.lean and not bank.fp32_only) else [0, 1, 2, 3, 4]
    assert [lv for lv, _ in spans] == want * 2 and all(ms > 0 for _, ms in spans)
    bank.mmpdf_launch(d, 5, 3, 3, 0.8)          # detached: nothing recorded
    torch.cuda.synchronize()
    assert len(t.read()) == 2 * len(want)
    t.close()


# ---- BASELINE.json full sizes: size-independent properties --------------------------------------------
@pytest.fixture(scope="module")
def config2(dic):
    np.random.seed(0)
    sigs, truth = [], []
    for snr, sp in itertools.product([0, 5, 10, 15], [2, 3, 4, 5]):
        for _ in range(200):
            s, infos = dic.generateConstrainedTestSignal(738, sp, snr, 10, 0.75)
            sigs.append(s)
            truth.append(infos)
    sigs = np.array(sigs)
    return sigs, truth, dic.device_bank().mmpdf_batch(sigs, 5, 3, 3, 0.8, want_coef=True)


def test_config2_full_batch_properties(dic, config2):
    sigs, truth, res = config2
    S = len(sigs)
    assert S == 3200 and res.stats[0] == 0 and res.stats[1] == 0
    bank64 = dic.getLocalDictionary()
    # (1) a child never has a larger MSE than its parent; the root's children improve on the signal
    for d in range(2, 6):
        for j in range(3):
            assert (res.node_mse[:, res.node_id(d, j)] <= res.node_mse[:, res.node_id(d - 1, j)] * (1 + 1e-12)).all()
    # (2) per-sparsity record = the minimum over that level, '<=' picks the last minimum
    for k in range(1, 6):
        lvl = res.node_mse[:, res.node_id(k, 0):res.node_id(k, 0) + 3]
        assert np.array_equal(res.rec_mse[:, k - 1], lvl.min(axis=1))
        last = 2 - np.argmin(lvl[:, ::-1], axis=1)
        assert np.array_equal(res.rec_leaf[:, k - 1, 0], last)
        assert np.array_equal(res.rec_leaf[:, k - 1, 1], np.argmin(lvl, axis=1))
    # (3) node MSE and coefficients agree with an independent fp64 normal-equation solve (every 16th signal)
    for s in range(0, S, 16):
        for j in range(3):
            sup = res.branch(s, 5, j)
            cols = np.zeros((738, 5))
            for q, (p, k) in enumerate(sup):
                cols[p:p + 492, q] = bank64[k]
            coef, *_ = np.linalg.lstsq(cols, sigs[s], rcond=None)
            r = sigs[s] - cols @ coef
            nid = res.node_id(5, j)
            assert abs(res.node_mse[s, nid] - np.mean(r ** 2)) <= 1e-9 * np.mean(r ** 2)
            assert np.allclose(res.node_coef[s, nid], coef, rtol=1e-7, atol=1e-10)
            # residual orthogonal to the selected atoms (what lsqr converges to in the reference)
            assert np.abs(cols.T @ r).max() <= 1e-10
    # (4) siblings differ and respect the dissimilarity rule at the root level
    assert (res.node_pos[:, 0] * 143 + res.node_atom[:, 0] != res.node_pos[:, 1] * 143 + res.node_atom[:, 1]).all()


def test_config2_depth_truncation_and_determinism(dic, config2):
    """the depth-k tree is the truncation of the depth-5 tree (what lets ONE tree serve every sparVar
    level), bit for bit, and a re-run is bit-identical."""
    sigs, _, res = config2
    bank = dic.device_bank()
    r3 = bank.mmpdf_batch(sigs, 3, 3, 3, 0.8)
    n3 = r3.node_pos.shape[1]
    assert np.array_equal(r3.node_pos, res.node_pos[:, :n3]) and np.array_equal(r3.node_atom, res.node_atom[:, :n3])
    assert np.array_equal(r3.node_mse, res.node_mse[:, :n3])
    assert np.array_equal(r3.rec_sel[:, :, :, :], res.rec_sel[:, :3, :3, :])
    again = bank.mmpdf_batch(sigs, 5, 3, 3, 0.8, want_coef=True)
    for name in ("node_pos", "node_atom", "node_mse", "node_coef", "rec_mse", "rec_leaf", "rec_sel"):
        assert np.array_equal(getattr(again, name), getattr(res, name)), name
    # OMP = MMP-DF with one connection: same selections as branch '1-1-1-1-1'
    omp = bank.mmpdf_batch(sigs, 5, 1, 1, 1.0)
    for d in range(1, 6):
        assert np.array_equal(omp.node_pos[:, d - 1], res.node_pos[:, res.node_id(d, 0)])
        assert np.array_equal(omp.node_atom[:, d - 1], res.node_atom[:, res.node_id(d, 0)])


def test_config2_fp16_filter_equals_fp32_path(dic, config2):
    """the two correlation paths select the same atoms on all 3200 signals; MSE to 1e-12"""
    sigs, _, res = config2
    bank = dic.device_bank()
    old = bank.fp32_only
    try:
        bank.fp32_only = not old
        other = bank.mmpdf_batch(sigs, 5, 3, 3, 0.8, want_coef=True)
    finally:
        bank.fp32_only = old
    assert other.stats[0] == 0
    assert np.array_equal(other.node_pos, res.node_pos) and np.array_equal(other.node_atom, res.node_atom)
    assert np.allclose(other.node_mse, res.node_mse, rtol=1e-12, atol=0)
    assert np.allclose(other.node_coef, res.node_coef, rtol=1e-10, atol=1e-14)


def test_config2_vs_oracle_sample(dic, config2, oracle_dict):
    """32 of the 3200 signals against the CPU oracle, every node."""
    from oracle import csc_oracle as O
    sigs, _, res = config2
    for s in range(11, 3200, 100):
        t = O.OracleTree(oracle_dict, sigs[s], 5, 3, 0.8, "direct")
        t.run(3)
        for path, (pos, atom, mse) in t.all_nodes().items():
            digits = [int(x) for x in path.split("-")]
            nid = res.node_id(len(digits), sum((o - 1) * 3 ** i for i, o in enumerate(digits)))
            assert (int(res.node_pos[s, nid]), int(res.node_atom[s, nid])) == (pos, atom), (s, path)
            assert abs(res.node_mse[s, nid] - mse) <= 1e-9 * mse


def test_config2_all_3200_vs_oracle(dic, config2):
    """EVERY signal of config 2, every node, against the CPU oracle's trees (tests/golden/config2_trees.npz, written
    by tests/golden/make_golden_config2.py with the oracle's 'direct' mode): positions and atoms bit-exact, MSE to
    1e-9 relative."""
    import hashlib
    sigs, _, res = config2
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "config2_trees.npz"))
    if hashlib.sha256(np.ascontiguousarray(sigs).tobytes()).hexdigest() != str(z["signals_sha256"]):
        pytest.skip("this NumPy draws a different config-2 batch than the one the fixture was made from")
    assert np.array_equal(res.node_pos, z["node_pos"]) and np.array_equal(res.node_atom, z["node_atom"])
    assert np.allclose(res.node_mse, z["node_mse"], rtol=1e-9, atol=0)


def test_recovery_of_true_atoms_at_high_snr(dic, config2):
    """sanity on the task itself: at 15 dB most true atoms are matched (position within 10, correlation >= 0.75) by the
    sparsity-matched record"""
    import csc_amd
    sigs, truth, res = config2
    hits = total = 0
    for s in range(2400, 3200, 10):
        k = len(truth[s])
        got = [dic._info(p, a) for p, a in res.rec_sel[s, k - 1, :k]]
        tp, _ = csc_amd.computeMaxTruePositives(dic, 738, truth[s], got, 10, 0.75)
        hits, total = hits + tp, total + k
    assert hits / total > 0.5


def test_gather_records_nccl_world1(dic, golden_signals):
    """the RCCL gather path with a single rank (the 1-GPU box cannot host two NCCL ranks)"""
    import torch.distributed as dist
    from csc_amd import distributed as D
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29600 + os.getpid() % 300))
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        local = np.arange(35, dtype=np.uint8).reshape(7, 5)
        assert np.array_equal(D.gather_records(local, 7), local)
        res, _ = dic._batch(golden_signals["signals"][:5], 3, 3, 3, 0.8, sharded=True)
        ref, _ = dic._batch(golden_signals["signals"][:5], 3, 3, 3, 0.8)
        assert np.array_equal(res.node_pos, ref.node_pos) and np.array_equal(res.rec_mse, ref.rec_mse)
    finally:
        dist.destroy_process_group()


def test_nan_and_inf_signals_do_not_fault(dic, golden_signals):
    """NaN / Inf samples: the reference's correlations become all-NaN and np.argmax returns 0; here the
    selection falls back to activation 0 as well, MSE is NaN, and the neighbours in the batch are untouched."""
    sigs = golden_signals["signals"][:4].copy()
    sigs[1, 100] = np.nan
    sigs[2, 5] = np.inf
    bank = dic.device_bank()
    res = bank.mmpdf_batch(sigs, 3, 3, 3, 0.8)
    ref = bank.mmpdf_batch(golden_signals["signals"][:4], 3, 3, 3, 0.8)
    for s in (0, 3):
        assert np.array_equal(res.node_pos[s], ref.node_pos[s]) and np.array_equal(res.node_mse[s], ref.node_mse[s])
    assert np.isnan(res.node_mse[1]).all() and res.node_pos[1, 0] == 0 and res.node_atom[1, 0] == 0
    assert not np.isfinite(res.node_mse[2]).any()
    assert (res.node_pos >= 0).all() and (res.node_pos < 247).all() and (res.node_atom >= 0).all() and (res.node_atom < 143).all()


def test_odd_shapes_vs_oracle():
    """tiny dictionary (6 atoms), P = 2 and P = 1 positions, atom length not a multiple of 32"""
    import csc_amd
    from oracle import csc_oracle as O
    b, y = [0.01, 0.03], [0.012, 0.02, 0.03]
    d = csc_amd.ZSDictionary.from_values(b, y, [1])
    od = O.OracleDictionary(b, y, [1])
    L = od.L
    rng = np.random.default_rng(11)
    for P in (1, 2, 37):
        N = L + P - 1
        sigs = rng.standard_normal((3, N)) * 0.05
        for s in range(3):
            sigs[s, rng.integers(0, P):][:L] += od.bank[rng.integers(6)]
        Kx = min(3, P * 6)
        res, _ = d._batch(sigs, Kx, 2, 3, 0.8, want_coef=True)
        for s in range(3):
            t = O.OracleTree(od, sigs[s], Kx, 2, 0.8, "direct")
            t.run(3)
            undecided = []          # subtrees below a selection that rounding decides (see below)
            for path, (pos, atom, mse) in sorted(t.all_nodes().items(), key=lambda kv: len(kv[0])):
                if any(path.startswith(u) for u in undecided):
                    continue
                digits = [int(x) for x in path.split("-")]
                nid = res.node_id(len(digits), sum((o - 1) * 2 ** i for i, o in enumerate(digits)))
                got = (int(res.node_pos[s, nid]), int(res.node_atom[s, nid]))
                if got != (pos, atom):
                    # with one or two positions the residual can be orthogonal to every atom that is left with a
                    # non-negative correlation: the arg-max is then between values of size 1e-16 ||r|| and belongs to
                    # the rounding, not to the algorithm -- accept any such pick, do not follow it further
                    node = t.root
                    for o in digits[:-1]:
                        node = node.children[o - 1]
                    m = node.corr.copy()                      # correlations / ||r||, earlier siblings suppressed
                    for sib in node.children[:digits[-1] - 1]:
                        m *= sib.sim_mask
                    assert m.max() - m[got[0], got[1]] <= 1e-12, (P, s, path, got)
                    undecided.append(path)
                    continue
                assert abs(res.node_mse[s, nid] - mse) <= 1e-9 * mse + 1e-30


@pytest.mark.gpu
def test_long_atoms_vs_oracle():
    """atoms longer than 512 samples (592 and 736): beyond the window the fp16 filter's LDS copies hold, so every
    variant must correlate in fp32 there -- same trees from all three, equal to the oracle's"""
    import csc_amd
    from oracle import csc_oracle as O
    rng = np.random.default_rng(77)
    for ys in ([0.02, 0.05, 0.06], [0.03, 0.08]):
        b = [0.005, 0.03]
        d = csc_amd.ZSDictionary.from_values(b, ys, [1])
        od = O.OracleDictionary(b, ys, [1])
        K, L = od.bank.shape
        assert L > 544
        P = 200
        N = L + P - 1
        S = 600
        sigs = 0.05 * rng.standard_normal((S, N))
        for s in range(S):
            for _ in range(3):
                p = int(rng.integers(0, P))
                sigs[s, p:p + L] += rng.uniform(0.3, 1.0) * od.bank[int(rng.integers(K))]
        bank = d.device_bank()
        out = {}
        for name, fp32_only, lean in (("lean", False, True), ("resident", False, False), ("fp32", True, True)):
            bank.fp32_only, bank.lean = fp32_only, lean
            out[name] = bank.mmpdf_batch(sigs, 4, 2, 3, 0.8, want_coef=True)
            assert out[name].stats[0] == 0
        for name in ("resident", "fp32"):
            assert np.array_equal(out[name].node_pos, out["lean"].node_pos), (name, L)
            assert np.array_equal(out[name].node_atom, out["lean"].node_atom), (name, L)
        res = out["lean"]
        for s in range(0, S, 100):
            t = O.OracleTree(od, sigs[s], 4, 2, 0.8, "direct")
            t.run(3)
            for path, (pos, atom, mse) in t.all_nodes().items():
                digits = [int(x) for x in path.split("-")]
                nid = res.node_id(len(digits), sum((o - 1) * 2 ** i for i, o in enumerate(digits)))
                assert (int(res.node_pos[s, nid]), int(res.node_atom[s, nid])) == (pos, atom), (L, s, path)
                assert abs(res.node_mse[s, nid] - mse) <= 1e-9 * mse + 1e-30


@pytest.mark.gpu
def test_random_shapes_all_variants_agree():
    """random dictionaries (6-120 atoms, atom lengths not multiples of 32), signal lengths (1-256 positions) and tree
    shapes: the lean fp16 kernel, the register-resident fp16 kernel and the fp32-only path give the same trees, and
    a sample of them equals the oracle's"""
    import csc_amd
    from oracle import csc_oracle as O
    rng = np.random.default_rng(2024)
    b_all, y_all = np.arange(0.005, 0.06, 0.005), np.arange(0.010, 0.036, 0.002)
    for trial in range(8):
        b = sorted(rng.choice(b_all, rng.integers(2, 9), replace=False).tolist())
        y = sorted(rng.choice(y_all, rng.integers(3, 14), replace=False).tolist())
        d = csc_amd.ZSDictionary.from_values(b, y, [1])
        od = O.OracleDictionary(b, y, [1])
        K, L = od.bank.shape
        P = int(rng.integers(1, 257))
        N = L + P - 1
        c, Kx = int(rng.integers(1, 4)), int(rng.integers(2, 6))
        B = int(rng.integers(1, 7))
        S = 600                                   # more than 512 nodes per level: the lean kernel is in play
        sigs = 0.05 * rng.standard_normal((S, N))
        for s in range(S):
            for _ in range(int(rng.integers(1, 4))):
                p = int(rng.integers(0, P))
                sigs[s, p:p + L] += rng.uniform(0.3, 1.0) * od.bank[int(rng.integers(K))]
        Kx = min(Kx, P * K)
        bank = d.device_bank()
        out = {}
        for name, fp32_only, lean in (("lean", False, True), ("resident", False, False), ("fp32", True, True)):
            bank.fp32_only, bank.lean = fp32_only, lean
            out[name] = bank.mmpdf_batch(sigs, Kx, c, B, 0.8, want_coef=True)
            assert out[name].stats[0] == 0, (trial, name)
        for name in ("resident", "fp32"):
            assert np.array_equal(out[name].node_pos, out["lean"].node_pos), (trial, name, K, L, P, c, Kx, B)
            assert np.array_equal(out[name].node_atom, out["lean"].node_atom), (trial, name)
            assert np.allclose(out[name].node_mse, out["lean"].node_mse, rtol=1e-12, atol=0)
        res = out["lean"]
        for s in range(0, S, 150):
            t = O.OracleTree(od, sigs[s], Kx, c, 0.8, "direct")
            t.run(B)
            for path, (pos, atom, mse) in t.all_nodes().items():
                digits = [int(x) for x in path.split("-")]
                nid = res.node_id(len(digits), sum((o - 1) * c ** i for i, o in enumerate(digits)))
                assert (int(res.node_pos[s, nid]), int(res.node_atom[s, nid])) == (pos, atom), (trial, s, path)
                assert abs(res.node_mse[s, nid] - mse) <= 1e-9 * mse + 1e-30


def test_random_shapes_matching_pursuit_vs_oracle():
    """plain matching pursuit on random dictionaries / signal lengths, 600 signals per call (the lean kernel takes
    the single-child levels): the selected activation and the MSE of every iteration against the oracle"""
    import csc_amd
    from oracle import csc_oracle as O
    rng = np.random.default_rng(77)
    b_all, y_all = np.arange(0.005, 0.06, 0.005), np.arange(0.010, 0.036, 0.002)
    for trial in range(4):
        b = sorted(rng.choice(b_all, rng.integers(2, 9), replace=False).tolist())
        y = sorted(rng.choice(y_all, rng.integers(3, 14), replace=False).tolist())
        d = csc_amd.ZSDictionary.from_values(b, y, [1])
        od = O.OracleDictionary(b, y, [1])
        K, L = od.bank.shape
        P = int(rng.integers(2, 257))
        N = L + P - 1
        iters = int(rng.integers(2, 6))
        S = 600
        sigs = 0.05 * rng.standard_normal((S, N))
        for s in range(S):
            for _ in range(int(rng.integers(1, 4))):
                p = int(rng.integers(0, P))
                sigs[s, p:p + L] += rng.uniform(0.3, 1.0) * od.bank[int(rng.integers(K))]
        pos, atom, mse, mask = d.device_bank().mp_batch(sigs, iters)
        for s in range(0, S, 120):
            prev = {}
            for i, (m, approx) in enumerate(O.mp_steps(od, sigs[s], iters, "direct")):
                changed = [k for k, v in m.items() if prev.get(k) != v]
                assert len(changed) == 1
                p_ref, k_ref = divmod(changed[0], K)
                assert (int(pos[s, i]), int(atom[s, i])) == (p_ref, k_ref), (trial, s, i)
                want = float(np.mean((sigs[s] - approx) ** 2))
                assert abs(mse[s, i] - want) <= 1e-9 * want, (trial, s, i)
                prev = m


def test_device_generator_constraints_and_snr(dic):
    """csc_generate_signals (the on-device twin of generateConstrainedSignalsDB, reference dictionary.py:271-365).
    Its random stream is Philox, not NumPy's global generator, so the stream is not comparable ("parity unpinned" for
    the draws); what the reference's generator guarantees is checked instead, on every signal: the number of atoms,
    positions inside [0, P), |delta pos| >= pos_err_threshold and normalised inner product <= corr_err_threshold for
    every pair, signal = unit-amplitude atoms + noise whose power follows the reference's SNR rule
    (var(clean[|clean| > 1e-3]) / 10^(snr/10)), uniform draws, determinism in (seed, signal id)."""
    bank64 = dic.getLocalDictionary()
    K, L = bank64.shape
    N, P = 738, 247
    g = dic.generateConstrainedSignalsDevice(150, N, [2, 3, 4, 5], [0, 5, 10, 15], 10, 0.75, seed=5)
    sig, truth = g["signals"].cpu().numpy(), g["truth"]
    S = len(sig)
    assert S == 2400 and list(g["ids"][:3]) == [0, 1, 2] and list(g["snr"][[0, 599, 600]]) == [0, 0, 5]
    assert list(g["sparsity"][[0, 149, 150, 599]]) == [2, 2, 3, 5]
    norms = np.linalg.norm(bank64, axis=1)
    ratio = []
    for s in range(S):
        pl = [(int(p), int(k)) for p, k in truth[s] if p >= 0]
        assert len(pl) == g["sparsity"][s] and all(0 <= p < P and 0 <= k < K for p, k in pl)
        clean = np.zeros(N)
        for p, k in pl:
            clean[p:p + L] += bank64[k]
        for i in range(len(pl)):
            for j in range(i):
                (p1, k1), (p2, k2) = pl[i], pl[j]
                assert abs(p1 - p2) >= 10
                a1, a2 = np.zeros(N), np.zeros(N)
                a1[p1:p1 + L], a2[p2:p2 + L] = bank64[k1], bank64[k2]
                assert a1 @ a2 / (norms[k1] * norms[k2]) <= 0.75 + 1e-12
        noise = sig[s] - clean
        want = np.var(clean[np.abs(clean) > 1e-3]) / 10 ** (g["snr"][s] / 10)
        ratio.append(np.mean(noise ** 2) / want)
    ratio = np.array(ratio)
    # 738 Gaussian samples per signal: the per-signal power ratio scatters by sqrt(2/738) = 5 %; its mean by 0.1 %
    assert abs(ratio.mean() - 1) < 0.01 and 0.75 < ratio.min() and ratio.max() < 1.3
    # noise is white and centred; atoms and positions are drawn over the whole range
    allnoise = np.concatenate([sig[s] - sum(np.pad(bank64[k], (p, N - L - p)) for p, k in truth[s] if p >= 0)
                               for s in range(0, S, 40)])
    z = allnoise.reshape(-1, N) / allnoise.reshape(-1, N).std(axis=1, keepdims=True)
    assert abs(z.mean()) < 0.02 and abs(np.mean(z[:, 1:] * z[:, :-1])) < 0.02 and abs(np.mean(z ** 4) - 3) < 0.15
    atoms = truth[..., 1][truth[..., 1] >= 0]
    pos = truth[..., 0][truth[..., 0] >= 0]
    assert len(np.unique(atoms)) == K and pos.min() < 5 and pos.max() > P - 6
    # determinism: same (seed, first_index) -> same bits; a signal depends on its id only; another seed differs
    g2 = dic.generateConstrainedSignalsDevice(150, N, [2, 3, 4, 5], [0, 5, 10, 15], 10, 0.75, seed=5)
    assert np.array_equal(g2["signals"].cpu().numpy(), sig) and np.array_equal(g2["truth"], truth)
    g3 = dic.generateConstrainedSignalsDevice(150, N, [2], [0], 10, 0.75, seed=5, first_index=0)
    assert np.array_equal(g3["signals"].cpu().numpy(), sig[:150])
    g4 = dic.generateConstrainedSignalsDevice(150, N, [2], [0], 10, 0.75, seed=6)
    assert not np.array_equal(g4["truth"], truth[:150, :2])
    # and the solver takes the tensor as it is
    res = dic.mmpdfSparVarBatch(g["signals"][:64], 3, 0.8, 3, 5)
    assert len(res) == 64 and np.isfinite(res.mse).all()


def test_torch_custom_ops_match_the_engine(golden_signals):
    """torch.ops.csc_amd.* (csc_amd/ops.py: torch.library registrations over the C ABI, the route BASELINE.json's north
    star names) give the engine's results bit for bit."""
    import torch
    import csc_amd
    import csc_amd.ops  # noqa: F401  (registers the ops)
    d = csc_amd.ZSDictionary.from_values(np.arange(0.005, 0.06, 0.005), np.arange(0.010, 0.036, 0.002), [1])
    bank = d.device_bank()
    b64 = bank.bank64
    b32 = torch.ops.csc_amd.bank_pack(b64)
    assert torch.equal(b32.view(torch.int32), bank.bank32t.view(torch.int32))      # bit for bit (the image holds ints too)
    sim = torch.ops.csc_amd.similarity_table(b64, 738, 0.8)
    assert torch.equal(sim, bank.similarity_table(738, 0.8))
    sig = torch.from_numpy(golden_signals["signals"][:24]).to(bank.device)
    pos, atom, mse, rec = torch.ops.csc_amd.mmpdf_batch(b32, b64, sim, sig, 5, 3, 3, 0)
    ref = bank.mmpdf_launch(sig, 5, 3, 3, 0.8, want_coef=False)
    assert torch.equal(pos, ref["node_pos"]) and torch.equal(atom, ref["node_atom"]) and torch.equal(mse, ref["node_mse"])
    assert torch.equal(rec, ref["rec_buf"])
    # OMP through the op: no similarity table
    pos1, _, _, _ = torch.ops.csc_amd.mmpdf_batch(b32, b64, None, sig, 3, 1, 1, 0)
    assert torch.equal(pos1[:, 0], pos[:, 0])
    # the ops hand raw pointers to the C ABI: wrong dtype / size / device / layout must raise, not be re-interpreted
    with pytest.raises(TypeError):
        torch.ops.csc_amd.mmpdf_batch(b32, b64, sim, sig.float(), 5, 3, 3, 0)
    with pytest.raises(TypeError):
        torch.ops.csc_amd.mmpdf_batch(b32.double(), b64, sim, sig, 5, 3, 3, 0)
    with pytest.raises(TypeError):
        torch.ops.csc_amd.mmpdf_batch(b32, b64.float(), sim, sig, 5, 3, 3, 0)
    with pytest.raises(TypeError):
        torch.ops.csc_amd.mmpdf_batch(b32, b64, sim.long(), sig, 5, 3, 3, 0)
    with pytest.raises(ValueError):
        torch.ops.csc_amd.mmpdf_batch(b32[:-4], b64, sim, sig, 5, 3, 3, 0)
    with pytest.raises(ValueError):
        torch.ops.csc_amd.mmpdf_batch(b32, b64, sim[:-1], sig, 5, 3, 3, 0)
    with pytest.raises(ValueError):
        torch.ops.csc_amd.mmpdf_batch(b32, b64, sim, sig[:, :400], 5, 3, 3, 0)          # shorter than an atom
    with pytest.raises(TypeError):
        torch.ops.csc_amd.bank_pack(b64.float())
    with pytest.raises(ValueError):
        torch.ops.csc_amd.bank_pack(b64.t())                                            # not contiguous
    # a non-contiguous view of the signals is copied, not refused
    wide = torch.zeros((24, 800), dtype=torch.float64, device=bank.device)
    wide[:, :738] = sig
    pos2, _, _, rec2 = torch.ops.csc_amd.mmpdf_batch(b32, b64, sim, wide[:, :738], 5, 3, 3, 0)
    assert torch.equal(pos2, pos) and torch.equal(rec2, rec)



def test_reference_defaults_crowded_bands(dic, config2):
    """The reference's default tree (connections 3, branches 10, max sparsity 10: 92 nodes per signal) on the 3200
    config-2 signals: deep in the tree the residual is flat and 129 ... 256 entries sit in the fp16 band of the maximum;
    the chain kernel re-scores them in two halves.  The fp16 path and the fp32-only path must give the same trees, and
    nearly no chain may be left to the fp32 follow-up kernel (92 per 3200 signals were, when the list held 128)."""
    sigs = config2[0]
    bank = dic.device_bank()
    old = (bank.fp32_only, bank.lean)
    out = {}
    try:
        for name, fp32_only in (("fp16", False), ("fp32", True)):
            bank.fp32_only, bank.lean = fp32_only, True
            out[name] = bank.mmpdf_batch(sigs, 10, 3, 10, 0.8, want_coef=False)
    finally:
        bank.fp32_only, bank.lean = old
    a, r = out["fp16"], out["fp32"]
    assert np.array_equal(a.node_pos, r.node_pos) and np.array_equal(a.node_atom, r.node_atom)
    assert np.allclose(a.node_mse, r.node_mse, rtol=1e-11, atol=0)
    assert a.stats[0] == 0 and a.stats[3] <= 10, a.stats
