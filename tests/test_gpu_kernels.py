"""GPU parity tests of the individual entry points of libcsc_hip.so (through the C ABI via
csc_amd.engine) against the CPU oracle and the golden fixtures generated from the reference.

Bars: indices / positions / masks bit-exact; fp32 correlations within the fp32 accumulation bound
stated at the assert; fp64 MSE within 1e-9 relative (north-star asks 1e-5).
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_json

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["fp16_filter", "fp32_only", "no_lean"])
def bank(request, golden_bank):
    """both ways of getting a node's correlations: the fp16 matrix-core filter in front of the float64 re-score
    (default) and fp32 correlations only"""
    from csc_amd.engine import DeviceBank
    b = DeviceBank(golden_bank["bank"])
    b.fp32_only = request.param == "fp32_only"
    b.lean = request.param != "no_lean"
    return b


def test_bank_pack(bank, golden_bank):
    """packed bank = column tiles sorted by support, blocked [slab][column tile][32 rows][16 atoms],
    followed by the permutation and the per-slab non-zero block masks"""
    b = golden_bank["bank"]
    K, L = b.shape
    raw = bank.bank32t.cpu().numpy()
    nslab = 16
    tiles = raw[:nslab * 9 * 512].reshape(nslab, 9, 32, 16)
    meta = raw[nslab * 9 * 512:].view(np.int32)
    perm, masks = meta[:144], meta[144:144 + nslab]
    assert sorted(perm.tolist()) == [-1] + list(range(K)) and perm[0] == -1
    first = (b != 0).argmax(axis=1)
    assert (np.diff(first[perm[1:]]) <= 0).all()                       # narrowest support first
    img = tiles.transpose(0, 2, 1, 3).reshape(nslab * 32, 144)          # [row][column]
    assert np.array_equal(img[:L, 1:], b.T.astype(np.float32)[:, perm[1:]])
    assert not img[L:].any() and not img[:, 0].any()
    for s in range(nslab):
        want = sum(1 << ni for ni in range(9) if tiles[s, ni].any())
        assert masks[s] == want
    # the zero padding that is skipped: ~30 % of the (slab, column tile) blocks
    active = sum(9 - (int(m) & -int(m)).bit_length() + 1 for m in masks if m)
    assert active < 0.75 * nslab * 9
    # the same blocks in fp16, [slab][column tile][k-group 4][column 16][8 rows], then max_k ||a_k||^2
    off = nslab * 9 * 512 + 208
    t16 = raw[off:off + nslab * 9 * 256].view(np.float16).reshape(nslab, 9, 4, 16, 8)
    img16 = t16.transpose(0, 2, 4, 1, 3).reshape(nslab * 32, 144)      # [row][column]
    err = np.abs(img16.astype(np.float64) - img)                        # round to nearest: 2^-11 relative, subnormals kept
    assert (err <= 2.0 ** -11 * np.abs(img) + 2.0 ** -25).all() and (img16 != 0).sum() > 0.99 * (img != 0).sum()
    amax2 = float(raw[off + nslab * 9 * 256])
    want = float((b * b).sum(axis=1).max())
    assert want <= amax2 <= want * (1 + 1e-6)


def test_correlate_matches_reference(bank, golden_signals):
    z = np.load(os.path.join(GOLDEN, "rootcorr.npz"))
    sigs = golden_signals["signals"][z["sig_idx"]]
    got = bank.correlate(sigs).cpu().numpy()
    assert got.shape == z["corr"].shape
    # fp32 fmaf chain over L = 492 terms: |err| <= ~2e-7 * sum|a*b| <= 2e-7 * ||r|| (measured ~1e-7)
    for g, ref, s in zip(got, z["corr"], sigs):
        assert np.abs(g - ref).max() <= 3e-6 * np.linalg.norm(s)
        assert np.unravel_index(g.argmax(), g.shape) == np.unravel_index(ref.argmax(), ref.shape)


def test_correlate_batch_vs_oracle(bank, golden_signals, oracle_dict):
    sigs = golden_signals["signals"]
    got = bank.correlate(sigs).cpu().numpy()
    for i in (0, 13, 63):
        ref = oracle_dict.correlations(sigs[i], "direct")
        assert np.abs(got[i] - ref).max() <= 3e-6 * np.linalg.norm(sigs[i])


def test_similarity_table(bank):
    K, P = bank.K, 738 - bank.L + 1
    for thr in (0.8, 0.4, 0.95):
        bits = bank.similarity_table(738, thr).cpu().numpy().view(np.uint32).reshape(K, K, -1)
        for rec in load_json("simmask.json"):
            if rec["thr"] != thr:
                continue
            row = bits[rec["atom"]]                                    # [K][W]
            dp = np.arange(P) - rec["pos"] + P - 1                      # [P]
            sup = (row[:, dp >> 5] >> (dp & 31).astype(np.uint32)[None, :]) & 1          # [K][P]
            zeros = np.flatnonzero(sup.T.ravel())                        # flattened (p, k)
            assert zeros.tolist() == rec["zeros"], (rec["atom"], rec["pos"], thr)


def _compare_tree(res, s, ref, c, K):
    """every node (pos, atom, mse) of the reference tree dump vs the GPU node tables"""
    for path, (pos, atom, mse, gap) in ref["nodes"].items():
        digits = [int(x) for x in path.split("-")]
        depth = len(digits)
        j = sum((d - 1) * c ** i for i, d in enumerate(digits))
        nid = res.node_id(depth, j)
        got = (int(res.node_pos[s, nid]), int(res.node_atom[s, nid]))
        assert got == (pos, atom), (s, path, got, (pos, atom), gap)
        assert abs(res.node_mse[s, nid] - mse) <= 1e-9 * mse, (s, path)


@pytest.mark.parametrize("key,c,K,B,thr,sl", [
    ("c3_K5_B3_thr0.8", 3, 5, 3, 0.8, slice(0, None, 1)),
    ("c3_K5_B10_thr0.8", 3, 5, 10, 0.8, slice(0, None, 4)),
    ("c2_K4_B16_thr0.4", 2, 4, 16, 0.4, slice(1, None, 8)),
])
def test_mmpdf_trees_vs_reference(bank, golden_signals, key, c, K, B, thr, sl):
    refs = load_json("trees.json")[key]
    sigs = golden_signals["signals"][sl]
    res = bank.mmpdf_batch(sigs, K, c, B, thr)
    assert res.stats[0] == 0 and res.stats[1] == 0
    n_nodes = sum(res.level_sizes)
    for s, ref in enumerate(refs):
        assert len(ref["nodes"]) == n_nodes
        _compare_tree(res, s, ref, c, K)
        # per-sparsity winner of the FULL-depth tree vs the reference record of the same tree
        leaf = int(res.rec_leaf[s, K - 1, 0])
        rec = ref["record"]
        if res.path_str(K, leaf) != rec["path"]:
            # two leaves with the same support have mathematically equal MSE (rounding decides)
            assert abs(res.rec_mse[s, K - 1] - rec["mse"]) <= 1e-12 * rec["mse"]
        else:
            assert [p for p, _ in res.branch(s, K, leaf)] == [a["x"] for a in rec["atoms"]]
        assert abs(res.rec_mse[s, K - 1] - rec["mse"]) <= 1e-9 * rec["mse"]


def test_mmpdf_deep_trees(bank, golden_signals):
    deep = load_json("deep.json")
    sigs = golden_signals["signals"][3::8]
    res = bank.mmpdf_batch(sigs, 7, 4, 4, 0.8)
    assert res.level_sizes == [4] * 7
    for s, ref in enumerate(deep["c4_K7_B4_thr0.8"]):
        _compare_tree(res, s, ref, 4, 7)
    sigs = golden_signals["signals"][3::16]
    res = bank.mmpdf_batch(sigs, 7, 4, 64, 0.8)
    assert res.level_sizes == [4, 16, 64, 64, 64, 64, 64] and sum(res.level_sizes) == 340
    for s, ref in enumerate(deep["c4_K7_B64_thr0.8"]):
        assert len(ref["nodes"]) == 340
        _compare_tree(res, s, ref, 4, 7)


# ---- the fp16 filter at its edges: scales outside its range, bands too crowded to settle ------------------
def _oracle_nodes_match(res, od, sigs, K, c, B, thr, rel=1e-9):
    from oracle import csc_oracle as O
    for s, sig in enumerate(sigs):
        t = O.OracleTree(od, sig, K, c, thr, "direct")
        t.run(B)
        for path, (pos, atom, mse) in t.all_nodes().items():
            digits = [int(x) for x in path.split("-")]
            nid = res.node_id(len(digits), sum((o - 1) * c ** i for i, o in enumerate(digits)))
            assert (int(res.node_pos[s, nid]), int(res.node_atom[s, nid])) == (pos, atom), (s, path)
            assert abs(res.node_mse[s, nid] - mse) <= rel * mse + 1e-300, (s, path)


@pytest.mark.parametrize("scale", [1e-4, 3e-3, 30.0, 300.0])
def test_scaled_dictionary_vs_oracle(golden_bank, golden_signals, oracle_dict, scale):
    """atom norms 1e-4 / 3e-3 / 30 / 300: the filter's error bound scales with max ||a_k|| (30); at 3e-3 most atom
    samples are subnormal in fp16 (the bound counts 2^-25 absolute for those: the fp16 MFMA honours subnormal
    inputs); outside [1e-3, 1e2] the nodes go straight to the fp32 product.  Selections equal the oracle's on the
    scaled bank"""
    import copy
    from csc_amd.engine import DeviceBank
    od = copy.copy(oracle_dict)
    od.bank = oracle_dict.bank * scale
    bank = DeviceBank(od.bank)
    sigs = golden_signals["signals"][5::16]
    res = bank.mmpdf_batch(sigs, 4, 3, 3, 0.8)
    assert res.stats[0] == 0
    _oracle_nodes_match(res, od, sigs, 4, 3, 3, 0.8)


@pytest.mark.parametrize("amp", [1e-17, 1e16])
def test_extreme_signal_amplitudes_vs_oracle(bank, golden_signals, oracle_dict, amp):
    """||r||^2 outside [1e-30, 1e30]: no fp16 scaling is attempted, the fp32 product decides"""
    sigs = golden_signals["signals"][2::16] * amp
    res = bank.mmpdf_batch(sigs, 3, 2, 2, 0.8)
    _oracle_nodes_match(res, oracle_dict, sigs, 3, 2, 2, 0.8)


def test_crowded_band_falls_back_to_fp32(golden_bank, golden_signals, oracle_dict):
    """143 near-copies of one atom (relative differences 1e-3): every atom sits inside the fp16 band of the best
    one, more than the filter's candidate list holds, so the nodes are multiplied again in fp32 (stats[3]) and
    still select what the oracle selects"""
    import copy
    from csc_amd.engine import DeviceBank
    rng = np.random.default_rng(5)
    base = oracle_dict.bank[70]
    od = copy.copy(oracle_dict)
    od.bank = base[None, :] * (1.0 + 1e-3 * rng.standard_normal((143, 1))) + 1e-4 * rng.standard_normal((143, base.size)) * (base != 0)
    sigs = golden_signals["signals"][1::16]
    for fp32_only in (False, True):
        bank = DeviceBank(od.bank)
        bank.fp32_only = fp32_only
        res = bank.mmpdf_batch(sigs, 2, 1, 1, 1.0)
        assert res.stats[0] == 0
        assert (res.stats[3] > 0) == (not fp32_only)
        _oracle_nodes_match(res, od, sigs, 2, 1, 1, 1.0)


def test_noiseless_signals_both_paths(bank, golden_bank, oracle_dict):
    """exact 3-atom signals: the first three levels equal the oracle's; below that the residual is rounding noise
    (its arg-max is not a property of the algorithm), but the two correlation paths must still agree bit for bit
    because both decide on the same float64 re-score"""
    from csc_amd.engine import DeviceBank
    rng = np.random.default_rng(3)
    b = golden_bank["bank"]
    sigs = np.zeros((6, 738))
    for s in range(6):
        for p, k, c in zip(rng.choice(247, 3, replace=False), rng.choice(143, 3, replace=False), (1.0, 0.7, 0.4)):
            sigs[s, p:p + b.shape[1]] += c * b[k]
    res = bank.mmpdf_batch(sigs, 5, 1, 1, 1.0)
    other = DeviceBank(b)
    other.fp32_only = not bank.fp32_only
    res2 = other.mmpdf_batch(sigs, 5, 1, 1, 1.0)
    assert np.array_equal(res.node_pos, res2.node_pos) and np.array_equal(res.node_atom, res2.node_atom)
    assert np.array_equal(res.node_mse, res2.node_mse)
    _oracle_nodes_match(type("R", (), {"node_id": res.node_id, "node_pos": res.node_pos[:, :3], "node_atom": res.node_atom[:, :3],
                                       "node_mse": res.node_mse[:, :3]})(), oracle_dict, sigs, 3, 1, 1, 1.0, rel=1e-6)


def test_random_shapes_correlate_project_match_vs_oracle():
    """the stand-alone entry points on random dictionaries (fewer atoms than padding columns included) and signal
    lengths: csc_correlate, csc_project_batch, csc_match_batch against the oracle"""
    from csc_amd.engine import DeviceBank
    from helpers import OracleBank
    from oracle import csc_oracle as O
    rng = np.random.default_rng(314)
    b_all, y_all = np.arange(0.005, 0.06, 0.005), np.arange(0.010, 0.036, 0.002)
    for trial in range(5):
        b = sorted(rng.choice(b_all, rng.integers(1, 9), replace=False).tolist())
        y = sorted(rng.choice(y_all, rng.integers(2, 14), replace=False).tolist())
        od = O.OracleDictionary(b, y, [1])
        K, L = od.bank.shape
        P = int(rng.integers(1, 400))
        N = L + P - 1
        bank, ob = DeviceBank(od.bank), OracleBank(od)
        sigs = rng.standard_normal((7, N))
        # correlations
        got = bank.correlate(sigs).cpu().numpy()
        for s in range(7):
            want = od.correlations(sigs[s], "direct")
            assert got[s].shape == want.shape == (P, K)
            assert np.abs(got[s] - want).max() <= 3e-6 * np.linalg.norm(sigs[s]) * np.sqrt((od.bank ** 2).sum(1).max())
        # projections on random supports (some padded, some with a repeated activation)
        kmax = int(rng.integers(1, 7))
        M = 12
        sup = np.full((M, kmax, 2), -1, np.int32)
        for m in range(M):
            n = int(rng.integers(1, kmax + 1))
            sup[m, :n, 0] = rng.integers(0, P, n)
            sup[m, :n, 1] = rng.integers(0, K, n)
            if n > 1 and rng.random() < 0.3:
                sup[m, n - 1] = sup[m, 0]
        idx = rng.integers(0, 7, M).astype(np.int32)
        coef, mse, approx, _ = bank.project_batch(sigs, sup, idx, want_approx=True)
        for m in range(M):
            s_ = [(int(p), int(k)) for p, k in sup[m] if p >= 0]
            uniq = list(dict.fromkeys(s_))
            ap, c = od.project(sigs[idx[m]], uniq, N, "direct")
            want_mse = float(np.mean((sigs[idx[m]] - ap) ** 2))
            if np.linalg.cond(np.array([[np.dot(od.place(k1, p1, N), od.place(k2, p2, N)) for (p2, k2) in uniq]
                                        for (p1, k1) in uniq])) < 1e8:
                assert abs(mse[m] - want_mse) <= 1e-8 * want_mse, (trial, m)
                assert np.abs(approx[m] - ap).max() <= 1e-7 * np.abs(ap).max() + 1e-12
        # matching
        kt, ka = int(rng.integers(1, 6)), int(rng.integers(1, 6))
        T = np.stack([np.stack([rng.integers(0, P, kt), rng.integers(0, K, kt)], 1) for _ in range(20)]).astype(np.int32)
        A = T[:, :min(kt, ka)].copy()
        A = np.concatenate([A, np.stack([np.stack([rng.integers(0, P, ka - A.shape[1]), rng.integers(0, K, ka - A.shape[1])], 1)
                                         for _ in range(20)]).astype(np.int32)], 1) if ka > A.shape[1] else A
        A[:, :, 0] = np.clip(A[:, :, 0] + rng.integers(-3, 4, A.shape[:2]), 0, P - 1)
        got_tp = bank.match_batch(T, A, 2, 0.7)
        assert np.array_equal(got_tp, ob.match_batch(T, A, 2, 0.7)), trial
