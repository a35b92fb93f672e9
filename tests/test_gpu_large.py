"""GPU parity of the TILED level kernel (dictionaries beyond one 256 x 144 register tile).

* BASELINE config 3: the 9570-atom dictionary (L = 518, N = 777, P = 260) against whole-tree dumps
  generated from the reference (tests/golden/make_golden_large.py); the 40 MB bank is rebuilt here and
  pinned by its sha256.
* a 2-row-block x 3-column-tile dictionary against the CPU oracle with 3 connections / 10 branches, so
  that sibling suppression, pruning and candidate collection cross unit boundaries.
"""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, load_json

pytestmark = pytest.mark.gpu


def _check_nodes(res, s, nodes, c):
    for path, node in nodes.items():
        pos, atom, mse = node[0], node[1], node[2]
        digits = [int(x) for x in path.split("-")]
        nid = res.node_id(len(digits), sum((o - 1) * c ** i for i, o in enumerate(digits)))
        got = (int(res.node_pos[s, nid]), int(res.node_atom[s, nid]))
        assert got == (pos, atom), (s, path, got, (pos, atom))
        assert abs(res.node_mse[s, nid] - mse) <= 1e-9 * mse, (s, path)


@pytest.fixture(scope="module")
def large_dic():
    import csc_amd
    return csc_amd.ZSDictionary.from_values(np.arange(0.005, 0.06, 0.0005), np.arange(0.010, 0.036, 0.0003), [1])


def test_config3_trees_vs_reference(large_dic):
    ref = load_json("large.json")
    z = np.load(os.path.join(GOLDEN, "large.npz"))
    assert (len(large_dic), large_dic.getAtomsLength()) == (ref["K"], ref["L"]) == (9570, 518)
    bank = large_dic.getLocalDictionary()
    assert hashlib.sha256(np.ascontiguousarray(bank).tobytes()).hexdigest() == ref["bank_sha256"]
    sigs = z["signals"]
    assert sigs.shape[1] == ref["N"] == 777
    res, _ = large_dic._batch(sigs, 5, 3, 3, 0.8, want_coef=True)
    assert res.stats[0] == 0 and res.stats[1] == 0
    for s, tree in enumerate(ref["c3_K5_B3_thr0.8"]):
        assert len(tree["nodes"]) == 15
        _check_nodes(res, s, tree["nodes"], 3)
        rec = tree["record"]
        leaf = int(res.rec_leaf[s, 4, 0])
        assert abs(res.rec_mse[s, 4] - rec["mse"]) <= 1e-9 * rec["mse"]
        if res.path_str(5, leaf) == rec["path"]:
            got = large_dic._sparvar_records(res, s, 0.0)[4]
            assert got["atoms"] == rec["atoms"]
    # correlations of the stand-alone entry point on the same dictionary (67 column tiles, 2 row blocks)
    c = large_dic.device_bank().correlate(sigs[:1]).cpu().numpy()[0]
    win = np.lib.stride_tricks.sliding_window_view(sigs[0], 518)
    want = win @ bank.T
    assert c.shape == want.shape == (260, 9570)
    assert np.abs(c - want).max() <= 3e-6 * np.linalg.norm(sigs[0])


def test_multi_unit_tree_vs_oracle():
    """K = 330 atoms (3 column tiles, the last partly filled), P = 300 positions (2 row blocks),
    connections 3, branches 10, sparsity 4, dissimilarity 0.7: every node against the CPU oracle."""
    import csc_amd
    from oracle import csc_oracle as O
    b, y = np.arange(0.005, 0.06, 0.0025), np.arange(0.010, 0.036, 0.0018)
    d = csc_amd.ZSDictionary.from_values(b, y, [1])
    od = O.OracleDictionary(b, y, [1])
    assert np.array_equal(d.getLocalDictionary(), od.bank)
    K, L = od.K, od.L
    assert K == 22 * 15
    N = L + 299
    rng = np.random.default_rng(5)
    sigs = []
    for i in range(6):
        sig = 0.02 * rng.standard_normal(N)
        for _ in range(3):
            sig[rng.integers(0, 300):][:L] += rng.uniform(0.5, 1.5) * od.bank[rng.integers(K)]
        sigs.append(sig)
    sigs = np.array(sigs)
    res, _ = d._batch(sigs, 4, 3, 10, 0.7, want_coef=True)
    assert res.level_sizes == [3, 9, 10, 10] and res.stats[0] == 0
    for s in range(len(sigs)):
        t = O.OracleTree(od, sigs[s], 4, 3, 0.7, "direct")
        t.run(10)
        nodes = t.all_nodes()
        assert len(nodes) == 32
        _check_nodes(res, s, nodes, 3)
    # OMP on the same dictionary: amplitudes against the oracle's projection
    res1, _ = d._batch(sigs[:2], 3, 1, 1, 1.0, want_coef=True)
    for s in range(2):
        sup = res1.branch(s, 3, 0)
        _, coef = od.project(sigs[s], sup, N, "direct")
        assert np.allclose(res1.node_coef[s, res1.node_id(3, 0), :3], coef, rtol=1e-9)


def test_random_tiled_shapes_vs_oracle():
    """random dictionaries / signal lengths beyond one register tile (more than 144 atoms and/or more than 256
    positions, remainder row blocks of every size): every node of a few signals against the CPU oracle"""
    import csc_amd
    from oracle import csc_oracle as O
    rng = np.random.default_rng(99)
    for trial in range(5):
        nb, ny = int(rng.integers(6, 20)), int(rng.integers(6, 20))
        b = np.linspace(0.005, 0.058, nb).tolist()
        y = np.linspace(0.010, 0.035, ny).tolist()
        d = csc_amd.ZSDictionary.from_values(b, y, [1])
        od = O.OracleDictionary(b, y, [1])
        K, L = od.bank.shape
        P = int(rng.integers(257, 700)) if (K <= 144 or rng.random() < 0.6) else int(rng.integers(20, 257))
        N = L + P - 1
        c, Kx, B = int(rng.integers(1, 4)), int(rng.integers(2, 5)), int(rng.integers(1, 6))
        sigs = 0.03 * rng.standard_normal((5, N))
        for s in range(5):
            for _ in range(3):
                p = int(rng.integers(0, P))
                sigs[s, p:p + L] += rng.uniform(0.4, 1.2) * od.bank[int(rng.integers(K))]
        res, _ = d._batch(sigs, Kx, c, B, 0.75, want_coef=True)
        assert res.stats[0] == 0
        for s in range(5):
            t = O.OracleTree(od, sigs[s], Kx, c, 0.75, "direct")
            t.run(B)
            _check_nodes(res, s, {k: v for k, v in t.all_nodes().items()}, c)


def test_chunked_batch_is_position_invariant():
    """BASELINE config 5's shape (far more signals than one device call takes): 70 000 signals = the 3200 config-2
    signals repeated, through mmpdfSparVarBatch, which cuts them into calls of 32 768.  A signal's records must not
    depend on where in the batch (which call, which workgroup, next to which neighbours) it sits: every repetition
    equals the first, and the first 3200 equal a plain 3200-signal call."""
    import bench
    dic = bench.canonical_dictionary()
    base = bench.make_signals(dic, 0)
    S0 = len(base)
    n = 70_000
    sig = np.empty((n, base.shape[1]))
    for c0 in range(0, n, S0):
        sig[c0:min(n, c0 + S0)] = base[:min(n, c0 + S0) - c0]
    big = dic.mmpdfSparVarBatch(sig, 3, 0.8, 3, 5)
    ref = dic.mmpdfSparVarBatch(base, 3, 0.8, 3, 5)
    assert big.mse.shape == (n, 5)
    for c0 in range(0, n, S0):
        m = min(n, c0 + S0) - c0
        assert np.array_equal(big.mse[c0:c0 + m], ref.mse[:m]), c0
        assert np.array_equal(big.leaf[c0:c0 + m], ref.leaf[:m]), c0
        assert np.array_equal(big.pos[c0:c0 + m], ref.pos[:m]), c0
        assert np.array_equal(big.atom[c0:c0 + m], ref.atom[:m]), c0
    assert big.path(n - 1, 5) == ref.path((n - 1) % S0, 5)


def test_candidate_overflow_same_choice_in_register_and_tiled_kernels():
    """More entries in the band than the candidate list holds (143 identical atoms: at every position the
    143 correlations are bit-identical, so the maximum is shared by 143 entries): every kernel falls back to the
    fp32 arg-max, lowest flattened index among the entries EQUAL to the maximum -- np.argmax's first occurrence
    on the float64 values, which tie exactly here.  Register-resident kernel (P = 247), the same with the fp16
    filter off, and the tiled kernel (the signal zero-extended to P = 260 positions) must agree with float64."""
    import csc_amd
    from csc_amd.engine import DeviceBank
    d = csc_amd.ZSDictionary.from_values([0.03], [0.02], [1])
    atom = d.getLocalDictionary()[0]
    L = len(atom)
    bank64 = np.tile(atom, (143, 1))
    rng = np.random.default_rng(11)
    N = L + 246
    sigs = rng.standard_normal((6, N)) * 0.01
    for i in range(6):
        sigs[i, 30 * i + 5:30 * i + 5 + L] += atom
    ext = np.concatenate([sigs, np.zeros((6, 13))], axis=1)            # P = 260: tiled kernel
    want = [int(np.argmax(np.lib.stride_tricks.sliding_window_view(s, L) @ bank64.T)) for s in sigs]
    assert want == [(30 * i + 5) * 143 for i in range(6)]
    bank = DeviceBank(bank64)
    got = {}
    for name, (x, fp32) in {"register": (sigs, False), "register fp32": (sigs, True), "tiled": (ext, False)}.items():
        bank.fp32_only = fp32
        r = bank.mmpdf_batch(x, 1, 1, 1, 1.0)
        assert r.stats[0] == 6, (name, r.stats)                         # every selection overflowed
        got[name] = (r.node_pos[:, 0].astype(np.int64) * 143 + r.node_atom[:, 0]).tolist()
    assert got["register"] == got["register fp32"] == got["tiled"] == want, got


def test_three_product_filter_equals_fp32_path(large_dic):
    """Dictionaries beyond one register tile take a node's correlations on the fp16 matrix cores as a three-product
    filter (two-term fp16 operands, mmp_level_tiled16_kernel) and re-score its band in float64; with
    CSC_FLAG_FP32_ONLY the fp32 kernel does everything.  Same selections, bit-identical MSE and coefficients (the
    least squares behind the choice is the same code), on BASELINE config 3's dictionary and on a small multi-unit
    one (remainder row tile, partly filled column tile, 3 connections); and the filter settles (nearly) every node
    itself: stats[3] counts the nodes handed to the fp32 kernel."""
    import csc_amd
    z = np.load(os.path.join(GOLDEN, "large.npz"))
    cases = [(large_dic, z["signals"], (5, 3, 3, 0.8))]
    b, y = np.arange(0.005, 0.06, 0.0025), np.arange(0.010, 0.036, 0.0018)
    d2 = csc_amd.ZSDictionary.from_values(b, y, [1])
    rng = np.random.default_rng(17)
    L2 = d2.getAtomsLength()
    bank2 = d2.getLocalDictionary()
    for P in (260, 272, 200):                                     # remainder tile of 4 and of 16 rows; a single row block
        sig = 0.02 * rng.standard_normal((12, L2 + P - 1))
        for srow in sig:
            for _ in range(3):
                srow[rng.integers(0, P):][:L2] += rng.uniform(0.5, 1.5) * bank2[rng.integers(len(bank2))]
        cases.append((d2, sig, (4, 3, 10, 0.7)))
    for dic, sigs, (Kx, c, B, thr) in cases:
        bank = dic.device_bank()
        try:
            bank.fp32_only = False
            got = bank.mmpdf_batch(sigs, Kx, c, B, thr, want_coef=True)
            bank.fp32_only = True
            ref = bank.mmpdf_batch(sigs, Kx, c, B, thr, want_coef=True)
        finally:
            bank.fp32_only = False
        assert ref.stats[3] == 0 and got.stats[0] == 0
        n_nodes = len(sigs) * (1 + sum(got.level_sizes[:-1]))
        assert got.stats[3] <= 0.05 * n_nodes, (got.stats, n_nodes)
        for name in ("node_pos", "node_atom", "node_mse", "node_coef", "rec_mse", "rec_leaf", "rec_sel"):
            assert np.array_equal(getattr(got, name), getattr(ref, name)), (name, len(bank.bank64))



def test_config3_at_full_size_fp16_front_equals_fp32_path(large_dic):
    """BASELINE config 3 at the size SURVEY 8(d) asks for: 4096 constrained signals (4 atoms, 10 dB) on the 9570-atom
    dictionary, c3 / B3 / K5.  The default path (screen + three-product kernels) and the fp32-only path give identical
    records and node tables, and the fp16 front settles every node itself."""
    np.random.seed(3)
    sigs = np.array([large_dic.generateConstrainedTestSignal(777, 4, 10, 10, 0.75)[0] for _ in range(4096)])
    bank = large_dic.device_bank()
    try:
        bank.fp32_only = False
        got = bank.mmpdf_batch(sigs, 5, 3, 3, 0.8, want_coef=False)
        bank.fp32_only = True
        ref = bank.mmpdf_batch(sigs, 5, 3, 3, 0.8, want_coef=False)
    finally:
        bank.fp32_only = False
    assert got.stats[0] == 0 and got.stats[3] <= 4, got.stats
    for name in ("node_pos", "node_atom", "node_mse", "rec_mse", "rec_leaf", "rec_sel"):
        assert np.array_equal(getattr(got, name), getattr(ref, name)), name


def test_tiled_fp16_front_equals_fp32_on_random_dictionaries():
    """Random dictionaries beyond one register tile (224 ... 1200 atoms; 40 ... 600 positions: one row block with and
    without the 16-row remainder tile, two and three row blocks (one workgroup per CU), partly filled last bank tiles; 1 ... 4 children per node; noisy, noiseless, near-tie and huge
    signals): the default path (screen + three-product fp16 kernels in front of the fp32 tiled kernel) and the fp32-only
    path give the same trees (scripts/fuzz_tiled.py)."""
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    cwd = os.getcwd()
    os.chdir(ROOT)
    try:
        import fuzz_tiled
        assert fuzz_tiled.run(14, 7, verbose=False) == 0
    finally:
        os.chdir(cwd)


def test_similarity_table_packed_equals_float64_kernel():
    """csc_similarity_table_packed (fp32 matrix cores, float64 where the fp32 value is too close to the threshold to
    call) against csc_similarity_table (float64 throughout): the same bits, on the canonical dictionary, on a
    330-atom dictionary with P = 260 (several bank tiles, three blocks of shifts) and on thresholds that sit in the
    middle of the cross-correlations' range (thousands of undecided entries per tile: the queue and its overflow path)."""
    import ctypes as C
    import torch
    import csc_amd
    from csc_amd import _lib

    def ptr(t):
        return C.c_void_p(t.data_ptr())

    cases = [(csc_amd.ZSDictionary.from_values(np.arange(0.005, 0.06, 0.005), np.arange(0.010, 0.036, 0.002), [1]), 738, (0.8, 0.3, 1e-9, -0.5)),
             (csc_amd.ZSDictionary.from_values(np.arange(0.005, 0.06, 0.0025), np.arange(0.010, 0.036, 0.0018), [1]), None, (0.8, 0.65))]
    for d, N, thrs in cases:
        bank = d.device_bank()
        lib = bank.lib
        if N is None:
            N = bank.L + 259
        W = lib.csc_sim_words(N, bank.L)
        for thr in thrs:
            a = torch.full((bank.K * bank.K * W,), 0x5a5a5a5a, dtype=torch.int32, device=bank.device)
            b = torch.full_like(a, 0x33333333)
            stream = C.c_void_p(torch.cuda.current_stream(bank.device).cuda_stream)
            _lib.check(lib.csc_similarity_table(ptr(bank.bank64), bank.K, bank.L, N, thr, ptr(a), stream), "csc_similarity_table")
            _lib.check(lib.csc_similarity_table_packed(ptr(bank.bank32t), ptr(bank.bank64), bank.K, bank.L, N, thr, ptr(b), stream),
                       "csc_similarity_table_packed")
            torch.cuda.synchronize(bank.device)
            assert torch.equal(a, b), (bank.K, N, thr, int((a != b).sum()))
            assert int(a.count_nonzero()) > 0 or thr > 0.9


def test_lowrank_screen_switches_give_identical_trees(large_dic):
    """The low-rank screen in front of the tiled fp16 kernels (csc_bank_lowrank, CSC_FLAG_LOWRANK) and the atom summary of
    the similarity table that tells its later-sibling launches which columns have similarity rows worth reading
    (csc_bank_simatoms, CSC_FLAG_SIMATOMS) change which units are multiplied and which rows are looked up, never what is
    selected: with both, with the screen alone and with neither, every output is the same -- on BASELINE config 3's
    dictionary (golden signals, siblings at the root) and on a 330-atom dictionary with 3 connections / 10 branches
    (siblings on every level, the last one included, where the siblings share one table row), P = 260, 272 and 256
    (remainder tile of 4, of 16 and of no rows).  The summary itself is checked against a restatement from the table."""
    import torch
    import csc_amd
    z = np.load(os.path.join(GOLDEN, "large.npz"))
    cases = [(large_dic, z["signals"], (5, 3, 3, 0.8))]
    d2 = csc_amd.ZSDictionary.from_values(np.arange(0.005, 0.06, 0.0025), np.arange(0.010, 0.036, 0.0018), [1])
    rng = np.random.default_rng(23)
    L2 = d2.getAtomsLength()
    bank2 = d2.getLocalDictionary()
    for P in (260, 272, 256):
        sig = 0.02 * rng.standard_normal((24, L2 + P - 1))
        for srow in sig:
            for _ in range(3):
                srow[rng.integers(0, P):][:L2] += rng.uniform(0.5, 1.5) * bank2[rng.integers(len(bank2))]
        cases.append((d2, sig, (4, 3, 10, 0.7)))
    cases.append((d2, cases[1][1][:12], (3, 4, 16, 0.7)))             # a fourth child: more earlier picks than the gathered bands take
    names = ("node_pos", "node_atom", "node_mse", "node_coef", "rec_mse", "rec_leaf", "rec_sel")
    for dic, sigs, (Kx, c, B, thr) in cases:
        bank = dic.device_bank()
        assert bank.lowrank and bank.simatoms
        try:
            both = bank.mmpdf_batch(sigs, Kx, c, B, thr, want_coef=True)
            bank.simatoms = False
            screen = bank.mmpdf_batch(sigs, Kx, c, B, thr, want_coef=True)
            bank.lowrank = False
            plain = bank.mmpdf_batch(sigs, Kx, c, B, thr, want_coef=True)
        finally:
            bank.lowrank, bank.simatoms = True, True
        assert both.stats[0] == 0 and both.stats[3] == screen.stats[3] == plain.stats[3]
        for name in names:
            assert np.array_equal(getattr(both, name), getattr(plain, name)), (name, bank.K, "summary")
            assert np.array_equal(getattr(screen, name), getattr(plain, name)), (name, bank.K, "screen")
        if bank.K > 1000:
            continue
        # the atom summary (the last K x ceil(K / 32) words of the image) of the table the launch above installed
        N = sigs.shape[1]
        P = N - bank.L + 1
        W = bank.lib.csc_sim_words(N, bank.L)
        table = bank.similarity_table(N, thr).cpu().numpy().reshape(bank.K, bank.K, W)
        similar = (table != 0).any(axis=2)                              # [k][k']
        raw = bank.bank32t.cpu().numpy()
        sa_w = (bank.K + 31) // 32
        tail = raw[len(raw) - bank.K * (sa_w + 2):]
        got = tail[:bank.K * sa_w].view(np.uint32).reshape(bank.K, sa_w)
        bits = np.unpackbits(got.view(np.uint8), axis=1, bitorder="little")[:, :bank.K].astype(bool)
        assert np.array_equal(bits, similar)
        assert similar.diagonal().all() and 0.01 < similar.mean() < 0.9
        # ... and, behind it, the lowest / highest bit set in any of an atom's rows
        rng_got = tail[bank.K * sa_w:].view(np.int32).reshape(2, bank.K)
        anyrow = np.bitwise_or.reduce(table.view(np.uint32), axis=1)          # [k][W]
        shifts = np.unpackbits(anyrow.view(np.uint8), axis=1, bitorder="little")
        want_lo = shifts.argmax(axis=1)
        want_hi = shifts.shape[1] - 1 - shifts[:, ::-1].argmax(axis=1)
        assert np.array_equal(rng_got[0], want_lo) and np.array_equal(rng_got[1], want_hi)
        assert (want_lo <= P - 1).all() and (want_hi >= P - 1).all()           # (an atom is similar to itself at shift 0)
