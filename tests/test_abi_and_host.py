"""CPU tests: the C-ABI library loads and exports every symbol include/csc_hip.h declares; host
logic of the package (tree shape, path mapping, record packing, sharding, atom bank, generator)."""
import hashlib
import itertools
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_json


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "csc_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(csc_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from csc_amd import _lib
    lib = _lib.lib()
    names = _declared_symbols()
    assert len(names) >= 19
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(_lib.EXPORTS) == names
    assert lib.csc_abi_version() == 8


def test_header_cites_reference_for_every_entry_point():
    src = open(os.path.join(ROOT, "include", "csc_hip.h")).read()
    assert src.count("src/csc/") >= 10


def test_tree_shape_closed_form():
    from csc_amd import _lib
    assert _lib.tree_shape(5, 3, 3) == ([3, 3, 3, 3, 3], 15)
    assert _lib.tree_shape(5, 3, 10) == ([3, 9, 10, 10, 10], 42)
    assert _lib.tree_shape(7, 4, 64) == ([4, 16, 64, 64, 64, 64, 64], 340)
    assert _lib.tree_shape(3, 1, 1) == ([1, 1, 1], 3)
    with pytest.raises(_lib.CscHipError, match="branches"):
        _lib.tree_shape(5, 3, 0)
    with pytest.raises(_lib.CscHipError, match="max_sparsity"):
        _lib.tree_shape(17, 3, 3)
    with pytest.raises(_lib.CscHipError, match="connections"):
        _lib.tree_shape(5, 9, 3)


def test_sizes_helpers():
    from csc_amd import _lib
    lib = _lib.lib()
    # per 144-atom tile: 16 slabs x 9 column blocks x (32 x 16) floats + 208 ints of metadata + the same blocks in fp16;
    # then max_k ||a_k||^2 (+ spare) and ||a_k||^2 as doubles
    # ... and the bank as a scaled two-term fp16 number (hi + lo) for the tiled kernel's three-product filter
    assert lib.csc_bank32_elems(143, 492) == 512 * 144 + 208 + 256 * 144 + 16 + 2 * 144 + 512 * 144
    # more than one tile: + the low-rank section (the basis as two-term fp16 fragments: 16 k-steps x 4 column blocks x 512
    # halves, hi and lo; per 16-column block of the bank 2 k-steps of fp16 coefficient fragments = 2 x 256 floats; one
    # slack float per column) + ceil(K / 32) words of atom summary per atom + lowest / highest similar shift per atom
    assert lib.csc_bank32_elems(145, 492) == (2 * (512 * 144 + 208 + 256 * 144) + 16 + 2 * 146 + 2 * 512 * 144
                                              + 16 * 4 * 512 + 2 * 9 * 2 * 256 + 2 * 144 + 145 * 5 + 2 * 145)
    assert lib.csc_sim_words(738, 492) == 2 * ((2 * 247 - 1 + 63) // 64)
    assert lib.csc_sim_words(100, 492) == 0
    # per frontier node: N residual samples + the packed Cholesky factor (15 doubles at depth 5)
    # + max_sparsity coefficients, two generations
    # (+ one flag byte per frontier node, rounded up to 16)
    assert lib.csc_mmpdf_workspace_bytes(10, 738, 5, 3, 3) == 2 * 10 * 3 * (738 + 15 + 5) * 8 + 32


def test_candidate_paths_match_reference_table():
    from csc_amd import MMPTree
    from csc_amd.engine import path_digits
    for key, paths in load_json("paths.json").items():
        c, K = map(int, key.split("_"))
        for n, p in enumerate(paths, start=1):
            assert list(MMPTree.getCandidatePath(K, c, n)) == p
            assert MMPTree.getCandidateNumber(K, c, p) == n
            assert path_digits(n - 1, K, c) == p
    assert MMPTree.getComputedNodesFromPath((1, 1, 1)) == 3
    assert MMPTree.getComputedNodesFromPath((2, 1, 1)) == 3
    assert MMPTree.getComputedNodesFromPath((1, 2, 1)) == 2
    assert MMPTree.getComputedNodesFromPath((1, 1, 2)) == 1
    assert sum(MMPTree.getComputedNodesFromTree(3, 2)) == 2 + 4 + 8


def test_record_roundtrip():
    from csc_amd import _lib
    from csc_amd.engine import BatchResult
    rng = np.random.default_rng(1)
    for want_coef in (False, True):
        sizes, NN = _lib.tree_shape(4, 3, 5)
        S = 6
        r = BatchResult(4, 3, 5, sizes, [0, 3, 8, 13], rng.integers(0, 247, (S, NN)).astype(np.int32),
                        rng.integers(0, 143, (S, NN)).astype(np.int32), rng.random((S, NN)),
                        rng.random((S, NN, 4)) if want_coef else None, rng.random((S, 4)),
                        rng.integers(0, 5, (S, 4, 2)).astype(np.int32),
                        rng.integers(-1, 247, (S, 4, 4, 2)).astype(np.int32), np.zeros(4, np.int32))
        raw = r.to_records()
        assert raw.dtype == np.uint8 and raw.shape[0] == S
        back = BatchResult.from_records(raw[[4, 1]], 4, 3, 5, want_coef)
        assert back.level_offsets == [0, 3, 8, 13]
        for name in BatchResult._FIELDS:
            a, b = getattr(r, name), getattr(back, name)
            if a is None:
                assert b is None
            else:
                assert np.array_equal(a[[4, 1]], b), name


def test_branch_and_path_indexing():
    from csc_amd.engine import BatchResult
    # 2 connections, depth 3, 5 branches: level sizes 2, 4, 5
    pos = np.arange(11, dtype=np.int32)[None]
    r = BatchResult(3, 2, 5, [2, 4, 5], [0, 2, 6], pos, pos + 100, np.zeros((1, 11)), None, None, None, None, None)
    assert r.path_str(3, 4) == "1-1-2"
    assert r.branch(0, 3, 4) == [(0, 100), (2, 102), (10, 110)]       # j=4: level1 j%2=0, level2 j%4=0, level3 j=4
    assert r.branch(0, 3, 3) == [(1, 101), (5, 105), (9, 109)]
    assert r.path_str(2, 3) == "2-2"


def test_shard_ranges_cover_in_order():
    from csc_amd.distributed import shard_range, shard_sizes
    for n, w in itertools.product([0, 1, 7, 8, 3200, 1000003], [1, 2, 3, 8]):
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        assert max(shard_sizes(n, w)) - min(shard_sizes(n, w)) <= 1


def test_atom_bank_bit_identical_to_reference(golden_bank):
    import csc_amd
    d = csc_amd.ZSDictionary.from_values(np.arange(0.005, 0.06, 0.005), np.arange(0.010, 0.036, 0.002), [1])
    assert str(d) == str(golden_bank["dict_str"]) and len(d) == 143 and d.getAtomsLength() == 492
    assert hashlib.sha256(np.ascontiguousarray(d.getLocalDictionary()).tobytes()).hexdigest() == str(golden_bank["sha256"])
    assert np.array_equal([a.sigma for a in d.getAtoms()], golden_bank["sigma"])
    assert np.array_equal([a.var for a in d.getAtoms()], golden_bank["var"])
    assert csc_amd.ZSAtom is csc_amd.CSCAtom and csc_amd.ZSDictionary is csc_amd.CSCDictionary
    a = d.getAtom(17)
    assert (a.b, a.y, a.sigma) in d and (a.b, a.y, 1) not in d and a in d and d.getActivationIdxFromAtom(a, 3) == 3 * 143 + 17
    assert d.getActivationIdxFromParams({"x": 5, "b": a.b, "y": a.y, "s": 1}) == 5 * 143 + 17
    assert d.getAtomFromActivationIdx(5 * 143 + 17) == (a, 5)
    with pytest.raises(TypeError):
        3 in d


def test_generator_reproduces_reference_config1_signals():
    """the product's constrained generator draws exactly what the reference drew (seed 0)"""
    import csc_amd
    ref = load_json("omp200.json")
    d = csc_amd.ZSDictionary.from_values(np.arange(0.005, 0.06, 0.005), np.arange(0.010, 0.036, 0.002), [1])
    np.random.seed(ref["seed"])
    sigs = [d.generateConstrainedTestSignal(738, 3, 10, 10, 0.75) for _ in range(200)]
    arr = np.array([s for s, _ in sigs])
    assert hashlib.sha256(arr.tobytes()).hexdigest() == ref["signals_sha256"]
    assert [[{k: (int(v) if k == "x" else v) for k, v in a.items()} for a in infos] for _, infos in sigs] == ref["truth"]


def test_no_cpu_fallback_without_gpu():
    """the product path must fail loudly, not fall back, when there is no GPU"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import csc_amd
    d = csc_amd.ZSDictionary.from_values([0.01], [0.02], [1])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        d.omp(np.zeros(2000), 1)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "csc_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), fn


def test_torch_custom_ops_registered_with_shape_inference():
    """csc_amd.ops registers torch.ops.csc_amd.{bank_pack, similarity_table, mmpdf_batch}; their fake implementations
    infer the output shapes without a GPU (what torch.compile / export need)."""
    import torch
    import csc_amd.ops  # noqa: F401
    from torch._subclasses.fake_tensor import FakeTensorMode
    with FakeTensorMode():
        b64 = torch.empty((143, 492), dtype=torch.float64, device="cuda")
        b32 = torch.ops.csc_amd.bank_pack(b64)
        sim = torch.ops.csc_amd.similarity_table(b64, 738, 0.8)
        sig = torch.empty((10, 738), dtype=torch.float64, device="cuda")
        pos, atom, mse, rec = torch.ops.csc_amd.mmpdf_batch(b32, b64, sim, sig, 5, 3, 3, 0)
    from csc_amd import _lib
    assert b32.shape == (_lib.lib().csc_bank32_elems(143, 492),) and sim.shape == (143 * 143 * 16,)
    assert pos.shape == atom.shape == mse.shape == (10, 15) and rec.shape == (10, 280) and rec.dtype == torch.uint8


def test_json_writer_threads_write_the_same_bytes(tmp_path, monkeypatch):
    """Lists of >= 8192 records are formatted by several threads, each chunk written at its own offset: the file must be
    the one a single thread writes (and the one json.loads reads back), in append mode too; a record that names an atom
    outside the dictionary fails the call with -3."""
    import ctypes as C
    import json
    from csc_amd import _lib
    lib = _lib.lib()
    S, K, c, A = 20011, 5, 3, 143
    rng = np.random.default_rng(4)
    rec = np.zeros((S, K * (16 + 8 * K)), dtype=np.uint8)
    mse = rng.random((S, K)) * 10.0 ** rng.integers(-8, 3, (S, K))
    leaf = rng.integers(0, 3, (S, K, 2)).astype(np.int32)
    sel = np.zeros((S, K, K, 2), dtype=np.int32)
    sel[..., 0] = rng.integers(0, 247, (S, K, K))
    sel[..., 1] = rng.integers(0, A, (S, K, K))
    rec[:, :8 * K] = mse.view(np.uint8).reshape(S, -1)
    rec[:, 8 * K:16 * K] = leaf.view(np.uint8).reshape(S, -1)
    rec[:, 16 * K:] = sel.view(np.uint8).reshape(S, -1)
    ids = np.arange(S, dtype=np.int64)[::-1].copy()
    snr = rng.integers(0, 16, S).astype(np.float64)
    isint = (rng.random(S) < 0.5).astype(np.uint8)
    b, y, s_ = rng.random(A), rng.random(A), np.ones(A)
    p = lambda x: x.ctypes.data_as(C.c_void_p)

    def write(path, threads, append=0, records=rec):
        monkeypatch.setenv("CSC_JSON_THREADS", str(threads))
        return lib.csc_sparvar_json_write(str(path).encode(), append, p(records), S, K, c, p(ids), p(snr), p(isint),
                                          C.c_double(0.25), p(b), p(y), p(s_), A, 1, 1)

    one, many = tmp_path / "one.json", tmp_path / "many.json"
    assert write(one, 1) == 0 and write(many, 5) == 0
    data = open(one, "rb").read()
    assert data == open(many, "rb").read()
    got = json.loads(b'{"mmp": ' + data)["mmp"]
    assert len(got) == S and got[0]["id"] == S - 1 and got[-1]["results"][4]["mse"] == mse[-1, 4]
    assert [a["x"] for a in got[7]["results"][2]["atoms"]] == sel[7, 2, :3, 0].tolist()
    head = b'{\n    "mmp": '
    open(many, "wb").write(head)
    assert write(many, 3, append=1) == 0
    assert open(many, "rb").read() == head + data
    bad = rec.copy()
    bad[S - 5, 16 * K + 4:16 * K + 8] = np.array([A], dtype=np.int32).view(np.uint8)
    assert write(tmp_path / "bad.json", 4, records=bad) == -3
    assert b"outside the dictionary" in lib.csc_json_last_error()
