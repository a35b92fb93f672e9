"""GPU test of csc_mmpdf_batch_static, the entry point that allocates nothing (include/csc_hip.h): called through
the C ABI on two batches in a row -- the second smaller, so that the static device scratch still holds the first
call's residuals and chain flags -- its node tables and records must equal csc_mmpdf_batch's bit for bit, and the
calls themselves must not allocate device memory.  Collected last (test_zz_*): the round's end runs pytest -x."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

NAMES = ("node_pos", "node_atom", "node_mse", "node_coef", "rec_buf", "stats")


def _ptr(t):
    return C.c_void_p(t.data_ptr() if t is not None else 0)


def test_static_scratch_entry_equals_workspace_entry(golden_signals):
    import torch
    import csc_amd
    from csc_amd import _lib
    d = csc_amd.ZSDictionary.from_values(np.arange(0.005, 0.06, 0.005), np.arange(0.010, 0.036, 0.002), [1])
    bank = d.device_bank()
    lib, dev = bank.lib, bank.device
    assert lib.csc_static_scratch_bytes() == 64 << 20
    sigs = golden_signals["signals"]
    big = torch.from_numpy(np.concatenate([sigs, sigs[::-1] * 0.5, sigs * 2.0])).to(dev)      # 192 signals
    small = torch.from_numpy(np.ascontiguousarray(sigs[5:42])).to(dev)                        # 37 signals
    calls = 0
    for (c, B, K) in ((3, 3, 5), (4, 64, 4)):
        for sig in ((big, small) if B == 3 else (big[:60], small[:20])):    # second call: smaller input, dirty scratch
            ref = bank.mmpdf_launch(sig, K, c, B, 0.8, want_coef=True)
            # fresh output tensors with the same layout; no workspace is passed
            S, N = sig.shape
            _, NN = _lib.tree_shape(K, c, B)
            mine = {"node_pos": torch.full((S, NN), -7, dtype=torch.int32, device=dev),
                    "node_atom": torch.full((S, NN), -7, dtype=torch.int32, device=dev),
                    "node_mse": torch.full((S, NN), -7.0, dtype=torch.float64, device=dev),
                    "node_coef": torch.full((S, NN, K), -7.0, dtype=torch.float64, device=dev),
                    "rec_buf": torch.zeros((S, K * (16 + 8 * K)), dtype=torch.uint8, device=dev),
                    "stats": torch.zeros(4, dtype=torch.int32, device=dev)}
            sim = bank.similarity_table(N, 0.8)
            torch.cuda.synchronize(dev)
            before = torch.cuda.memory_stats(dev)["allocation.all.allocated"], torch.cuda.mem_get_info(dev)[0]
            rc = lib.csc_mmpdf_batch_static(_ptr(bank.bank32t), _ptr(bank.bank64), _ptr(sim), bank.K, bank.L, _ptr(sig), S, N,
                                            K, c, B, 4, _ptr(mine["node_pos"]), _ptr(mine["node_atom"]),
                                            _ptr(mine["node_mse"]), _ptr(mine["node_coef"]), _ptr(mine["rec_buf"]), None,
                                            None, _ptr(mine["stats"]), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            _lib.check(rc, "csc_mmpdf_batch_static")
            torch.cuda.synchronize(dev)
            after = torch.cuda.memory_stats(dev)["allocation.all.allocated"], torch.cuda.mem_get_info(dev)[0]
            assert after[0] == before[0], "the call went through torch's allocator"
            calls += 1
            if calls > 1:                                          # (the first call may page the library's image in)
                assert after[1] >= before[1] - (1 << 20), "device memory was allocated by the call"
            for n in NAMES:
                # stats[2], the number of re-scored candidates, is a diagnostic: a group's candidates are taken against
                # the maximum the other waves have PUBLISHED so far, so a few entries more or fewer are re-scored from
                # run to run (never one too few); everything that reaches an output is bit-identical
                if n == "stats":
                    assert torch.equal(mine[n][[0, 1, 3]], ref[n][[0, 1, 3]]), (n, c, B, K, S)
                    assert abs(int(mine[n][2]) - int(ref[n][2])) <= max(64, int(ref[n][2]) // 50), (n, c, B, K, S)
                else:
                    assert torch.equal(mine[n], ref[n]), (n, c, B, K, S)
    # sizes beyond the static array are refused with a clear error, nothing is launched
    S_big = 4000
    sig = torch.zeros((S_big, 738), dtype=torch.float64, device=dev)
    need = lib.csc_mmpdf_workspace_bytes(S_big, 738, 5, 3, 3)
    assert need > lib.csc_static_scratch_bytes()
    rc = lib.csc_mmpdf_batch_static(_ptr(bank.bank32t), _ptr(bank.bank64), _ptr(bank.similarity_table(738, 0.8)), bank.K,
                                    bank.L, _ptr(sig), S_big, 738, 5, 3, 3, 0, _ptr(sig), _ptr(sig), _ptr(sig), None, None,
                                    None, None, None, None)
    assert rc == -10 and b"static" in lib.csc_last_error()
    # the high-level switch goes the same way
    a = bank.mmpdf_launch(small, 5, 3, 3, 0.8, want_coef=False, static_scratch=True)
    b = bank.mmpdf_launch(small, 5, 3, 3, 0.8, want_coef=False)
    assert torch.equal(a["rec_buf"], b["rec_buf"])


def test_static_scratch_entry_beyond_one_register_tile():
    """The same entry point on a dictionary of 330 atoms and 260 positions (screen + three-product fp16 kernels in front
    of the fp32 tiled kernel: the screen's bounds travel through the static scratch too), two calls in a row, the second
    smaller: records equal to the workspace entry's."""
    import torch
    import csc_amd
    d = csc_amd.ZSDictionary.from_values(np.arange(0.005, 0.06, 0.0025), np.arange(0.010, 0.036, 0.0018), [1])
    bank64 = d.getLocalDictionary()
    K, L = bank64.shape
    P = 260
    rng = np.random.default_rng(3)
    sigs = 0.02 * rng.standard_normal((48, L + P - 1))
    for row in sigs:
        for _ in range(3):
            row[rng.integers(0, P):][:L] += rng.uniform(0.5, 1.5) * bank64[rng.integers(K)]
    bank = d.device_bank()
    for n in (48, 17):
        sig = torch.from_numpy(np.ascontiguousarray(sigs[:n])).to(bank.device)
        a = bank.mmpdf_launch(sig, 4, 3, 10, 0.7, want_coef=True, static_scratch=True)
        b = bank.mmpdf_launch(sig, 4, 3, 10, 0.7, want_coef=True)
        for name in NAMES:
            assert torch.equal(a[name], b[name]), (name, n)


def test_static_scratch_entry_under_stream_capture(golden_signals):
    """What the entry point is for: the whole tree search of a batch captured ONCE into a HIP graph (no allocation, no
    synchronisation, no host read-back inside the call) and replayed on new signals -- node tables and records equal to
    csc_mmpdf_batch's on both replays."""
    import torch
    import csc_amd
    from csc_amd import _lib
    d = csc_amd.ZSDictionary.from_values(np.arange(0.005, 0.06, 0.005), np.arange(0.010, 0.036, 0.002), [1])
    bank = d.device_bank()
    lib, dev = bank.lib, bank.device
    sigs = golden_signals["signals"]
    first = torch.from_numpy(np.concatenate([sigs, sigs[::-1] * 0.5])).to(dev)                 # 128 signals
    second = torch.from_numpy(np.concatenate([sigs * 3.0, sigs[::-1]])).to(dev)
    K, c, B = 5, 3, 3
    S, N = first.shape
    _, NN = _lib.tree_shape(K, c, B)
    sig = torch.zeros_like(first)
    mine = {"node_pos": torch.zeros((S, NN), dtype=torch.int32, device=dev),
            "node_atom": torch.zeros((S, NN), dtype=torch.int32, device=dev),
            "node_mse": torch.zeros((S, NN), dtype=torch.float64, device=dev),
            "node_coef": torch.zeros((S, NN, K), dtype=torch.float64, device=dev),
            "rec_buf": torch.zeros((S, K * (16 + 8 * K)), dtype=torch.uint8, device=dev),
            "stats": torch.zeros(4, dtype=torch.int32, device=dev)}
    sim = bank.similarity_table(N, 0.8)
    bank.mmpdf_launch(first[:8], K, c, B, 0.8, want_coef=True, static_scratch=True)            # modules loaded, attributes set
    torch.cuda.synchronize(dev)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        mine["stats"].zero_()
        rc = lib.csc_mmpdf_batch_static(_ptr(bank.bank32t), _ptr(bank.bank64), _ptr(sim), bank.K, bank.L, _ptr(sig), S, N,
                                        K, c, B, 4, _ptr(mine["node_pos"]), _ptr(mine["node_atom"]),
                                        _ptr(mine["node_mse"]), _ptr(mine["node_coef"]), _ptr(mine["rec_buf"]), None,
                                        None, _ptr(mine["stats"]), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    _lib.check(rc, "csc_mmpdf_batch_static under capture")
    for data in (first, second):
        sig.copy_(data)
        for n in NAMES[:-1]:
            mine[n].fill_(0)
        graph.replay()
        torch.cuda.synchronize(dev)
        ref = bank.mmpdf_launch(data, K, c, B, 0.8, want_coef=True)
        for n in NAMES[:-1]:
            assert torch.equal(mine[n], ref[n]), n
        assert torch.equal(mine["stats"][[0, 1, 3]], ref["stats"][[0, 1, 3]])
