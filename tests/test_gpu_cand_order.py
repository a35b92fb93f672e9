"""The chain kernel takes its candidates against a running maximum that its four waves publish as they go: WHICH
entries reach the raw list depends on the order the tasks finish in, but the list filtered by the final maximum, and
with it every record, must not.  This test pins that by forcing the two extreme orders in development builds of the
library (csc_amd/csrc/Makefile, `make candcheck`):

  cand0  the shipped order (whatever the hardware gives), plus the dump of the lists
  cand1  the loosest: a task's threshold sees that task's own maximum only (as if it always finished first)
  cand2  the tightest: every task sees the final maximum (the second pass is forced on every node)

on BASELINE.json's config 2 (1024 signals of its 16 cells) and on the c=3 / B=10 / K=10 tree of the golden signals:
identical records in all three and in the shipped library; per node the same final candidate set; and the raw lists
ordered tight <= shipped <= loose.  Each build runs in a process of its own (CSC_HIP_LIB)."""
import itertools
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import ctypes as C, os, sys
import numpy as np
root, src, dst = sys.argv[1:4]
sys.path.insert(0, root)
import torch
import csc_amd
from csc_amd import _lib
d = csc_amd.ZSDictionary.from_values(np.arange(0.005, 0.06, 0.005), np.arange(0.010, 0.036, 0.002), [1])
bank = d.device_bank()
L = bank.lib
assert any(os.path.basename(os.environ["CSC_HIP_LIB"]) in line for line in open("/proc/self/maps"))
L.csc_dbg_cand_read.restype, L.csc_dbg_cand_read.argtypes = C.c_int, [C.c_void_p, C.c_int]
W = L.csc_dbg_cand_width()
data = np.load(src)
out = {}
L.csc_dbg_cand_read(None, 0)                                        # clears the dump
for tag, sig, (K, c, B) in (("c2", data["c2"], (5, 3, 3)), ("deep", data["deep"], (10, 3, 10))):
    res = bank.mmpdf_launch(torch.from_numpy(sig).to(bank.device), K, c, B, 0.8, want_coef=True)
    torch.cuda.synchronize()
    for n in ("node_pos", "node_atom", "node_mse", "node_coef", "rec_buf", "stats"):
        out[tag + "_" + n] = res[n].cpu().numpy()
    dump = np.empty((16384, W), np.int32)
    assert L.csc_dbg_cand_read(dump.ctypes.data, 16384) == 16384
    out[tag + "_dump"] = dump
np.savez(dst, **out)
'''


def test_running_maximum_publication_orders(tmp_path, golden_signals):
    import torch
    import csc_amd
    d = csc_amd.ZSDictionary.from_values(np.arange(0.005, 0.06, 0.005), np.arange(0.010, 0.036, 0.002), [1])
    libs = [os.path.join(ROOT, "csc_amd", "csrc", f"libcsc_hip_cand{i}.so") for i in range(3)]
    assert all(os.path.exists(p) for p in libs), "make -C csc_amd/csrc candcheck"
    np.random.seed(11)
    c2 = np.array([d.generateConstrainedTestSignal(738, sp, snr, 10, 0.75)[0]
                   for snr, sp in itertools.product([0, 5, 10, 15], [2, 3, 4, 5]) for _ in range(64)])
    src = tmp_path / "signals.npz"
    np.savez(src, c2=c2, deep=np.ascontiguousarray(golden_signals["signals"]))
    script = tmp_path / "cand_worker.py"
    script.write_text(WORKER)
    got = []
    for i, lib in enumerate(libs):
        dst = tmp_path / f"cand{i}.npz"
        r = subprocess.run([sys.executable, str(script), ROOT, str(src), str(dst)], env=dict(os.environ, CSC_HIP_LIB=lib),
                           capture_output=True, timeout=600)
        assert r.returncode == 0, r.stderr.decode()[-3000:]
        got.append(np.load(dst))
    bank = d.device_bank()
    for tag, sig, (K, c, B) in (("c2", c2, (5, 3, 3)), ("deep", golden_signals["signals"], (10, 3, 10))):
        ship = bank.mmpdf_launch(torch.from_numpy(np.ascontiguousarray(sig)).to(bank.device), K, c, B, 0.8, want_coef=True)
        for n in ("node_pos", "node_atom", "node_mse", "node_coef", "rec_buf"):
            ref = ship[n].cpu().numpy()
            for i in range(3):
                assert np.array_equal(got[i][f"{tag}_{n}"], ref), (tag, n, i)
        for i in range(3):
            assert np.array_equal(got[i][f"{tag}_stats"][[0, 1, 3]], ship["stats"].cpu().numpy()[[0, 1, 3]]), (tag, i)
        shipped, loose, tight = (got[i][f"{tag}_dump"] for i in range(3))
        written = shipped[:, 1] >= 0
        assert written.sum() >= 4 * len(sig), (tag, int(written.sum()))          # the chain kernel did run these trees
        assert np.array_equal(written, loose[:, 1] >= 0) and np.array_equal(written, tight[:, 1] >= 0), tag
        # raw lists: every threshold of the tight order >= the shipped order's >= the loose order's (a forced second
        # pass starts from an empty list, so its count IS the final count)
        assert (tight[written, 0] <= shipped[written, 0]).all() and (shipped[written, 0] <= loose[written, 0]).all(), tag
        assert (loose[written, 0] > shipped[written, 0]).any(), tag                 # the forced orders do differ
        # final lists: the same SET of table indices in all three orders (the order inside a list is the order of arrival)
        assert np.array_equal(shipped[written, 1], loose[written, 1]) and np.array_equal(shipped[written, 1], tight[written, 1])
        assert shipped[written, 1].min() >= 1 and shipped[written, 1].max() <= 256
        for row in np.nonzero(written)[0]:
            n = shipped[row, 1]
            a, b, t = (np.sort(x[row, 2:2 + n]) for x in (shipped, loose, tight))
            assert np.array_equal(a, b) and np.array_equal(a, t), (tag, int(row))
