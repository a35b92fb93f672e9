"""The low-rank screen's bounds against the product they stand in for.

csc_amd/csrc/Makefile's `lrcheck` library is the shipped code with one difference: screen16_kernel multiplies EVERY
16-column block instead of only those the low-rank bounds leave in reach, and prints a line "LRCHK ..." for every unit
whose single-term maximum exceeds the upper bound screen_lr_kernel gave it (plus the rounding bound of the single-term
product) and for every node whose maximum stays below the lower bound.  None may appear -- on BASELINE config 3's
dictionary (golden signals: three children at the root, the later ones under suppression through the gathered
similarity bands) and on a 330-atom dictionary with 3 and with 4 children per node on every level (the fourth child takes
the block-by-block lookup), P = 260 / 272 / 256.  The library runs in a process of its own (CSC_HIP_LIB); its "LRNEED"
lines show that the screen ran at all."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys
import numpy as np
root, golden = sys.argv[1:3]
sys.path.insert(0, root)
import torch
import csc_amd
large = csc_amd.ZSDictionary.from_values(np.arange(0.005, 0.06, 0.0005), np.arange(0.010, 0.036, 0.0003), [1])
z = np.load(os.path.join(golden, "large.npz"))
bank = large.device_bank()
assert any(os.path.basename(os.environ["CSC_HIP_LIB"]) in line for line in open("/proc/self/maps"))
assert bank.lowrank and bank.simatoms
bank.mmpdf_batch(z["signals"], 5, 3, 3, 0.8, want_coef=False)
d2 = csc_amd.ZSDictionary.from_values(np.arange(0.005, 0.06, 0.0025), np.arange(0.010, 0.036, 0.0018), [1])
rng = np.random.default_rng(23)
L2 = d2.getAtomsLength()
bank2 = d2.getLocalDictionary()
b2 = d2.device_bank()
assert b2.lowrank
for P, (K, c, B) in ((260, (4, 3, 10)), (272, (4, 3, 10)), (256, (4, 3, 10)), (260, (3, 4, 16))):
    sig = 0.02 * rng.standard_normal((12, L2 + P - 1))
    for srow in sig:
        for _ in range(3):
            srow[rng.integers(0, P):][:L2] += rng.uniform(0.5, 1.5) * bank2[rng.integers(len(bank2))]
    b2.mmpdf_batch(sig, K, c, B, 0.7, want_coef=False)
    b2.simatoms = not b2.simatoms                                    # both forms of the later siblings' product
torch.cuda.synchronize()
print("WORKER DONE")
'''


def test_lowrank_bounds_hold_against_the_full_product():
    lib = os.path.join(ROOT, "csc_amd", "csrc", "libcsc_hip_lrcheck.so")
    assert os.path.exists(lib), "make -C csc_amd/csrc lrcheck"
    env = dict(os.environ, CSC_HIP_LIB=lib)
    r = subprocess.run([sys.executable, "-c", WORKER, ROOT, GOLDEN], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "WORKER DONE" in r.stdout, r.stderr[-2000:]
    bad = [line for line in r.stdout.splitlines() if "LRCHK" in line]
    assert not bad, bad[:10]
    assert sum("LRNEED" in line for line in r.stdout.splitlines()) > 100
