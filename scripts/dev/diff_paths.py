"""Development aid: where do the chain kernel's node MSEs differ from the fp32-only path's (bit level)?"""
import sys
import numpy as np
sys.path.insert(0, ".")
import csc_amd
d = csc_amd.ZSDictionary.from_values(np.arange(0.005, 0.06, 0.005), np.arange(0.010, 0.036, 0.002), [1])
sigs = np.load("tests/golden/signals.npz")["signals"]
bank = d.device_bank()
for (K, c, B) in ((5, 1, 1), (5, 3, 3)):
    bank.fp32_only = False
    a = bank.mmpdf_batch(sigs, K, c, B, 0.8, want_coef=True)
    bank.fp32_only = True
    b = bank.mmpdf_batch(sigs, K, c, B, 0.8, want_coef=True)
    bank.fp32_only = False
    print(K, c, B, "pos equal", np.array_equal(a.node_pos, b.node_pos), "stats", a.stats, b.stats)
    for lvl in range(1, K + 1):
        ids = [a.node_id(lvl, j) for j in range(a.level_sizes[lvl - 1])]
        dm = a.node_mse[:, ids] != b.node_mse[:, ids]
        dc = a.node_coef[:, ids] != b.node_coef[:, ids]
        rel = np.abs(a.node_mse[:, ids] - b.node_mse[:, ids]) / np.abs(b.node_mse[:, ids])
        print("  level", lvl, "mse differs in", int(dm.sum()), "of", dm.size, "max rel", rel.max(), "coef differs", int(dc.any(axis=-1).sum()))
