import sys, os, json, numpy as np
sys.path.insert(0, ".")
import csc_amd
z = np.load("tests/golden/large.npz")
large = csc_amd.ZSDictionary.from_values(np.arange(0.005, 0.06, 0.0005), np.arange(0.010, 0.036, 0.0003), [1])
res, _ = large._batch(z["signals"][:4], 5, 3, 3, 0.8, want_coef=True)
import torch; torch.cuda.synchronize()
print("done", res.stats)
