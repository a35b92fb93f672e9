"""Development aid: selection statistics of one config-2 batch (candidates per selection, nodes the table
kernel handed to the GEMM launch)."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import bench
dic = bench.canonical_dictionary()
sig = bench.make_signals(dic, 0)
res, _ = dic._batch(sig, 5, 3, 3, 0.8)
st = np.asarray(res.stats)
sel = sig.shape[0] * 15
print("stats", st.tolist())
print(f"candidates per selection {st[2] / sel:.2f}; nodes handed to the GEMM launch {st[3]} of {sig.shape[0] * 12}")
