"""Development aid: per-phase shader-clock shares of the level kernels (needs `make -C csc_amd/csrc prof`,
run with CSC_HIP_LIB=csc_amd/csrc/libcsc_hip_prof.so).  Clocks are thread 0's, summed over workgroups."""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from csc_amd import _lib
dic = bench.canonical_dictionary()
sig = bench.make_signals(dic, 0)
bank = dic.device_bank()
if "--fp32" in sys.argv:            # fp32 correlations only
    bank.fp32_only = True
if "--no-lean" in sys.argv:
    bank.lean = False
d = torch.from_numpy(sig).cuda()
out = bank.mmpdf_launch(d, 5, 3, 3, 0.8, want_coef=False)
L = _lib.lib()
buf = (C.c_ulonglong * 32)()
L.csc_prof_read(buf, 1)
bank.mmpdf_launch(d, 5, 3, 3, 0.8, want_coef=False, out=out)
torch.cuda.synchronize()
L.csc_prof_read(buf, 1)
v = np.array(list(buf), dtype=np.float64)
names = ["prologue", "gemm", "suppress", "max + candidates", "-", "rescore + choose",
         "gram dots", "solve", "residual + write", "tail", "(suppress: wait for the siblings' waves)",
         "(suppress: similarity row -> LDS)"]
lean = "--fp32" not in sys.argv and "--no-lean" not in sys.argv
rows = (("mmp_level_kernel: level 0" + ("" if lean else " - 4"), 0, 3200 if lean else 3200 * 13),
        ("mmp_level_lean_kernel: levels 1 - 4 (gemm = sweep 1, max + candidates = sweep 2)", 16, 3200 * 12 if lean else 1))
for title, base, nodes in rows:
    w = v[base:base + 12]
    tot = w.sum()
    print(title)
    if base == 16:
        names = ["prologue", "build r16 + product", "re-score: dots + wave sums (before the barrier)", "wait for the product's last wave + scan of the raw list", "gram dots + prefetch (before the barrier)", "re-score: barrier + choice",
                 "gram: barrier", "solve", "residual: barrier + node tables", "tail", "residual update (before the barrier)", "-"]
    for n, x in zip(names, w):
        print(f"  {n:60s} {x / nodes:10.0f} clk/node  {100 * x / max(tot, 1):5.1f} %")
    print(f"  total {tot / nodes:.0f} clk/node")

if lean:
    n = 3200 * 12
    print("chain product, wave 0 per node: %.0f clk for %.1f slab steps (128 clk of MFMAs each)" % (v[28] / n, v[31] / n))
