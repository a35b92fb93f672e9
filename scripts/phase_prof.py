"""Development aid: per-phase shader-clock shares of mmp_level_kernel (needs `make -C csc_amd/csrc prof`,
run with CSC_HIP_LIB=csc_amd/csrc/libcsc_hip_prof.so)."""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from csc_amd import _lib
dic = bench.canonical_dictionary()
sig = bench.make_signals(dic, 0)
bank = dic.device_bank()
d = torch.from_numpy(sig).cuda()
out = bank.mmpdf_launch(d, 5, 3, 3, 0.8, want_coef=False)
L = _lib.lib()
buf = (C.c_ulonglong * 16)()
L.csc_prof_read(buf, 1)
bank.mmpdf_launch(d, 5, 3, 3, 0.8, want_coef=False, out=out)
L.csc_prof_read(buf, 1)
v = np.array(list(buf), dtype=np.float64)
names = ["prologue", "gemm", "argmax+cands", "rescore+choose", "gram dots", "solve", "residual+write", "suppress/tail"]
nodes = 3200 * 13
tot = v[:8].sum()
for n, x in zip(names, v):
    print(f"{n:16s} {x / nodes:10.0f} clk/node  {100 * x / tot:5.1f} %")
print("total", tot / nodes, "clk/node")
