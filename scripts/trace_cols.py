"""Development aid: where a wave of the chain kernel's column-owner product spends its clocks (prof build): ring fill,
slab loops, tile epilogues; per wave, with its number of tiles and slabs.  Usage (GPU box):
    CSC_HIP_LIB=$PWD/csc_amd/csrc/libcsc_hip_prof.so python scripts/trace_cols.py"""
import ctypes as C, sys, os
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from csc_amd import _lib
dic = bench.canonical_dictionary()
sig = bench.make_signals(dic, 0)
bank = dic.device_bank()
d = torch.from_numpy(sig).cuda()
S = int(sys.argv[1]) if len(sys.argv) > 1 else 3200
d = d[:S].contiguous()
out = bank.mmpdf_launch(d, 5, 3, 3, 0.8, want_coef=False)
L = _lib.lib()
cap = 1 << 18
buf = (C.c_uint * (4 * cap))()
torch.cuda.synchronize()
L.csc_prof_trace_read(buf, cap)
bank.mmpdf_launch(d, 5, 3, 3, 0.8, want_coef=False, out=out)
torch.cuda.synchronize()
n = L.csc_prof_trace_read(buf, cap)
t = np.frombuffer(buf, dtype=np.uint32)[:4 * n].reshape(n, 4).astype(np.int64)
w = t[(t[:, 0] >> 16) == 0xffff]
print(f"{len(w)} wave records of {n}")
for wave in range(4):
    m = (w[:, 0] & 15) == wave
    x = w[m]
    nit = (x[:, 0] >> 4) & 15
    nsl = (x[:, 0] >> 8) & 255
    print(f"wave {wave}: tiles {np.bincount(nit)[:6]}, slabs {nsl.mean():.1f}; fill {x[:, 1].mean():.0f} clk, "
          f"loops {x[:, 2].mean():.0f} clk ({(x[:, 2] / np.maximum(nsl, 1)).mean():.0f} per slab), epilogues {x[:, 3].mean():.0f} clk")
