"""Development aid: clocks of screen_lr_kernel's phases (wave 0) on config 3 (CSC_HIP_LIB=...libcsc_hip_prof.so)."""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, ".")
import csc_amd
from csc_amd import _lib
large = csc_amd.ZSDictionary.from_values(np.arange(0.005, 0.06, 0.0005), np.arange(0.010, 0.036, 0.0003), [1])
bank = large.device_bank()
sig, _ = bank.generate(np.full(1024, 4, np.int32), np.full(1024, 10.0), 777, 10, 0.75, seed=3)
out = bank.mmpdf_launch(sig, 5, 3, 3, 0.8, want_coef=False)
L = _lib.lib()
buf = (C.c_ulonglong * 32)()
L.csc_prof_read(buf, 1)
bank.mmpdf_launch(sig, 5, 3, 3, 0.8, want_coef=False, out=out)
torch.cuda.synchronize()
L.csc_prof_read(buf, 1)
v = np.array(list(buf), dtype=np.float64)
n = 1024 * 13
if "--screen16" in sys.argv:      # library built with -DCSC_PHASE_PROF -DSCREEN16_PROF
    n = 1024 * 15
    print("screen16_kernel per node (wave 0): prologue %.0f clk, table + need %.0f clk, blocks %.0f clk" % (v[28] / n, v[29] / n, v[30] / n))
else:
    print("screen_lr_kernel per node (wave 0): T = R U %.0f clk, C~ = T V + bounds %.0f clk" % (v[29] / n, v[30] / n))
