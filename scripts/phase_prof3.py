"""Development aid: where a slab of the tiled fp16 three-product loop spends its clocks (wave 0 of every workgroup,
20-bit shader cycle counter; needs `make -C csc_amd/csrc prof`, run with CSC_HIP_LIB=.../libcsc_hip_prof.so)."""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, ".")
import csc_amd
from csc_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
large = csc_amd.ZSDictionary.from_values(np.arange(0.005, 0.06, 0.0005), np.arange(0.010, 0.036, 0.0003), [1])
np.random.seed(3)
sigs = np.array([large.generateConstrainedTestSignal(777, 4, 10, 10, 0.75)[0] for _ in range(n)])
bank = large.device_bank()
d = torch.from_numpy(sigs).cuda()
out = bank.mmpdf_launch(d, 5, 3, 3, 0.8, want_coef=False)
L = _lib.lib()
buf = (C.c_ulonglong * 32)()
L.csc_prof_read(buf, 1)
bank.mmpdf_launch(d, 5, 3, 3, 0.8, want_coef=False, out=out)
torch.cuda.synchronize()
L.csc_prof_read(buf, 1)
v = np.array(list(buf), dtype=np.float64)
slabs = v[7]
names = ["wait for the fragments (lgkmcnt)", "MFMA block 1 (issue)", "wait for the copies (vmcnt)", "barrier",
         "issue the copies", "MFMA block 2 (issue)", "request the next fragments", "MFMA block 3 (issue)"]
tot = v[8:16].sum()
print(f"{slabs:.0f} slabs, {tot / slabs:.0f} clocks per slab (wave 0); per signal: {slabs / n:.0f} slabs, three-product units: "
      f"sweep 1 {v[5] / n:.0f}, sweep 2 {v[6] / n:.0f}")
for nme, x in zip(names, v[8:16]):
    print(f"  {nme:36s} {x / slabs:8.0f} clk/slab  {100 * x / tot:5.1f} %")

names2 = ["prologue + fp16 copies", "screen + order", "sweep 1 (three-product)", "sweep 2 + candidates", "-", "rescore + choose",
          "gram dots", "solve", "residual + write", "tail"]
nodes = 13 * n
tot2 = v[16:26].sum()
print(f"phases, clocks per node (thread 0): total {tot2 / nodes:.0f}")
for nme, x in zip(names2, v[16:26]):
    print(f"  {nme:36s} {x / nodes:10.0f}  {100 * x / tot2:5.1f} %")


units = v[5] + v[6]
if units:
    print(f"three-product units: {units / nodes:.1f} per node, clocks per unit (thread 0):")
    for nme, x in zip(["product (first wait included)", "begin the next unit + masks / suppression", "maximum + barriers",
                       "candidates + between units"], v[0:4]):
        print(f"  {nme:44s} {x / units:8.0f}")
