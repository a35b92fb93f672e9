"""Development aid (prof build, CSC_GEMM_ONLY=1): level-kernel launch times with the children skipped."""
import sys, torch
sys.path.insert(0, ".")
import bench
from csc_amd import _lib
dic = bench.canonical_dictionary()
bank = dic.device_bank()
d = torch.from_numpy(bench.make_signals(dic, 0)).cuda()
out = bank.mmpdf_launch(d, 5, 3, 3, 0.8, want_coef=False)
torch.cuda.synchronize()
t = _lib.LevelTimer(64)
with t:
    for _ in range(3):
        bank.mmpdf_launch(d, 5, 3, 3, 0.8, want_coef=False, out=out)
    torch.cuda.synchronize()
sp = t.read()
for lv in range(5):
    ms = sum(m for l, m in sp if l == lv) / 3
    n = 3200 * (1 if lv == 0 else 3)
    print(f"level {lv}: {ms:.3f} ms  {ms * 1e-3 * 2.4e9 * 256 / n:.0f} clk/node/CU  MFMA-issue {103424 / (ms * 1e-3 * 2.4e9 * 256 / n):.2f}")
