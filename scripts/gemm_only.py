"""Development aid: MFMA rate of the correlation GEMM alone (csc_correlate), no selection / LS."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import bench
dic = bench.canonical_dictionary()
bank = dic.device_bank()
S = 9600
sig = torch.randn(S, 738, dtype=torch.float64, device="cuda")
out = bank.correlate(sig)
torch.cuda.synchronize()
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); out = bank.correlate(sig); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    issued = S * 101 * 4 * 32 * 2048 / (ms * 1e-3) / 1e12
    print(f"{ms:.3f} ms  {S / ms * 1e3:.0f} nodes/s  dense {S * 2 * 247 * 143 * 492 / ms / 1e9:.1f} TF  issued {issued:.1f} TF ({issued / 157.3:.2f} of peak)")
