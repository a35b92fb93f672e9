"""Ad-hoc timing of csc_mmpdf_batch on synthetic signals (development aid, not the bench)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from csc_amd.engine import DeviceBank

S = int(sys.argv[1]) if len(sys.argv) > 1 else 3200
bank64 = np.load("tests/golden/bank.npz")["bank"]
db = DeviceBank(bank64)
rng = np.random.default_rng(0)
K, L = bank64.shape
N = 738
sig = np.zeros((S, N))
for s in range(S):
    for _ in range(3):
        k = rng.integers(K); p = rng.integers(N - L + 1)
        sig[s, p:p + L] += bank64[k]
sig += 0.01 * rng.standard_normal(sig.shape)
d = torch.from_numpy(sig).cuda()
out = db.mmpdf_launch(d, 5, 3, 3, 0.8)
torch.cuda.synchronize()
for it in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    db.mmpdf_launch(d, 5, 3, 3, 0.8, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(f"S={S} {ms:.3f} ms  {S / ms * 1e3:.0f} signals/s  {S * 13 * 34.76e6 / ms / 1e9:.1f} TFLOP/s algorithmic", flush=True)
print("stats", out["stats"].cpu().numpy())
