"""Development aid: config 2's 3200-signal batches on one stream against two streams taking alternate batches (the
root level of one batch under the chains of the other)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import csc_amd
d = csc_amd.ZSDictionary.from_values(np.arange(0.005, 0.06, 0.005), np.arange(0.010, 0.036, 0.002), [1])
bank = d.device_bank()
S = 3200
sig, _ = bank.generate(np.full(S, 4, np.int32), np.full(S, 10.0), 738, 10, 0.75, seed=1)
outs = [bank.mmpdf_launch(sig, 5, 3, 3, 0.8, want_coef=False) for _ in range(4)]
torch.cuda.synchronize()
def run(nstream, steps=600, prio=None):
    streams = [torch.cuda.Stream(priority=(prio[k] if prio else 0)) for k in range(nstream)]
    for warm in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            with torch.cuda.stream(streams[k % nstream]):
                bank.mmpdf_launch(sig, 5, 3, 3, 0.8, want_coef=False, out=outs[k % nstream])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return S * steps / dt
for n in (1, 2, 3, 4, 2, 3):
    print(n, "stream(s): %.3f M signals/s" % (run(n) / 1e6))
for prio in ((-1, -1), (-1, 0), (0, 0)):       # stream priorities: no effect expected on equal work
    print("2 streams, priorities", prio, ": %.3f M signals/s" % (run(2, prio=prio) / 1e6))
