"""The deep-tree configurations (reference defaults c3 / B10 / K10; config 4a c4 / B4 / K7) on the 3200 config-2 signals,
for profiling: python scripts/run_deep.py [2b|4a|4b] [reps]"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
which = sys.argv[1] if len(sys.argv) > 1 else "2b"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
Kmax, c, B = {"2b": (10, 3, 10), "4a": (7, 4, 4), "4b": (7, 4, 64)}[which]
dic = bench.canonical_dictionary()
sig = bench.make_signals(dic, 0)
if which == "4b":
    sig = sig[::4]
bank = dic.device_bank()
d = torch.from_numpy(sig).cuda()
out = bank.mmpdf_launch(d, Kmax, c, B, 0.8, want_coef=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    bank.mmpdf_launch(d, Kmax, c, B, 0.8, want_coef=False, out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
print(json.dumps({"config": which, "signals": len(sig), "seconds": dt, "signals_per_s": len(sig) / dt,
                  "stats": out["stats"].cpu().tolist()}))
