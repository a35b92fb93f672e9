"""BASELINE.json config 3 alone (9570-atom dictionary, c3/B3/K5), for profiling:
    python3 scripts/run_config3.py [signals] [reps] [--fp32] [--no-lowrank] [--no-simatoms]
prints one JSON line with the throughput; under rocprofv3 the kernels of the timed calls are what is traced."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, __file__.rsplit("/", 2)[0])
import csc_amd        # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 2
large = csc_amd.ZSDictionary.from_values(np.arange(0.005, 0.06, 0.0005), np.arange(0.010, 0.036, 0.0003), [1])
np.random.seed(3)
sigs = np.array([large.generateConstrainedTestSignal(777, 4, 10, 10, 0.75)[0] for _ in range(n)])
bank = large.device_bank()
bank.fp32_only = "--fp32" in sys.argv
if "--no-lowrank" in sys.argv:
    bank.lowrank = False
if "--no-simatoms" in sys.argv:
    bank.simatoms = False
d = torch.from_numpy(sigs).cuda()
out = bank.mmpdf_launch(d, 5, 3, 3, 0.8, want_coef=False)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(reps):
    bank.mmpdf_launch(d, 5, 3, 3, 0.8, want_coef=False, out=out)
torch.cuda.synchronize()
dt = (time.time() - t0) / reps
print(json.dumps({"config": "3", "signals": n, "fp32_only": bank.fp32_only, "lowrank": bank.lowrank, "seconds": dt, "signals_per_s": n / dt,
                  "stats": out["stats"].cpu().tolist()}))
