"""Calibration of the CPU baseline (BUILD CONTAINER ONLY -- imports the reference from /root/reference).

Times the imported reference's mmpdfSparVarFromDict and the oracle's 'faithful' mode (the thing bench.py's
cpu_baseline times on the GPU box) on the same 16 seeded signals, one process each; BASELINE.md asks for
agreement within +-10 %.  Run:  PYTHONDONTWRITEBYTECODE=1 python oracle/calibrate.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_golden import ref_dict            # noqa: E402
from oracle import csc_oracle as O          # noqa: E402

z = np.load(os.path.join(ROOT, "tests", "golden", "signals.npz"))
sigs = z["signals"][::4]
d_ref, d_or = ref_dict(), O.canonical_dictionary()
t_ref = t_or = 0.0
for i, s in enumerate(sigs):
    sd = {"id": i, "snr": 0, "signal": s.tolist()}
    t0 = time.time(); a = d_ref.mmpdfSparVarFromDict(sd, 3, 0.8, 3, 5); t_ref += time.time() - t0
    t0 = time.time(); b = O.mmpdf_sparvar(d_or, {"id": i, "snr": 0, "signal": s}, 3, 0.8, 3, 5, "faithful"); t_or += time.time() - t0
    assert [r["path"] for r in a["results"]] == [r["path"] for r in b["results"]]
n = len(sigs)
print(f"reference {t_ref / n:.3f} s/signal, oracle faithful {t_or / n:.3f} s/signal, ratio {t_or / t_ref:.3f} ({n} signals, 1 process)")
