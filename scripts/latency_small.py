"""Latency of small calls (the reference's notebook path solves ONE signal, main.py:5): config-2 parameters,
S = 1, 2, 8, 32, 128, 512 signals per call.  Usage (GPU box): python scripts/latency_small.py"""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench
dic = bench.canonical_dictionary()
sig = bench.make_signals(dic, 0)
bank = dic.device_bank()
d_all = torch.from_numpy(sig).cuda()
out = {}
for S in (1, 2, 8, 32, 128, 512):
    d = d_all[:S].contiguous()
    o = bank.mmpdf_launch(d, 5, 3, 3, 0.8, want_coef=False)
    for _ in range(20):
        bank.mmpdf_launch(d, 5, 3, 3, 0.8, want_coef=False, out=o)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        bank.mmpdf_launch(d, 5, 3, 3, 0.8, want_coef=False, out=o)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200
    out[S] = {"us_per_call": round(dt * 1e6, 1), "signals_per_s": round(S / dt)}
# one signal through the public single-signal method (host array in, result dict out)
t0 = time.perf_counter()
for _ in range(20):
    dic.mmpdfSparVarFromDict({"id": 0, "signal": sig[0].tolist(), "snr": 10}, 3, 0.8, 3, 5)
out["mmpdfSparVarFromDict, one signal, host in / dict out"] = {"us_per_call": round((time.perf_counter() - t0) / 20 * 1e6, 1)}
print(json.dumps(out))
