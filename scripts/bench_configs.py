"""Throughput of the other BASELINE.json configurations (parity cases, not the bench line):
config 1 (OMP, 200 signals, sparsity 3), config 3 (9570-atom dictionary), config 4 (deep trees).
Writes one JSON object per line; run on the GPU box:  python scripts/bench_configs.py > gpurun_out/configs.json"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bench          # noqa: E402
import csc_amd        # noqa: E402
from csc_amd import _lib   # noqa: E402

PEAK = bench.FP32_MFMA_PEAK_TFLOPS


def run(name, dic, sigs, Kmax, c, B, thr, reps=3):
    bank = dic.device_bank()
    d = torch.from_numpy(sigs).cuda()
    t0 = time.time()
    out = bank.mmpdf_launch(d, Kmax, c, B, thr, want_coef=False)
    torch.cuda.synchronize()
    first = time.time() - t0
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        bank.mmpdf_launch(d, Kmax, c, B, thr, want_coef=False, out=out)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e-3)
    S, N = sigs.shape
    sizes, NN = _lib.tree_shape(Kmax, c, B)
    n_sel = 1 + sum(sizes[:-1])
    P = N - bank.L + 1
    flops = n_sel * 2.0 * P * bank.K * bank.L
    print(json.dumps({"config": name, "signals": S, "K": bank.K, "L": bank.L, "N": N, "nodes_per_signal": NN,
                      "seconds": best, "signals_per_s": S / best, "algorithmic_tflops": S * flops / best / 1e12,
                      "frac_fp32_mfma_peak": S * flops / best / 1e12 / PEAK, "first_call_s": first,
                      "stats": out["stats"].cpu().tolist()}), flush=True)


def constrained(dic, n, N, sparsity, snr, seed):
    np.random.seed(seed)
    return np.array([dic.generateConstrainedTestSignal(N, sparsity, snr, 10, 0.75)[0] for _ in range(n)])


ONLY_LARGE = "--only-large" in sys.argv
small = bench.canonical_dictionary()
if ONLY_LARGE:
    sys.argv.append("--skip-small")
if "--skip-small" not in sys.argv:
  run("1: OMP 200 signals sparsity 3 SNR 10", small, constrained(small, 200, 738, 3, 10, 0), 3, 1, 1, 1.0)
  sig2 = bench.make_signals(small, 0)
  run("2: sparVar 3200 signals c3 B3 K5", small, sig2, 5, 3, 3, 0.8)
  run("4a: deep tree c4 B4 K7, 3200 signals", small, sig2, 7, 4, 4, 0.8)
  run("4b: deep tree c4 B64 K7 (341 nodes), 800 signals", small, sig2[::4], 7, 4, 64, 0.8)
  run("2b: sparVar c3 B10 K10 (reference defaults), 3200 signals", small, sig2, 10, 3, 10, 0.8)
  # config 5, one GPU's share: 125 000 signals (1M over 8 GPUs) in ONE call, config-2 recipe tiled with noise
  rng = np.random.default_rng(7)
  sig5 = np.tile(sig2, (40, 1))[:125000] + 0.01 * rng.standard_normal((125000, 738))
  run("5 (one GPU's share): 125000 signals c3 B3 K5", small, sig5, 5, 3, 3, 0.8, reps=2)
  del sig5
if "--no-large" not in sys.argv:
    t0 = time.time()
    large = csc_amd.ZSDictionary.from_values(np.arange(0.005, 0.06, 0.0005), np.arange(0.010, 0.036, 0.0003), [1])
    print(json.dumps({"large_dictionary_build_s": time.time() - t0}), flush=True)
    n = int(sys.argv[sys.argv.index("--large-signals") + 1]) if "--large-signals" in sys.argv else 512
    sig3 = constrained(large, n, 777, 4, 10, 3)
    t0 = time.time()
    large.device_bank().similarity_table(777, 0.8)
    torch.cuda.synchronize()
    print(json.dumps({"large_similarity_table_s": time.time() - t0}), flush=True)
    run("3: 9570-atom dictionary c3 B3 K5, %d signals" % n, large, sig3, 5, 3, 3, 0.8, reps=2)
    large.device_bank().fp32_only = True
    run("3 (CSC_FLAG_FP32_ONLY): 9570-atom dictionary c3 B3 K5, %d signals" % n, large, sig3, 5, 3, 3, 0.8, reps=1)
