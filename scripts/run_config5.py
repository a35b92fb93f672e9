"""BASELINE config 5 on the hardware at hand: 10^6 synthetic signals through the array API
(mmpdfSparVarBatch: host arrays in, chunks of 32768 to the GPU, records out).  On one GPU this is the
whole million; under torch.distributed.run each rank takes its contiguous share and rank 0 gathers."""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dic = bench.canonical_dictionary()
base = bench.make_signals(dic, 0)
rng = np.random.default_rng(11)
t0 = time.time()
sig = np.empty((n, base.shape[1]))
for c0 in range(0, n, len(base)):
    c1 = min(n, c0 + len(base))
    sig[c0:c1] = base[:c1 - c0] + 0.01 * rng.standard_normal((c1 - c0, base.shape[1]))
gen_s = time.time() - t0
dic.device_bank().similarity_table(738, 0.8)
dic.mmpdfSparVarBatch(sig[:4096], 3, 0.8, 3, 5)          # warm-up (allocator, kernels)
t0 = time.time()
b = dic.mmpdfSparVarBatch(sig, 3, 0.8, 3, 5)
wall = time.time() - t0
t0 = time.time()
b.save_npz("/tmp/config5_records.npz")
save_s = time.time() - t0
print(json.dumps({"config": "5: %d signals, MMP-DF c3 B3 K5, one GPU, host arrays in / records out" % n,
                  "signals": n, "wall_s": wall, "signals_per_s_end_to_end": n / wall, "generate_s": gen_s,
                  "save_npz_s": save_s, "mse_mean_by_sparsity": b.mse.mean(axis=0).tolist(),
                  "first_record": b.record(0, 0, 0)["results"][4]["path"]}))
