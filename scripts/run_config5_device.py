"""BASELINE config 5 end to end on one GPU with the round-2 pieces either side of the solve: 10^6 constrained signals
GENERATED ON THE DEVICE (generateConstrainedSignalsDevice, chunks of 64 000), solved where they lie (mmpdfSparVarRecords),
the records saved as the uncompressed binary file, and the first 100 000 written as the reference's sparVar JSON by the
streaming writer.  Usage (GPU box): python scripts/run_config5_device.py [signals]"""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from csc_amd.dictionary import SparVarBatch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dic = bench.canonical_dictionary()
per_cell = 4000                                   # x 4 SNR levels x 4 sparsity levels = 64 000 signals per chunk
chunk = per_cell * 16
dic.device_bank().similarity_table(738, 0.8)
g = dic.generateConstrainedSignalsDevice(200, 738, [2, 3, 4, 5], [0, 5, 10, 15], 10, 0.75, seed=1)      # warm-up
dic.mmpdfSparVarRecords(g["signals"], 3, 0.8, 3, 5)
torch.cuda.synchronize()
t_gen = t_solve = 0.0
rows, ids, snrs = [], [], []
done = 0
while done < n:
    t0 = time.perf_counter()
    g = dic.generateConstrainedSignalsDevice(per_cell, 738, [2, 3, 4, 5], [0, 5, 10, 15], 10, 0.75, seed=1, first_index=done)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    r = dic.mmpdfSparVarRecords(g["signals"], 3, 0.8, 3, 5)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    t_gen += t1 - t0
    t_solve += t2 - t1
    take = min(chunk, n - done)
    rows.append(r[:take].cpu())
    ids.append(np.asarray(g["ids"])[:take])
    snrs.append(np.asarray(g["snr"])[:take])
    done += take
t0 = time.perf_counter()
allrows = torch.cat(rows).numpy()
batch = SparVarBatch.from_records(dic, allrows, 3, 3, 5)
t_host = time.perf_counter() - t0
t0 = time.perf_counter()
batch.save_records("/tmp/config5_records.npy")
t_bin = time.perf_counter() - t0
m = min(n, 100_000)
sub = SparVarBatch.from_records(dic, allrows[:m], 3, 3, 5)
t0 = time.perf_counter()
sub.save_json("/tmp/config5_first.json", ids=np.concatenate(ids)[:m].tolist(), snrs=np.concatenate(snrs)[:m].tolist())
t_json = time.perf_counter() - t0
out = {"config": "5 on one GPU, device generator -> solve -> records", "signals": n,
       "generate_s": t_gen, "solve_s": t_solve, "records_to_host_s": t_host, "save_records_s": t_bin,
       "binary_bytes": os.path.getsize("/tmp/config5_records.npy"),
       "json_signals": m, "save_json_s": t_json, "json_bytes": os.path.getsize("/tmp/config5_first.json"),
       "json_records_per_s": m / t_json, "generate_signals_per_s": n / t_gen, "solve_signals_per_s": n / t_solve,
       "end_to_end_signals_per_s (generate + solve + binary save)": n / (t_gen + t_solve + t_host + t_bin),
       "mse_mean_by_sparsity": batch.mse.mean(axis=0).tolist()}
print(json.dumps(out))
os.remove("/tmp/config5_records.npy"); os.remove("/tmp/config5_records.npy.json"); os.remove("/tmp/config5_first.json")
