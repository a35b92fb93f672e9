#!/usr/bin/env python3
"""bench.py -- signals/s of batched MMP-DF sparVar (connections 3, branches 3, max sparsity 5,
dissimilarity 0.8) on MI355X, the metric of BASELINE.json.

    python bench.py --gpus 1 --steps 1000 --warmup 100
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], SURVEY.md section 8d "Config 2"): the 3200-signal sparVar sweep --
200 constrained synthetic signals per (snr, sparsity) in product([0,5,10,15],[2,3,4,5]), N = 738,
canonical 143-atom dictionary (L = 492), generator seed = rank.  One step = one pass of the hot path over
that batch with the signals already resident in HBM, through the PRODUCT's own sharded function
(ZSDictionary.mmpdfSparVarRecords(sharded=True): csc_mmpdf_batch -- level launches + the per-sparsity
reduction -- then, with N > 1, the ONE gather of the fixed-size device-resident sparVar records on rank 0
over RCCL); with N > 1 every rank holds its own 3200 signals (weak scaling) and a step's gather is only
ENQUEUED (async_gather=True, two sets of receive buffers): it runs under the next step's solve and is waited
for before its buffers are used again, and all of them before the clock stops.

Reported next to `value`:
  roofline      matrix-core roofline of the level kernels, launch durations measured live with HIP events on
                the launch stream (csc_timer in include/csc_hip.h).  A node's correlations are taken on the
                fp16 matrix cores as a filter and the entries in its rounding band are re-scored in float64:
                algorithmic flops per signal = n_sel * 2*P*K*L, n_sel = 13 (dense semantics of the reference,
                the atoms' zero padding NOT discounted), against the dense fp16 MFMA peak; `issued_frac` is
                the fraction of that peak really issued (all-zero bank blocks are skipped).
  fp32_path     the same batch with CSC_FLAG_FP32_ONLY (every node correlated on v_mfma_f32_16x16x4_f32, the
                path BASELINE.json's north star describes), against the fp32 MFMA peak; identical records.
  config5       BASELINE.json configs[4]: 125 000 signals per rank (the config-2 clean signals repeated with
                fresh noise, seed = base + rank), solved in chunks, ONE gather at the end.
  configs       (N = 1) the other BASELINE.json configurations as driver-timed legs: config 1 (200 OMP signals),
                config 3 (9570-atom dictionary, 4096 signals; similarity-table build reported separately), config 4
                (deep trees c4/B4/K7 and c4/B64/K7) -- signals/s, dense-equivalent and issued matrix-core fractions,
                stats, and whether a subsample's records equal the fp32-only path's.
  cpu_baseline  the oracle's reference-faithful mode (oaconvolve over all atoms + lsqr defaults + one
                fresh tree per sparsity = the reference's cost structure; kind "port" because the
                reference's files cannot travel to the GPU box) on a bounded sample, rank 0, N = 1 only.
"""
import os

# CPU-baseline workers (spawned below with this variable set) must size their BLAS / FFT pools BEFORE numpy
# loads: one thread per worker process.
if os.environ.get("CSC_BENCH_CPU_WORKER") == "1":
    for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
        os.environ[_v] = "1"

import argparse
import itertools
import json
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np

CONNECTIONS, BRANCHES, MAX_SPARSITY, DISSIMILARITY = 3, 3, 5, 0.8
SIGNAL_LENGTH, PER_CELL = 738, 200
SNR_LEVELS, SPARSITY_LEVELS = [0, 5, 10, 15], [2, 3, 4, 5]
FP32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
FP16_MFMA_PEAK_TFLOPS = 2500.0     # same guide, "Peak BF16/FP16 MFMA": ~2.5 PF dense
HBM_PEAK_GBS = 8000.0              # same guide, "HBM3E peak BW": 8.0 TB/s spec
CPU_SAMPLE = [i * PER_CELL + o for i in range(16) for o in (7, 107)]    # two signals of every (snr, sparsity) cell: ~25 CPU-s
CONFIG5_PER_RANK = 125000
TRAFFIC_PROFILE = os.path.join("profiles", "r03_hbm_traffic.json")


def canonical_dictionary():
    import csc_amd
    return csc_amd.ZSDictionary.from_values(np.arange(0.005, 0.06, 0.005), np.arange(0.010, 0.036, 0.002), [1])


def make_signals(dic, seed, per_cell=PER_CELL, with_parts=False):
    """Config-2 batch with the product's own generator (reference src/csc/dictionary.py:271-365).
    ``with_parts``: also the clean signals and the noise sigma of each (for the config-5 leg)."""
    np.random.seed(seed)
    out, clean, sigma = [], [], []
    for snr, sp in itertools.product(SNR_LEVELS, SPARSITY_LEVELS):
        for _ in range(per_cell):
            s, infos = dic.generateConstrainedTestSignal(SIGNAL_LENGTH, sp, snr, 10, 0.75)
            out.append(s)
            if with_parts:
                c = np.zeros(SIGNAL_LENGTH)
                for a in infos:
                    c += dic.atoms[dic._atom_index(a)].getAtomInSignal(SIGNAL_LENGTH, a["x"])
                clean.append(c)
                active = c[np.abs(c) > 1e-3]
                sigma.append(np.sqrt(np.var(active) / (10 ** (snr / 10))))   # reference :318-322
    if with_parts:
        return np.array(out), np.array(clean), np.array(sigma)
    return np.array(out)


def _cpu_job(sig):
    from oracle import csc_oracle as O
    d = _cpu_job.dic if hasattr(_cpu_job, "dic") else O.canonical_dictionary()
    _cpu_job.dic = d
    t0 = time.time()
    O.mmpdf_sparvar(d, {"id": 0, "snr": 0, "signal": sig}, CONNECTIONS, DISSIMILARITY, BRANCHES, MAX_SPARSITY,
                    mode="faithful")
    return time.time() - t0


def _cpu_warm(_):
    from oracle import csc_oracle as O
    _cpu_job.dic = O.canonical_dictionary()
    return int(os.environ.get("OMP_NUM_THREADS", "-1"))


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(signals):
    """Reference-faithful CPU path on len(CPU_SAMPLE) signals, one single-threaded process per signal; next to
    it the same on ONE process alone (2 signals), so that contention between the workers shows."""
    import multiprocessing as mp
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(len(CPU_SAMPLE), avail))
    sample = [signals[i] for i in CPU_SAMPLE]
    saved = {v: os.environ.get(v) for v in ("CSC_BENCH_CPU_WORKER", "OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS",
                                            "MKL_NUM_THREADS")}
    # the spawned workers inherit these and re-import this module: thread pools are sized before numpy loads
    os.environ.update(CSC_BENCH_CPU_WORKER="1", OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    try:
        ctx = mp.get_context("spawn")
        with ctx.Pool(1) as pool:
            threads = pool.map(_cpu_warm, [0])
            solo = pool.map(_cpu_job, sample[:2], chunksize=1)
        with ctx.Pool(cores) as pool:
            pool.map(_cpu_warm, range(cores))                    # imports + dictionary build, untimed
            t0 = time.time()
            per = pool.map(_cpu_job, sample, chunksize=1)
            wall = time.time() - t0
    finally:
        for v, x in saved.items():
            if x is None:
                os.environ.pop(v, None)
            else:
                os.environ[v] = x
    return {"value": len(sample) / wall, "unit": "signals/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
            "sample": f"{len(sample)} of the {len(signals)} signals (indices {CPU_SAMPLE}), oracle mode "
                      f"'faithful' (oaconvolve + lsqr, one tree per sparsity = 50 nodes/signal), one "
                      f"single-threaded process per signal (OMP_NUM_THREADS={threads[0]} in the workers), "
                      f"{sum(per):.1f} CPU-s, {np.mean(per):.2f} s/signal/core",
            "one_process": {"s_per_signal": float(np.mean(solo)), "signals": len(solo),
                            "note": "the same job on one process with the other cores idle"},
            "host_cores_available": avail}


def product_blocks(bank):
    """Live (32-row slab x 16-column tile) bank blocks per node product, from the bank image's block masks:
    (the register-resident kernel's five runs of column tiles 6.. / 3.. / all, the chain kernel's per-tile slab
    ranges, number of slabs).  One block = 16 v_mfma_f32_16x16x32_f16 (16 384 flop each) over 256 positions."""
    nslab = (bank.L + 31) // 32
    off = nslab * 9 * 512
    meta = bank.bank32t[off:off + 208].view(__import__("torch").int32).cpu().numpy()
    masks = [int(m) for m in meta[144:144 + nslab]]
    first = [(m & -m).bit_length() - 1 if m else 9 for m in masks]
    p6 = 0
    while p6 < nslab and first[p6] >= 6:
        p6 += 1
    p3 = p6
    while p3 < nslab and first[p3] >= 3:
        p3 += 1
    q6 = nslab
    while q6 > p3 and first[q6 - 1] >= 6:
        q6 -= 1
    q3 = q6
    while q3 > p3 and first[q3 - 1] >= 3:
        q3 -= 1
    runs = 3 * p6 + 6 * (p3 - p6) + 9 * (q3 - p3) + 6 * (q6 - q3) + 3 * (nslab - q6)
    per_tile = 0
    for c in range(9):
        live = [q for q in range(nslab) if (masks[q] >> c) & 1]
        per_tile += (live[-1] + 1 - live[0]) if live else 0
    fp32_blocks = sum(9 - first[q] for q in range(nslab) if masks[q])
    return runs, per_tile, nslab, fp32_blocks


def issued_fp16_flops_per_signal(bank, N, max_sparsity, connections, branches):
    """MFMA flops the default path issues per signal (one register tile: K <= 144, P <= 256), following
    run_levels' dispatch in csc_hip.hip: one product per selecting node; nodes of single-child levels run on the
    chain kernel (per-tile ranges), the others on the register-resident kernel (five runs)."""
    from csc_amd import _lib
    runs, per_tile, _, _ = product_blocks(bank)
    sizes, _ = _lib.tree_shape(max_sparsity, connections, branches)
    F = [1] + list(sizes)
    lean = max_sparsity > 1 and N <= 1024 and ((bank.L + 31) // 32) * 32 <= 512
    flops, chain = 0.0, False
    for d in range(max_sparsity):
        if lean and (connections == 1 or connections ** d >= F[d + 1]):
            chain = True
        flops += F[d] * (per_tile if chain else runs) * 16 * 16384.0
    return flops


def config_leg(name, dic, d_sig, Kmax, c, B, thr, reps, n_sub, extra=None):
    """One BASELINE configuration as a timed leg: `reps` device calls on resident signals."""
    import torch
    from csc_amd import _lib
    bank = dic.device_bank()
    S, N = d_sig.shape
    out = bank.mmpdf_launch(d_sig, Kmax, c, B, thr, want_coef=False)           # warm-up; also this run's stats
    torch.cuda.synchronize(bank.device)
    stats = out["stats"].cpu().numpy().tolist()
    t0 = time.perf_counter()
    for _ in range(reps):
        bank.mmpdf_launch(d_sig, Kmax, c, B, thr, want_coef=False, out=out)
    torch.cuda.synchronize(bank.device)
    dt = (time.perf_counter() - t0) / reps
    sizes, NN = _lib.tree_shape(Kmax, c, B)
    n_sel = 1 + sum(sizes[:-1])
    P = N - bank.L + 1
    dense = n_sel * 2.0 * P * bank.K * bank.L
    one_tile = bank.K <= 144 and P <= 256
    issued = issued_fp16_flops_per_signal(bank, N, Kmax, c, B) if one_tile else None
    sub = d_sig[:n_sub]
    a = bank.mmpdf_launch(sub, Kmax, c, B, thr, want_coef=False)
    bank.fp32_only = True
    try:
        b = bank.mmpdf_launch(sub, Kmax, c, B, thr, want_coef=False)
    finally:
        bank.fp32_only = False
    same = bool(torch.equal(a["rec_buf"], b["rec_buf"]) and torch.equal(a["node_pos"], b["node_pos"])
                and torch.equal(a["node_atom"], b["node_atom"]))
    leg = {"config": name, "value": S / dt, "unit": "signals/s", "signals": S, "ms_per_call": dt * 1e3, "calls": reps,
           "atoms": bank.K, "atom_length": bank.L, "signal_length": N, "nodes_per_signal": NN, "selecting_nodes": n_sel,
           "dense_equivalent_tflops": S * dense / dt / 1e12,
           "dense_equivalent_over_fp16_peak": S * dense / dt / 1e12 / FP16_MFMA_PEAK_TFLOPS,
           "dense_equivalent_over_fp32_peak": S * dense / dt / 1e12 / FP32_MFMA_PEAK_TFLOPS,
           "issued_fp16_tflops": None if issued is None else S * issued / dt / 1e12,
           "issued_over_fp16_peak": None if issued is None else S * issued / dt / 1e12 / FP16_MFMA_PEAK_TFLOPS,
           "stats": {"fp32_argmax_fallbacks": stats[0], "duplicate_columns": stats[1],
                     "rescored_candidates_per_selection": stats[2] / max(1, S * NN), "nodes_repeated_in_fp32": stats[3]},
           "identical_to_fp32": same, "identical_to_fp32_signals": int(sub.shape[0])}
    if extra:
        leg.update(extra)
    return leg


def other_configs(dic, d_sig2):
    """BASELINE.json configs[0], [2], [3] on one GPU (configs[1] is the line itself, configs[4] is `config5`)."""
    import torch
    import csc_amd
    legs = []
    np.random.seed(0)
    sig1 = np.array([dic.generateConstrainedTestSignal(SIGNAL_LENGTH, 3, 10, 10, 0.75)[0] for _ in range(200)])
    dev = dic.device_bank().device
    legs.append(config_leg("1: convolutional OMP, 200 constrained signals, sparsity 3, SNR 10 dB (a grid of 200 workgroups "
                           "cannot fill 256 CUs: latency-bound)", dic, torch.from_numpy(sig1).to(dev), 3, 1, 1, 1.0, 20, 200))
    legs.append(config_leg("4a: deep tree, connections 4 / branches 4 / max sparsity 7 (28 nodes), 3200 signals",
                           dic, d_sig2, 7, 4, 4, DISSIMILARITY, 5, 256))
    legs.append(config_leg("4b: deep tree with connection dedup, connections 4 / branches 64 / max sparsity 7 (340 nodes), "
                           "800 signals", dic, d_sig2[::4].contiguous(), 7, 4, 64, DISSIMILARITY, 3, 128))
    t0 = time.perf_counter()
    large = csc_amd.ZSDictionary.from_values(np.arange(0.005, 0.06, 0.0005), np.arange(0.010, 0.036, 0.0003), [1])
    lbank = large.device_bank()
    torch.cuda.synchronize(dev)
    t_dict = time.perf_counter() - t0
    n3 = 4096
    sig3, _ = lbank.generate(np.full(n3, 4, np.int32), np.full(n3, 10.0), 777, 10, 0.75, seed=3)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    lbank.similarity_table(777, DISSIMILARITY)
    torch.cuda.synchronize(dev)
    t_sim = time.perf_counter() - t0
    legs.append(config_leg("3: large dictionary (b step 0.0005 x y step 0.0003: 9570 atoms of 518 samples), MMP-DF connections 3 / "
                           "branches 3 / max sparsity 5, 4096 device-generated constrained signals (N = 777, sparsity 4, SNR 10 dB)",
                           large, sig3, MAX_SPARSITY, CONNECTIONS, BRANCHES, DISSIMILARITY, 2, 64,
                           extra={"dictionary_build_and_pack_s": t_dict, "similarity_table_build_s": t_sim,
                                  "note": "low-rank screen (rank 64, rigorous per-atom bounds) + single-term fp16 screen of the "
                                          "blocks in reach + three-product fp16 kernel, float64 re-score; the issued flops "
                                          "depend on the blocks and units each selection visits (not counted here)"}))
    del sig3, large, lbank
    torch.cuda.empty_cache()
    return legs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)    # a step is ~1.3 ms: the timed region lasts > 1 s, clocks
    ap.add_argument("--warmup", type=int, default=100)    # settled (100 steps)
    ap.add_argument("--per-cell", type=int, default=PER_CELL, help="signals per (snr, sparsity) cell per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the fp32-only comparison run (profiling)")
    ap.add_argument("--no-config5", action="store_true", help="skip the 125 000-signals-per-rank leg")
    ap.add_argument("--no-configs", action="store_true", help="skip the config 1 / 3 / 4 legs (N = 1)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="gloo: CPU rehearsal of the multi-rank control flow (tests/test_bench_world2.py injects a CPU "
                         "stand-in for the device bank); no kernels, no roofline")
    ap.add_argument("--streams", type=int, default=2, choices=(1, 2),
                    help="HIP streams the timed steps alternate between (2: the root level of one batch runs under the "
                         "chains of the batch before it; 1: one batch after the other)")
    ap.add_argument("--fp32-only", action="store_true",
                    help="diagnostic (profiling the fp32 leg): the timed loop itself runs with CSC_FLAG_FP32_ONLY")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N > 1")

    dic = canonical_dictionary()
    signals, clean, sigma = make_signals(dic, seed=rank, per_cell=a.per_cell, with_parts=True)
    S, N = signals.shape

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline(signals if a.per_cell == PER_CELL else make_signals(dic, 0))   # before any GPU init

    import torch
    import torch.distributed as dist
    from csc_amd import _lib
    from csc_amd import distributed as D
    gpu = a.backend == "nccl"
    dev = torch.device("cuda", local_rank) if gpu else torch.device("cpu")
    use_dist = world > 1 or "RANK" in os.environ          # launched by torch.distributed.run (also with 1 rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if gpu:
            torch.cuda.set_device(local_rank)             # for init_process_group(device_id=...); csc_amd picks
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # cuda:LOCAL_RANK by itself
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    bank = dic.device_bank()
    if gpu:
        assert bank.device == dev == D.local_device(), (bank.device, dev)
    else:
        a.no_fp32_leg = a.no_config5 = a.no_configs = True
    if a.fp32_only:
        bank.fp32_only = True
        a.no_fp32_leg = True

    def sync():
        if gpu:
            torch.cuda.synchronize(dev)

    K, L = bank.K, bank.L
    P = N - L + 1
    sizes, NN = _lib.tree_shape(MAX_SPARSITY, CONNECTIONS, BRANCHES)
    n_sel = 1 + sum(sizes[:-1])
    flops_per_signal = n_sel * 2.0 * P * K * L
    rec_bytes = MAX_SPARSITY * (16 + 8 * MAX_SPARSITY)

    def solve(sig, **kw):
        """The product's sharded sparVar call: device signals in, device records out (rank 0: all ranks')."""
        return dic.mmpdfSparVarRecords(sig, CONNECTIONS, DISSIMILARITY, BRANCHES, MAX_SPARSITY, sharded=True,
                                       presharded=True, **kw)

    # PCIe-inclusive pass (host signals in, host records out) -- reported, never `value`
    t0 = time.time()
    d_sig = torch.from_numpy(signals).to(dev)
    host_rec = bank.mmpdf_records(d_sig, MAX_SPARSITY, CONNECTIONS, BRANCHES, DISSIMILARITY)[0].cpu()
    first_call_s = time.time() - t0
    t0 = time.time()
    d_sig2 = torch.from_numpy(signals).to(dev)
    host_rec = bank.mmpdf_records(d_sig2, MAX_SPARSITY, CONNECTIONS, BRANCHES, DISSIMILARITY)[0].cpu()
    pcie_s = time.time() - t0
    del d_sig2

    # ranks that really take part (one all_reduce of ones: a rank that died or never joined shows here)
    ranks_seen = 1
    if use_dist:
        one = torch.ones(1, dtype=torch.int64, device=dev)
        dist.all_reduce(one)
        ranks_seen = int(one.item())

    shard_sizes = [S] * world
    # two sets of receive buffers: step k's gather is enqueued into set k % 2 and runs under step k + 1's solve
    recv = [[torch.empty((S, rec_bytes), dtype=torch.uint8, device=dev) for _ in range(world)] for _ in range(2)] \
        if (use_dist and rank == 0) else [None, None]
    pending = [None, None]

    # The steps alternate between two HIP streams (--streams 2, the default): every batch is a chain of dependent
    # launches -- the root level on 2 workgroups per CU, then the chains on 5 -- whose first and last workgroups leave CUs
    # idle; with the next batch on the other stream its root level fills them.  Step k uses stream k % 2, receive-buffer
    # set k % 2 and (through torch's allocator) its own workspace and outputs; all of it is waited for before the clock stops.
    import contextlib
    streams = [torch.cuda.Stream(dev) for _ in range(a.streams)] if gpu else []

    def on_stream(k):
        return torch.cuda.stream(streams[k % len(streams)]) if len(streams) > 1 else contextlib.nullcontext()

    def step(k):
        """One step: wait for the gather that last used this buffer set, solve, enqueue this step's gather."""
        with on_stream(k):
            rows = pending[k & 1].wait() if pending[k & 1] is not None else None
            pending[k & 1] = solve(d_sig, shard_sizes=shard_sizes, recv=recv[k & 1], async_gather=True)
        return rows

    def drain():
        rows = None
        for q in (0, 1):
            if pending[q] is not None:
                r = pending[q].wait()
                rows = r if r is not None else rows
                pending[q] = None
        return rows

    for k in range(a.warmup):
        step(k)
    drain()
    timer = _lib.LevelTimer(capacity=a.steps * MAX_SPARSITY) if gpu else None
    sync()
    if use_dist:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    if timer is not None:
        timer.__enter__()
    for k in range(a.steps):
        step(k)
    rows = pending[(a.steps - 1) & 1].wait() if a.steps > 0 else None     # the LAST step's rows (rank 0)
    pending[(a.steps - 1) & 1] = None
    drain()
    sync()
    if use_dist:
        dist.barrier()
    sync()
    elapsed = time.perf_counter() - t0
    spans = []
    if timer is not None:
        timer.__exit__(None, None, None)
        spans = timer.read()
        timer.close()

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    if rank == 0:      # rank 0's own records come first in the gathered rows
        assert rows is not None and rows.shape == (world * S, rec_bytes), (None if rows is None else rows.shape)
        assert torch.equal(rows[:S].cpu(), host_rec), "gathered records differ from the single-call records"
    own, stats_t = bank.mmpdf_records(d_sig, MAX_SPARSITY, CONNECTIONS, BRANCHES, DISSIMILARITY)
    stats = stats_t.cpu().numpy().tolist()

    # the same batches on ONE stream (outside the timed region, no collective): the level kernels' launch times when
    # nothing else shares the chip with them -- the roofline of the kernels themselves, next to the timed region's
    spans_1, one_elapsed, steps_1 = [], 0.0, 0
    if gpu and len(streams) > 1 and rank == 0:
        out_1 = bank.mmpdf_launch(d_sig, MAX_SPARSITY, CONNECTIONS, BRANCHES, DISSIMILARITY, want_coef=False)
        steps_1 = max(2, min(300, a.steps))
        for _ in range(min(50, steps_1)):
            bank.mmpdf_launch(d_sig, MAX_SPARSITY, CONNECTIONS, BRANCHES, DISSIMILARITY, want_coef=False, out=out_1)
        torch.cuda.synchronize(dev)
        timer_1 = _lib.LevelTimer(capacity=steps_1 * MAX_SPARSITY)
        t0 = time.perf_counter()
        with timer_1:
            for _ in range(steps_1):
                bank.mmpdf_launch(d_sig, MAX_SPARSITY, CONNECTIONS, BRANCHES, DISSIMILARITY, want_coef=False, out=out_1)
            torch.cuda.synchronize(dev)
        one_elapsed = time.perf_counter() - t0
        spans_1 = timer_1.read()
        timer_1.close()
        del out_1

    # the same batch correlated in fp32 only (CSC_FLAG_FP32_ONLY): the "% of fp32 MFMA peak" leg of
    # BASELINE.json's metric; outside the timed region, no collective
    spans_m, fp32_elapsed, msteps, same = [], 0.0, 0, None
    if not a.no_fp32_leg:
        bank.fp32_only = True
        out_m = bank.mmpdf_launch(d_sig, MAX_SPARSITY, CONNECTIONS, BRANCHES, DISSIMILARITY, want_coef=False)
        torch.cuda.synchronize(dev)
        msteps = max(2, min(20, a.steps))
        timer_m = _lib.LevelTimer(capacity=msteps * MAX_SPARSITY)
        t0 = time.perf_counter()
        with timer_m:
            for _ in range(msteps):
                bank.mmpdf_launch(d_sig, MAX_SPARSITY, CONNECTIONS, BRANCHES, DISSIMILARITY, want_coef=False, out=out_m)
            torch.cuda.synchronize(dev)
        fp32_elapsed = time.perf_counter() - t0
        spans_m = timer_m.read()
        timer_m.close()
        bank.fp32_only = False
        same = bool(torch.equal(out_m["rec_buf"], own))

    # ---- config 5: 125 000 signals per rank, chunked solve, ONE gather at the end ---------------------------
    config5 = None
    if not a.no_config5:
        g = torch.Generator(device=dev)
        g.manual_seed(1000 + rank)
        reps = (CONFIG5_PER_RANK + S - 1) // S
        d_clean = torch.from_numpy(clean).to(dev).repeat(reps, 1)[:CONFIG5_PER_RANK]
        d_sigma = torch.from_numpy(sigma).to(dev).repeat(reps)[:CONFIG5_PER_RANK]
        big = d_clean + d_sigma[:, None] * torch.randn(d_clean.shape, generator=g, dtype=torch.float64, device=dev)
        del d_clean
        sizes5 = [CONFIG5_PER_RANK] * world
        solve(big, shard_sizes=sizes5)                     # untimed: allocator warm-up
        sync()
        if use_dist:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        nrep = 3
        for _ in range(nrep):
            rows5 = solve(big, shard_sizes=sizes5)
        sync()
        if use_dist:
            dist.barrier()
        sync()
        t5 = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if use_dist:
            dist.all_reduce(t5, op=dist.ReduceOp.MAX)
        if rank == 0:
            assert rows5.shape == (world * CONFIG5_PER_RANK, rec_bytes)
            config5 = {"value": world * CONFIG5_PER_RANK * nrep / float(t5.item()), "unit": "signals/s",
                       "signals_per_rank": CONFIG5_PER_RANK, "total_signals": world * CONFIG5_PER_RANK,
                       "ms_per_pass": float(t5.item()) / nrep * 1e3, "passes": nrep,
                       "chunk": dic.PIPELINE_CHUNK, "gathers_per_pass": 1 if use_dist else 0,
                       "note": "config-2 clean signals repeated with fresh Gaussian noise at each signal's SNR "
                               "(seed 1000 + rank), resident in HBM; mmpdfSparVarRecords(sharded=True): chunked "
                               "solve, then one gather of the %d-byte records" % rec_bytes}
        del big

    if rank == 0:
        ms_step = elapsed / a.steps * 1e3
        value = world * S * a.steps / elapsed
        # a span = the launches of one tree level, or of a whole run of levels when one launch walks the chains
        # below the branching levels (tagged with its first level); `launch_ms` = level-kernel time per tree level
        if not gpu:
            # gloo rehearsal: the control flow ran (barrier, async gathers, asserts above); no kernel, no roofline
            print(json.dumps({"metric": "signals/sec for MMP-DF (3 branches, sparsity<=5)", "value": value,
                              "unit": "signals/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                              "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                              "dtype": "f64 (CPU stand-in)", "data": "synthetic", "ranks_seen": ranks_seen,
                              "config": {"workload": "REHEARSAL of bench.py's multi-rank control flow on the gloo "
                                                     "backend with a CPU stand-in bank: not a measurement",
                                         "signals_per_gpu": S}}), flush=True)
            if use_dist:
                dist.barrier()
                dist.destroy_process_group()
            return
        launch_ms = float(sum(ms for _, ms in spans)) / (a.steps * MAX_SPARSITY)
        per_level = {str(d): float(np.mean([ms for lv, ms in spans if lv == d]))
                     for d in sorted({lv for lv, _ in spans})}
        # ---- algorithmic flops: every selecting node correlates its residual with all
        # K atoms at all P shifts, 2*P*K*L flops, n_sel = 13 nodes per signal over the 5 tree levels
        flops_per_launch = S * flops_per_signal / MAX_SPARSITY
        # With two streams a launch of the timed region shares the chip with the other stream's launches (and waits for
        # its workgroup slots): its span is not the kernel's own time.  The roofline of the KERNELS is then taken from the
        # single-stream leg behind the timed region (the same batches, `roofline.measured_in`); the timed region's own
        # spans and its overall rate are reported beside it (`roofline.timed_region`).
        timed_region = None
        if spans_1:
            timed_region = {"launch_ms_avg": launch_ms, "launch_ms_by_level": per_level, "launches_timed": len(spans),
                            "dense_equivalent_tflops": world * S * a.steps * flops_per_signal / elapsed / 1e12 / world,
                            "note": "HIP-event spans of the timed region's launches (two streams: they overlap, their sum "
                                    "exceeds the elapsed time) and the algorithmic flops of all its steps over the "
                                    "elapsed time, per GPU"}
            timed_region["dense_equivalent_over_fp16_peak"] = timed_region["dense_equivalent_tflops"] / FP16_MFMA_PEAK_TFLOPS
            launch_ms = float(sum(ms for _, ms in spans_1)) / (steps_1 * MAX_SPARSITY)
            per_level = {str(d): float(np.mean([ms for lv, ms in spans_1 if lv == d]))
                         for d in sorted({lv for lv, _ in spans_1})}
        achieved = flops_per_launch / (launch_ms * 1e-3) / 1e12
        # MFMA flops actually issued by the fp16 filter (issued_fp16_flops_per_signal: the root on the register-resident
        # kernel, the chains' nodes on the chain kernel with per-tile slab ranges)
        blocks16, blocks_chain, nslab, blocks = product_blocks(bank)
        issued_per_signal = issued_fp16_flops_per_signal(bank, N, MAX_SPARSITY, CONNECTIONS, BRANCHES)
        issued16 = S * issued_per_signal / MAX_SPARSITY / (launch_ms * 1e-3) / 1e12
        fp32_path = None
        if spans_m:
            launch_ms_m = float(sum(ms for _, ms in spans_m)) / (msteps * MAX_SPARSITY)
            achieved_m = flops_per_launch / (launch_ms_m * 1e-3) / 1e12
            # fp32 MFMA flops issued: blocks holding only the atoms' zero padding are skipped, every other
            # block costs 4 waves x 32 v_mfma_f32_16x16x4_f32 (2048 flop each) per node
            issued = S * n_sel * blocks * 4 * 32 * 2048.0 / MAX_SPARSITY / (launch_ms_m * 1e-3) / 1e12
            fp32_path = {
                "note": "same batch with CSC_FLAG_FP32_ONLY: every node correlates its residual on "
                        "v_mfma_f32_16x16x4_f32 (DeviceBank.fp32_only = True); identical records: %s" % same,
                "value": S * msteps / fp32_elapsed, "unit": "signals/s",
                "roofline": {"bound": "mfma", "achieved": issued, "peak": FP32_MFMA_PEAK_TFLOPS,
                             "unit": "TFLOP/s", "frac": issued / FP32_MFMA_PEAK_TFLOPS,
                             "dense_equivalent_tflops": achieved_m,
                             "dense_equivalent_over_peak": achieved_m / FP32_MFMA_PEAK_TFLOPS,
                             "launch_ms_avg": launch_ms_m,
                             "launch_ms_by_level": {str(d): float(np.mean([ms for lv, ms in spans_m if lv == d]))
                                                    for d in sorted({lv for lv, _ in spans_m})},
                             "flops_per_launch_dense": flops_per_launch,
                             "blocks_per_node": blocks, "of_dense_blocks": nslab * 9,
                             "note": "`achieved` / `frac` count the fp32 MFMA flops really issued (bank blocks that "
                                     "are all zero padding are skipped); `dense_equivalent_*` counts the "
                                     "reference's dense 2*P*K*L flops per node over the same time and can "
                                     "exceed 1"}}
        traffic, traffic_src = None, None
        tf = os.path.join(ROOT, TRAFFIC_PROFILE)
        if os.path.exists(tf):
            with open(tf) as f:
                traffic = json.load(f).get("bytes_per_launch")
            traffic_src = TRAFFIC_PROFILE + " (static: rocprofv3 PMC passes over this command, taken when the " \
                                            "profile was committed; not measured by this run)"
        # HBM side of the roofline: bytes per launch from the committed PMC profile over the live launch time
        hbm_gbs = None if traffic is None else traffic / (launch_ms * 1e-3) / 1e9
        configs = None
        if world == 1 and not a.no_configs and not a.fp32_only:
            configs = other_configs(dic, d_sig)
        line = {
            "metric": "signals/sec for MMP-DF (3 branches, sparsity<=5)", "value": value, "unit": "signals/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (diagnostic run: --fp32-only)" if a.fp32_only else "f16",
            "data": "synthetic",
            "config": {"workload": "config 2: 3200-signal sparVar sweep (sparsity 2-5 x SNR 0-15 dB, 200 each, "
                                   "constrained), MMP-DF connections 3 / branches 3 / max sparsity 5 / "
                                   "dissimilarity 0.8, per GPU",
                       "signals_per_gpu": S, "signal_length": N, "atoms": K, "atom_length": L,
                       "nodes_per_signal": NN,
                       "parallelism": (f"signals sharded over {world} ranks (one per GPU), no traffic during the "
                                       f"solve, one RCCL gather of the {rec_bytes}-byte device-resident sparVar "
                                       "records per step (ZSDictionary.mmpdfSparVarRecords(sharded=True, "
                                       "async_gather=True)): a step's gather is enqueued into one of two receive-buffer "
                                       "sets and runs under the next step's solve; every gather is waited for before "
                                       "the clock stops (the per-pass form with ONE gather per 125 000 signals per "
                                       "rank is `config5`)")
                       if world > 1 else "single GPU (ZSDictionary.mmpdfSparVarRecords)",
                       "streams": len(streams),
                       "overlap": ("the steps alternate between two HIP streams: a batch is a chain of dependent launches "
                                   "(root level, then the chains), and the next batch's root level runs in what the "
                                   "chains' last workgroups leave idle; every step's work is complete before the clock stops")
                       if len(streams) > 1 else "none: one batch after the other on one stream"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP16_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "hbm_gb_per_s": hbm_gbs, "hbm_peak_gb_per_s": HBM_PEAK_GBS,
                         "hbm_frac": None if hbm_gbs is None else hbm_gbs / HBM_PEAK_GBS,
                         "issued_tflops": issued16, "issued_frac": issued16 / FP16_MFMA_PEAK_TFLOPS,
                         "kernel": "mmp_level_lean_kernel (levels 1-4, one launch) + mmp_level_kernel (level 0)",
                         "launch_ms_avg": launch_ms, "launch_ms_by_level": per_level,
                         "measured_in": ("single-stream leg: %d steps of the same batch, one after the other on one stream, "
                                         "behind the timed region (%.3f ms per step, %.0f signals/s); "
                                         "profiles/r03_kernel_stats.csv is rocprofv3 of `bench.py --streams 1`"
                                         % (steps_1, one_elapsed / steps_1 * 1e3, S * steps_1 / one_elapsed))
                         if spans_1 else "the timed region",
                         "timed_region": timed_region,
                         "flops_per_launch": flops_per_launch, "flops_per_signal": flops_per_signal,
                         "launches_timed": len(spans_1) if spans_1 else len(spans),
                         "frac_of_fp32_mfma_peak": achieved / FP32_MFMA_PEAK_TFLOPS,
                         "blocks_per_node": {"register_resident_kernel": blocks16, "chain_kernel": blocks_chain},
                         "of_dense_blocks": nslab * 9,
                         "note": ("`achieved` / `frac` / `launch_ms_*` are the level kernels' own launch times (see "
                                  "`measured_in`): in the timed region two streams share the chip; "
                                  if spans_1 else "") +
                                 "correlations on the fp16 matrix cores as a filter (arithmetic dtype f16, fp32 "
                                 "accumulation), selection decided by a float64 re-score of the entries in the "
                                 "fp16 rounding band; `achieved` / `frac` count the reference's dense 2*P*K*L flops "
                                 "per selecting node (algorithmic flops per launch / measured launch time), "
                                 "`issued_*` the v_mfma_f32_16x16x32_f16 flops issued (all-padding bank blocks are "
                                 "skipped; a rare second pass of a node is not counted); the answer "
                                 "to BASELINE.json's '% fp32 MFMA peak' is fp32_path"},
            "fp32_path": fp32_path,
            "config5": config5,
            "configs": configs,
            "ranks_seen": ranks_seen,
            "cpu_baseline": cpu,
            "pcie_inclusive": {"value": S / pcie_s, "unit": "signals/s",
                               "note": "one call: host f64 signals -> HBM, solve, sparVar records -> host",
                               "first_call_s": first_call_s},
            "fp32_argmax_fallbacks": stats[0], "duplicate_columns": stats[1],
            "rescored_candidates_per_selection": stats[2] / (S * NN),
            "nodes_repeated_in_fp32": stats[3],
        }
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
