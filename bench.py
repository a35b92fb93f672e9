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
over RCCL); with N > 1 every rank holds its own 3200 signals (weak scaling).

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
CPU_SAMPLE = [i * PER_CELL + o for i in range(16) for o in (7, 107)]    # two signals of every (snr, sparsity) cell: ~25 CPU-s
CONFIG5_PER_RANK = 125000
TRAFFIC_PROFILE = os.path.join("profiles", "r02_hbm_traffic.json")


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
    return {"value": len(sample) / wall, "unit": "signals/s", "cores": cores, "kind": "port",
            "sample": f"{len(sample)} of the {len(signals)} signals (indices {CPU_SAMPLE}), oracle mode "
                      f"'faithful' (oaconvolve + lsqr, one tree per sparsity = 50 nodes/signal), one "
                      f"single-threaded process per signal (OMP_NUM_THREADS={threads[0]} in the workers), "
                      f"{sum(per):.1f} CPU-s, {np.mean(per):.2f} s/signal/core",
            "one_process": {"s_per_signal": float(np.mean(solo)), "signals": len(solo),
                            "note": "the same job on one process with the other cores idle"},
            "host_cores_available": avail}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)    # a step is ~1.3 ms: the timed region lasts > 1 s, clocks
    ap.add_argument("--warmup", type=int, default=100)    # settled (100 steps)
    ap.add_argument("--per-cell", type=int, default=PER_CELL, help="signals per (snr, sparsity) cell per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the fp32-only comparison run (profiling)")
    ap.add_argument("--no-config5", action="store_true", help="skip the 125 000-signals-per-rank leg")
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
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or "RANK" in os.environ          # launched by torch.distributed.run (also with 1 rank)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        torch.cuda.set_device(local_rank)                 # for init_process_group(device_id=...); csc_amd picks
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # cuda:LOCAL_RANK by itself
    bank = dic.device_bank()
    assert bank.device == dev == D.local_device(), (bank.device, dev)
    if a.fp32_only:
        bank.fp32_only = True
        a.no_fp32_leg = True
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

    shard_sizes = [S] * world
    recv = [torch.empty((S, rec_bytes), dtype=torch.uint8, device=dev) for _ in range(world)] \
        if (use_dist and rank == 0) else None
    step_kw = dict(shard_sizes=shard_sizes, recv=recv)

    rows = None
    for _ in range(a.warmup):
        rows = solve(d_sig, **step_kw)
    timer = _lib.LevelTimer(capacity=a.steps * MAX_SPARSITY)
    torch.cuda.synchronize(dev)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    with timer:
        for _ in range(a.steps):
            rows = solve(d_sig, **step_kw)
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    spans = timer.read()
    timer.close()

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    if rank == 0:      # rank 0's own records come first in the gathered rows
        assert rows is not None and rows.shape == (world * S, rec_bytes)
        assert torch.equal(rows[:S].cpu(), host_rec), "gathered records differ from the single-call records"
    own, stats_t = bank.mmpdf_records(d_sig, MAX_SPARSITY, CONNECTIONS, BRANCHES, DISSIMILARITY)
    stats = stats_t.cpu().numpy().tolist()

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
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        nrep = 3
        for _ in range(nrep):
            rows5 = solve(big, shard_sizes=sizes5)
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)
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
        launch_ms = float(sum(ms for _, ms in spans)) / (a.steps * MAX_SPARSITY)
        per_level = {str(d): float(np.mean([ms for lv, ms in spans if lv == d]))
                     for d in sorted({lv for lv, _ in spans})}
        # ---- algorithmic flops: every selecting node correlates its residual with all
        # K atoms at all P shifts, 2*P*K*L flops, n_sel = 13 nodes per signal over the 5 tree levels
        flops_per_launch = S * flops_per_signal / MAX_SPARSITY
        achieved = flops_per_launch / (launch_ms * 1e-3) / 1e12
        # MFMA flops actually issued by the fp16 filter: per slab the column tiles from 6, 3 or 0 on (the kernel's five
        # runs over the support-sorted bank, corr_tile_gemm16), 4 waves x 4 v_mfma_f32_16x16x32_f16 (16384 flop) per
        # (slab, tile) and node
        nslab = (L + 31) // 32
        off = nslab * 9 * 512
        meta = bank.bank32t[off:off + 208].view(torch.int32).cpu().numpy()
        first = [(int(m) & -int(m)).bit_length() - 1 if m else 9 for m in meta[144:144 + nslab]]
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
        blocks16 = 3 * p6 + 6 * (p3 - p6) + 9 * (q3 - p3) + 6 * (q6 - q3) + 3 * (nslab - q6)
        issued16 = S * n_sel * blocks16 * 4 * 4 * 16384.0 / MAX_SPARSITY / (launch_ms * 1e-3) / 1e12
        fp32_path = None
        if spans_m:
            launch_ms_m = float(sum(ms for _, ms in spans_m)) / (msteps * MAX_SPARSITY)
            achieved_m = flops_per_launch / (launch_ms_m * 1e-3) / 1e12
            # fp32 MFMA flops issued: blocks holding only the atoms' zero padding are skipped, every other
            # block costs 4 waves x 32 v_mfma_f32_16x16x4_f32 (2048 flop each) per node
            blocks = sum(9 - (int(m) & -int(m)).bit_length() + 1 for m in meta[144:144 + nslab] if m)
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
                                       "records per step (ZSDictionary.mmpdfSparVarRecords(sharded=True))")
                       if world > 1 else "single GPU (ZSDictionary.mmpdfSparVarRecords)"},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": FP16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / FP16_MFMA_PEAK_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                         "issued_tflops": issued16, "issued_frac": issued16 / FP16_MFMA_PEAK_TFLOPS,
                         "kernel": "mmp_level_lean_kernel (levels 1-4, one launch) + mmp_level_kernel (level 0)",
                         "launch_ms_avg": launch_ms, "launch_ms_by_level": per_level,
                         "flops_per_launch": flops_per_launch, "flops_per_signal": flops_per_signal,
                         "launches_timed": len(spans),
                         "frac_of_fp32_mfma_peak": achieved / FP32_MFMA_PEAK_TFLOPS,
                         "blocks_per_node": blocks16, "of_dense_blocks": nslab * 9,
                         "note": "correlations on the fp16 matrix cores as a filter (arithmetic dtype f16, fp32 "
                                 "accumulation), selection decided by a float64 re-score of the entries in the "
                                 "fp16 rounding band; `achieved` / `frac` count the reference's dense 2*P*K*L flops "
                                 "per selecting node (algorithmic flops per launch / measured launch time), "
                                 "`issued_*` the v_mfma_f32_16x16x32_f16 flops issued when no column group is "
                                 "multiplied twice (lower bound; all-padding column tiles are skipped); the answer "
                                 "to BASELINE.json's '% fp32 MFMA peak' is fp32_path"},
            "fp32_path": fp32_path,
            "config5": config5,
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
