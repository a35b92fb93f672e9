"""CPU rehearsal of bench.py's multi-rank control flow (world_size 2, gloo): the code that first meets several GPUs
at round end -- process-group set-up, the ranks_seen all_reduce, the two sets of receive buffers, the async gathers
enqueued per step and waited for before reuse, the elapsed-time all_reduce, rank 0's shape / equality asserts and the
ONE JSON line -- runs here once with tests/helpers.OracleBank standing in for the device bank
(`bench.py --backend gloo`).  No kernel runs; the numbers mean nothing."""
import json
import os
import subprocess
import sys

from conftest import ROOT

WORKER = r'''
import os, sys
root, rank, world, port = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
import torch
import bench
from helpers import OracleBank
from oracle import csc_oracle as O
real = bench.canonical_dictionary


def stand_in():
    d = real()
    bank = OracleBank(O.canonical_dictionary())
    bank.device = torch.device("cpu")
    d.device_bank = lambda: bank
    return d


bench.canonical_dictionary = stand_in
sys.argv = ["bench.py", "--gpus", str(world), "--steps", "3", "--warmup", "1", "--per-cell", "1", "--backend", "gloo",
            "--no-cpu-baseline"]
bench.main()
'''


def test_bench_multi_rank_control_flow_gloo_world2(tmp_path):
    script = tmp_path / "bench_worker.py"
    script.write_text(WORKER)
    port = str(31500 + os.getpid() % 2000)
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), "2", port], stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE) for r in range(2)]
    outs = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1].decode()[-2000:] for o in outs]
    lines = [l for l in outs[0][0].decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, outs[0][0].decode()
    assert not [l for l in outs[1][0].decode().splitlines() if l.startswith("{")]      # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    assert line["scaling"] == "weak" and line["value"] > 0 and line["config"]["signals_per_gpu"] == 16
