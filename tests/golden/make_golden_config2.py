"""Expected trees for ALL 3200 signals of BASELINE.json's config 2, from the CPU oracle (mode 'direct', itself
pinned against the reference by tests/test_oracle_golden.py).

    python tests/golden/make_golden_config2.py        # ~4 min on 8 cores -> tests/golden/config2_trees.npz

The signals are regenerated from seed 0 with the product's host-side generator (the reference's draw order,
src/csc/dictionary.py:271-365); their sha256 is stored so that the GPU test can tell whether it regenerated the
same batch (same NumPy RNG stream) before it compares.  Node (level d, index j) is column off_d + j, as in
include/csc_hip.h."""
import hashlib
import itertools
import multiprocessing as mp
import os
import sys

for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ[v] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np


def config2_signals():
    import csc_amd
    d = csc_amd.ZSDictionary.from_values(np.arange(0.005, 0.06, 0.005), np.arange(0.010, 0.036, 0.002), [1])
    np.random.seed(0)
    sigs = []
    for snr, sp in itertools.product([0, 5, 10, 15], [2, 3, 4, 5]):
        for _ in range(200):
            sigs.append(d.generateConstrainedTestSignal(738, sp, snr, 10, 0.75)[0])
    return np.array(sigs)


def tree(sig):
    from oracle import csc_oracle as O
    od = tree.od if hasattr(tree, "od") else O.canonical_dictionary()
    tree.od = od
    t = O.OracleTree(od, sig, 5, 3, 0.8, "direct")
    t.run(3)
    pos, atom, mse = np.zeros(15, np.int16), np.zeros(15, np.int16), np.zeros(15)
    for path, (p, a, m) in t.all_nodes().items():
        digits = [int(x) for x in path.split("-")]
        nid = 3 * (len(digits) - 1) + sum((o - 1) * 3 ** i for i, o in enumerate(digits))
        pos[nid], atom[nid], mse[nid] = p, a, m
    return pos, atom, mse


if __name__ == "__main__":
    sigs = config2_signals()
    with mp.get_context("spawn").Pool(min(16, os.cpu_count() or 1)) as pool:
        out = pool.map(tree, list(sigs), chunksize=8)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "config2_trees.npz"),
                        node_pos=np.array([o[0] for o in out]), node_atom=np.array([o[1] for o in out]),
                        node_mse=np.array([o[2] for o in out]),
                        signals_sha256=hashlib.sha256(np.ascontiguousarray(sigs).tobytes()).hexdigest())
    print("wrote config2_trees.npz", len(out))
