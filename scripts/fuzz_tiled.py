"""Development aid: random dictionaries BEYOND one register tile (K > 144 atoms and / or P > 256 positions: the screen +
three-product fp16 kernels in front of the fp32 tiled kernel) against the fp32-only path, which must give identical
trees: row blocks with and without a remainder tile, partly filled last bank tiles, short and long signals, one to four
children per node.  Usage (GPU box): python scripts/fuzz_tiled.py [trials] [seed].  One line per trial; exit code 1 on a
mismatch."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import csc_amd

P_CHOICES = [40, 200, 256, 257, 260, 272, 273, 300, 512, 515, 528, 529, 600]


def run(trials, seed, verbose=True):
  """-> number of trials whose trees differ between the default (fp16 front) and the fp32-only path"""
  rng = np.random.default_rng(seed)
  bad = 0
  for trial in range(trials):
      nb, ny = int(rng.integers(8, 40)), int(rng.integers(8, 40))
      b = np.linspace(0.005, 0.0595, nb)
      y = np.linspace(0.010, 0.0355, ny)
      d = csc_amd.ZSDictionary.from_values(b, y, [1])
      bank64 = d.getLocalDictionary()
      K, L = bank64.shape
      P = int(P_CHOICES[trial % len(P_CHOICES)]) if trial < 2 * len(P_CHOICES) else int(rng.integers(1, 700))
      if K <= 144 and P <= 256:
          P = 300
      N = L + P - 1
      c, Kx, B = int(rng.integers(1, 5)), int(rng.integers(2, 6)), int(rng.integers(1, 7))
      S = int(rng.integers(8, 48))
      kind = ("noisy", "noiseless", "pairs", "huge")[trial % 4]
      sigs = (0.0 if kind == "noiseless" else 0.05) * rng.standard_normal((S, N))
      for s in range(S):
          for _ in range(int(rng.integers(1, 5))):
              p, k = int(rng.integers(0, P)), int(rng.integers(K))
              amp = rng.uniform(0.3, 1.0)
              sigs[s, p:p + L] += amp * bank64[k]
              if kind == "pairs":
                  p2 = min(P - 1, p + int(rng.integers(0, 3)))
                  sigs[s, p2:p2 + L] += amp * bank64[min(K - 1, k + 1)]
      if kind == "huge":
          sigs *= 1e4
      bank = d.device_bank()
      out = {}
      for name, fp32_only in (("fp16", False), ("fp32", True)):
          bank.fp32_only = fp32_only
          out[name] = bank.mmpdf_batch(sigs, Kx, c, B, 0.8, want_coef=True)
      a, r = out["fp16"], out["fp32"]
      same = np.array_equal(a.node_pos, r.node_pos) and np.array_equal(a.node_atom, r.node_atom)
      close = np.allclose(a.node_mse, r.node_mse, rtol=1e-10, atol=0, equal_nan=True)
      info = dict(trial=trial, kind=kind, K=K, L=L, P=P, c=c, Kx=Kx, B=B, S=S, same=bool(same), mse_close=bool(close),
                  stats={n: [int(x) for x in o.stats] for n, o in out.items()})
      if not same:
          # a pick decided by rounding may differ between the paths only where two float64 correlations tie to ~1e-13: the
          # fp16 path re-scores in float64 and is the reference's choice; report and count
          diff = np.argwhere(a.node_pos != r.node_pos)
          info["first_diff"] = diff[:3].tolist()
          bad += 1
      if verbose:
          print(json.dumps(info), flush=True)
  return bad


if __name__ == "__main__":
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t_start = time.time()
    bad = run(trials, seed)
    print(json.dumps({"trials": trials, "mismatching_trials": bad, "seconds": round(time.time() - t_start, 1)}))
    sys.exit(1 if bad else 0)
