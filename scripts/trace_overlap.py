"""Development aid: do the four chains that share a CU take their products at the same time?  The prof build
(`make -C csc_amd/csrc prof`) records (CU, start, end) of every node's sweep 1 in the chain kernel; this prints, per CU,
how many sweeps overlap on average and a timeline of one CU.  Usage (GPU box):
    CSC_HIP_LIB=$PWD/csc_amd/csrc/libcsc_hip_prof.so python scripts/trace_overlap.py"""
import ctypes as C, sys
import numpy as np, torch
sys.path.insert(0, ".")
import bench
from csc_amd import _lib
dic = bench.canonical_dictionary()
sig = bench.make_signals(dic, 0)
bank = dic.device_bank()
d = torch.from_numpy(sig).cuda()
out = bank.mmpdf_launch(d, 5, 3, 3, 0.8, want_coef=False)
L = _lib.lib()
cap = 1 << 18
buf = (C.c_uint * (4 * cap))()
for _ in range(3):
    bank.mmpdf_launch(d, 5, 3, 3, 0.8, want_coef=False, out=out)
torch.cuda.synchronize()
L.csc_prof_trace_read(buf, cap)                      # drop the warm-up
bank.mmpdf_launch(d, 5, 3, 3, 0.8, want_coef=False, out=out)
torch.cuda.synchronize()
n = L.csc_prof_trace_read(buf, cap)
t = np.frombuffer(buf, dtype=np.uint32)[:4 * n].reshape(n, 4).astype(np.int64)
hw, t0, t1, blk = t[:, 0], t[:, 1], t[:, 2], t[:, 3] & 0x00ffffff
xcc = (t[:, 3] >> 24) & 0xf
dur = (t1 - t0) & 0xffffffff
cu = (hw >> 8) & 0xf | ((hw >> 12) & 0x1) << 4 | ((hw >> 13) & 0x7) << 5 | xcc << 8
print(f"{n} sweeps, {len(np.unique(cu))} CUs, sweep duration mean {dur.mean():.0f} / median {np.median(dur):.0f} clocks "
      f"(min {dur.min()}, max {dur.max()})")
conc, share = [], []
for c in np.unique(cu)[:64]:
    m = cu == c
    a, b = t0[m], (t0[m] + dur[m])
    base = a.min()
    a, b = a - base, b - base                         # (32-bit wrap ignored: a launch is ~2.5 M clocks)
    ev = sorted([(x, 1) for x in a] + [(x, -1) for x in b])
    last, k, w, busy, span = 0, 0, 0.0, 0.0, 0.0
    hist = np.zeros(8)
    for x, dk in ev:
        hist[min(k, 7)] += x - last
        last, k = x, k + dk
    tot = hist.sum()
    conc.append((hist * np.arange(8)).sum() / max(hist[1:].sum(), 1))
    share.append(hist / tot)
share = np.mean(share, axis=0)
print("time share with k sweeps in flight on a CU, k = 0..5:", np.round(share[:6], 3))
print(f"mean number of sweeps in flight while any is: {np.mean(conc):.2f}")
c = np.unique(cu)[5]
m = cu == c
o = np.argsort(t0[m])
base = t0[m].min()
print("one CU, first 24 sweeps (start, end, duration in k clocks; block, depth):")
for i in o[:24]:
    print(f"  {(t0[m][i] - base) / 1e3:8.1f} {(t0[m][i] - base + dur[m][i]) / 1e3:8.1f} {dur[m][i] / 1e3:6.1f}   block {blk[m][i]:5d} depth {t[m][i, 3] >> 28}")
