"""Development aid: does running the two halves of a batch on two streams (half B's level 0 under half A's chains)
pay?  Usage (GPU box): python scripts/overlap_probe.py"""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import bench
dic = bench.canonical_dictionary()
sig = bench.make_signals(dic, 0)
bank = dic.device_bank()
d = torch.from_numpy(sig).cuda()
S = len(sig)
def timed(fn, reps=50):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
out = bank.mmpdf_launch(d, 5, 3, 3, 0.8, want_coef=False)
one = timed(lambda: bank.mmpdf_launch(d, 5, 3, 3, 0.8, want_coef=False, out=out))
res = {"one_stream_ms": one * 1e3, "signals_per_s": S / one}
for parts in (2, 4):
    streams = [torch.cuda.Stream() for _ in range(parts)]
    cuts = [S * i // parts for i in range(parts + 1)]
    halves = [d[cuts[i]:cuts[i + 1]].contiguous() for i in range(parts)]
    outs = [bank.mmpdf_launch(h, 5, 3, 3, 0.8, want_coef=False) for h in halves]
    torch.cuda.synchronize()
    def run():
        for st, h, o in zip(streams, halves, outs):
            with torch.cuda.stream(st):
                bank.mmpdf_launch(h, 5, 3, 3, 0.8, want_coef=False, out=o)
    t = timed(run)
    res["%d_streams_ms" % parts] = t * 1e3
    res["%d_streams_signals_per_s" % parts] = S / t
one = timed(lambda: bank.mmpdf_launch(d, 5, 3, 3, 0.8, want_coef=False, out=out), reps=50)
res["one_stream_again_ms"] = one * 1e3
res["one_stream_again_signals_per_s"] = S / one
print(json.dumps(res))
