"""Development aid: build time of the config-3 similarity table, float64 kernel against the packed-bank entry."""
import ctypes as C, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import csc_amd
from csc_amd import _lib
d = csc_amd.ZSDictionary.from_values(np.arange(0.005, 0.06, 0.0005), np.arange(0.010, 0.036, 0.0003), [1])
bank = d.device_bank()
lib = bank.lib
N = 777
W = lib.csc_sim_words(N, bank.L)
ptr = lambda t: C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream(bank.device).cuda_stream)
b = torch.empty(bank.K * bank.K * W, dtype=torch.int32, device=bank.device)
for rep in range(2):
    torch.cuda.synchronize(); t = time.perf_counter()
    _lib.check(lib.csc_similarity_table_packed(ptr(bank.bank32t), ptr(bank.bank64), bank.K, bank.L, N, 0.8, ptr(b), st), "packed")
    torch.cuda.synchronize(); print("packed: %.3f s" % (time.perf_counter() - t), flush=True)
if "--check" in sys.argv:
    a = torch.empty_like(b)
    torch.cuda.synchronize(); t = time.perf_counter()
    _lib.check(lib.csc_similarity_table(ptr(bank.bank64), bank.K, bank.L, N, 0.8, ptr(a), st), "f64")
    torch.cuda.synchronize(); print("float64: %.3f s" % (time.perf_counter() - t))
    print("equal:", bool(torch.equal(a, b)), "set bits:", int(sum(int(x.item()) for x in [torch.sum((a.view(torch.uint8)).to(torch.int64) != 0)])))
