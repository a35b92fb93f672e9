"""Development aid: how wide is the band of shifts at which an atom of BASELINE config 3's dictionary is similar to any
other (the set bits of its similarity rows), per word of 32 shifts."""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
import csc_amd
large = csc_amd.ZSDictionary.from_values(np.arange(0.005, 0.06, 0.0005), np.arange(0.010, 0.036, 0.0003), [1])
bank = large.device_bank()
N, thr = 777, 0.8
W = bank.lib.csc_sim_words(N, bank.L)
tab = bank.similarity_table(N, thr).view(bank.K, bank.K, W)
torch.cuda.synchronize()
nz = torch.stack([(tab[:, :, w] != 0).any(dim=1) for w in range(W)], dim=1).cpu().numpy()   # [K][W]
lo = nz.argmax(axis=1); hi = W - 1 - nz[:, ::-1].argmax(axis=1)
print("words per row", W, "P", N - bank.L + 1)
print("words in use per pick atom: min %d median %d max %d" % ((hi - lo + 1).min(), np.median(hi - lo + 1), (hi - lo + 1).max()))
print("lowest word used %d .. %d, highest %d .. %d" % (lo.min(), lo.max(), hi.min(), hi.max()))
frac = torch.stack([(tab[k] != 0).any(dim=1).float().mean() for k in range(0, bank.K, 97)]).mean().item()
print("fraction of atoms similar to a pick at some shift: %.3f" % frac)
# bit-level range for a sample of picks
for k in range(0, bank.K, 1201):
    row = tab[k].cpu().numpy().view(np.uint32)              # [K][W]
    orr = np.bitwise_or.reduce(row, axis=0)
    bits = np.unpackbits(orr.view(np.uint8), bitorder="little")
    idx = np.nonzero(bits)[0]
    print("pick", k, "set shifts dp - (P-1) in [%d, %d]" % (idx.min() - (N - bank.L), idx.max() - (N - bank.L)), "count", len(idx))
