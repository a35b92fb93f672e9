"""Development aid: cross-check of the tiled fp16 kernel's barrier-free screen (screen_blocks16) against the per-unit
single-term product (gemm16x1_group), unit by unit and bit for bit.  Build the checking library first
(`make -C csc_amd/csrc check`), then on the GPU box:

    CSC_HIP_LIB=$PWD/csc_amd/csrc/libcsc_hip_check.so python scripts/check_screen.py [signals]

The kernel prints one line per unit whose two maxima differ; no such line = the screen is right."""
import sys
import numpy as np
sys.path.insert(0, ".")
import csc_amd
import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
z = np.load("tests/golden/large.npz")
large = csc_amd.ZSDictionary.from_values(np.arange(0.005, 0.06, 0.0005), np.arange(0.010, 0.036, 0.0003), [1])
res, _ = large._batch(z["signals"][:n], 5, 3, 3, 0.8, want_coef=True)
torch.cuda.synchronize()
print("done; stats", res.stats)
