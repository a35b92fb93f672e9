"""Calls of csc_mmpdf_batch on different HIP streams, each with its own workspace and outputs, may overlap -- bench.py's
timed steps do (two streams, the next batch's root level under the chains of the batch before).  Nothing in the
library is shared between such calls: two different batches launched alternately on two streams, several times over
and without a synchronisation in between, give exactly what each gives alone."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_overlapping_calls_on_two_streams_equal_single_calls():
    import torch
    import csc_amd
    d = csc_amd.ZSDictionary.from_values(np.arange(0.005, 0.06, 0.005), np.arange(0.010, 0.036, 0.002), [1])
    bank = d.device_bank()
    names = ("node_pos", "node_atom", "node_mse", "rec_buf")
    sigs, alone = [], []
    for seed in (5, 6):
        sig, _ = bank.generate(np.full(1600, 4, np.int32), np.full(1600, 10.0), 738, 10, 0.75, seed=seed)
        sigs.append(sig)
        out = bank.mmpdf_launch(sig, 5, 3, 3, 0.8, want_coef=False)
        torch.cuda.synchronize()
        alone.append({n: out[n].clone() for n in names})
    assert not torch.equal(alone[0]["rec_buf"], alone[1]["rec_buf"])
    streams = [torch.cuda.Stream(bank.device) for _ in range(2)]
    outs = [None, None]
    torch.cuda.synchronize()
    for k in range(12):
        with torch.cuda.stream(streams[k & 1]):
            outs[k & 1] = bank.mmpdf_launch(sigs[k & 1], 5, 3, 3, 0.8, want_coef=False, out=outs[k & 1])
    torch.cuda.synchronize()
    for q in (0, 1):
        for n in names:
            assert torch.equal(outs[q][n], alone[q][n]), (q, n)
        assert outs[q]["stats"].cpu().numpy()[0] == 0
