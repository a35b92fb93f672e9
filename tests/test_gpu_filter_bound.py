"""GPU test of the bound the fp16 matrix-core filter rests on.

The level kernels take a node's correlations on v_mfma_f32_16x16x32_f16 from the residual scaled to unit norm
and re-score in float64 every entry within 2 eps of the computed maximum; the selection is the float64 arg-max
ONLY IF every entry of that product is within eps of the float64 value, with
    eps = (2^-10 + 2^-14 + 1e-6) am + 2^-25 sqrt(L) (1 + am),   am = max_k ||a_k||   (units of ||residual||).
csc_filter_product writes the product (taken by the kernels' own loops) so that the bound itself is checked
here against float64, not just its consequence: golden residuals at every depth, and operands built to be
bad for it (one-signed products, alternating signs, fp16 rounding midpoints, windows that are half subnormal
in fp16, banks scaled to am = 3e-3 / 1 / 30).  The worst error/eps ratios are written to
gpurun_out/filter_bound_margins.json when that directory exists.
"""
import json
import os

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

L, N, P = 492, 738, 247
MARGINS = {}


def _bank(bank64):
    from csc_amd.engine import DeviceBank
    return DeviceBank(np.ascontiguousarray(bank64, dtype=np.float64))


def _ratio(bank, bank64, sigs, tag):
    """max over entries of |fp16 product - float64| / eps, per variant; asserts <= 1."""
    sigs = np.ascontiguousarray(sigs, dtype=np.float64)
    worst = 0.0
    for variant in (0, 1):
        for c0 in range(0, len(sigs), 64):
            chunk = sigs[c0:c0 + 64]
            prod, eps, nrm = bank.filter_product(chunk, variant)
            assert np.isfinite(prod).all(), (tag, variant)
            assert np.allclose(nrm, np.linalg.norm(chunk, axis=1), rtol=1e-14)
            win = np.lib.stride_tricks.sliding_window_view(chunk, bank64.shape[1], axis=1)      # [B, P, L]
            ref = (win @ bank64.T) / nrm[:, None, None]
            r = (np.abs(prod.astype(np.float64) - ref).max(axis=(1, 2)) / eps).max()
            worst = max(worst, float(r))
    MARGINS[tag] = worst
    assert worst <= 1.0, (tag, worst)
    return worst


@pytest.fixture(scope="module")
def canon():
    import csc_amd
    d = csc_amd.ZSDictionary.from_values(np.arange(0.005, 0.06, 0.005), np.arange(0.010, 0.036, 0.002), [1])
    return d, d.getLocalDictionary()


def test_bound_on_golden_residuals_every_depth(canon, golden_signals):
    """the 64 golden signals and the residual of every node that is correlated (levels 1..4 of c3/B3/K5)"""
    d, bank64 = canon
    sigs = golden_signals["signals"]
    res, _ = d._batch(sigs, 5, 3, 3, 0.8, want_coef=True)
    rows = [s for s in sigs]
    for s in range(len(sigs)):
        for depth in range(1, 5):
            for j in range(3):
                r = sigs[s].copy()
                coef = res.node_coef[s, res.node_id(depth, j)]
                for (p, k), c in zip(res.branch(s, depth, j), coef):
                    r[p:p + L] -= c * bank64[k]
                rows.append(r)
    assert len(rows) == 64 * 13
    w = _ratio(d.device_bank(), bank64, np.array(rows), "golden residuals, every depth")
    assert w < 0.5          # real data sits far inside the worst-case bound


@pytest.mark.parametrize("am", [3e-3, 1.0, 30.0])
def test_bound_on_adversarial_operands(canon, am):
    _, canon64 = canon
    rng = np.random.default_rng(5)
    base = canon64 * am
    ulp = 2.0 ** -10                                   # fp16 spacing in [1, 2)

    def midpoints(x):
        """every non-zero value moved to the nearest fp16 rounding midpoint of its binade (odd multiple of half an
        fp16 ulp), where rounding costs the full 2^-11 relative"""
        out = np.zeros_like(x)
        nz = x != 0
        e = np.floor(np.log2(np.abs(x[nz])))
        m = np.abs(x[nz]) / 2.0 ** e                    # [1, 2)
        m = (np.floor(m / ulp) + 0.5) * ulp
        out[nz] = np.sign(x[nz]) * m * 2.0 ** e
        return out

    alt = (-1.0) ** np.arange(L)
    banks = {
        "canonical": base,
        "all positive": np.abs(base),
        "alternating": np.abs(base) * alt,
        "midpoints": midpoints(np.abs(base)),
    }
    noise = rng.standard_normal((6, N))
    half_tiny = np.abs(noise.copy())
    half_tiny[:, N // 2:] *= 2.0 ** -17                # after scaling to unit norm: subnormal in fp16
    sig_sets = {
        "noise": noise,
        "all positive": np.abs(noise) + 0.5,
        "alternating": (np.abs(noise) + 0.5) * ((-1.0) ** np.arange(N)),
        "constant": np.ones((2, N)),
        "half subnormal": half_tiny,
        "midpoints x 2^k norm": None,
    }
    # a signal of fp16 midpoints whose norm is a power of two keeps them midpoints after the kernel's scaling
    mp = midpoints(np.abs(noise[:2]) + 0.5)
    for row in mp:                                     # the last sample brings the norm to a power of two
        rest = float(np.sum(row[:-1] ** 2))
        row[-1] = np.sqrt(4.0 ** np.ceil(np.log2(rest) / 2 + 1e-9) - rest)
    sig_sets["midpoints x 2^k norm"] = mp
    worst = {}
    for bname, b64 in banks.items():
        bank = _bank(b64)
        for sname, sg in sig_sets.items():
            worst[(bname, sname)] = _ratio(bank, b64, sg, f"am={am}: bank {bname} / signal {sname}")
    # an atom shifted through the window against itself: the largest entries of the matrix
    b64 = banks["all positive"]
    sg = np.zeros((4, N))
    for i, k in enumerate((0, 50, 100, 142)):
        sg[i, 17 * i:17 * i + L] = b64[k]
    _ratio(_bank(b64), b64, sg, f"am={am}: placed atoms")


def test_accumulation_term_alone(canon):
    """Operands that ARE fp16 numbers (bank rounded to fp16 beforehand; a signal of 256 entries +-1, norm 16, so
    the scaled signal is +-2^-4 exactly): every product is exact in fp32 and the whole error is the matrix cores'
    fp32 accumulation over the window, which the bound budgets at 2^-14 am.  One-signed operands make the sum grow
    monotonically, the case in which an adder that truncated instead of rounding would use the budget up."""
    _, canon64 = canon
    rng = np.random.default_rng(9)
    for am in (1.0, 30.0):
        b16 = (np.abs(canon64) * am).astype(np.float16).astype(np.float64)
        sg = np.zeros((8, N))
        for i in range(8):
            on = rng.choice(N, 256, replace=False)
            sg[i, on] = 1.0 if i < 4 else rng.choice([-1.0, 1.0], 256)
        sg[3, :] = 0.0
        sg[3, 100:356] = 1.0                           # a dense one-signed block: the longest monotone sums
        bank = _bank(b16)
        amx = np.sqrt((b16 * b16).sum(axis=1).max())
        worst = 0.0
        for variant in (0, 1):
            prod, eps, nrm = bank.filter_product(sg, variant)
            assert np.array_equal(nrm, np.full(8, 16.0))
            win = np.lib.stride_tricks.sliding_window_view(sg, L, axis=1)
            ref = (win @ b16.T) / 16.0
            worst = max(worst, float(np.abs(prod.astype(np.float64) - ref).max() / (2.0 ** -14 * amx)))
        MARGINS[f"am={am}: accumulation alone, error / (2^-14 am)"] = worst
        assert worst <= 0.5, worst                    # at least a factor of two inside its budget


def test_bound_margin_report():
    """writes the collected worst ratios (error / eps) for DESIGN.md; the accumulation term of eps keeps a factor
    of two as long as every ratio is <= 0.5 + operand term share"""
    assert MARGINS, "run the whole module"
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "filter_bound_margins.json"), "w") as f:
            json.dump({"worst_error_over_eps": MARGINS, "max": max(MARGINS.values())}, f, indent=1)
    assert max(MARGINS.values()) <= 1.0
