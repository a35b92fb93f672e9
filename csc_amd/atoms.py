"""Zatsepin-Shcherbinin atom model (host side, evaluated once per dictionary).

Mirror of the reference class ``CSCAtom`` (``ZSAtom`` in its README / wheel), reference
src/csc/atoms.py:6-172: same constructor, attributes and methods, so notebooks and the workbench
that build atoms from ``{'b','y','s'}`` dicts keep working.  The values feed the HBM-resident atom
bank of csc_amd.engine.DeviceBank; nothing here is on the per-signal path.
"""
from __future__ import annotations

from typing import Tuple

import numpy as np


def der_zs_formula(b, sigma, x, y):
    """x-derivatives (dHr/dx, dHa/dx) of the Zatsepin-Shcherbinin leakage field of a defect of
    half-width ``b`` seen at lift-off ``y`` (reference src/csc/atoms.py:6-33)."""
    left, right = (x + b) ** 2 + y ** 2, (x - b) ** 2 + y ** 2
    den = left * right
    dden = 2 * (x + b) * right + left * 2 * (x - b)
    scale = sigma * b

    n_r, dn_r = -8 * scale * x * y, -8 * scale * y
    der_hr = (dn_r * den - n_r * dden) / den ** 2

    n_a, dn_a = -4 * scale * (x ** 2 - y ** 2 - b ** 2), -4 * scale * 2 * x
    der_ha = (dn_a * den - n_a * dden) / den ** 2
    return der_hr, der_ha


class CSCAtom:
    STEP = 0.001
    SUPPORT_THRESHOLD = 0.001

    def __init__(self, b, y, sigma):
        self.step = CSCAtom.STEP
        self.support_threshold = CSCAtom.SUPPORT_THRESHOLD
        lo, hi = -b - 1, b + 1
        grid = np.linspace(lo, hi, int((hi - lo) / self.step))
        profile, _ = der_zs_formula(b, sigma, grid, y)
        norm = np.linalg.norm(profile)
        # unit L2 norm over the FULL grid; sigma is rescaled by the same factor (atoms.py:51-52)
        values = profile / norm
        self.b, self.y, self.sigma = b, y, sigma / norm
        self.params = {"b": b, "y": y, "sigma": self.sigma}
        # keep [first rising edge, last falling edge) of the indicator (values > threshold)
        edges = np.diff((values > CSCAtom.SUPPORT_THRESHOLD).astype(int))
        first = np.argwhere(edges == 1).flatten()[0]
        last = np.argwhere(edges == -1).flatten()[-1]
        self.x = grid[first:last]
        self.values = values[first:last]
        self.support_width = len(self.values)
        self.var = np.var(self.values)
        self.position = self.x - self.x[0]

    @classmethod
    def from_dict(cls, params) -> "CSCAtom":
        return cls(params["b"], params["y"], params["s"])

    def __str__(self) -> str:
        return "b={:.3f}, y={:.3f}, sigma={:.3e}".format(self.params["b"], self.params["y"], self.params["sigma"])

    def __eq__(self, other: object) -> bool:
        """Shape parameters only (b, y) -- reference src/csc/atoms.py:79-85."""
        if not isinstance(other, CSCAtom):
            return False
        return self.params["b"] == other.params["b"] and self.params["y"] == other.params["y"]

    __hash__ = None

    def __getitem__(self, key):
        return self.params.get(key, None)

    def __setitem__(self, key, value):
        self.params[key] = value

    def __len__(self) -> int:
        return len(self.values)

    def __call__(self, *args, **kwargs):
        return self.getSignal(*args, **kwargs)

    def __truediv__(self, other):
        if not isinstance(other, (int, float)):
            raise TypeError("Unsupported operand type for /: 'CSCDictionary' and '{}'".format(type(other).__name__))
        self.sigma /= other
        self.params["sigma"] = self.sigma
        self.values /= other
        return self

    def copy(self) -> "CSCAtom":
        return CSCAtom(self.params["b"], self.params["y"], self.params["sigma"])

    def set_param(self, key, value):
        self.params[key] = value

    def get_param(self, key):
        return self.params[key]

    def getInfos(self) -> dict:
        return self.params

    def getPlots(self) -> Tuple[np.ndarray, np.ndarray]:
        return self.position, self.values

    def padBothSides(self, signal_length):
        """Zero-pad to ``signal_length`` samples, floor(half) on the left (atoms.py:130-143)."""
        assert signal_length >= len(self), "signal_length must be greater than or equal to self.support_width"
        extra = signal_length - len(self)
        n_left = extra // 2
        n_right = extra - n_left
        self.values = np.pad(self.values, (n_left, n_right), mode="constant", constant_values=0)
        self.x_pad_left = self.x[0] - self.step * np.arange(n_left, 0, -1)
        self.x_pad_right = self.x[-1] + self.step * np.arange(1, n_right + 1)
        self.x = np.concatenate([self.x_pad_left, self.x, self.x_pad_right])
        self.position = self.x - self.x[0]

    @staticmethod
    def noiseVarForSNR(var, snr) -> float:
        return var / (10 ** (snr / 10))

    def getNoiseVarToSNR(self, snr) -> float:
        return CSCAtom.noiseVarForSNR(self.var, snr)

    def getSignal(self, *args, **kwargs) -> np.ndarray:
        return self.values

    def getAtomInSignal(self, signal_length, offset) -> np.ndarray:
        """The atom placed at ``offset`` in a zero signal (atoms.py:162-172)."""
        out = np.zeros(signal_length)
        out[offset:offset + len(self)] = self.values
        return out


ZSAtom = CSCAtom   # README / wheel spelling of the same class
