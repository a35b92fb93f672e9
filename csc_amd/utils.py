"""JSON helpers and the atom-matching metric used next to the sparVar records
(reference src/csc/utils.py:7-114).  Host side, off the device path."""
from __future__ import annotations

import datetime
from typing import Dict, List, Tuple

import numpy as np

from .atoms import CSCAtom


def handle_non_serializable(obj):
    """``json.dump(default=...)`` hook: NumPy scalars / arrays -> Python (reference utils.py:7-15)."""
    if isinstance(obj, np.integer):
        return int(obj)
    if isinstance(obj, np.floating):
        return float(obj)
    if isinstance(obj, np.ndarray):
        return obj.tolist()
    raise TypeError("Unserializable object {} of type {}".format(obj, type(obj)))


def get_today_date_str():
    """'%y%m%d' stamp of the DB headers (reference utils.py:17-21)."""
    return datetime.datetime.now().strftime("%y%m%d")


def _placed(dictionary, signal_length, atoms):
    out = []
    for info in atoms:
        a = CSCAtom.from_dict(info)
        a.padBothSides(dictionary.getAtomsLength())
        out.append(a.getAtomInSignal(signal_length, info["x"]))
    return np.array(out).reshape(len(atoms), signal_length)


def _match(dictionary, signal_length, true_atoms, approx_atoms, pos_err_threshold, corr_err_threshold):
    """Hungarian assignment on the 0/1 cost 'position within pos_err AND |<a_true, a_approx>| >=
    corr_err' (reference utils.py:34-64)."""
    from scipy import optimize
    tx = np.array([a["x"] for a in true_atoms])
    ax = np.array([a["x"] for a in approx_atoms])
    dpos = np.abs(tx[:, None] - ax)
    corr = np.abs(_placed(dictionary, signal_length, true_atoms)
                  @ _placed(dictionary, signal_length, approx_atoms).T)
    cost = np.ones_like(dpos)
    cost[(dpos <= pos_err_threshold) & (corr >= corr_err_threshold)] = 0
    rows, cols = optimize.linear_sum_assignment(cost)
    return cost, rows, cols


def computeMaxTruePositives(dictionary, signal_length, true_atoms: List[Dict], approx_atoms: List[Dict],
                            pos_err_threshold: float, corr_err_threshold: float, verbose: bool = False):
    """(max number of true positives, [(true, approx)] matched pairs) -- reference utils.py:23-69."""
    cost, rows, cols = _match(dictionary, signal_length, true_atoms, approx_atoms, pos_err_threshold,
                              corr_err_threshold)
    matched = [(true_atoms[i], approx_atoms[j]) for i, j in zip(rows, cols)]
    return np.sum(cost[rows, cols] == 0), matched


def getMatchedAtoms(dictionary, signal_length, true_atoms: List[Dict], approx_atoms: List[Dict],
                    pos_err_threshold: float, corr_err_threshold: float,
                    verbose: bool = False) -> List[Tuple[Dict, Dict]]:
    """Matched pairs only (reference utils.py:71-114)."""
    _, rows, cols = _match(dictionary, signal_length, true_atoms, approx_atoms, pos_err_threshold,
                           corr_err_threshold)
    return [(true_atoms[i], approx_atoms[j]) for i, j in zip(rows, cols)]
