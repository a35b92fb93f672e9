"""``conv_multipath_matching_pursuit``: the notebook entry point (reference src/csc/main.py:5-21)."""
import numpy as np

from .atoms import CSCAtom
from .dictionary import CSCDictionary


def conv_multipath_matching_pursuit(signal: np.ndarray, dictionary: CSCDictionary, nb_atoms: int,
                                    nb_mmp_branches: int) -> list:
    """MMP-DF with dissimilarity 0.4 and connections = branches = ``nb_mmp_branches``; returns
    (sum of the unit atoms of the best branch, atom infos)."""
    return dictionary.mmpdf(signal=np.array(signal, dtype=np.float64), sparsity_level=nb_atoms,
                            connections_level=nb_mmp_branches, dissimilarity=0.4,
                            nb_branches=nb_mmp_branches, verbose=False)
