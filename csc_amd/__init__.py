"""csc_amd -- MI355X-native batched convolutional OMP / MMP-DF behind the API of yanisgomes/csc.

Same public names as the reference package (``src/csc/__init__.py``), both spellings of the atom /
dictionary classes (``CSC*`` in its sources, ``ZS*`` in its README and wheel).  Compute goes through
``csrc/libcsc_hip.so`` (C ABI in ``include/csc_hip.h``); there is no CPU solver here.
"""
from .atoms import CSCAtom, ZSAtom, der_zs_formula
from .dictionary import CSCDictionary, ZSDictionary, SparVarBatch
from .mmp import MMPNode, MMPTree
from .main import conv_multipath_matching_pursuit
from .workbench import CSCWorkbench
from .utils import computeMaxTruePositives, getMatchedAtoms, handle_non_serializable, get_today_date_str

__all__ = ["CSCAtom", "ZSAtom", "CSCDictionary", "ZSDictionary", "SparVarBatch", "MMPNode", "MMPTree", "CSCWorkbench",
           "conv_multipath_matching_pursuit", "computeMaxTruePositives", "getMatchedAtoms",
           "handle_non_serializable", "get_today_date_str", "der_zs_formula"]
__version__ = "0.1.0"
