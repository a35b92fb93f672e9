"""``CSCWorkbench`` shell: signal-DB loading and the hooks the solvers need.

The reference's workbench survives only as bytecode; its ~100 analysis / plotting
methods read result JSONs and never call a solver, so they are outside the accelerated path.  This
shell keeps the four entry points that sit on it (constructor, ``load_data``, ``set_dictionary``,
``signalDictFromId``; wheel ``csc/workbench.py:17-64``) plus batch runners that hand the loaded DB to
the GPU pipelines.
"""
from __future__ import annotations

import json
from typing import Optional

from .dictionary import CSCDictionary


class CSCWorkbench:
    def __init__(self, signals_path: str):
        self.signals_path = signals_path
        self.dictionary: Optional[CSCDictionary] = None
        self.signals = None
        self.loaded_signals = False
        self.signals_data = None

    def load_data(self):
        """Load the signals DB once (same layout as generateConstrainedSignalsDB writes)."""
        if self.loaded_signals:
            return
        with open(self.signals_path, "r") as f:
            self.signals_data = json.load(f)
        self.signals = self.signals_data["signals"]
        self._by_id = {sd["id"]: sd for sd in self.signals}
        self.loaded_signals = True

    def set_dictionary(self, dictionary: CSCDictionary):
        self.dictionary = dictionary

    def signalDictFromId(self, id: int) -> dict:
        """The signal dict with this id, or None."""
        self.load_data()
        return self._by_id.get(id)

    # ---- batch runners on the loaded DB -------------------------------------------------------------
    def _need_dictionary(self):
        if self.dictionary is None:
            raise RuntimeError("set_dictionary() must be called first")
        return self.dictionary

    def runMMPDFSparVar(self, output_filename: str, connections: int = 3, dissimilarity: float = 0.8,
                        branches: int = 10, max_sparsity: int = 10):
        self._need_dictionary().mmpdfSparVarPipelineFromDB(self.signals_path, output_filename, 1, connections,
                                                           dissimilarity, branches, max_sparsity)

    def runOMPSparVar(self, output_filename: str, max_sparsity: int):
        self._need_dictionary().ompSparVarPipelineFromDB(self.signals_path, output_filename, 1, max_sparsity)
