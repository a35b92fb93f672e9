"""``MMPTree`` / ``MMPNode``: the reference's tree API (reference src/csc/mmp.py:25-733) as a view on
the node tables that ONE device call fills.

The reference grows the tree node by node, depth first (``MMPDFBranchFromPath`` :388-403), each node
paying an lsqr solve and two FFT correlations.  Here ``runMMPDF`` issues one ``csc_mmpdf_batch`` call:
the level-synchronous frontier kernel expands every node of a level in one launch, and the tree
object below only indexes the result (closed form of :303-313: node j of level d has parent
j mod c^(d-1), sibling rank j div c^(d-1)).  Plotting / offline analytics of the reference's class
are out of scope.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np


class MMPNode:
    """Read-only view of one tree node (reference src/csc/mmp.py:25-277)."""

    def __init__(self, tree, depth: int, index: int):
        self._tree, self.depth, self.index = tree, depth, index
        self.dictionary = tree.dictionary
        self.dissimilarity_threshold = tree.dissimilarityMMPNode
        if depth == 0:
            self.atom_pos = self.atom_idx = self.atom = self.atom_info = None
        else:
            res = tree._result()
            nid = res.node_id(depth, index)
            self.atom_pos = int(res.node_pos[0, nid])
            self.atom_idx = int(res.node_atom[0, nid])
            self.atom = self.dictionary.getAtom(self.atom_idx)
            self.atom_info = {"x": self.atom_pos, "b": self.atom.b, "y": self.atom.y, "s": self.atom.sigma}

    # ---- structure ----------------------------------------------------------------------------------
    @property
    def parent(self):
        if self.depth == 0:
            return None
        return self._tree._node(self.depth - 1, self.index % self._tree.connections ** (self.depth - 1))

    @property
    def children(self) -> List["MMPNode"]:
        t = self._tree
        if self.depth >= t.sparsity or t._res is None:
            return []
        stride, width = t.connections ** self.depth, t._res.level_sizes[self.depth]
        return [t._node(self.depth + 1, self.index + r * stride) for r in range(t.connections)
                if self.index + r * stride < width]

    def isRoot(self) -> bool:
        return self.depth == 0

    def isLeaf(self) -> bool:
        return len(self.children) == 0

    def getChildren(self) -> List["MMPNode"]:
        return self.children

    def getAtom(self):
        return self.atom

    def getChildrenIndex(self) -> int:
        """1-based rank among the parent's children (0 for the root)."""
        if self.depth == 0:
            return 0
        return self.index // self._tree.connections ** (self.depth - 1) + 1

    def getGenealogyIndex(self) -> List[int]:
        c, j = self._tree.connections, self.index
        out = []
        for _ in range(self.depth):
            out.append(j % c + 1)
            j //= c
        return out

    def getGenealogy(self) -> List["MMPNode"]:
        out, n = [], self
        while not n.isRoot():
            out.append(n)
            n = n.parent
        return out

    def getGenealogyStr(self) -> str:
        return "ROOT" + "".join(f" -> {i}" for i in self.getGenealogyIndex())

    def getFullBranchPath(self) -> str:
        return "-".join(str(i) for i in self.getGenealogyIndex())

    # ---- numerics -----------------------------------------------------------------------------------
    def getMSE(self) -> float:
        if self.depth == 0:
            return float(np.mean(np.asarray(self._tree.signal, dtype=np.float64) ** 2))
        res = self._tree._result()
        return float(res.node_mse[0, res.node_id(self.depth, self.index)])

    def getDepth(self) -> int:
        """Number of distinct activations of the branch (reference :111-112)."""
        return len(set(self._support()))

    def _support(self):
        return self._tree._result().branch(0, self.depth, self.index) if self.depth else []

    def getFullBranchAtoms(self) -> List[dict]:
        return [self._tree.dictionary._info(p, k) for p, k in self._support()]

    def getAtomSignal(self):
        if self.depth == 0:
            return None
        return self.atom.getAtomInSignal(signal_length=len(self._tree.signal), offset=self.atom_pos)

    @property
    def atom_signal(self):
        return self.getAtomSignal()

    def buildSignalRecovery(self) -> np.ndarray:
        """Sum of the UNIT atoms of the branch -- LS amplitudes are not applied (reference :248-255)."""
        out = np.zeros(len(self._tree.signal))
        L = self._tree.dictionary.getAtomsLength()
        for p, k in self._support():
            out[p:p + L] += self._tree.dictionary.getAtom(k)()
        return out

    def getCoefficients(self) -> np.ndarray:
        """LS amplitudes of the branch against the original signal, root -> this node (the lsqr
        solution the reference computes and discards, :66)."""
        res = self._tree._result()
        return res.node_coef[0, res.node_id(self.depth, self.index), :self.depth].copy()

    def getResidual(self) -> np.ndarray:
        sig = np.asarray(self._tree.signal, dtype=np.float64).copy()
        L = self._tree.dictionary.getAtomsLength()
        for (p, k), c in zip(self._support(), self.getCoefficients() if self.depth else []):
            sig[p:p + L] -= c * self._tree.dictionary.getAtom(k)()
        return sig

    @property
    def residual(self):
        return self.getResidual()

    def getDelay(self) -> float:
        return self._tree._delay

    def getMMPDFComputeTime(self) -> float:
        return self._tree._delay


class MMPTree:
    def __init__(self, dictionary, signal: np.ndarray, sparsity: int, connections: int, dissimilarity: float = 0.8):
        self.dictionary = dictionary
        self.signal = signal
        self.sparsity = sparsity
        self.connections = connections
        self.dissimilarityMMPNode = dissimilarity
        self._res, self._delay, self._cache = None, 0.0, {}
        self.root = MMPNode(self, 0, 0)
        self.init_structure()

    def init_structure(self):
        self.leaves_nodes = []
        self.leaves_paths = []

    # ---- candidate numbering (reference :303-324) ---------------------------------------------------
    @staticmethod
    def getCandidatePath(sparsity, connections, candidate_number: int) -> Tuple[int]:
        """Base-``connections`` digits of candidate_number - 1, least significant (= root level) first."""
        rest, digits = candidate_number - 1, []
        for _ in range(sparsity):
            rest, d = divmod(rest, connections)
            digits.append(d + 1)
        return tuple(digits)

    @staticmethod
    def getCandidateNumber(sparsity, connections, path: List[int]) -> int:
        assert len(path) == sparsity, "The path must have the same length as the sparsity"
        return 1 + sum((order - 1) * connections ** level for level, order in enumerate(path))

    @staticmethod
    def getComputedNodesFromPath(path: List[int]) -> int:
        """Nodes a depth-first walk must newly create for ``path`` (reference :326-335)."""
        n = 1
        for order in list(path)[:0:-1]:
            if order != 1:
                break
            n += 1
        return n

    @staticmethod
    def getComputedNodesFromTree(sparsity: int, connections: int):
        return [MMPTree.getComputedNodesFromPath(MMPTree.getCandidatePath(sparsity, connections, n))
                for n in range(1, connections ** sparsity + 1)]

    # ---- the search ---------------------------------------------------------------------------------
    def _result(self):
        if self._res is None:
            raise RuntimeError("runMMPDF has not been called on this tree")
        return self._res

    def _node(self, depth, index) -> MMPNode:
        key = (depth, index)
        if key not in self._cache:
            self._cache[key] = MMPNode(self, depth, index) if depth else self.root
        return self._cache[key]

    def runMMPDF(self, branches_number: int, verbose=False):
        """Candidates 1..min(branches_number, connections**sparsity) (reference :405-424), all levels
        expanded on the GPU in one call."""
        assert branches_number > 0, "branches_number must be greater than 0"
        self.init_structure()
        self._cache = {}
        self._res, self._delay = self.dictionary._batch(self.signal, self.sparsity, self.connections,
                                                        branches_number, self.dissimilarityMMPNode,
                                                        want_coef=True)
        nb = min(branches_number, self.connections ** self.sparsity)
        for n in range(1, nb + 1):
            path = MMPTree.getCandidatePath(self.sparsity, self.connections, n)
            self.leaves_paths.append(path)
            self.leaves_nodes.append(self._node(self.sparsity, n - 1))
            if verbose:
                print("\nBRANCH n°{} exploring path : {} | computed nodes = {}".format(
                    n, path, MMPTree.getComputedNodesFromPath(path)))

    def MMPDFBranchFromPath(self, path: List[int], verbose=False) -> MMPNode:
        """Node at the end of ``path`` (reference :388-403); the tree must have been run."""
        j = sum((order - 1) * self.connections ** level for level, order in enumerate(path))
        if j >= self._result().level_sizes[len(path) - 1]:
            raise IndexError(f"path {tuple(path)} is outside the {self._res.branches} explored branches")
        return self._node(len(path), j)

    def getResult(self) -> Tuple[np.ndarray, List[dict]]:
        """(sum of unit atoms, infos) of the FIRST minimum-MSE leaf (np.argmin, reference :458-467)."""
        leaf = self.leaves_nodes[int(self._result().rec_leaf[0, self.sparsity - 1, 1])]
        return leaf.buildSignalRecovery(), leaf.getFullBranchAtoms()

    def buildMMPTreeDict(self) -> Dict:
        """{path: {'mse','atoms','delay'}} over the leaves (reference :564-575)."""
        return {"-".join(str(p) for p in path): {"mse": leaf.getMSE(), "atoms": leaf.getFullBranchAtoms(),
                                                 "delay": leaf.getMMPDFComputeTime()}
                for path, leaf in zip(self.leaves_paths, self.leaves_nodes)}

    def buildMMPDFResultDict(self, compute_time_type: str = "local") -> dict:
        """{'mse','path','atoms','delay'} of the LAST minimum-MSE leaf ('<=', reference :577-606)."""
        res = self._result()
        j = int(res.rec_leaf[0, self.sparsity - 1, 0])
        leaf = self.leaves_nodes[j]
        return {"mse": leaf.getMSE(), "path": "-".join(str(p) for p in self.leaves_paths[j]),
                "atoms": leaf.getFullBranchAtoms(), "delay": self._delay}

    def printLeaves(self):
        for i, leaf in enumerate(self.leaves_nodes):
            print(f"Leaf {i + 1} : {leaf.getGenealogyStr()} | MSE = {leaf.getMSE()}")

    # ---- helpers on stored tree dicts (reference :608-733) ------------------------------------------------
    @staticmethod
    def getTreeParamsFromMMPTreeDict(mmp_tree_dict: dict) -> Tuple[int, int]:
        paths = [[int(x) for x in p.split("-")] for p in mmp_tree_dict]
        return len(paths[0]), max(max(p) for p in paths)

    @staticmethod
    def shrinkMMPTreeDict(mmp_tree_dict: dict, max_branches: int):
        k, c = MMPTree.getTreeParamsFromMMPTreeDict(mmp_tree_dict)
        for path in [p for p in mmp_tree_dict
                     if MMPTree.getCandidateNumber(k, c, [int(x) for x in p.split("-")]) > max_branches]:
            del mmp_tree_dict[path]
        return mmp_tree_dict

    @staticmethod
    def getSubTreeFromMMPTreeDict(mmp_tree_dict: dict, dictionary, signal: np.ndarray, sparsity: int,
                                  verbose: bool = False) -> dict:
        """Depth-``sparsity`` truncation of a stored tree, each distinct sub-path re-projected
        (reference :641-669); all projections of the tree go to the GPU as one batch."""
        subs = {}
        for path, leaf in mmp_tree_dict.items():
            sub = "-".join(path.split("-")[:sparsity])
            if sub not in subs:
                subs[sub] = leaf["atoms"][:sparsity]
        keys = list(subs)
        mses = dictionary.projectionMSEBatch(signal, [subs[k] for k in keys])
        return {k: {"atoms": subs[k], "mse": m} for k, m in zip(keys, mses)}

    @staticmethod
    def mmpdfCandidateFromMMPTreeDict(mmp_tree_dict: dict, dictionary, signal: np.ndarray,
                                      candidate_sparsity: int, verbose: bool = False):
        """(atoms, mse) of the best depth-``candidate_sparsity`` sub-branch, FIRST strict minimum
        (reference :671-703)."""
        k, _ = MMPTree.getTreeParamsFromMMPTreeDict(mmp_tree_dict)
        sub = MMPTree.getSubTreeFromMMPTreeDict(mmp_tree_dict, dictionary, signal, candidate_sparsity)
        best, best_mse = "-".join(["1"] * min(candidate_sparsity, k)), np.inf
        for path, leaf in sub.items():
            if leaf["mse"] < best_mse:
                best, best_mse = path, leaf["mse"]
        return sub[best]["atoms"], best_mse

    @staticmethod
    def ompCandidateFromMMPTreeDict(mmp_tree_dict: dict, dictionary, signal: np.ndarray,
                                    candidate_sparsity: int, verbose: bool = False):
        """(atoms, mse) of the OMP branch '1-1-...-1' cut at ``candidate_sparsity`` (reference :705-733)."""
        k, _ = MMPTree.getTreeParamsFromMMPTreeDict(mmp_tree_dict)
        atoms = mmp_tree_dict["-".join(["1"] * k)]["atoms"][:min(candidate_sparsity, k)]
        (mse,) = dictionary.projectionMSEBatch(signal, [atoms])
        return atoms, mse
