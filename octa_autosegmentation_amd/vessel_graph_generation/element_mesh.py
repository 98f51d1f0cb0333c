"""The reference's element containers (vessel_graph_generation/element_mesh.py:11-232) for code that pokes a mesh directly.

The simulator itself never goes through per-element calls: its neighbour queries are whole-phase HIP kernels (csrc/sim_core.h).
These classes keep the reference's interface -- `SpacePartitioner` with `extend / add / find_elements_in_distance /
get_all_elements / find_nearest_element / delete / delete_all / reassign`, `KD_Tree` with the parallel `elements` / `points` lists,
`CoordKdTree` for position tuples, `NodeKdTree` for objects with `.position` -- with the reference's results:

* `find_nearest_element`: the exact nearest element (lowest index on an exact tie) if its distance is <= max_dist, else None;
* `find_elements_in_distance`: the closed ball, in the order scipy's cKDTree reports it -- the order of `tree.indices`, which the
  simulator's O2 -> CO2 step depends on. That order comes from the SAME device code the simulator uses
  (`octa_sim_kat_kd_order`: the level-synchronous introselect restatement, checked against scipy in tests/test_sim_gpu.py); it is
  rebuilt lazily after a mutation, as the reference rebuilds its cKDTree on every mutation (:97-119, :180-211);
* batch forms `find_nearest_elements(positions, max_dist)` / `find_elements_in_distances(positions, distance)` for callers that can
  hand over many queries at once (one distance matrix per call instead of one tree walk per point).

Distances are evaluated in double precision on the host (numpy), so results do not depend on a device; only the ball ORDER of sets
larger than one leaf (16 points) needs the GPU, and asking for it without one raises like every other device call of this package.
"""
from abc import ABC, abstractmethod
from typing import Generic, List, Optional, TypeVar

import numpy as np

from .. import _native

T = TypeVar("T")
_LEAF = 16          # cKDTree's default leafsize: smaller sets are one leaf, reported in insertion order
_KD_CAP = 13312     # points the device order takes per call (the simulator's sink capacity)


class SpacePartitioner(ABC, Generic[T]):
    @abstractmethod
    def extend(self, positions: List[T]): ...

    @abstractmethod
    def add(self, element: T): ...

    @abstractmethod
    def find_elements_in_distance(self, pos, distance: float) -> List[T]: ...

    @abstractmethod
    def get_all_elements(self) -> List[T]: ...

    @abstractmethod
    def find_nearest_element(self, pos, max_dist=np.inf) -> Optional[T]: ...

    @abstractmethod
    def delete(self, element: T): ...

    @abstractmethod
    def delete_all(self, elemens: List[T]): ...

    @abstractmethod
    def _get_position(self, el: T): ...

    @abstractmethod
    def reassign(self, x: float): ...


class KD_Tree(SpacePartitioner, Generic[T]):
    def __init__(self) -> None:
        self.elements: List[T] = []
        self.points: list = []
        self._pts = None         # float64 [n, 3] of self.points, rebuilt after a mutation
        self._rank = None        # position of every point in cKDTree's tree.indices

    # ---- mutation (every one invalidates the index, as in the reference)
    def _dirty(self):
        self._pts = self._rank = None

    def update_kdTree(self):     # the reference's name: rebuilds eagerly
        self._dirty()
        self._index()

    def extend(self, elements: List[T]):
        if not elements:
            return
        self.elements.extend(elements)
        self.points.extend([self._get_position(e) for e in elements])
        self._dirty()

    def add(self, element: T):
        self.elements.append(element)
        self.points.append(self._get_position(element))
        self._dirty()

    def delete(self, element: T):
        try:
            idx = self.elements.index(element)      # == on tuples, identity on Node objects (element_mesh.py:186)
        except ValueError:
            return
        del self.elements[idx]
        del self.points[idx]
        self._dirty()

    def delete_all(self, elements: List[T]):
        if not elements:
            return
        to_remove = []
        for e in set(elements):
            try:
                to_remove.append(self.elements.index(e))
            except ValueError:
                pass
        for idx in sorted(set(to_remove), reverse=True):
            del self.elements[idx]
            del self.points[idx]
        self._dirty()

    def reassign(self, x: float):
        pass

    def get_all_elements(self) -> List[T]:
        return list(self.elements)

    # ---- index
    def _index(self):
        if self._pts is None:
            self._pts = np.asarray(self.points, dtype=np.float64).reshape(-1, 3)
        return self._pts

    def _ranks(self):
        """rank[i] = position of point i in scipy's tree.indices (leafsize 16, compact, median splits)."""
        if self._rank is None:
            pts = self._index()
            n = len(pts)
            if n <= _LEAF:
                self._rank = np.arange(n)
            else:
                if n > _KD_CAP:
                    raise ValueError(f"KD_Tree: the device order takes at most {_KD_CAP} points, got {n}")
                idx = np.zeros(n, np.int32)
                _native.check(_native.lib().octa_sim_kat_kd_order(_native.ctx(), pts.ctypes.data, n, None, idx.ctypes.data), "octa_sim_kat_kd_order")
                rank = np.empty(n, np.int64)
                rank[idx] = np.arange(n)
                self._rank = rank
        return self._rank

    # ---- queries
    def find_elements_in_distances(self, positions, distance: float) -> List[List[T]]:
        pts = self._index()
        q = np.asarray(positions, dtype=np.float64).reshape(-1, 3)
        if len(pts) == 0:
            return [[] for _ in q]
        d = np.linalg.norm(pts[None, :, :] - q[:, None, :], axis=2)
        out = []
        rank = None
        for row in d:
            hit = np.flatnonzero(row <= float(distance))
            if len(hit) > 1:
                rank = self._ranks() if rank is None else rank
                hit = hit[np.argsort(rank[hit], kind="stable")]
            out.append([self.elements[int(i)] for i in hit])
        return out

    def find_elements_in_distance(self, pos, distance: float) -> List[T]:
        return self.find_elements_in_distances([pos], distance)[0]

    def find_nearest_elements(self, positions, max_dist=np.inf) -> List[Optional[T]]:
        pts = self._index()
        q = np.asarray(positions, dtype=np.float64).reshape(-1, 3)
        if len(pts) == 0:
            return [None for _ in q]
        d = np.linalg.norm(pts[None, :, :] - q[:, None, :], axis=2)
        best = d.argmin(axis=1)                      # first (lowest-index) minimum
        return [self.elements[int(b)] if np.isfinite(d[i, b]) and d[i, b] <= max_dist else None for i, b in enumerate(best)]

    def find_nearest_element(self, pos, max_dist=np.inf) -> Optional[T]:
        return self.find_nearest_elements([pos], max_dist)[0]

    def _iter_points(self):
        return self.points


class CoordKdTree(KD_Tree):
    """Positions as elements (the O2 / CO2 meshes of the reference's Greenhouse)."""

    def _get_position(self, el):
        return el


class NodeKdTree(KD_Tree):
    """Objects with a `.position` (the node meshes)."""

    def _get_position(self, el):
        return el.position
