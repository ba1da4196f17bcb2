"""Mesh containers (reference: src/pnmol/mesh.py)."""

from functools import cached_property

import numpy as np
import scipy.spatial


def read_bbox(points):
    """One (min, max) row per spatial dimension (mesh.py:177-184)."""
    points = np.asarray(points, dtype=np.float64)
    return np.stack([points.min(axis=0), points.max(axis=0)], axis=1)


class RectangularMesh:
    def __init__(self, points, bbox=None):
        self.points = np.asarray(points, dtype=np.float64)
        self.bbox = np.asarray(bbox, dtype=np.float64) if bbox is not None else read_bbox(self.points)
        self._tree = scipy.spatial.KDTree(data=self.points)

    @classmethod
    def from_bbox_1d(cls, bbox, step=None, num=None):  # mesh.py:85-98
        bbox = np.asarray(bbox, dtype=np.float64)
        if int(step is None) + int(num is None) != 1:
            raise ValueError("Provide exactly one of step or num.")
        if step is not None:
            num = int((bbox[1] - bbox[0]) / step) + 1
        return cls(np.linspace(bbox[0], bbox[1], num=num, endpoint=True).reshape(-1, 1))

    @classmethod
    def from_bbox_2d(cls, bbox, steps=None, nums=None):  # mesh.py:100-130
        bbox = np.asarray(bbox, dtype=np.float64)
        if int(steps is None) + int(nums is None) != 1:
            raise ValueError("Provide exactly one of step or num.")
        if steps is not None:
            num_y = int((bbox[1, 0] - bbox[0, 0]) / steps[0]) + 1
            num_x = int((bbox[1, 1] - bbox[0, 1]) / steps[1]) + 1
        else:
            num_y, num_x = nums
        Y = np.linspace(bbox[0, 0], bbox[1, 0], num=num_y, endpoint=True)
        X = np.linspace(bbox[0, 1], bbox[1, 1], num=num_x, endpoint=True)
        Xm, Ym = np.meshgrid(X, Y)
        return cls(np.stack([Xm.ravel(), Ym.ravel()], axis=1))

    def __len__(self):
        return len(self.points)

    def __getitem__(self, key):
        return self.points.__getitem__(key)

    def __repr__(self):
        return f"{type(self).__name__}({self.points!r})"

    shape = property(lambda self: self.points.shape)
    ndim = property(lambda self: self.points.ndim)
    dimension = property(lambda self: self.points.shape[-1])

    @property
    def fill_distance(self):
        return np.amin(scipy.spatial.distance_matrix(self.points, self.points))

    def neighbours(self, point, num):  # mesh.py:132-139
        if num <= 0:
            raise ValueError("num >= 1 required!")
        _, indices = self._tree.query(x=point, k=num)
        return self.points[indices], indices

    def _on_boundary(self):  # mesh.py:141-154
        on = np.zeros(len(self.points), dtype=bool)
        for k in range(self.points.shape[1]):
            on |= (self.points[:, k] == self.bbox[k, 0]) | (self.points[:, k] == self.bbox[k, 1])
        return on

    @cached_property
    def boundary(self):
        on = self._on_boundary()
        return self.points[on], on, np.nonzero(on)[0]

    @cached_property
    def interior(self):  # mesh.py:156-169
        off = ~self._on_boundary()
        return self.points[off], off, np.nonzero(off)[0]

    @cached_property
    def boundary_projection_matrix(self):  # mesh.py:171-175
        return np.eye(self.points.shape[0])[self._on_boundary(), :]
