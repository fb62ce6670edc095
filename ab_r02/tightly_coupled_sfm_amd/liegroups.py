"""Stand-in for the slice of utiasSTARS/liegroups the reference calls (an absent, unpinned third-party
dependency): `SE3.from_matrix(M, normalize=True)`, `.inv()`, `.dot()`, `.log()` -> [rho(3), phi(3)] translation first,
`SE3.exp(xi)`, `.as_matrix()`, `.trans`, `.rot`; and the same for `SO3`.  Call sites: data/kitti_loader_stereo.py:135-146,
kitti_loader.py:330-348, validate.py:65-71, ...

The arithmetic is the library's own closed-form SE(3) code (include/tcsfm.h: tcsfm_se3_exp / _log / _mul / _inv, double
precision, host side -- the same routines the solve kernel uses on the device).  PARITY UNPINNED: the reference has no
tests or fixtures for liegroups; these classes are checked for self-consistency (tests/test_abi_cpu.py).
"""
from __future__ import annotations

import numpy as np

from . import engine as _E


def _orthonormalise(R):
    U, _, Vt = np.linalg.svd(np.asarray(R, dtype=np.float64))
    D = np.eye(3); D[2, 2] = np.sign(np.linalg.det(U @ Vt))
    return U @ D @ Vt


class SO3:
    dof, dim = 3, 3

    def __init__(self, mat):
        self.mat = np.asarray(mat, dtype=np.float64).reshape(3, 3).copy()

    @classmethod
    def identity(cls):
        return cls(np.eye(3))

    @classmethod
    def from_matrix(cls, mat, normalize=False):
        return cls(_orthonormalise(mat) if normalize else mat)

    @classmethod
    def exp(cls, phi):
        return cls(_E.se3_exp(np.concatenate([np.zeros(3), np.asarray(phi, dtype=np.float64).reshape(3)]))[:, :3])

    def log(self):
        return _E.se3_log(np.concatenate([self.mat, np.zeros((3, 1))], 1))[3:]

    def inv(self):
        return SO3(self.mat.T)

    def as_matrix(self):
        return self.mat.copy()

    def dot(self, other):
        if isinstance(other, SO3):
            return SO3(self.mat @ other.mat)
        other = np.asarray(other, dtype=np.float64)
        return (other @ self.mat.T) if other.ndim == 2 else self.mat @ other


class SE3:
    dof, dim = 6, 4

    def __init__(self, rot, trans=None):
        if trans is None:                      # a 3x4 / 4x4 matrix
            M = np.asarray(rot, dtype=np.float64)
            rot, trans = M[:3, :3], M[:3, 3]
        self.rot = rot if isinstance(rot, SO3) else SO3(rot)
        self.trans = np.asarray(trans, dtype=np.float64).reshape(3).copy()

    def _T(self):
        return np.concatenate([self.rot.mat, self.trans[:, None]], 1)

    @classmethod
    def identity(cls):
        return cls(np.eye(3), np.zeros(3))

    @classmethod
    def from_matrix(cls, mat, normalize=False):
        M = np.asarray(mat, dtype=np.float64)
        return cls(SO3.from_matrix(M[:3, :3], normalize=normalize), M[:3, 3])

    @classmethod
    def exp(cls, xi):
        return cls(_E.se3_exp(xi))

    def log(self):
        """[rho, phi]: translation part first, as the reference's pose vectors are built (kitti_loader_stereo.py:135-147)"""
        return _E.se3_log(self._T())

    def inv(self):
        return SE3(_E.se3_inv(self._T()))

    def as_matrix(self):
        return np.vstack([self._T(), [0.0, 0.0, 0.0, 1.0]])

    def dot(self, other):
        if isinstance(other, SE3):
            return SE3(_E.se3_mul(self._T(), other._T()))
        other = np.asarray(other, dtype=np.float64)
        if other.ndim == 2:                    # points [N,3] (or homogeneous [N,4])
            return other[:, :3] @ self.rot.mat.T + self.trans if other.shape[1] == 3 else other @ self.as_matrix().T
        return self.rot.mat @ other[:3] + self.trans if other.shape[0] == 3 else self.as_matrix() @ other
