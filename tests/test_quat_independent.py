"""Independent pin of the quaternion helpers.

The three Isaac Gym helpers the reference's planner calls (`quat_rotate_inverse`, `quat_apply`, `normalize`;
legged_gym/envs/base/legged_robot_dtc.py:72-73, 86, 120, 154 and legged_gym/utils/math.py:8-12) are not vendored by
the reference, so both the oracle (`oracle/quat.py`) and the capture harness that generated `tests/golden/scorer.npz`
(`tests/golden/_ref_harness.py`) restate their published formulas.  A transcription slip there would pin the oracle,
the fixture and the kernel to the same wrong formula -- this test checks both restatements against an implementation
that shares no code with them: scipy.spatial.transform.Rotation (float64), on 1e5 random unit quaternions.
CPU only."""
import os
import sys

import numpy as np
import pytest
import torch
from scipy.spatial.transform import Rotation

from oracle import quat as Q

N = 100_000
TOL = 2e-6


@pytest.fixture(scope="module")
def data():
    rng = np.random.default_rng(20260928)
    q = rng.normal(size=(N, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    v = rng.normal(size=(N, 3)) * rng.uniform(0.1, 5.0, size=(N, 1))
    return q.astype(np.float32), v.astype(np.float32)


def _rel(a, b, scale):
    return float(np.max(np.abs(a.astype(np.float64) - b) / scale))


def test_oracle_quat_rotate_inverse_vs_scipy(data):
    q, v = data
    want = Rotation.from_quat(q.astype(np.float64)).inv().apply(v.astype(np.float64))       # scipy: (x, y, z, w) order
    got = Q.quat_rotate_inverse(q, v)
    assert _rel(got, want, np.linalg.norm(v, axis=1, keepdims=True)) <= TOL


def test_oracle_yaw_quat_and_apply_vs_scipy(data):
    q, v = data
    zq, wq = Q.yaw_quat(q)
    # math.py:8-12: zero the x / y components, renormalise -> a pure rotation about z
    yaw = Rotation.from_quat(np.stack([np.zeros(N), np.zeros(N), zq.astype(np.float64), wq.astype(np.float64)], axis=1))
    p = np.concatenate([v[:, :2].astype(np.float64), np.zeros((N, 1))], axis=1)
    want = yaw.apply(p)
    x, y = Q.apply_yaw_xy(zq[:, None], wq[:, None], v[:, :1], v[:, 1:2])   # [N,1] x [N,1] broadcast
    got = np.stack([x[:, 0], y[:, 0]], axis=1)
    assert _rel(got, want[:, :2], np.linalg.norm(p, axis=1, keepdims=True) + 1e-30) <= TOL
    assert np.max(np.abs(want[:, 2])) < 1e-12                     # a z rotation keeps the plane
    # the renormalised (z, w) pair is a unit quaternion
    np.testing.assert_allclose(zq.astype(np.float64) ** 2 + wq.astype(np.float64) ** 2, 1.0, atol=3e-7)


def test_capture_harness_stubs_vs_scipy(data):
    """The stubs the golden-capture harness installs as `isaacgym.torch_utils` (the reference's own code ran on them)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import _ref_harness as H
    q, v = data
    qt, vt = torch.from_numpy(q), torch.from_numpy(v)
    rot = Rotation.from_quat(q.astype(np.float64))
    scale = np.linalg.norm(v, axis=1, keepdims=True)
    assert _rel(H._quat_apply(qt, vt).numpy(), rot.apply(v.astype(np.float64)), scale) <= TOL
    assert _rel(H._quat_rotate_inverse(qt, vt).numpy(), rot.inv().apply(v.astype(np.float64)), scale) <= TOL
    n = H._normalize(vt).numpy()
    assert _rel(n, v.astype(np.float64) / scale, 1.0) <= TOL
    tiny = torch.zeros(4, 3)
    assert torch.equal(H._normalize(tiny), tiny)                  # clamp(min=1e-9): no division by zero


def test_oracle_sincos_vs_numpy():
    x = np.linspace(-50.0, 50.0, 200_001).astype(np.float32)
    s, c = Q.sincos(x)
    assert np.max(np.abs(s.astype(np.float64) - np.sin(x.astype(np.float64)))) <= 3e-7
    assert np.max(np.abs(c.astype(np.float64) - np.cos(x.astype(np.float64)))) <= 3e-7
