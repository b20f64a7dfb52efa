"""float32 quaternion / trig helpers of the scorer oracle (TEST INFRASTRUCTURE ONLY).

Restates the published Isaac Gym Preview 4 `isaacgym.torch_utils` helpers the reference
calls (`quat_rotate_inverse`, `quat_apply`, `normalize`; call sites
legged_gym/envs/base/legged_robot_dtc.py:72-73,120 and legged_gym/utils/math.py:8-12).
Isaac Gym is not vendored by the reference and its version is unpinned
(requirements.txt:1) -> "parity unpinned" at this boundary.

Every operation is an individually rounded IEEE float32 op in a fixed order (no FMA), which
is exactly what the HIP kernel executes (compiled with -ffp-contract=off), so GPU results
can be compared bit for bit.  `sincos` is a fixed Cody-Waite + minimax-polynomial routine
shared verbatim (same constants, same order) with csrc/foothold.hip.
"""
import numpy as np

F = np.float32

# pi/2 split into three parts (first two have few mantissa bits so k*P is exact)
_P1 = F(1.5703125)
_P2 = F(4.837512969970703125e-4)
_P3 = F(7.54978995489188216e-8)
_TWO_OVER_PI = F(0.636619772367581343)
_S1, _S2, _S3 = F(-1.6666654611e-1), F(8.3321608736e-3), F(-1.9515295891e-4)
_C1, _C2, _C3 = F(4.166664568298827e-2), F(-1.388731625493765e-3), F(2.443315711809948e-5)


def sincos(theta):
    """(sin, cos) of a float32 array; <= ~2 ulp for |theta| < 1e4."""
    x = np.asarray(theta, dtype=F)
    k = np.rint(x * _TWO_OVER_PI).astype(F)
    r = ((x - k * _P1) - k * _P2) - k * _P3
    r2 = r * r
    s = r + (r * r2) * (_S1 + r2 * (_S2 + r2 * _S3))
    c = (F(1.0) - F(0.5) * r2) + (r2 * r2) * (_C1 + r2 * (_C2 + r2 * _C3))
    q = k.astype(np.int64) & 3
    sin = np.where(q == 0, s, np.where(q == 1, c, np.where(q == 2, -s, -c)))
    cos = np.where(q == 0, c, np.where(q == 1, -s, np.where(q == 2, -c, s)))
    return sin.astype(F), cos.astype(F)


def quat_rotate_inverse(q, v):
    """q [N,4] (xyzw), v [N,3]:  v*(2w^2-1) - 2w*(qv x v) + 2*qv*(qv.v)"""
    qx, qy, qz, qw = (q[:, i] for i in range(4))
    vx, vy, vz = (v[:, i] for i in range(3))
    s = F(2.0) * (qw * qw) - F(1.0)
    ax, ay, az = vx * s, vy * s, vz * s
    cx = qy * vz - qz * vy
    cy = qz * vx - qx * vz
    cz = qx * vy - qy * vx
    bx, by, bz = (cx * qw) * F(2.0), (cy * qw) * F(2.0), (cz * qw) * F(2.0)
    dot = (qx * vx + qy * vy) + qz * vz
    ex, ey, ez = (qx * dot) * F(2.0), (qy * dot) * F(2.0), (qz * dot) * F(2.0)
    return np.stack([(ax - bx) + ex, (ay - by) + ey, (az - bz) + ez], axis=1).astype(F)


def yaw_quat(q):
    """zero x,y and renormalise with clamp(norm, 1e-9) (math.py:8-12) -> (z', w')."""
    qz, qw = q[:, 2], q[:, 3]
    n = np.sqrt(qz * qz + qw * qw)
    n = np.maximum(n, F(1e-9))
    return (qz / n).astype(F), (qw / n).astype(F)


def apply_yaw_xy(zq, wq, px, py):
    """quat_apply((0,0,z',w'), (px,py,0)).xy with the reference's op order:
    t = 2*(qv x p);  out = (p + w*t) + (qv x t).   zq,wq [N,1]; px,py [1,P] -> [N,P]."""
    t0 = -(zq * py) * F(2.0)
    t1 = (zq * px) * F(2.0)
    x = (px + wq * t0) + (-(zq * t1))
    y = (py + wq * t1) + (zq * t0)
    return x.astype(F), y.astype(F)
