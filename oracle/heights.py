"""numpy restatement of LeggedRobot._get_heights (TEST INFRASTRUCTURE ONLY).

Follows legged_gym/envs/base/legged_robot.py:1279-1317: yaw-rotate the 693 base-frame grid
points, add the base position, `+ border_size`, `/ horizontal_scale`, truncate to int64
(`.long()`), clip to [0, dim-2], take the min of the three int16 samples
(px,py), (px+1,py), (px,py+1), scale by vertical_scale.  float32, fixed op order (the same
order csrc/heights.hip executes), so a GPU implementation can be compared bit for bit.
"""
import numpy as np

from . import quat

F = np.float32


def sample_indices(root_states, points_x, points_y, border_size, horizontal_scale, rows, cols):
    rs = np.asarray(root_states, dtype=F)
    X = np.asarray(points_x, dtype=F)
    Y = np.asarray(points_y, dtype=F)
    zq, wq = quat.yaw_quat(rs[:, 3:7])
    px = np.repeat(X, len(Y))[None, :]
    py = np.tile(Y, len(X))[None, :]
    ax, ay = quat.apply_yaw_xy(zq[:, None], wq[:, None], px, py)
    wx = ((ax + rs[:, 0:1]) + F(border_size)) / F(horizontal_scale)
    wy = ((ay + rs[:, 1:2]) + F(border_size)) / F(horizontal_scale)
    ix = np.clip(np.trunc(wx).astype(np.int64), 0, rows - 2)
    iy = np.clip(np.trunc(wy).astype(np.int64), 0, cols - 2)
    return ix, iy


def get_heights(height_samples, root_states, points_x, points_y, border_size=20.0,
                horizontal_scale=0.05, vertical_scale=0.005):
    """height_samples int16 [rows, cols] -> measured_heights float32 [N, 693]."""
    hs = np.asarray(height_samples)
    ix, iy = sample_indices(root_states, points_x, points_y, border_size, horizontal_scale, *hs.shape)
    h = np.minimum(np.minimum(hs[ix, iy], hs[ix + 1, iy]), hs[ix, iy + 1])
    return (h.astype(F) * F(vertical_scale)).astype(F)
