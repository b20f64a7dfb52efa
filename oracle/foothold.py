"""numpy float32 restatement of the Raibert + terrain-score foothold planner
(TEST INFRASTRUCTURE ONLY).

Follows legged_gym/envs/base/legged_robot_dtc.py:
  :98-120   Raibert nominal footholds  (rotate_positions :35-54)
  :123-148  terrain score (clamp, central differences, mean / unbiased var, thresholds)
  :152-164  distance to the nominal foothold, 0.16 m radius mask
  :169-175  combine + exception mask
  :180-181  per-leg argmin over the 693 grid points (topk k=1, smallest; ties -> lowest index)
  :184-201  observation decode (with the reference's swapped x/y table indexing) + world position
Constants: legged_robot_config.py:210 (dt), lite3_dtc_config.py:109 (decimation) -> t_stance 0.02.

Canonical operation order.  The reference's own result is insensitive to the reduction order
of the two per-env reductions (mean / var of 693 values; SURVEY.md F5: three orders gave
identical indices on 32768/32768 cases), so this oracle FIXES the order to the one the HIP
kernels use -- per lane two running sums over its float4 chunks (see wave_sum) followed by an
xor-butterfly (offsets 1,2,4,8,16,32 = the kernel's DPP quad / half-row / row / cross-row steps) -- and every other operation is a single correctly
rounded float32 op in reference order.  The HIP kernel is therefore expected to agree with
this file BIT FOR BIT on every output; this file in turn is pinned against the imported
reference by tests/golden/scorer_*.npz (knife-edge policy: SURVEY.md §8c G4).
"""
import numpy as np

from . import quat

F = np.float32
NX, NY, NP = 33, 21, 693
_BFLY = [np.arange(64) ^ o for o in (1, 2, 4, 8, 16, 32)]


def wave_sum(v, valid=None):
    """Sum over the last axis (693) in the kernel's canonical order.  v [N,693] float32.
    Lane k owns the float4 chunks q = k + 64 j (elements 4q .. 4q+3, zero past 693) and keeps two running sums --
    elements 0, 2 of every chunk in one, elements 1, 3 in the other (a packed add), chunks in ascending j, both
    starting from +0 -- adds the two, and the 64 lane sums are folded by the xor-butterfly."""
    n = v.shape[0]
    nch = (NP + 255) // 256
    pad = np.zeros((n, nch * 256), dtype=F)
    pad[:, :NP] = v
    ch = pad.reshape(n, nch, 64, 4)
    se = np.zeros((n, 64), dtype=F)
    so = np.zeros((n, 64), dtype=F)
    for j in range(nch):
        se = se + ch[:, j, :, 0]
        so = so + ch[:, j, :, 1]
        se = se + ch[:, j, :, 2]
        so = so + ch[:, j, :, 3]
    p = se + so
    for perm in _BFLY:
        p = p + p[:, perm]
    return p[:, 0]


def plan(measured_heights, root_states, thigh_pos, commands, points_x, points_y,
         t_stance=0.02, fdbk_gain=0.03, want_debug=False):
    """All inputs float32 numpy.  Returns dict with idx [N,4] int64, foothold_obs [N,8],
    optimal_footholds_world [N,4,3], pred_footholds [N,4,3], pred_footholds_to_robot [N,4,3]
    (+ foothold_score [N,693,4], nominal_idx [N,4], slope [N,33,21] when want_debug)."""
    mh = np.ascontiguousarray(measured_heights, dtype=F)
    rs = np.asarray(root_states, dtype=F)
    th = np.asarray(thigh_pos, dtype=F)
    cmd = np.asarray(commands, dtype=F)
    X = np.asarray(points_x, dtype=F)
    Y = np.asarray(points_y, dtype=F)
    N = mh.shape[0]
    base = rs[:, 0:3]
    q = rs[:, 3:7]

    # ---- Raibert heuristic (:98-120)
    v_body = quat.quat_rotate_inverse(q, rs[:, 7:10])
    sn, cs = quat.sincos(cmd[:, 2])
    h2b = th - base[:, None, :]
    rx = cs[:, None] * h2b[:, :, 0] + (-sn[:, None]) * h2b[:, :, 1]
    ry = sn[:, None] * h2b[:, :, 0] + cs[:, None] * h2b[:, :, 1]
    rot = np.stack([rx, ry, h2b[:, :, 2]], axis=2).astype(F)
    p_sh = base[:, None, :] + rot
    cmd_lin = np.stack([cmd[:, 0], cmd[:, 1], np.zeros(N, dtype=F)], axis=1)
    p_sym = F(t_stance / 2) * v_body + F(fdbk_gain) * (v_body - cmd_lin)
    pred = (p_sh + p_sym[:, None, :]).astype(F)
    rel = pred - base[:, None, :]
    pred_to_robot = np.stack([quat.quat_rotate_inverse(q, rel[:, l, :]) for l in range(4)], axis=1)

    # ---- terrain score (:123-148)
    g = mh - base[:, 2:3]
    exc = (g > F(1.0)) | (g < F(-1.0))
    g = np.clip(g, F(-0.5), F(0.5))
    G = g.reshape(N, NX, NY)
    dx = np.empty_like(G)
    dy = np.empty_like(G)
    dx[:, 1:-1, :] = (G[:, 2:, :] - G[:, :-2, :]) / F(0.1)
    dx[:, 0, :] = (G[:, 1, :] - G[:, 0, :]) / F(0.05)
    dx[:, -1, :] = (G[:, -1, :] - G[:, -2, :]) / F(0.05)
    dy[:, :, 1:-1] = (G[:, :, 2:] - G[:, :, :-2]) / F(0.1)
    dy[:, :, 0] = (G[:, :, 1] - G[:, :, 0]) / F(0.05)
    dy[:, :, -1] = (G[:, :, -1] - G[:, :, -2]) / F(0.05)
    slope = np.sqrt(dx * dx + dy * dy).astype(F)
    mean = wave_sum(g) / F(NP)
    dev = g - mean[:, None]
    rough = np.abs(dev)
    var = wave_sum(dev * dev) / F(NP - 1)
    edge = np.minimum(np.maximum(np.sqrt(var), F(0.0)), F(0.3)).astype(F)
    s_raw = (F(0.2) * edge[:, None] + slope.reshape(N, NP)) + F(0.3) * rough
    s = np.where(s_raw < F(0.1), s_raw, F(10.0)).astype(F)

    # ---- distance to nominal (:152-164)
    zq, wq = quat.yaw_quat(q)
    px = np.repeat(X, NY)[None, :]
    py = np.tile(Y, NX)[None, :]
    ax, ay = quat.apply_yaw_xy(zq[:, None], wq[:, None], px, py)
    hx = ax + base[:, 0:1]
    hy = ay + base[:, 1:2]
    ddx = pred[:, None, :, 0] - hx[:, :, None]
    ddy = pred[:, None, :, 1] - hy[:, :, None]
    d = np.sqrt(ddx * ddx + ddy * ddy).astype(F)
    d = np.where(d < F(0.16), d, F(10.0)).astype(F)

    # ---- combine (:169-175) and argmin (:180-181)
    tot = s[:, :, None] * F(0.2) + d * F(0.8)
    tot = np.where(exc[:, :, None], F(10.0), tot).astype(F)
    idx = np.argmin(tot, axis=1).astype(np.int64)      # first occurrence == lowest index on ties

    # ---- decode (:184-201)
    Ytile = np.tile(Y, 4)
    obs = np.concatenate([X[idx % NY], Ytile[idx // NY]], axis=1).astype(F)
    rows = np.arange(N)[:, None]
    world = np.stack([hx[rows, idx], hy[rows, idx], mh[rows, idx]], axis=2).astype(F)
    out = dict(idx=idx, foothold_obs=obs, optimal_footholds_world=world, pred_footholds=pred,
               pred_footholds_to_robot=pred_to_robot.astype(F))
    if want_debug:
        out.update(foothold_score=tot, nominal_idx=np.argmin(d, axis=1).astype(np.int64),
                   slope=slope, heights_world_xy=np.stack([hx, hy], axis=2))
        # runner-up gap per (env, leg): used for knife-edge tagging
        srt = np.sort(tot, axis=1)
        out["gap"] = (srt[:, 1, :] - srt[:, 0, :]).astype(F)
        marg = np.minimum(np.abs(s_raw - F(0.1)).min(axis=1), np.abs(np.abs(mh - base[:, 2:3]) - F(1.0)).min(axis=1))
        dm = np.abs(np.sqrt(ddx * ddx + ddy * ddy) - F(0.16)).min(axis=1)
        out["threshold_margin"] = np.minimum(marg[:, None], dm).astype(F)
    return out


def rewards(foot_positions, optimal_footholds_world, contact_filt):
    """legged_robot_dtc.py:577-586 (_reward_tracking_optimal_footholds) and :536-539 (_reward_foothold_miss)."""
    fp = np.asarray(foot_positions, dtype=F)
    ow = np.asarray(optimal_footholds_world, dtype=F)
    d = fp[:, :, :2] - ow[:, :, :2]
    dis = np.sqrt(d[:, :, 0] * d[:, :, 0] + d[:, :, 1] * d[:, :, 1]).astype(F)
    per_foot = -np.log(F(0.8) + dis).astype(F)
    filt = np.where(np.asarray(contact_filt).astype(bool), per_foot, F(0.0)).astype(F)
    tracking = ((filt[:, 0] + filt[:, 1]) + filt[:, 2]) + filt[:, 3]
    miss = np.where(fp[:, :, 2].min(axis=1) < 0, F(1.0), F(0.0)).astype(F)
    return tracking.astype(F), miss
