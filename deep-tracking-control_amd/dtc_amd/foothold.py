"""Host side of the foothold planner (drop-in for legged_gym/envs/base/legged_robot_dtc.py:98-201).

`plan(...)` returns exactly the attributes the reference block writes on `self`
(pred_footholds, pred_footholds_to_robot, optimal_foothold_indice, foothold_obs,
optimal_footholds_world and, on request, foothold_score / nominal_footholds_indice / slope /
heights_world), computed by ONE fused HIP kernel (csrc/foothold.hip).  `patch_env(env)` shows
the one-line integration into an env's `post_physics_step`.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

from . import _ffi
from .synthetic import MEASURED_POINTS_X, MEASURED_POINTS_Y


@dataclass
class GridConfig:
    """cfg.terrain.measured_points_{x,y} + cfg.sim.dt * cfg.control.decimation."""
    points_x: tuple = tuple(MEASURED_POINTS_X)
    points_y: tuple = tuple(MEASURED_POINTS_Y)
    t_stance: float = 0.005 * 4
    fdbk_gain: float = 0.03

    def c_struct(self) -> _ffi.DtcGridCfg:
        c = _ffi.DtcGridCfg()
        c.nx, c.ny = len(self.points_x), len(self.points_y)
        if not (2 <= c.nx <= 64 and 2 <= c.ny <= 32):
            raise ValueError(f"grid {c.nx}x{c.ny} unsupported (nx <= 64, ny <= 32)")
        c.t_stance, c.fdbk_gain = self.t_stance, self.fdbk_gain
        for i, v in enumerate(self.points_x):
            c.x[i] = v
        for i, v in enumerate(self.points_y):
            c.y[i] = v
        return c

    @property
    def num_points(self) -> int:
        return len(self.points_x) * len(self.points_y)


def plan(measured_heights: torch.Tensor, root_states: torch.Tensor, thigh_pos: torch.Tensor,
         commands: torch.Tensor, grid: GridConfig | None = None, want_score: bool = False,
         want_debug: bool = False) -> dict:
    """measured_heights [N,P], root_states [N,13], thigh_pos [N,4,3], commands [N,4] (device fp32).

    Returns a dict keyed by the reference's attribute names."""
    grid = grid or GridConfig()
    N, P = measured_heights.shape
    if P != grid.num_points:
        raise ValueError(f"measured_heights has {P} points, grid has {grid.num_points}")
    if root_states.shape != (N, 13) or thigh_pos.shape != (N, 4, 3) or commands.shape[0] != N or commands.shape[1] < 3:
        raise ValueError("bad input shapes")
    dev = measured_heights.device
    f32 = dict(dtype=torch.float32, device=dev)
    mh = measured_heights.contiguous()
    rs = root_states.contiguous()
    th = thigh_pos.contiguous()
    cmd = commands[:, :4].contiguous() if commands.shape[1] >= 4 else torch.nn.functional.pad(commands, (0, 4 - commands.shape[1]))
    idx = torch.empty(N, 1, 4, dtype=torch.int64, device=dev)
    obs = torch.empty(N, 8, **f32)
    world = torch.empty(N, 4, 3, **f32)
    pred = torch.empty(N, 4, 3, **f32)
    p2r = torch.empty(N, 4, 3, **f32)
    score = torch.empty(N, P, 4, **f32) if want_score or want_debug else None
    nom = torch.empty(N, 4, dtype=torch.int64, device=dev) if want_debug else None
    slope = torch.empty(N, len(grid.points_x), len(grid.points_y), **f32) if want_debug else None
    hw = torch.empty(N, P, 3, **f32) if want_debug else None
    cfg = grid.c_struct()
    rc = _ffi.lib().dtc_foothold_plan(_ffi.cptr(mh, torch.float32), _ffi.cptr(rs, torch.float32),
                                      _ffi.cptr(th, torch.float32), _ffi.cptr(cmd, torch.float32), cfg,
                                      _ffi.ptr(idx), _ffi.ptr(obs), _ffi.ptr(world), _ffi.ptr(pred), _ffi.ptr(p2r),
                                      _ffi.ptr(score), _ffi.ptr(nom), _ffi.ptr(slope), _ffi.ptr(hw), N, _ffi.stream())
    _ffi.check(rc, "dtc_foothold_plan")
    out = dict(optimal_foothold_indice=idx, foothold_obs=obs, optimal_footholds_world=world,
               pred_footholds=pred, pred_footholds_to_robot=p2r)
    if score is not None:
        out["foothold_score"] = score
    if want_debug:
        out.update(nominal_footholds_indice=nom, slope=slope, heights_world=hw)
    return out


def get_heights(height_samples: torch.Tensor, root_states: torch.Tensor, grid: GridConfig | None = None,
                border_size: float = 20.0, horizontal_scale: float = 0.05, vertical_scale: float = 0.005):
    """LeggedRobot._get_heights (legged_gym/envs/base/legged_robot.py:1279-1317) for all envs:
    int16 height table [rows, cols] -> measured_heights [N, P]."""
    grid = grid or GridConfig()
    assert height_samples.dtype == torch.int16 and height_samples.dim() == 2
    N = root_states.shape[0]
    out = torch.empty(N, grid.num_points, dtype=torch.float32, device=root_states.device)
    rc = _ffi.lib().dtc_get_heights(_ffi.cptr(height_samples.contiguous()), height_samples.shape[0],
                                    height_samples.shape[1], _ffi.cptr(root_states.contiguous(), torch.float32),
                                    grid.c_struct(), border_size, horizontal_scale, vertical_scale, _ffi.ptr(out), N,
                                    _ffi.stream())
    _ffi.check(rc, "dtc_get_heights")
    return out


def rewards(foot_positions: torch.Tensor, optimal_footholds_world: torch.Tensor, contact_filt: torch.Tensor):
    """(_reward_tracking_optimal_footholds, _reward_foothold_miss) of legged_robot_dtc.py:577-586 / :536-539."""
    N = foot_positions.shape[0]
    dev = foot_positions.device
    tracking, miss = torch.empty(N, device=dev), torch.empty(N, device=dev)
    c = contact_filt.to(torch.uint8).contiguous()
    rc = _ffi.lib().dtc_foothold_rewards(_ffi.cptr(foot_positions.contiguous(), torch.float32),
                                         _ffi.cptr(optimal_footholds_world.contiguous(), torch.float32), _ffi.ptr(c),
                                         _ffi.ptr(tracking), _ffi.ptr(miss), N, _ffi.stream())
    _ffi.check(rc, "dtc_foothold_rewards")
    return tracking, miss


def patch_env(env, grid: GridConfig | None = None):
    """Attach `env.plan_footholds()` that performs lines :98-201 of post_physics_step on `env`'s
    own buffers (rigid_body_state, base_pos/quat via root_states, commands, measured_heights)."""
    grid = grid or GridConfig(tuple(env.cfg.terrain.measured_points_x), tuple(env.cfg.terrain.measured_points_y),
                              env.cfg.sim.dt * env.cfg.control.decimation)

    def plan_footholds():
        thigh = env.rigid_body_state.view(env.num_envs, env.num_bodies, 13)[:, env.thigh_indices, 0:3]
        out = plan(env.measured_heights, env.root_states, thigh, env.commands, grid)
        env.hip_positions = thigh
        for k, v in out.items():
            setattr(env, k, v)
        return out

    env.plan_footholds = plan_footholds
    return env
