"""Host side of the foothold planner (drop-in for legged_gym/envs/base/legged_robot_dtc.py:98-201).

`plan(...)` returns exactly the attributes the reference block writes on `self`
(pred_footholds, pred_footholds_to_robot, optimal_foothold_indice, foothold_obs,
optimal_footholds_world and, on request, foothold_score / nominal_footholds_indice / slope /
heights_world), computed by ONE fused HIP kernel (csrc/foothold.hip); `plan_from_table(...)` additionally samples the
terrain height field (`_get_heights`) inside the same launch.  `patch_env(env)` shows the one-line integration into an
env's `post_physics_step`.
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


def plan_from_table(height_samples: torch.Tensor, root_states: torch.Tensor, thigh_pos: torch.Tensor,
                    commands: torch.Tensor, grid: GridConfig | None = None, border_size: float = 20.0,
                    horizontal_scale: float = 0.05, vertical_scale: float = 0.005) -> dict:
    """`_get_heights` (legged_robot.py:1279-1317) + the foothold block (legged_robot_dtc.py:98-201) in ONE launch: the
    planner samples the int16 terrain table itself and also returns `measured_heights` [N,P] for the observations.
    Same results as `get_heights(...)` followed by `plan(...)`, bit for bit."""
    grid = grid or GridConfig()
    assert height_samples.dtype == torch.int16 and height_samples.dim() == 2
    N, dev = root_states.shape[0], root_states.device
    rs = root_states.contiguous()
    th = thigh_pos.contiguous()
    cmd = commands.contiguous()
    if rs.shape != (N, 13) or th.shape != (N, 4, 3) or cmd.shape[0] != N or cmd.shape[1] < 3:
        raise ValueError("bad input shapes")
    cmd = cmd[:, :4].contiguous() if cmd.shape[1] >= 4 else torch.nn.functional.pad(cmd, (0, 4 - cmd.shape[1]))
    mh = torch.empty(N, grid.num_points, dtype=torch.float32, device=dev)
    idx = torch.empty(N, 1, 4, dtype=torch.int64, device=dev)
    obs = torch.empty(N, 8, dtype=torch.float32, device=dev)
    world, pred, p2r = (torch.empty(N, 4, 3, dtype=torch.float32, device=dev) for _ in range(3))
    hs = height_samples.contiguous()
    rc = _ffi.lib().dtc_foothold_plan_from_table(_ffi.cptr(hs), hs.shape[0], hs.shape[1], border_size, horizontal_scale,
                                                 vertical_scale, _ffi.cptr(rs, torch.float32), _ffi.cptr(th, torch.float32),
                                                 _ffi.cptr(cmd, torch.float32), grid.c_struct(), _ffi.ptr(mh), _ffi.ptr(idx),
                                                 _ffi.ptr(obs), _ffi.ptr(world), _ffi.ptr(pred), _ffi.ptr(p2r), N, _ffi.stream())
    _ffi.check(rc, "dtc_foothold_plan_from_table")
    return dict(measured_heights=mh, optimal_foothold_indice=idx, foothold_obs=obs,
                optimal_footholds_world=world, pred_footholds=pred, pred_footholds_to_robot=p2r)


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


@dataclass
class ObsConfig:
    """Scales / sizes consumed by compute_observations and check_termination (defaults = Lite3DTCCfg:
    legged_robot_config.py:181-188, lite3_dtc_config.py:139, legged_robot_dtc.py:244-246, 278)."""
    ang_vel: float = 0.25
    dof_pos: float = 1.0
    dof_vel: float = 0.05
    height_measurements: float = 5.0
    force: float = 0.005
    lin_vel: float = 2.0
    base_height_target: float = 0.32
    height_noise: float = 0.1
    term_height: float = 0.15
    num_dof: int = 12
    num_foothold_obs: int = 8
    num_points: int = len(MEASURED_POINTS_X) * len(MEASURED_POINTS_Y)
    term_row0: int = 10 * 21
    term_row1: int = (33 - 10) * 21

    def c_struct(self) -> _ffi.DtcObsCfg:
        c = _ffi.DtcObsCfg()
        for k in ("ang_vel", "dof_pos", "dof_vel", "height_measurements", "force", "base_height_target", "height_noise",
                  "term_height", "num_dof", "num_foothold_obs", "num_points", "term_row0", "term_row1"):
            setattr(c, k, getattr(self, k))
        c.commands_scale[0], c.commands_scale[1], c.commands_scale[2] = self.lin_vel, self.lin_vel, self.ang_vel
        return c


def compute_observations(base_ang_vel, projected_gravity, commands, dof_pos, default_dof_pos, dof_vel, actions,
                         foothold_obs, root_states, measured_heights, forces, height_noise_offset=None, u_obs=None,
                         noise_scale_vec=None, u_heights=None, cfg: ObsConfig | None = None, where=None, out=None):
    """LeggedRobotDTC.compute_observations (legged_robot_dtc.py:255-288) as one kernel.  `forces` is the env's
    [N, num_bodies, 3] force buffer (body 0 is used); `u_obs` / `u_heights` are the uniform [0,1) draws the reference
    takes with torch.rand_like (pass None to omit the noise term).  Returns dict(obs_buf, privileged_obs_buf, heights).
    `where` ([N] bool / uint8) + `out` (the dict a previous call or `EnvStep` returned): only the rows with where != 0 are
    recomputed, in place -- the pass after `reset_idx` when `EnvStep` has written the rows of the envs that were not reset."""
    cfg = cfg or ObsConfig()
    N, dev, P = root_states.shape[0], root_states.device, cfg.num_points
    n_obs = 9 + 3 * cfg.num_dof + cfg.num_foothold_obs
    if where is not None and out is None:
        raise ValueError("compute_observations(where=...) updates rows in place: pass the buffers as out=")
    if out is not None:
        obs, priv, heights = out["obs_buf"], out["privileged_obs_buf"], out["heights"]
        if obs.shape != (N, n_obs) or priv.shape != (N, 2 * P + 3) or heights.shape != (N, P) or not (
                obs.is_contiguous() and priv.is_contiguous() and heights.is_contiguous()):
            raise ValueError("out= buffers do not match the observation shapes")
    else:
        obs = torch.empty(N, n_obs, device=dev)
        priv = torch.empty(N, 2 * P + 3, device=dev)
        heights = torch.empty(N, P, device=dev)
    wh = where.to(torch.uint8).contiguous() if where is not None else None
    forces = forces.contiguous().float()
    ld_f = forces.stride(0) if forces.dim() > 1 else 3          # floats between consecutive envs (num_bodies * 3)
    keep = [t.contiguous().float() if t is not None else None for t in
            (base_ang_vel, projected_gravity, commands, dof_pos, default_dof_pos.reshape(-1), dof_vel, actions, foothold_obs,
             root_states, measured_heights)]
    opt = [t.contiguous().float() if t is not None else None for t in (height_noise_offset, u_obs, noise_scale_vec, u_heights)]
    rc = _ffi.lib().dtc_compute_observations_where(*[_ffi.ptr(t) for t in keep], _ffi.ptr(forces), ld_f, _ffi.ptr(opt[0]),
                                                   _ffi.ptr(opt[1]), _ffi.ptr(opt[2]), _ffi.ptr(opt[3]), cfg.c_struct(),
                                                   _ffi.ptr(obs), _ffi.ptr(priv), _ffi.ptr(heights), _ffi.ptr(wh), N, _ffi.stream())
    _ffi.check(rc, "dtc_compute_observations_where")
    return dict(obs_buf=obs, privileged_obs_buf=priv, heights=heights)


def check_termination(contact_forces, termination_contact_indices, episode_length_buf, max_episode_length,
                      projected_gravity, root_states, measured_heights, cfg: ObsConfig | None = None):
    """LeggedRobotDTC.check_termination (legged_robot_dtc.py:229-248).  Returns (reset_buf, time_out_buf) as bool
    tensors and the base-height mean of the last test."""
    cfg = cfg or ObsConfig()
    N, dev = root_states.shape[0], root_states.device
    cf = contact_forces.contiguous().float()
    idx = termination_contact_indices.to(device=dev, dtype=torch.int32).contiguous()
    ep = episode_length_buf.to(torch.int64).contiguous()
    reset, tout = torch.empty(N, dtype=torch.uint8, device=dev), torch.empty(N, dtype=torch.uint8, device=dev)
    mean = torch.empty(N, device=dev)
    g, r, h = projected_gravity.contiguous().float(), root_states.contiguous().float(), measured_heights.contiguous().float()
    rc = _ffi.lib().dtc_check_termination(_ffi.ptr(cf), cf.shape[1], _ffi.ptr(idx), idx.numel(), _ffi.ptr(ep),
                                          int(max_episode_length), _ffi.ptr(g), _ffi.ptr(r), _ffi.ptr(h), cfg.c_struct(),
                                          _ffi.ptr(reset), _ffi.ptr(tout), _ffi.ptr(mean), N, _ffi.stream())
    _ffi.check(rc, "dtc_check_termination")
    return reset.bool(), tout.bool(), mean


class EnvStep:
    """One env step's post-physics block as ONE launch (`dtc_env_post_physics`; BASELINE configs[3] -- 4096 envs, where each of the
    separate calls is launch-bound): [`_get_heights`] + the foothold block (legged_robot_dtc.py:98-201) + `check_termination`
    (:229-248) + the two foothold rewards (:577-586, :536-539) + `compute_observations` (:255-288), on persistent output buffers.
    Same bits as `plan[_from_table]` + `check_termination` + `rewards` + `compute_observations` (tests/test_hip_envstep.py).

    The reference resets envs between the rewards and the observations (`reset_idx`, :209-211): the observation rows returned are
    those of the state passed in; after resetting, call `compute_observations(..., where=reset_buf, out=<this result>)`."""

    def __init__(self, num_envs: int, device, grid: GridConfig | None = None, cfg: ObsConfig | None = None):
        self.N, self.device = num_envs, torch.device(device)
        self.grid, self.cfg = grid or GridConfig(), cfg or ObsConfig()
        N, P, f = num_envs, self.grid.num_points, dict(dtype=torch.float32, device=self.device)
        n_obs = 9 + 3 * self.cfg.num_dof + self.cfg.num_foothold_obs
        u8 = dict(dtype=torch.uint8, device=self.device)
        self.out = dict(measured_heights=torch.empty(N, P, **f), optimal_foothold_indice=torch.empty(N, 1, 4, dtype=torch.int64, device=self.device),
                        foothold_obs=torch.empty(N, 8, **f), optimal_footholds_world=torch.empty(N, 4, 3, **f),
                        pred_footholds=torch.empty(N, 4, 3, **f), pred_footholds_to_robot=torch.empty(N, 4, 3, **f),
                        reset_buf=torch.empty(N, **u8), time_out_buf=torch.empty(N, **u8), height_mean=torch.empty(N, **f),
                        rew_tracking_optimal_footholds=torch.empty(N, **f), rew_foothold_miss=torch.empty(N, **f),
                        obs_buf=torch.empty(N, n_obs, **f), privileged_obs_buf=torch.empty(N, 2 * P + 3, **f), heights=torch.empty(N, P, **f))
        self._grid_c, self._cfg_c = self.grid.c_struct(), self.cfg.c_struct()

    def __call__(self, *, root_states, thigh_pos, commands, contact_forces, termination_contact_indices, episode_length_buf,
                 max_episode_length, projected_gravity, foot_positions, contact_filt, base_ang_vel, dof_pos, default_dof_pos, dof_vel,
                 actions, forces, measured_heights=None, height_samples=None, border_size=20.0, horizontal_scale=0.05,
                 vertical_scale=0.005, height_noise_offset=None, u_obs=None, noise_scale_vec=None, u_heights=None) -> dict:
        N, o = self.N, self.out
        if (measured_heights is None) == (height_samples is None):
            raise ValueError("EnvStep: pass exactly one of measured_heights (an input) and height_samples (the terrain table)")
        f32 = lambda t: t.contiguous().float()            # noqa: E731  (no copies for the env's own fp32 buffers)
        rs, th = f32(root_states), f32(thigh_pos)
        cmd = f32(commands[:, :4]) if commands.shape[1] >= 4 else torch.nn.functional.pad(commands.float(), (0, 4 - commands.shape[1]))
        if rs.shape != (N, 13) or th.shape != (N, 4, 3) or cmd.shape[0] != N:
            raise ValueError("bad input shapes")
        cf, fr = f32(contact_forces), f32(forces)
        keep = [rs, th, cmd, cf, fr]
        st = _ffi.DtcEnvStep()
        if height_samples is not None:
            assert height_samples.dtype == torch.int16 and height_samples.dim() == 2
            hs = height_samples.contiguous()
            keep.append(hs)
            st.height_samples, st.rows, st.cols = _ffi.ptr(hs), hs.shape[0], hs.shape[1]
            st.border_size, st.horizontal_scale, st.vertical_scale = border_size, horizontal_scale, vertical_scale
            mh = o["measured_heights"]
        else:
            mh = f32(measured_heights)
            if mh.shape != (N, self.grid.num_points):
                raise ValueError(f"measured_heights must be ({N}, {self.grid.num_points})")
            keep.append(mh)
        # the index list is static in an env: converted once per tensor object (the conversion is a launch of its own)
        key = (id(termination_contact_indices), termination_contact_indices._version)
        if getattr(self, "_tidx_key", None) != key:
            self._tidx_key, self._tidx = key, termination_contact_indices.to(device=self.device, dtype=torch.int32).contiguous()
            self._tidx_src = termination_contact_indices          # keeps id() unique while cached
        tidx = self._tidx
        ep = episode_length_buf.to(torch.int64).contiguous()
        cflt = contact_filt.contiguous()
        cflt = cflt.view(torch.uint8) if cflt.dtype == torch.bool else cflt.to(torch.uint8)      # bool is one byte of 0 / 1: no copy
        named = dict(root_states=rs, thigh_pos=th, commands=cmd, measured_heights=mh, idx=o["optimal_foothold_indice"],
                     foothold_obs=o["foothold_obs"], opt_world=o["optimal_footholds_world"], pred=o["pred_footholds"],
                     pred_to_robot=o["pred_footholds_to_robot"], contact_forces=cf, termination_contact_indices=tidx,
                     episode_length_buf=ep, projected_gravity=f32(projected_gravity), reset_buf=o["reset_buf"],
                     time_out_buf=o["time_out_buf"], height_mean=o["height_mean"], foot_positions=f32(foot_positions), contact_filt=cflt,
                     rew_tracking=o["rew_tracking_optimal_footholds"], rew_miss=o["rew_foothold_miss"], base_ang_vel=f32(base_ang_vel),
                     dof_pos=f32(dof_pos), default_dof_pos=f32(default_dof_pos.reshape(-1)), dof_vel=f32(dof_vel), actions=f32(actions),
                     forces=fr, height_noise_offset=None if height_noise_offset is None else f32(height_noise_offset),
                     u_obs=None if u_obs is None else f32(u_obs), noise_scale_vec=None if noise_scale_vec is None else f32(noise_scale_vec),
                     u_heights=None if u_heights is None else f32(u_heights), obs_buf=o["obs_buf"],
                     privileged_obs_buf=o["privileged_obs_buf"], heights=o["heights"])
        for k, t in named.items():
            setattr(st, k, _ffi.ptr(t))
        st.max_episode_length, st.num_bodies, st.n_term = int(max_episode_length), cf.shape[1], tidx.numel()
        st.ld_forces = fr.stride(0) if fr.dim() > 1 else 3
        rc = _ffi.lib().dtc_env_post_physics(st, self._grid_c, self._cfg_c, N, _ffi.stream())
        _ffi.check(rc, "dtc_env_post_physics")
        res = dict(o)
        res["measured_heights"] = mh
        return res


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

    def measure_and_plan_footholds():
        """`self.measured_heights = self._get_heights()` (legged_robot.py:388 / :1279-1317) and the foothold block in one
        launch; for terrains with a height field (`env.height_samples`, `env.terrain.cfg`)."""
        thigh = env.rigid_body_state.view(env.num_envs, env.num_bodies, 13)[:, env.thigh_indices, 0:3]
        tc = env.terrain.cfg
        out = plan_from_table(env.height_samples, env.root_states, thigh, env.commands, grid, tc.border_size,
                              tc.horizontal_scale, tc.vertical_scale)
        env.hip_positions = thigh
        for k, v in out.items():
            setattr(env, k, v)
        return out

    env.plan_footholds = plan_footholds
    env.measure_and_plan_footholds = measure_and_plan_footholds
    return env
