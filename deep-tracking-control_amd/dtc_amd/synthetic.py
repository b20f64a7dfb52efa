"""Deterministic synthetic inputs for the hot path.

Isaac Gym physics is out of scope, so both parity tests and `bench.py` run on synthetic /
pre-recorded rollout buffers.  Everything here is a pure function of (shape, seed) on a
`torch.Generator`, so the CPU oracle and the HIP path can be fed identical numbers.
"""
from __future__ import annotations

import math

import torch

NUM_OBS, NUM_PRIV, NUM_HIST, NUM_ACT, N_POINTS = 53, 1389, 265, 12, 693

# Lite3DTCCfg.terrain (legged_gym/envs/lite3/lite3_dtc_config.py:32-36)
MEASURED_POINTS_X = [round(-0.8 + 0.05 * i, 2) for i in range(33)]
MEASURED_POINTS_Y = [round(-0.5 + 0.05 * i, 2) for i in range(21)]


def _gen(seed: int, device="cpu") -> torch.Generator:
    return torch.Generator(device=device).manual_seed(seed)


def rollout(num_envs: int, num_steps: int = 24, seed: int = 4, device="cpu") -> dict:
    """A filled `[T, N, d]` rollout (field names = RolloutStorage attributes,
    rsl_rl/rsl_rl/storage/rollout_storage.py:57-97) + `last_values`."""
    g = _gen(seed, device)
    T, N = num_steps, num_envs
    rn = lambda *s: torch.randn(*s, generator=g, device=device)
    obs_seq = rn(T + 1, N, NUM_OBS)
    out = dict(
        observations=obs_seq[:T].contiguous(),
        next_observations=obs_seq[1:].contiguous(),
        privileged_observations=rn(T, N, NUM_PRIV).clamp_(-5.0, 5.0),
        observation_histories=rn(T, N, NUM_HIST),
        base_vel=rn(T, N, 3),
        rewards=0.1 * rn(T, N, 1),
        dones=(torch.rand(T, N, 1, generator=g, device=device) < 0.02).to(torch.uint8),
        mu=0.01 * rn(T, N, NUM_ACT),
        sigma=torch.ones(T, N, NUM_ACT, device=device),
        values=0.01 * rn(T, N, 1),
    )
    out["actions"] = out["mu"] + out["sigma"] * rn(T, N, NUM_ACT)
    a, m, s = out["actions"], out["mu"], out["sigma"]
    logp = -((a - m) ** 2) / (2 * s * s) - s.log() - math.log(math.sqrt(2 * math.pi))
    out["actions_log_prob"] = logp.sum(-1, keepdim=True)
    out["last_values"] = torch.zeros(N, 1, device=device)
    return out


def update_noise(num_envs: int, num_steps: int = 24, num_mini_batches: int = 4, num_epochs: int = 5,
                 seed: int = 123, device="cpu"):
    """The random draws one `PPO.update` consumes: the mini-batch permutation (one per update,
    rollout_storage.py:165) and the two reparameterisation noises per mini-batch
    (actor_critic_decoder.py:283)."""
    g = _gen(seed, device)
    mb = (num_envs * num_steps) // num_mini_batches
    steps = num_mini_batches * num_epochs
    perm = torch.randperm(num_mini_batches * mb, generator=g, device=device)
    eps1 = torch.randn(steps, mb, 16, generator=g, device=device)
    eps2 = torch.randn(steps, mb, 16, generator=g, device=device)
    return perm, eps1, eps2


def height_points() -> torch.Tensor:
    """[693, 3] base-frame sample grid, flat index i = ix*21 + iy
    (legged_gym/envs/base/legged_robot.py:1263-1277)."""
    x = torch.tensor(MEASURED_POINTS_X)
    y = torch.tensor(MEASURED_POINTS_Y)
    gx, gy = torch.meshgrid(x, y, indexing="ij")
    pts = torch.zeros(N_POINTS, 3)
    pts[:, 0] = gx.flatten()
    pts[:, 1] = gy.flatten()
    return pts


def scorer_inputs(num_envs: int, seed: int = 7, device="cpu") -> dict:
    """Mock env state for the foothold planner: root_states [N,13] (pos, quat
    xyzw, lin vel, ang vel), thigh positions [N,4,3] (FL,FR,HL,HR), commands [N,4],
    measured_heights [N,693] (stepping-stone-like, quantised to vertical_scale = 0.005)."""
    g = _gen(seed, device)
    N = num_envs
    ru = lambda *s: torch.rand(*s, generator=g, device=device)
    rn = lambda *s: torch.randn(*s, generator=g, device=device)
    root = torch.zeros(N, 13, device=device)
    root[:, 0] = 20.0 + 40.0 * ru(N)
    root[:, 1] = 20.0 + 10.0 * ru(N)
    root[:, 2] = 0.3 + 0.3 * ru(N)
    yaw = (2 * ru(N) - 1) * math.pi
    q = torch.stack([0.05 * rn(N), 0.05 * rn(N), torch.sin(yaw / 2), torch.cos(yaw / 2)], dim=1)
    root[:, 3:7] = q / q.norm(dim=1, keepdim=True)
    root[:, 7:13] = 0.5 * rn(N, 6)
    commands = 0.5 * rn(N, 4)
    off = torch.tensor([[0.17, 0.1, 0.0], [0.17, -0.1, 0.0], [-0.17, 0.1, 0.0], [-0.17, -0.1, 0.0]],
                       device=device)
    c, s = torch.cos(yaw), torch.sin(yaw)
    thigh = torch.empty(N, 4, 3, device=device)
    thigh[:, :, 0] = root[:, None, 0] + c[:, None] * off[None, :, 0] - s[:, None] * off[None, :, 1]
    thigh[:, :, 1] = root[:, None, 1] + s[:, None] * off[None, :, 0] + c[:, None] * off[None, :, 1]
    thigh[:, :, 2] = root[:, None, 2]
    flat = ru(N, N_POINTS) < 0.7
    steps = torch.randint(-400, 20, (N, N_POINTS), generator=g, device=device).float() * 0.005
    heights = (root[:, 2:3] - 0.32) + torch.where(flat, torch.zeros_like(steps), steps)
    return dict(root_states=root, thigh_pos=thigh.contiguous(), commands=commands,
                measured_heights=heights.contiguous())


def env_state(num_envs: int, seed: int = 13, device="cpu") -> dict:
    """Mock env state consumed by compute_observations / check_termination (legged_robot_dtc.py:229-288):
    the scorer inputs of `scorer_inputs(seed)` plus joint state, actions, contacts, forces and the uniform draws
    the reference takes from torch.rand_like (inputs here, so that both sides use the same numbers)."""
    d = scorer_inputs(num_envs, seed=seed, device=device)
    g = _gen(seed + 1000, device)
    N = num_envs
    ru = lambda *s: torch.rand(*s, generator=g, device=device)
    rn = lambda *s: torch.randn(*s, generator=g, device=device)
    grav = torch.stack([0.1 * rn(N), 0.1 * rn(N), -1.0 + 0.05 * rn(N).abs()], dim=1)
    grav[: N // 32, 2] = 0.3                                   # fallen over -> terminated by the gravity test
    d.update(base_ang_vel=0.5 * rn(N, 3), projected_gravity=grav, dof_pos=0.3 * rn(N, 12),
             default_dof_pos=torch.tensor([0.0, -0.8, 1.6] * 4, device=device), dof_vel=2.0 * rn(N, 12),
             actions=rn(N, 12), foothold_obs=0.5 * rn(N, 8), forces=10.0 * rn(N, 17, 3),
             height_noise_offset=0.02 * rn(N, 1).expand(N, N_POINTS).contiguous(),
             u_obs=ru(N, 53), u_heights=ru(N, N_POINTS),
             noise_scale_vec=torch.cat([0.05 * ru(45), torch.zeros(8, device=device)]),
             contact_forces=40.0 * rn(N, 17, 3), episode_length_buf=torch.randint(0, 1100, (N,), generator=g, device=device),
             termination_contact_indices=torch.tensor([0, 1, 5, 9, 13], device=device))
    # a few robots sunk into the terrain -> the base-height termination test fires
    d["root_states"][N // 32: N // 16, 2] -= 0.25
    return d
