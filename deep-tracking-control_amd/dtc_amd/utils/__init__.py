"""Trajectory batching helpers for the recurrent (GRU) policy path.

Same contracts as rsl_rl/rsl_rl/utils/utils.py:33-70, implemented as index maps (one scatter /
one masked gather, no per-trajectory Python loop) so they run as a handful of device ops:

    split_and_pad_trajectories(tensor [T,N,...], dones [T,N,1]) -> (padded [T, n_traj, ...], masks [T, n_traj])
    unpad_trajectories(padded [T, n_traj, D], masks [T, n_traj]) -> [T, N, D]

A trajectory ends at every `done` and at the last stored step.  The padded time dimension is
always T (the reference pads to the longest trajectory, which equals T whenever at least one env
ran a full rollout without a reset -- the only case in which its own `unpad` works).
"""
import torch

from .export import export_policy_as_jit  # noqa: F401  (legged_gym/utils/helpers.py:150)


def trajectory_index_map(dones):
    """dones [T,N,1] or [T,N] -> (traj_id [N*T], pos [N*T], lengths [n_traj]) in env-major order."""
    d = dones.reshape(dones.shape[0], dones.shape[1]).clone().to(torch.int64)
    d[-1] = 1
    flat = d.transpose(1, 0).reshape(-1)                     # env-major: index = n*T + t
    ends = flat.cumsum(0)
    traj_id = ends - flat                                     # number of trajectory ends strictly before
    n_traj = int(ends[-1])
    end_idx = flat.nonzero()[:, 0]
    start_idx = torch.cat((end_idx.new_zeros(1), end_idx[:-1] + 1))
    lengths = end_idx - start_idx + 1
    pos = torch.arange(flat.numel(), device=flat.device) - start_idx[traj_id]
    return traj_id, pos, lengths, n_traj


def split_and_pad_trajectories(tensor, dones):
    T = tensor.shape[0]
    traj_id, pos, lengths, n_traj = trajectory_index_map(dones)
    src = tensor.transpose(1, 0).reshape(-1, *tensor.shape[2:])
    padded = tensor.new_zeros(T, n_traj, *tensor.shape[2:])
    padded[pos, traj_id] = src
    masks = lengths > torch.arange(0, T, device=tensor.device).unsqueeze(1)
    return padded, masks


def unpad_trajectories(trajectories, masks):
    T = trajectories.shape[0]
    return trajectories.transpose(1, 0)[masks.transpose(1, 0)].view(-1, T, trajectories.shape[-1]).transpose(1, 0)


def true_indices(mask, count):
    """Flat indices of the True entries of `mask` in ascending order WITHOUT a host synchronisation: `count` = their number,
    known to the caller (torch.nonzero / boolean indexing must read it back from the device first).  A stable sort of the negated
    mask puts the True entries first, in their original order."""
    flat = mask.reshape(-1)
    return torch.argsort((~flat).to(torch.uint8), stable=True)[:count]
