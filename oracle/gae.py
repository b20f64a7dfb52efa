"""numpy float32 restatement of RolloutStorage.compute_returns (TEST INFRASTRUCTURE ONLY).

Follows rsl_rl/rsl_rl/storage/rollout_storage.py:138-152: backward-in-time GAE(lambda)
scan, then advantage normalisation with the *unbiased* std over all T*N samples.
The scan is elementwise per env with a fixed op order, so `returns` (and the un-normalised
advantages) are expected to match a GPU implementation bit for bit; mean/std are reductions
(fp64 accumulation here) so normalised advantages carry a ~1e-6 relative tolerance.
"""
import numpy as np

F = np.float32


def gae_scan(rewards, values, dones, last_values, gamma=0.99, lam=0.95):
    """rewards, values [T,N] float32; dones [T,N] uint8; last_values [N] -> returns [T,N]."""
    T, N = rewards.shape
    g, l = F(gamma), F(lam)
    returns = np.empty((T, N), dtype=F)
    adv = np.zeros(N, dtype=F)
    for t in reversed(range(T)):
        nv = last_values if t == T - 1 else values[t + 1]
        nnt = F(1.0) - dones[t].astype(F)
        delta = (rewards[t] + (nnt * g) * nv) - values[t]
        adv = delta + ((nnt * g) * l) * adv
        returns[t] = adv + values[t]
    return returns


def normalize_advantages(returns, values):
    a = (returns - values).astype(F)
    a64 = a.astype(np.float64)
    mean = a64.mean()
    std = a64.std(ddof=1)
    return ((a - F(mean)) / (F(std) + F(1e-8))).astype(F), float(mean), float(std)


def compute_returns(rewards, values, dones, last_values, gamma=0.99, lam=0.95):
    ret = gae_scan(rewards, values, dones, last_values, gamma, lam)
    adv, _, _ = normalize_advantages(ret, values)
    return ret, adv
