"""CPU oracle for the PPO-update + foothold-scorer hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU restatement (numpy for the scorer / GAE /
height sampling, torch-CPU for the neural update, because the reference's arithmetic *is*
torch-CPU) of the reference algorithms named in SURVEY.md §8(a).  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it, and only
as the checker / the timed CPU baseline.  The product path (`deep-tracking-control_amd/`)
never imports it and raises if the HIP library is missing.

Pinning status:
  * PPO update, GAE, gather, `_get_heights`: pinned against outputs of the imported
    reference (`tests/golden/make_golden.py` -> `tests/golden/*.npz`).
  * GRU actor-critic (config 3) and the GRU + CE-net composite (config 5, build-defined): modules, padding and
    BPTT gradients pinned against the imported reference classes (`gru.npz`, `composite.npz`); the training
    step around them is the ppo.py:288-338 loss block already pinned by `ppo.npz`.
  * Foothold scorer: pinned against `LeggedRobotDTC.post_physics_step` run on a mock env,
    EXCEPT for the three Isaac Gym quaternion helpers (`isaacgym.torch_utils`, un-vendored,
    version unpinned by the reference) whose published formulas are restated -> that
    boundary is "parity unpinned".
"""
