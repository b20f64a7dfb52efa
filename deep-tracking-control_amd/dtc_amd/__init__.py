"""dtc_amd -- MI355X-native PPO-update + foothold-scorer hot path of Deep-Tracking-Control.

Python host code with the reference's class surface (rsl_rl OnPolicyRunner / PPO /
RolloutStorage / ActorCriticDecoder, legged_gym foothold planner) over hand-written HIP kernels
behind a C ABI (include/dtc_hip.h, csrc/*.hip).  No CPU fallback: importing the compute entry
points without the built library raises.
"""
__version__ = "0.1.0"
