"""Run-to-run determinism of a recurrent update on the two-term fp16 GEMM path: identical runs in one process must end with the same
bits in every parameter, on the serial and on the overlapped schedule.  (How the address-reuse hazard of ops.Amax was found: the first
run of a process differed from the later ones by ~1e-8 in the critic's weights -- a published tensor's memory had returned to the caching
allocator and come back as another tensor with the same registry key; the registry now holds every published tensor until its phase
ends.  DESIGN.md 4.2d.)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dtc_amd.algorithms import RecurrentPPO  # noqa: E402
from dtc_amd.modules import ActorCriticRecurrent  # noqa: E402

DEV = "cuda:0"


def run(overlap):
    torch.manual_seed(0)
    ac = ActorCriticRecurrent(53, 1389, 12, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128],
                              rnn_type="lstm", rnn_hidden_size=128, rnn_num_layers=1)
    alg = RecurrentPPO(ac, device=DEV, learning_rate=1e-3)
    alg.overlap = overlap
    alg.init_storage(32, 24, [53], [1389], [12])
    g = torch.Generator(device=DEV).manual_seed(1)
    for t in range(24):
        obs = torch.randn(32, 53, device=DEV, generator=g)
        cobs = torch.randn(32, 1389, device=DEV, generator=g)
        torch.manual_seed(100 + t)
        alg.act(obs, cobs)
        dones = (torch.rand(32, device=DEV, generator=g) < 0.05)
        alg.process_env_step(0.1 * torch.randn(32, device=DEV, generator=g), dones, {})
    alg.compute_returns(torch.randn(32, 1389, device=DEV, generator=g))
    alg.update()
    return {k: t.clone() for k, t in ac.state_dict().items()}


if __name__ == "__main__":
    outs = [run(False), run(False), run(True), run(True)]
    ok = True
    for i in range(1, 4):
        bad = [(k, float((outs[0][k] - outs[i][k]).abs().max())) for k in outs[0] if not torch.equal(outs[0][k], outs[i][k])]
        print(f"run 0 (serial) vs run {i} ({'serial' if i < 2 else 'overlapped'}): {len(bad)} tensors differ", bad[:3])
        ok = ok and not bad
    sys.exit(0 if ok else 1)
