"""TorchScript exporters of the trained policy (legged_gym/utils/helpers.py:150-189 and
rsl_rl/rsl_rl/modules/actor_critic_decoder.py:616-666).

The exported modules are plain torch CPU modules built from COPIES of the trained weights (the deployment target is the
robot's CPU, not the trainer's GPU); nothing here launches a kernel.  Parity: tests/test_export.py loads every exported file
and compares it with `act_inference` / `act_expert` of the HIP path on a fixture batch (1e-5).

    export_policy_as_jit(actor_critic, path)
        ActorCritic                      -> <path>/policy_1.pt        scripted copy of `actor`           (helpers.py:155-160)
        ActorCriticRecurrent (LSTM/GRU)  -> <path>/policy_lstm_1.pt   PolicyExporterLSTM / PolicyExporterGRU (helpers.py:163-189;
                                            the reference's exporter "assumes LSTM: TODO add GRU" -- the GRU of BASELINE
                                            config 3 gets the same module with one state buffer)
        ActorCriticDecoder               -> <path>/policy_decoder_1.pt + <path>/terrain_encoder_1.pt: the deployment pair of
                                            actor_critic_decoder.py:616-666 (`PolicyExporter_export`, `TerrainEncoder`), whose
                                            hard-coded output paths become `path`
"""
from __future__ import annotations

import copy
import os

import torch
import torch.nn as nn


def _cpu_copy(module: nn.Module) -> nn.Module:
    """Deep copy on the CPU with parameters of its own (the trainer's parameters are views of one flat arena)."""
    m = copy.deepcopy(module).to("cpu")
    for p in m.parameters():
        p.data = p.data.clone()
        p.requires_grad_(False)
    return m.eval()


class PolicyExporterLSTM(nn.Module):
    """helpers.py:163-189: actor MLP + LSTM memory with the (h, c) state kept in buffers of batch size 1."""

    def __init__(self, actor_critic):
        super().__init__()
        self.actor = _cpu_copy(actor_critic.actor)
        self.is_recurrent = actor_critic.is_recurrent
        self.memory = _cpu_copy(actor_critic.memory_a.rnn)
        self.register_buffer("hidden_state", torch.zeros(self.memory.num_layers, 1, self.memory.hidden_size))
        self.register_buffer("cell_state", torch.zeros(self.memory.num_layers, 1, self.memory.hidden_size))

    def forward(self, x):
        out, (h, c) = self.memory(x.unsqueeze(0), (self.hidden_state, self.cell_state))
        self.hidden_state[:] = h
        self.cell_state[:] = c
        return self.actor(out.squeeze(0))

    @torch.jit.export
    def reset_memory(self):
        self.hidden_state[:] = 0.
        self.cell_state[:] = 0.

    def export(self, path):
        os.makedirs(path, exist_ok=True)
        path = os.path.join(path, "policy_lstm_1.pt")
        self.to("cpu")
        torch.jit.script(self).save(path)
        return path


class PolicyExporterGRU(nn.Module):
    """The same exporter for `rnn_type='gru'` (one state buffer); file name kept so that deployment scripts find it."""

    def __init__(self, actor_critic):
        super().__init__()
        self.actor = _cpu_copy(actor_critic.actor)
        self.is_recurrent = actor_critic.is_recurrent
        self.memory = _cpu_copy(actor_critic.memory_a.rnn)
        self.register_buffer("hidden_state", torch.zeros(self.memory.num_layers, 1, self.memory.hidden_size))

    def forward(self, x):
        out, h = self.memory(x.unsqueeze(0), self.hidden_state)
        self.hidden_state[:] = h
        return self.actor(out.squeeze(0))

    @torch.jit.export
    def reset_memory(self):
        self.hidden_state[:] = 0.

    def export(self, path):
        os.makedirs(path, exist_ok=True)
        path = os.path.join(path, "policy_lstm_1.pt")
        self.to("cpu")
        torch.jit.script(self).save(path)
        return path


class TerrainEncoder(nn.Module):
    """actor_critic_decoder.py:616-633: the terrain encoder alone (it runs at the height-map rate on the robot)."""

    def __init__(self, terrain_encoder):
        super().__init__()
        self.terrain_encoder = _cpu_copy(terrain_encoder)

    def forward(self, observations):
        return self.terrain_encoder(observations)

    def export(self, path):
        os.makedirs(path, exist_ok=True)
        path = os.path.join(path, "terrain_encoder_1.pt")
        torch.jit.script(self.eval()).save(path)
        return path


class PolicyExporterDecoder(nn.Module):
    """actor_critic_decoder.py:636-666 (`PolicyExporter_export`): the deployment policy of `act_teacher` -- CE-net encoder ->
    latent_mu, belief b_t = m + l_t * m with m = memory_mlp(cat[hist, l_t]), actor body -- as ONE module taking
    `observations_total = cat[obs, obs_history]` and the terrain latent.  `num_obs` (45 in the reference's robot build, 53 for
    the lite3 task of this repository's config) is the split point."""

    def __init__(self, cenet_encoder, actor_body, latent_mu, memory_mlp, num_obs: int):
        super().__init__()
        self.cenet_encoder = _cpu_copy(cenet_encoder)
        self.actor_body = _cpu_copy(actor_body)
        self.latent_mu = _cpu_copy(latent_mu)
        self.memory_mlp = _cpu_copy(memory_mlp)
        self.num_obs = int(num_obs)

    def forward(self, observations_total, latent_lidar):
        n = self.num_obs
        latent_e = self.cenet_encoder(observations_total[:, n:])
        latent = self.latent_mu(latent_e)
        b_t1 = self.memory_mlp(torch.cat((observations_total[:, n:], latent_lidar), dim=-1))
        b_t = b_t1 + torch.mul(latent_lidar, b_t1)
        return self.actor_body(torch.cat((observations_total[:, 0:n], latent[:, 3:], latent[:, :3], b_t), dim=-1))

    def export(self, path):
        os.makedirs(path, exist_ok=True)
        path = os.path.join(path, "policy_decoder_1.pt")
        torch.jit.script(self.eval()).save(path)
        return path


def export_policy_as_jit(actor_critic, path):
    """helpers.py:150-160.  Returns the list of files written."""
    if hasattr(actor_critic, "memory_a") and hasattr(actor_critic, "actor"):
        kind = type(actor_critic.memory_a.rnn).__name__
        exporter = PolicyExporterLSTM(actor_critic) if kind == "LSTM" else PolicyExporterGRU(actor_critic)
        return [exporter.export(path)]
    if hasattr(actor_critic, "actor"):
        os.makedirs(path, exist_ok=True)
        path = os.path.join(path, "policy_1.pt")
        torch.jit.script(_cpu_copy(actor_critic.actor)).save(path)
        return [path]
    if hasattr(actor_critic, "vae") and hasattr(actor_critic, "actor_body"):
        vae = actor_critic.vae
        pol = PolicyExporterDecoder(vae.cenet_encoder, actor_critic.actor_body, vae.latent_mu, vae.memory_mlp, actor_critic.num_obs)
        return [pol.export(path), TerrainEncoder(vae.terrain_encoder).export(path)]
    raise TypeError(f"export_policy_as_jit: no exporter for {type(actor_critic).__name__}")
