"""TorchScript exporters (legged_gym/utils/helpers.py:150-189, actor_critic_decoder.py:616-666).
CPU: every exported file loads with torch.jit.load and reproduces the torch modules it was scripted from.
GPU: the exported (CPU) module equals `act_inference` / `act_expert` of the HIP path on a fixture batch to 1e-5."""
import os

import numpy as np
import pytest
import torch

from dtc_amd.utils.export import export_policy_as_jit

DEV = "cuda:0"


def _mlp_model():
    from dtc_amd.modules import ActorCritic
    torch.manual_seed(5)
    return ActorCritic(48, 48, 12, actor_hidden_dims=[128, 64], critic_hidden_dims=[128, 64])


def _rec_model(kind, layers=1):
    from dtc_amd.modules import ActorCriticRecurrent
    torch.manual_seed(6)
    return ActorCriticRecurrent(53, 60, 12, actor_hidden_dims=[128, 64], critic_hidden_dims=[128, 64], activation="elu",
                                rnn_type=kind, rnn_hidden_size=64, rnn_num_layers=layers)


def _dec_model():
    from dtc_amd.modules import ActorCriticDecoder
    torch.manual_seed(7)
    return ActorCriticDecoder(53, 1389, 12)


def test_exported_files_reproduce_their_torch_modules(tmp_path):
    g = torch.Generator().manual_seed(1)
    ac = _mlp_model()
    (f,) = export_policy_as_jit(ac, str(tmp_path / "mlp"))
    assert os.path.basename(f) == "policy_1.pt"
    x = torch.randn(7, 48, generator=g)
    assert torch.equal(torch.jit.load(f)(x), ac.actor(x))
    for kind, layers in (("lstm", 2), ("gru", 1)):
        ac = _rec_model(kind, layers)
        (f,) = export_policy_as_jit(ac, str(tmp_path / kind))
        assert os.path.basename(f) == "policy_lstm_1.pt"
        m = torch.jit.load(f)
        hidden = None
        for t in range(3):
            x = torch.randn(1, 53, generator=g)
            out, hidden = ac.memory_a.rnn(x.unsqueeze(0), hidden)
            np.testing.assert_allclose(m(x).numpy(), ac.actor(out.squeeze(0)).detach().numpy(), rtol=1e-6, atol=1e-6)
        m.reset_memory()
        out, _ = ac.memory_a.rnn(x.unsqueeze(0), None)
        np.testing.assert_allclose(m(x).numpy(), ac.actor(out.squeeze(0)).detach().numpy(), rtol=1e-6, atol=1e-6)
    ac = _dec_model()
    pol, enc = export_policy_as_jit(ac, str(tmp_path / "dec"))
    assert (os.path.basename(pol), os.path.basename(enc)) == ("policy_decoder_1.pt", "terrain_encoder_1.pt")
    obs, hist, heights = torch.randn(5, 53, generator=g), torch.randn(5, 265, generator=g), torch.randn(5, 693, generator=g)
    vae = ac.vae
    with torch.no_grad():                                   # actor_critic_decoder.py:504-538 on the module's own torch layers
        latent = vae.latent_mu(vae.cenet_encoder(hist))
        l_t = vae.terrain_encoder(heights)
        b1 = vae.memory_mlp(torch.cat((hist, l_t), -1))
        want = ac.actor_body(torch.cat((obs, latent[:, 3:], latent[:, :3], b1 + l_t * b1), -1))
    got = torch.jit.load(pol)(torch.cat((obs, hist), -1), torch.jit.load(enc)(heights))
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
def test_exported_policies_equal_the_hip_inference_path(tmp_path):
    g = torch.Generator().manual_seed(2)
    # ActorCritic: act_inference
    ac = _mlp_model().to(DEV)
    x = torch.randn(64, 48, generator=g)
    want = ac.act_inference(x.to(DEV)).cpu()
    (f,) = export_policy_as_jit(ac, str(tmp_path / "mlp"))
    np.testing.assert_allclose(torch.jit.load(f)(x).numpy(), want.numpy(), rtol=1e-5, atol=1e-5)
    # ActorCriticRecurrent (GRU, BASELINE config 3's recurrence; LSTM, the reference's default): three rollout steps of ONE env
    for kind, layers in (("gru", 1), ("lstm", 2)):
        ac = _rec_model(kind, layers).to(DEV)
        (f,) = export_policy_as_jit(ac, str(tmp_path / kind))
        m = torch.jit.load(f)
        for t in range(3):
            x = torch.randn(1, 53, generator=g)
            want = ac.act_inference(x.to(DEV)).cpu()
            np.testing.assert_allclose(m(x).numpy(), want.numpy(), rtol=1e-5, atol=1e-5, err_msg=f"{kind} step {t}")
    # ActorCriticDecoder: act_expert (the policy get_inference_policy(env_t=True) hands out)
    ac = _dec_model().to(DEV)
    obs, hist, priv = torch.randn(128, 53, generator=g), torch.randn(128, 265, generator=g), torch.randn(128, 1389, generator=g)
    want = ac.act_expert(dict(obs=obs.to(DEV), obs_history=hist.to(DEV), privileged_obs=priv.to(DEV))).cpu()
    pol, enc = export_policy_as_jit(ac, str(tmp_path / "dec"))
    got = torch.jit.load(pol)(torch.cat((obs, hist), -1), torch.jit.load(enc)(priv[:, :693]))
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-5, atol=1e-5)
