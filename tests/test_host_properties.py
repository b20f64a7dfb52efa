"""Property tests (hypothesis) of the host-side logic: trajectory batching as index maps against the
oracle's pad_sequence restatement of rsl_rl/rsl_rl/utils/utils.py:33-70, and the mini-batch generator contract of
rollout_storage.py:162-214 (every sample exactly once per epoch, the same permutation for all epochs)."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from dtc_amd.storage import RolloutStorage
from dtc_amd.utils import split_and_pad_trajectories, trajectory_index_map, unpad_trajectories
from oracle import gru_ref as GR

CFG = dict(deadline=None, max_examples=40, suppress_health_check=list(HealthCheck), derandomize=True)


@settings(**CFG)
@given(T=st.integers(2, 30), N=st.integers(1, 40), D=st.integers(1, 7), p=st.sampled_from([0.0, 0.05, 0.3, 0.9]),
       seed=st.integers(0, 2 ** 16))
def test_split_pad_unpad_equal_the_reference_semantics(T, N, D, p, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(T, N, D, generator=g)
    dones = (torch.rand(T, N, 1, generator=g) < p).to(torch.uint8)
    dones[:, 0] = 0                                   # one env without resets: the reference pads to the longest trajectory
    want, want_masks = GR.split_and_pad(x, dones)
    got, masks = split_and_pad_trajectories(x, dones)
    assert torch.equal(masks, want_masks) and torch.equal(got, want)
    assert torch.equal(unpad_trajectories(got, masks), x) and torch.equal(GR.unpad(want, want_masks), x)
    traj_id, pos, lengths, n_traj = trajectory_index_map(dones)
    assert n_traj == want.shape[1] and int(lengths.sum()) == T * N and int(lengths.max()) == T
    assert torch.equal(lengths, want_masks.sum(0))


@pytest.mark.gpu                                       # the generator gathers through dtc_gather_rows: no CPU path
@settings(**{**CFG, "max_examples": 15})
@given(N=st.sampled_from([4, 8, 20]), T=st.integers(1, 6), nmb=st.sampled_from([1, 2, 4]), epochs=st.integers(1, 3),
       seed=st.integers(0, 2 ** 16))
def test_mini_batch_generator_visits_every_sample_once_per_epoch(N, T, nmb, epochs, seed):
    torch.manual_seed(seed)
    st_ = RolloutStorage(N, T, [53], [1389], [265], [12], "cuda:0")
    tag = torch.arange(T * N, dtype=torch.float32, device="cuda:0").view(T, N, 1)
    st_.observations[..., :1] = tag                    # sample id in the first observation column
    st_.rewards.copy_(tag)
    seen = []
    for batch in st_.mini_batch_generator(nmb, epochs):
        assert len(batch) == 16 and batch[13] == (None, None) and batch[14] is None
        ids = batch[0][:, 0].long()
        assert torch.equal(batch[15][:, 0].long(), ids)          # rew_buf_batch rides on the same indices
        seen.append(ids)
    assert len(seen) == nmb * epochs
    B = (T * N) // nmb
    for e in range(epochs):
        ep = torch.cat(seen[e * nmb:(e + 1) * nmb])
        assert ep.numel() == nmb * B and ep.unique().numel() == nmb * B
        assert torch.equal(ep, torch.cat(seen[:nmb]))            # one randperm per update, reused by all epochs (:165)


@pytest.mark.parametrize("kind", ["gru", "lstm"])
def test_recurrent_generator_on_index_maps_equals_the_reference_fixture_and_oracle(kind, golden):
    """RolloutStorage.reccurent_mini_batch_generator (host logic on index maps, CPU tensors here) against (a) the shapes /
    sums the reference's own generator produced (tests/golden/{gru,lstm}.npz) and (b) every tensor of the oracle's
    restatement of rollout_storage.py:217-267, incl. the (sic) actor states handed to the LSTM critic; `"own"` yields the
    critic's states instead."""
    import importlib
    import numpy as np
    from dtc_amd.storage import RolloutStorage
    from oracle import gru_ref as GR
    mod = importlib.import_module("tests.test_gru_path" if kind == "gru" else "tests.test_lstm_path")
    g = golden(kind)
    if kind == "gru":
        data, hid_a, hid_c = mod.gru_case()
        sa, sc = [hid_a], [hid_c]
    else:
        data, hid_a, hid_c = mod.lstm_case()
        sa, sc = list(hid_a), list(hid_c)
    n = data["observations"].shape[1]
    st = RolloutStorage(n, 24, [53], [1389], [1], [12], "cpu")
    for k, v in data.items():
        if k not in ("last_values", "observation_histories"):
            getattr(st, k).copy_(v)
    st.saved_hidden_states_a, st.saved_hidden_states_c = sa, sc
    ref = mod.oracle_storage(data) if kind == "gru" else mod.oracle_storage(data)
    want = list(GR.recurrent_batches(ref, hid_a, hid_c, 4))
    got = list(st.reccurent_mini_batch_generator(4, 2))
    assert len(got) == 8
    for i, (b, w) in enumerate(zip(got[:4], want)):
        obs, cobs, actions, values, adv, ret, logp, mu, sigma, (ha, hc), masks = b
        if f"mb{i}_shape" in g.files:
            shape = g[f"mb{i}_shape"]
            assert list(obs.shape) == list(shape[:3]) and list(masks.shape) == list(shape[3:5])
            if f"mb{i}_mask_sum" in g.files:
                assert int(masks.sum()) == int(g[f"mb{i}_mask_sum"][0])
                assert abs(obs.double().sum().item() - g[f"mb{i}_obs_sum"][0]) < 1e-6
        assert torch.equal(obs, w["obs"]) and torch.equal(cobs, w["cobs"]) and torch.equal(masks, w["masks"])
        assert torch.equal(actions, st.actions[:, w["sl"]]) and torch.equal(sigma, st.sigma[:, w["sl"]])
        flat = lambda h: list(h) if isinstance(h, (tuple, list)) else [h]
        for x, y in zip(flat(ha) + flat(hc), flat(w["hid_a"]) + flat(w["hid_c"])):
            assert torch.equal(x, y)
        for b2, b1 in zip(got[4 + i], b):                       # second epoch: the same batches
            for x, y in zip(flat(b2) if not torch.is_tensor(b2) else [b2], flat(b1) if not torch.is_tensor(b1) else [b1]):
                for u, v in zip(flat(x), flat(y)):
                    assert torch.equal(u, v)
    if kind == "lstm":
        st.lstm_critic_states = "own"
        b = next(iter(st.reccurent_mini_batch_generator(4, 1)))
        own = next(iter(GR.recurrent_batches(ref, hid_c, hid_c, 4)))["hid_a"]
        assert all(torch.equal(x, y) for x, y in zip(b[9][1], own)) and not torch.equal(b[9][1][0], b[9][0][0])
