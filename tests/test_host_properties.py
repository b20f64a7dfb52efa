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
