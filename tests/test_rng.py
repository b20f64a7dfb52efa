"""Device random draws of the update (csrc/rng.hip) against their numpy restatement (oracle/rng.py) and against what a
permutation / a standard normal sample must satisfy."""
import numpy as np
import pytest
import torch

from oracle import rng as OR

DEV = "cuda:0"


def test_oracle_philox_matches_the_published_known_answer_vectors():
    """Random123's kat_vectors for philox4x32-10 (Salmon et al.): counter / key all zeros, all ones, and the pi digits."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = OR.philox4x32_10(np.array(ctr, dtype=np.uint64), key[0], key[1])
        assert tuple(int(v) for v in got) == want


def test_oracle_randperm_is_a_permutation():
    for n in (1, 2, 3, 17, 1000, 1536, 98304):
        p = OR.randperm(n, 1234567)
        assert np.array_equal(np.sort(p), np.arange(n))


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 2, 3, 5, 1000, 1536, 4097, 98304, 786432])
def test_randperm_bit_exact_and_a_permutation(n):
    from dtc_amd import ops
    for seed in (0, 1234567, (1 << 62) - 3):
        p = ops.randperm(n, DEV, seed).cpu().numpy()
        assert np.array_equal(np.sort(p), np.arange(n))
        assert np.array_equal(p, OR.randperm(n, seed))
    if n >= 1000:
        a, b = ops.randperm(n, DEV, 1).cpu().numpy(), ops.randperm(n, DEV, 2).cpu().numpy()
        assert (a == b).mean() < 0.01 and (a == np.arange(n)).mean() < 0.01
        # no position bias: the mean displacement of a uniform permutation is n/3
        assert abs(np.abs(a - np.arange(n)).mean() / n - 1.0 / 3.0) < 0.03


@pytest.mark.gpu
def test_randn_matches_the_oracle_and_is_standard_normal():
    from dtc_amd import ops
    for n, seed, off in ((7, 5, 0), (4096, 99, 3), (20 * 24576 * 16, (1 << 61) + 17, 0)):
        x = ops.randn((n,), DEV, seed, off).cpu().numpy()
        ref = OR.randn(n, seed, off)
        assert np.isfinite(x).all()
        np.testing.assert_allclose(x, ref, rtol=2e-6, atol=2e-6)
    x = x.astype(np.float64)
    assert abs(x.mean()) < 2e-3 and abs(x.var() - 1.0) < 2e-3 and abs((x ** 4).mean() - 3.0) < 2e-2
    assert abs(np.corrcoef(x[:-1], x[1:])[0, 1]) < 2e-3
    # another key / another offset: other draws
    y = ops.randn((4096,), DEV, 100, 3).cpu().numpy()
    assert np.abs(y - OR.randn(4096, 99, 3)).max() > 1.0


@pytest.mark.gpu
def test_update_draws_come_from_the_device_generator_and_follow_torch_manual_seed():
    import sys
    from tests.test_hip_ppo import _pair
    outs = []
    for seed in (7, 7, 8):
        ref, alg = _pair(64)
        torch.manual_seed(seed)
        outs.append((alg.update(), alg.actor_critic.arena.flat.clone()))
    assert torch.equal(outs[0][1], outs[1][1]) and not torch.equal(outs[0][1], outs[2][1])
