"""GRU recurrence kernels (dtc_gru_fwd / dtc_gru_bwd + the input projection GEMMs) against torch.nn.GRU
on the CPU -- the module the reference's `Memory` wraps (actor_critic_recurrent.py:92-116).  GPU only."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("T,R,I,H", [(24, 13, 53, 512), (24, 300, 1389, 512), (1, 64, 53, 512), (5, 7, 20, 64)])
def test_gru_forward_backward_vs_torch(T, R, I, H):
    from dtc_amd import ops
    g = torch.Generator().manual_seed(T * 1000 + R)
    rnn = torch.nn.GRU(input_size=I, hidden_size=H, num_layers=1)
    with torch.no_grad():
        for p in rnn.parameters():
            p.copy_(torch.randn(p.shape, generator=g) / (p.shape[-1] ** 0.5 if p.dim() == 2 else 4.0))
    x = torch.randn(T, R, I, generator=g, requires_grad=True)
    h0 = (0.5 * torch.randn(1, R, H, generator=g)).requires_grad_(True)
    out, hT = rnn(x, h0)
    dout = torch.randn(T, R, H, generator=g)
    dout[T // 2:, : R // 3] = 0.0              # "padded" tail of some trajectories: no gradient there
    out.backward(dout)

    d = lambda t: t.detach().to(DEV).contiguous()
    W_ih, W_hh, b_ih, b_hh = d(rnn.weight_ih_l0), d(rnn.weight_hh_l0), d(rnn.bias_ih_l0), d(rnn.bias_hh_l0)
    xd = d(x).view(T * R, I)
    gi = torch.empty(T, R, 3 * H, device=DEV)
    ops.linear_fwd(xd, W_ih, b_ih, gi.view(T * R, 3 * H), None)
    hs_all = torch.empty(T + 1, R, H, device=DEV)
    gates, hn = torch.empty(T, R, 3 * H, device=DEV), torch.empty(T, R, H, device=DEV)
    ws = ops.workspace(ops.gru_workspace_bytes(T, R, H), DEV)
    ops.gru_fwd(gi, d(h0[0]), W_hh, b_hh, hs_all, gates, hn, ws)
    np.testing.assert_allclose(hs_all[1:].cpu().numpy(), out.detach().numpy(), rtol=2e-5, atol=2e-5)

    dgi = torch.empty(T, R, 3 * H, device=DEV)
    dW_hh, db_hh, dh0 = torch.empty(3 * H, H, device=DEV), torch.empty(3 * H, device=DEV), torch.empty(R, H, device=DEV)
    ops.gru_bwd(d(dout), hs_all, gates, hn, W_hh, dgi, dW_hh, db_hh, dh0, ws)
    dW_ih, db_ih = torch.empty(3 * H, I, device=DEV), torch.empty(3 * H, device=DEV)
    ws2 = ops.workspace(ops.wgrad_workspace_bytes(T * R, 3 * H, I), DEV)
    ops.linear_wgrad(dgi.view(T * R, 3 * H), xd, dW_ih, db_ih, ws2)
    dx = torch.empty(T * R, I, device=DEV)
    ops.linear_dgrad(dgi.view(T * R, 3 * H), W_ih, dx)

    def close(a, b, what):
        scale = float(b.abs().max()) + 1e-30
        err = float((a.cpu() - b).abs().max()) / scale
        assert err <= 5e-5, (what, err, scale)
    close(dW_hh, rnn.weight_hh_l0.grad, "dW_hh")
    close(db_hh, rnn.bias_hh_l0.grad, "db_hh")
    close(dW_ih, rnn.weight_ih_l0.grad, "dW_ih")
    close(db_ih, rnn.bias_ih_l0.grad, "db_ih")
    close(dh0, h0.grad[0], "dh0")
    close(dx.view(T, R, I), x.grad, "dx")


@pytest.mark.parametrize("R", [13, 1473])
def test_split_precision_step_kernels_vs_fp64(R):
    """csrc/gru_s3.hip through its own entry points: the fused forward step and the data-gradient chunks on the bf16 x 3 path
    against fp64, next to the single-pass fp32 kernels on the same inputs (same error level required)."""
    from dtc_amd import _ffi, ops
    lib = _ffi.lib()
    H = 512
    g = torch.Generator().manual_seed(R)
    W = (torch.randn(3 * H, H, generator=g) / H ** 0.5).to(DEV)
    b = torch.randn(3 * H, generator=g).to(DEV)
    hp = torch.randn(R, H, generator=g).to(DEV)
    gi = torch.randn(R, 3 * H, generator=g).to(DEV)
    img = torch.empty(int(lib.dtc_gru_s3_image_bytes(H)) // 8 + 1, dtype=torch.float64, device=DEV)
    outs = []
    for s3 in (False, True):
        h, gates, hn = (torch.full((R, H), float("nan"), device=DEV), torch.full((R, 3 * H), float("nan"), device=DEV),
                        torch.full((R, H), float("nan"), device=DEV))
        if s3:
            _ffi.check(lib.dtc_gru_s3_image(_ffi.cptr(W, torch.float32), _ffi.ptr(img), H, 0, _ffi.stream()), "image")
            _ffi.check(lib.dtc_gru_step_fwd_s3(_ffi.cptr(hp, torch.float32), _ffi.ptr(img), _ffi.cptr(b, torch.float32),
                                               _ffi.cptr(gi, torch.float32), _ffi.ptr(h), _ffi.ptr(gates), _ffi.ptr(hn), R, H,
                                               _ffi.stream()), "step")
        else:
            _ffi.check(lib.dtc_gru_step_fwd(_ffi.cptr(hp, torch.float32), _ffi.cptr(W, torch.float32), _ffi.cptr(b, torch.float32),
                                            _ffi.cptr(gi, torch.float32), _ffi.ptr(h), _ffi.ptr(gates), _ffi.ptr(hn), R, H,
                                            _ffi.stream()), "step")
        outs.append((h, gates, hn))
    gh = hp.double() @ W.double().T + b.double()
    r, z = torch.sigmoid(gi[:, :H].double() + gh[:, :H]), torch.sigmoid(gi[:, H:2 * H].double() + gh[:, H:2 * H])
    n = torch.tanh(gi[:, 2 * H:].double() + r * gh[:, 2 * H:])
    ref = ((1 - z) * n + z * hp.double(), torch.cat([r, z, n], 1), gh[:, 2 * H:])
    for k in range(3):
        e32 = float((outs[0][k].double() - ref[k]).abs().max())
        es3 = float((outs[1][k].double() - ref[k]).abs().max())
        print(f"gru step R={R} output {k}: fp32 MFMA err {e32:.2e}, split err {es3:.2e}")
        assert es3 <= 2.0 * e32 + 1e-6, (k, e32, es3)
    # data-gradient chunks: sum of the chunks == dgh W_hh
    dgh = torch.randn(R, 3 * H, generator=g).to(DEV)
    ref = dgh.double() @ W.double()
    _ffi.check(lib.dtc_gru_s3_image(_ffi.cptr(W, torch.float32), _ffi.ptr(img), H, 1, _ffi.stream()), "image")
    for nparts in (1, 3, 6):
        part = torch.full((nparts, R, H), float("nan"), device=DEV)
        _ffi.check(lib.dtc_gru_dgrad_parts_s3(_ffi.cptr(dgh, torch.float32), _ffi.ptr(img), _ffi.ptr(part), R * H, R, H, nparts,
                                              _ffi.stream()), "parts")
        p32 = torch.full((3, R, H), float("nan"), device=DEV)
        _ffi.check(lib.dtc_linear_dgrad_split(_ffi.cptr(dgh, torch.float32), 3 * H, _ffi.cptr(W, torch.float32), _ffi.ptr(p32), H, R * H,
                                              R, 3 * H, H, 3, _ffi.stream()), "split")
        es3 = float((part.double().sum(0) - ref).abs().max() / ref.abs().max())
        e32 = float((p32.double().sum(0) - ref).abs().max() / ref.abs().max())
        print(f"gru dgrad chunks R={R} nparts={nparts}: fp32 MFMA err {e32:.2e}, split err {es3:.2e}")
        assert es3 <= 2.0 * e32 + 2e-7, (nparts, e32, es3)


@pytest.mark.parametrize("T,R,H", [(24, 1473, 512), (6, 77, 128), (2, 40, 128), (5, 7, 64)])
def test_two_recurrences_in_one_launch_per_time_step_equal_the_single_calls_bit_for_bit(T, R, H):
    from dtc_amd import _ffi
    _ffi.lib().dtc_set_gru_seq(0)            # (the per-step kernels: the persistent launches have tests of their own below)
    try:
        _multi_equals_single(T, R, H)
    finally:
        _ffi.lib().dtc_set_gru_seq(-1)


def _multi_equals_single(T, R, H):
    """dtc_gru_fwd_multi / dtc_gru_bwd_multi (the actor's and the critic's Memory advanced together, actor_critic_recurrent.py:45-46):
    ONE launch per time step for both recurrences -- the same kernels on the same tiles, so every output equals the single calls' bit
    for bit (incl. dgh_all in the workspace, which the trainers' W_hh weight gradient reads).  (2, 40, 128) and (5, 7, 64) take the
    documented fall-back: the single calls one after the other."""
    from dtc_amd import _ffi, ops
    g = torch.Generator(device=DEV).manual_seed(T * 100 + R)
    rn = lambda *s: torch.randn(*s, device=DEV, generator=g)

    def make():
        return dict(gi=rn(T, R, 3 * H), h0=0.5 * rn(R, H), W=rn(3 * H, H) / H ** 0.5, b=0.2 * rn(3 * H), dhs=0.01 * rn(T, R, H))

    def outs():
        d = dict(hs=torch.empty(T + 1, R, H, device=DEV), gates=torch.empty(T, R, 3 * H, device=DEV), hn=torch.empty(T, R, H, device=DEV),
                 dgi=torch.empty(T, R, 3 * H, device=DEV), dh0=torch.empty(R, H, device=DEV))
        d["ws"] = ops.workspace(ops.gru_workspace_bytes(T, R, H), DEV)
        d["ws"].zero_()
        return d

    ins = [make(), make()]
    single, multi = [outs(), outs()], [outs(), outs()]
    for x, o in zip(ins, single):
        ops.gru_fwd(x["gi"], x["h0"], x["W"], x["b"], o["hs"], o["gates"], o["hn"], o["ws"])
        ops.gru_bwd(x["dhs"], o["hs"], o["gates"], o["hn"], x["W"], o["dgi"], None, None, o["dh0"], o["ws"])
    ops.gru_fwd_multi([(x["gi"], x["h0"], x["W"], x["b"], o["hs"], o["gates"], o["hn"], o["ws"]) for x, o in zip(ins, multi)])
    ops.gru_bwd_multi([(x["dhs"], o["hs"], o["gates"], o["hn"], x["W"], o["dgi"], o["dh0"], o["ws"]) for x, o in zip(ins, multi)])
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(single, multi)):
        for k in ("hs", "gates", "hn", "dgi", "dh0"):
            assert torch.equal(a[k], b[k]), (i, k)
        assert torch.equal(ops.gru_dgh_all(a["ws"], T, R, H), ops.gru_dgh_all(b["ws"], T, R, H)), (i, "dgh_all")
    assert not torch.equal(single[0]["hs"], single[1]["hs"])      # (the two recurrences are different problems)


def _gru_ref64(gi, h0, W, b):
    """torch.nn.GRU's recurrence in fp64 on the device (gate order r, z, n)."""
    T, R, H3 = gi.shape
    H = H3 // 3
    h = h0.double()
    hs, gates, hn = [h], [], []
    Wd, bd = W.double(), b.double()
    for t in range(T):
        gh = h @ Wd.T + bd
        g = gi[t].double()
        r, z = torch.sigmoid(g[:, :H] + gh[:, :H]), torch.sigmoid(g[:, H:2 * H] + gh[:, H:2 * H])
        n = torch.tanh(g[:, 2 * H:] + r * gh[:, 2 * H:])
        h = (1 - z) * n + z * h
        hs.append(h)
        gates.append(torch.cat([r, z, n], 1))
        hn.append(gh[:, 2 * H:])
    return torch.stack(hs), torch.stack(gates), torch.stack(hn)


@pytest.mark.parametrize("T,R", [(24, 1473), (24, 1600), (24, 2048), (3, 33), (2, 1), (24, 300)])
def test_persistent_recurrence_vs_fp64_and_the_per_step_launches(T, R):
    """csrc/gru_seq.hip (the forward recurrence as ONE persistent launch: W_hh slices resident in LDS as two-term fp16, row blocks meeting
    at counter barriers) against an fp64 recurrence, next to the per-step launches on the same inputs: same error level required (the
    two split the operands differently: no bit equality), every slot of every output written, status flag clear."""
    from dtc_amd import _ffi, ops
    lib = _ffi.lib()
    H = 512
    g = torch.Generator(device=DEV).manual_seed(T * 7 + R)
    rn = lambda *s: torch.randn(*s, device=DEV, generator=g)
    gi, h0, W, b = rn(T, R, 3 * H), 0.5 * rn(R, H).clamp(-1.9, 1.9), rn(3 * H, H) / H ** 0.5, 0.2 * rn(3 * H)
    ref = _gru_ref64(gi, h0, W, b)
    res = {}
    for seq in (1, 0):
        lib.dtc_set_gru_seq(seq)
        try:
            assert lib.dtc_gru_seq_supported(T, R, H, 1) == seq
            # two recurrences through dtc_gru_fwd_multi (ONE persistent launch that may take every CU: up to 2048 rows); the second one is
            # the same problem with its rows reversed -- both must come out right
            o = [[torch.full((T + 1, R, H), float("nan"), device=DEV), torch.full((T, R, 3 * H), float("nan"), device=DEV),
                  torch.full((T, R, H), float("nan"), device=DEV), ops.workspace(ops.gru_workspace_bytes(T, R, H), DEV)] for _ in range(2)]
            gi2, h02 = gi.flip(1).contiguous(), h0.flip(0).contiguous()
            ops.gru_fwd_multi([(gi, h0, W, b, *o[0]), (gi2, h02, W, b, *o[1])])
            torch.cuda.synchronize()
        finally:
            lib.dtc_set_gru_seq(-1)
        hs, gates, hn = o[0][:3]
        res[seq] = [float((a.double() - r).abs().max()) for a, r in zip((hs, gates, hn), ref)]
        assert all(np.isfinite(e) for e in res[seq]), (seq, res[seq])            # (a NaN left = a slot nobody wrote)
        for a, b2 in zip(o[0][:3], o[1][:3]):
            assert torch.equal(a, b2.flip(1)), seq                               # rows are independent: the reversed problem, reversed
    assert lib.dtc_gru_seq_status(1) == 0
    print(f"gru recurrence T={T} R={R}: max abs error vs fp64 (h, gates, gh_n): persistent {res[1]}, per-step {res[0]}")
    for k in range(3):
        assert res[1][k] <= 2.0 * res[0][k] + 2e-6, (k, res)


def test_two_persistent_recurrences_on_two_streams_meet():
    """Two single-recurrence persistent launches (dtc_gru_fwd) on two streams at the largest size they serve (1536 rows: 4 row blocks of
    384 rows x 32 unit tiles = 128 workgroups of one per CU each) -- both fit the chip at once, neither starves the other's barriers.  Results equal the
    launches run one after the other bit for bit; the status flag stays clear."""
    from dtc_amd import _ffi, ops
    lib = _ffi.lib()
    T, R, H = 24, 1536, 512
    lib.dtc_set_gru_seq(1)
    try:
        assert lib.dtc_gru_seq_supported(T, R, H, 0) == 1 and lib.dtc_gru_seq_supported(T, R + 1, H, 0) == 0
        _two_streams(lib, ops, T, R, H)
    finally:
        lib.dtc_set_gru_seq(-1)


def _two_streams(lib, ops, T, R, H):
    g = torch.Generator(device=DEV).manual_seed(11)
    rn = lambda *s: torch.randn(*s, device=DEV, generator=g)
    ins = [(rn(T, R, 3 * H), 0.5 * rn(R, H), rn(3 * H, H) / H ** 0.5, 0.2 * rn(3 * H)) for _ in range(2)]

    def run(streams):
        outs = []
        for (gi, h0, W, b), st in zip(ins, streams):
            hs = torch.empty(T + 1, R, H, device=DEV)
            gates, hn = torch.empty(T, R, 3 * H, device=DEV), torch.empty(T, R, H, device=DEV)
            ws = ops.workspace(ops.gru_workspace_bytes(T, R, H), DEV)
            torch.cuda.synchronize()
            with torch.cuda.stream(st):
                for _ in range(3):                                   # a few back-to-back launches per stream: they overlap with the other stream's
                    ops.gru_fwd(gi, h0, W, b, hs, gates, hn, ws)
            outs.append((hs, gates, hn, ws))
        torch.cuda.synchronize()
        return outs
    cur = torch.cuda.current_stream()
    serial = run([cur, cur])
    both = run([torch.cuda.Stream(), torch.cuda.Stream()])
    assert lib.dtc_gru_seq_status(1) == 0
    for a, b in zip(serial, both):
        for k in range(3):
            assert torch.equal(a[k], b[k]), k
