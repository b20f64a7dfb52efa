"""LSTM recurrence kernels (dtc_lstm_fwd / dtc_lstm_bwd + the input projection GEMMs) against torch.nn.LSTM on the CPU -- the
module the reference's `Memory` wraps by default (actor_critic_recurrent.py:93-97).  GPU only."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("T,R,I,H", [(24, 13, 53, 256), (24, 300, 1389, 512), (1, 64, 53, 256), (5, 7, 20, 64)])
def test_lstm_forward_backward_vs_torch(T, R, I, H):
    from dtc_amd import ops
    g = torch.Generator().manual_seed(T * 1000 + R)
    rnn = torch.nn.LSTM(input_size=I, hidden_size=H, num_layers=1)
    with torch.no_grad():
        for p in rnn.parameters():
            p.copy_(torch.randn(p.shape, generator=g) / (p.shape[-1] ** 0.5 if p.dim() == 2 else 4.0))
    x = torch.randn(T, R, I, generator=g, requires_grad=True)
    h0 = (0.5 * torch.randn(1, R, H, generator=g)).requires_grad_(True)
    c0 = (0.5 * torch.randn(1, R, H, generator=g)).requires_grad_(True)
    out, (hT, cT) = rnn(x, (h0, c0))
    dout = torch.randn(T, R, H, generator=g)
    dout[T // 2:, : R // 3] = 0.0              # "padded" tail of some trajectories: no gradient there
    out.backward(dout)

    d = lambda t: t.detach().to(DEV).contiguous()
    W_ih, W_hh, b_ih, b_hh = d(rnn.weight_ih_l0), d(rnn.weight_hh_l0), d(rnn.bias_ih_l0), d(rnn.bias_hh_l0)
    xd = d(x).view(T * R, I)
    gi = torch.empty(T, R, 4 * H, device=DEV)
    ops.linear_fwd(xd, W_ih, b_ih, gi.view(T * R, 4 * H), None)
    hs_all, cs_all = torch.empty(T + 1, R, H, device=DEV), torch.empty(T + 1, R, H, device=DEV)
    gates = torch.empty(T, R, 4 * H, device=DEV)
    ws = ops.workspace(ops.lstm_workspace_bytes(T, R, H), DEV)
    ops.lstm_fwd(gi, d(h0[0]), d(c0[0]), W_hh, b_hh, hs_all, cs_all, gates, ws)
    np.testing.assert_allclose(hs_all[1:].cpu().numpy(), out.detach().numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(cs_all[-1].cpu().numpy(), cT[0].detach().numpy(), rtol=2e-5, atol=2e-5)

    dgi = torch.empty(T, R, 4 * H, device=DEV)
    dW_hh, db_hh = torch.empty(4 * H, H, device=DEV), torch.empty(4 * H, device=DEV)
    dh0, dc0 = torch.empty(R, H, device=DEV), torch.empty(R, H, device=DEV)
    ops.lstm_bwd(d(dout), hs_all, cs_all, gates, W_hh, dgi, dW_hh, db_hh, dh0, dc0, ws)
    dW_ih, db_ih = torch.empty(4 * H, I, device=DEV), torch.empty(4 * H, device=DEV)
    ws2 = ops.workspace(ops.wgrad_workspace_bytes(T * R, 4 * H, I), DEV)
    ops.linear_wgrad(dgi.view(T * R, 4 * H), xd, dW_ih, db_ih, ws2)
    dx = torch.empty(T * R, I, device=DEV)
    ops.linear_dgrad(dgi.view(T * R, 4 * H), W_ih, dx)

    def close(a, b, what):
        scale = float(b.abs().max()) + 1e-30
        err = float((a.cpu() - b).abs().max()) / scale
        assert err <= 5e-5, (what, err, scale)
    close(dW_hh, rnn.weight_hh_l0.grad, "dW_hh")
    close(db_hh, rnn.bias_hh_l0.grad, "db_hh")
    close(dW_ih, rnn.weight_ih_l0.grad, "dW_ih")
    close(db_ih, rnn.bias_ih_l0.grad, "db_ih")
    close(dh0, h0.grad[0], "dh0")
    close(dc0, c0.grad[0], "dc0")
    close(dx.view(T, R, I), x.grad, "dx")


def test_lstm_entry_points_reject_bad_arguments():
    from dtc_amd import _ffi
    lib = _ffi.lib()
    assert lib.dtc_lstm_workspace(0, 4, 8) == 0 and lib.dtc_lstm_workspace(3, 4, 8) > 0
    assert lib.dtc_lstm_fwd(None, None, None, None, None, None, None, None, None, 2, 4, 8, None) != 0
    assert b"null" in lib.dtc_last_error()
    assert lib.dtc_lstm_bwd(None, None, None, None, None, None, None, None, None, None, None, 0, 4, 8, None) != 0
