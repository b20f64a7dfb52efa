"""Where a time step of the persistent GRU forward (csrc/gru_seq.hip) spends its time: per-workgroup time stamps of one launch at
BASELINE's size (T = 24, R = 1500, H = 512), alone and beside a second recurrence on another stream."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dtc_amd import _ffi, ops  # noqa: E402

DEV = "cuda:0"
T, R, H = 24, int(os.environ.get("TRACE_R", "1500")), 512
lib = _ffi.lib()
lib.dtc_set_gru_seq(1)
g = torch.Generator(device=DEV).manual_seed(11)
rn = lambda *s: torch.randn(*s, device=DEV, generator=g)  # noqa: E731
ins = [(rn(T, R, 3 * H), 0.5 * rn(R, H), rn(3 * H, H) / H ** 0.5, 0.2 * rn(3 * H)) for _ in range(2)]
outs = [(torch.empty(T + 1, R, H, device=DEV), torch.empty(T, R, 3 * H, device=DEV), torch.empty(T, R, H, device=DEV),
         ops.workspace(ops.gru_workspace_bytes(T, R, H), DEV)) for _ in range(2)]
nwg = 8 * 32


def launch(i):
    (gi, h0, W, b), (hs, gates, hn, ws) = ins[i], outs[i]
    ops.gru_fwd(gi, h0, W, b, hs, gates, hn, ws)


def launch_pair():
    ops.gru_fwd_multi([(i_[0], i_[1], i_[2], i_[3], o_[0], o_[1], o_[2], o_[3]) for i_, o_ in zip(ins, outs)])


def two_streams():
    s2 = torch.cuda.Stream()
    s2.wait_stream(torch.cuda.current_stream())
    launch(0)
    with torch.cuda.stream(s2):
        launch(1)
    torch.cuda.current_stream().wait_stream(s2)


for _ in range(3):
    two_streams()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    two_streams()
e1.record()
torch.cuda.synchronize()
print(f"two single-recurrence launches on two streams (R = {R}): {e0.elapsed_time(e1) / 5 * 1e3:.0f} us per pair = {e0.elapsed_time(e1) / 5 / T * 1e3:.1f} us per step-pair")
lib.dtc_set_gru_seq(0)
for _ in range(3):
    two_streams()
torch.cuda.synchronize()
e0.record()
for _ in range(5):
    two_streams()
e1.record()
torch.cuda.synchronize()
print(f"the same with the per-step launches: {e0.elapsed_time(e1) / 5 * 1e3:.0f} us per pair = {e0.elapsed_time(e1) / 5 / T * 1e3:.1f} us per step-pair")
lib.dtc_set_gru_seq(1)

for pair in (False, True):
    for _ in range(3):
        launch_pair() if pair else launch(0)
    torch.cuda.synchronize()
    trace = torch.zeros(nwg, T, 4, dtype=torch.int64, device=DEV)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    lib.dtc_gru_seq_trace(trace.data_ptr())
    e0.record()
    launch_pair() if pair else launch(0)      # (pair: the trace holds the SECOND recurrence's stamps, written last)
    lib.dtc_gru_seq_trace(None)
    e1.record()
    torch.cuda.synchronize()
    if not pair:
        trace = trace[:4 * 32]
    tr = trace.cpu().double() / 100.0          # us
    met, kdone, gdone, arrived = tr[..., 0], tr[..., 1], tr[..., 2], tr[..., 3]
    step = (met[:, 1:] - met[:, :-1]).mean()
    print(f"{'both recurrences in one launch (256 workgroups)' if pair else 'one recurrence, 128 workgroups (R = ' + str(R) + ')'}: launch {e0.elapsed_time(e1) * 1e3:.0f} us; per step {step:.1f} us = K loop "
          f"{(kdone - met).mean():.1f} + gates/stores {(gdone - kdone).mean():.1f} + drain+arrive {(arrived - gdone)[:, :-1].mean():.1f} + "
          f"wait for the block {(met[:, 1:] - arrived[:, :-1]).mean():.1f}; prologue {(met[:, 0] - met[:, 0].min()).mean():.1f} us skew; "
          f"slowest / fastest workgroup K loop {(kdone - met).mean(dim=1).max():.1f} / {(kdone - met).mean(dim=1).min():.1f}")
