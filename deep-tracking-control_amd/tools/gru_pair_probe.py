"""Do two GRU recurrences on two streams overlap?  gru_pair_probe.py: one recurrence alone, two back to back on one stream, two on two
(high-priority) streams -- forward and backward, T = 24, R = 1470, H = 512 (the recurrent workloads' mini-batch)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dtc_amd import ops  # noqa: E402

DEV = "cuda:0"
T, R, H = 24, 1470, 512
g = torch.Generator(device=DEV).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=DEV, generator=g)


def make():
    d = dict(gi=rn(T, R, 3 * H), h0=rn(R, H) * 0.1, W=rn(3 * H, H) / H ** 0.5, b=rn(3 * H) * 0.1, hs=torch.empty(T + 1, R, H, device=DEV),
             gates=torch.empty(T, R, 3 * H, device=DEV), hn=torch.empty(T, R, H, device=DEV), dhs=rn(T, R, H) * 0.01,
             dgi=torch.empty(T, R, 3 * H, device=DEV), dh0=torch.empty(R, H, device=DEV))
    d["ws"] = ops.workspace(ops.gru_workspace_bytes(T, R, H), DEV)
    return d


def fwd(d):
    ops.gru_fwd(d["gi"], d["h0"], d["W"], d["b"], d["hs"], d["gates"], d["hn"], d["ws"])


def bwd(d):
    ops.gru_bwd(d["dhs"], d["hs"], d["gates"], d["hn"], d["W"], d["dgi"], None, None, d["dh0"], d["ws"])


a, b = make(), make()
s1, s2 = torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=-1)


def timed(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def two_streams(f):
    def run():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur)
        s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            f(a)
        with torch.cuda.stream(s2):
            f(b)
        cur.wait_stream(s1)
        cur.wait_stream(s2)
    return run


for name, f in (("fwd", fwd), ("bwd", bwd)):
    one = timed(lambda: f(a))
    seq = timed(lambda: (f(a), f(b)))
    par = timed(two_streams(f))
    if name == "fwd":
        mul = timed(lambda: ops.gru_fwd_multi([(d["gi"], d["h0"], d["W"], d["b"], d["hs"], d["gates"], d["hn"], d["ws"]) for d in (a, b)]))
    else:
        mul = timed(lambda: ops.gru_bwd_multi([(d["dhs"], d["hs"], d["gates"], d["hn"], d["W"], d["dgi"], d["dh0"], d["ws"]) for d in (a, b)]))
    print(f"{name}: one recurrence {one:.3f} ms ({one / T * 1e3:.1f} us per time step), two on one stream {seq:.3f} ms, two on two streams {par:.3f} ms, "
          f"two in one launch per time step {mul:.3f} ms")
