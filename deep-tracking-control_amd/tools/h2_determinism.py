"""Debug: first GEMM call whose result differs between the first and the second identical run of a recurrent update (fp16 path)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dtc_amd import ops
from dtc_amd.algorithms import RecurrentPPO
from dtc_amd.modules import ActorCriticRecurrent
DEV = "cuda:0"
LOG, STATES, KEEP = [], [], {}
_fwd, _dgrad = ops.linear_fwd, ops.linear_dgrad
_dbg = (ctypes.c_uint32 * 6)()
_lib = ctypes.CDLL(os.environ["DTC_LIB"])


def digest(t):
    torch.cuda.synchronize()
    return int(t.contiguous().view(torch.int32).to(torch.int64).sum().item())


def fwd(X, W, b, Y, act=None, M=None, **kw):
    r = _fwd(X, W, b, Y, act, M=M, **kw)
    rows = Y.shape[0] if M is None else M
    LOG.append(("fwd", tuple(W.shape), rows, act, digest(Y[:rows, :W.shape[0]]), digest(X[:rows]) if torch.is_tensor(X) else "segs", digest(W)))
    return r


def dgrad(dZ, W, dX, Xsaved=None, act=None, M=None, **kw):
    torch.cuda.synchronize()
    _lib.dtc_h2_debug(_dbg)
    r = _dgrad(dZ, W, dX, Xsaved, act, M=M, **kw)
    torch.cuda.synchronize()
    _lib.dtc_h2_debug(_dbg)
    rows = dZ.shape[0] if M is None else M
    LOG.append(("dgrad", tuple(W.shape), rows, act, digest(dX[:rows]) if torch.is_tensor(dX) else "segs", digest(dZ[:rows]), digest(W),
                digest(Xsaved[:rows]) if Xsaved is not None else None, tuple(int(v) for v in _dbg)))
    if tuple(W.shape) == (512, 128):
        KEEP.setdefault(len(STATES), []).append((len(LOG) - 1, dZ[:rows].cpu(), W.cpu(), dX[:rows].cpu()))
    return r


ops.linear_fwd, ops.linear_dgrad = fwd, dgrad


def run():
    LOG.clear()
    torch.manual_seed(0)
    ac = ActorCriticRecurrent(53, 1389, 12, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128],
                              rnn_type="lstm", rnn_hidden_size=128, rnn_num_layers=1)
    alg = RecurrentPPO(ac, device=DEV, learning_rate=1e-3)
    alg.overlap = False
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
    STATES.append({k: t.clone() for k, t in ac.state_dict().items()})
    return list(LOG)


a, b = run(), run()
print(len(a), len(b), "calls")
for i, (x, y) in enumerate(zip(a, b)):
    if x != y:
        print("first differing call", i, "\n ", x, "\n ", y)
        break
else:
    print("all equal")
for (i0, dZ0, W0, dX0), (i1, dZ1, W1, dX1) in zip(KEEP[0], KEEP[1]):
    if not torch.equal(dX0, dX1):
        d = (dX0 - dX1)
        nz = d.nonzero()
        ref = dZ0.double() @ W0.double()
        print("call", i0, "inputs equal", torch.equal(dZ0, dZ1), torch.equal(W0, W1), "differing elements", nz.shape[0], "of", dX0.numel(),
              "rows", sorted(set(nz[:, 0].tolist()))[:10], "n rows", len(set(nz[:, 0].tolist())), "cols", sorted(set(nz[:, 1].tolist()))[:10], "n cols", len(set(nz[:, 1].tolist())))
        print("  max |diff|", float(d.abs().max()), "scale", float(ref.abs().max()), "err run0", float((dX0.double() - ref).abs().max()), "err run1", float((dX1.double() - ref).abs().max()))
        break
