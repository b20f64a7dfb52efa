"""Reference-import harness (golden generation ONLY; runs only where /root/reference exists).

Registers tiny stand-in modules for the third-party packages the reference imports but
which are absent from this image (params_proto, isaacgym, gym, tensorboard, cv2,
torchvision), then puts /root/reference on sys.path.  The stand-ins contain no reference
source; `isaacgym.torch_utils` restates the three published Isaac Gym Preview 4 quaternion
helpers the scorer calls (SURVEY.md §8c "parity unpinned" boundary).

Nothing in tests/, bench.py or smoke() imports this file at run time: it is used by
tests/golden/make_golden.py to produce the committed fixtures.
"""
import contextlib
import io
import sys
import types

import torch

REF_ROOT = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def _normalize(x, eps: float = 1e-9):
    return x / x.norm(p=2, dim=-1).clamp(min=eps, max=None).unsqueeze(-1)


def _quat_apply(a, b):
    shape = b.shape
    a = a.reshape(-1, 4)
    b = b.reshape(-1, 3)
    xyz = a[:, :3]
    t = xyz.cross(b, dim=-1) * 2
    return (b + a[:, 3:] * t + xyz.cross(t, dim=-1)).view(shape)


def _quat_rotate_inverse(q, v):
    shape = q.shape
    q_w = q[:, -1]
    q_vec = q[:, :3]
    a = v * (2.0 * q_w ** 2 - 1.0).unsqueeze(-1)
    b = torch.cross(q_vec, v, dim=-1) * q_w.unsqueeze(-1) * 2.0
    c = q_vec * torch.bmm(q_vec.view(shape[0], 1, 3), v.view(shape[0], 3, 1)).squeeze(-1) * 2.0
    return a - b + c


def _to_torch(x, dtype=torch.float, device="cpu", requires_grad=False):
    return torch.tensor(x, dtype=dtype, device=device, requires_grad=requires_grad)


def _torch_rand_float(lower, upper, shape, device):
    return (upper - lower) * torch.rand(*shape, device=device) + lower


def install():
    if "/root/reference/rsl_rl" in sys.path:
        return
    sys.dont_write_bytecode = True

    class PrefixProto:
        def __init_subclass__(cls, cli=False, **kw):
            super().__init_subclass__(**kw)

    _mod("params_proto")
    _mod("params_proto.proto", PrefixProto=PrefixProto)

    class SummaryWriter:
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass

    _mod("tensorboard")
    tb = _mod("torch.utils.tensorboard", SummaryWriter=SummaryWriter)
    torch.utils.tensorboard = tb

    class Wrapper:
        def __init__(self, env):
            self.env = env

        def __getattr__(self, k):
            return getattr(self.__dict__["env"], k)

    _mod("gym", Wrapper=Wrapper)
    _mod("cv2")
    tv = _mod("torchvision")
    tv.transforms = _mod("torchvision.transforms")
    ig = _mod("isaacgym")
    for sub in ("gymapi", "gymtorch", "gymutil", "terrain_utils"):
        setattr(ig, sub, _mod("isaacgym." + sub))
    tu = _mod(
        "isaacgym.torch_utils",
        torch=torch,
        normalize=_normalize,
        quat_apply=_quat_apply,
        quat_rotate_inverse=_quat_rotate_inverse,
        to_torch=_to_torch,
        torch_rand_float=_torch_rand_float,
    )
    tu.__all__ = ["torch", "normalize", "quat_apply", "quat_rotate_inverse", "to_torch", "torch_rand_float"]
    ig.torch_utils = tu
    sys.path.insert(0, REF_ROOT)
    sys.path.insert(0, REF_ROOT + "/rsl_rl")


@contextlib.contextmanager
def quiet():
    """The reference prints every parameter in PPO.__init__ (ppo.py:73-75)."""
    with contextlib.redirect_stdout(io.StringIO()):
        yield
