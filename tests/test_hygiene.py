"""Source hygiene of the package (CPU): no name is defined twice in one scope (a botched merge once left RolloutStorage with
two class bodies, the later one silently winning), and RolloutStorage.get_statistics follows rollout_storage.py:154-160."""
import ast
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "deep-tracking-control_amd")


def _py_files():
    for base in (os.path.join(PKG, "dtc_amd"), os.path.join(ROOT, "oracle")):
        for d, _dirs, files in os.walk(base):
            for f in files:
                if f.endswith(".py"):
                    yield os.path.join(d, f)
    for f in ("bench.py", "__graft_entry__.py", os.path.join("deep-tracking-control_amd", "build.py")):
        yield os.path.join(ROOT, f)


def _duplicates(tree):
    out = []
    for node in ast.walk(tree):
        if not isinstance(node, (ast.Module, ast.ClassDef)):
            continue
        seen = {}
        for child in node.body:
            if isinstance(child, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
                decorated = any(isinstance(d, ast.Attribute) and d.attr in ("setter", "getter", "deleter") or
                                isinstance(d, ast.Name) and d.id == "overload" for d in child.decorator_list)
                if child.name in seen and not decorated:
                    out.append((getattr(node, "name", "<module>"), child.name, seen[child.name], child.lineno))
                seen[child.name] = child.lineno
    return out


def test_no_scope_defines_a_name_twice():
    bad = {}
    for path in _py_files():
        dup = _duplicates(ast.parse(open(path).read(), path))
        if dup:
            bad[os.path.relpath(path, ROOT)] = dup
    assert not bad, f"duplicate definitions (scope, name, first line, second line): {bad}"


def test_get_statistics_matches_reference_formula():
    from dtc_amd.storage import RolloutStorage
    T, N = 24, 16
    st = RolloutStorage(N, T, [5], [7], [10], [3], device="cpu")
    g = torch.Generator().manual_seed(5)
    st.dones.copy_((torch.rand(T, N, 1, generator=g) < 0.1).to(torch.uint8))
    st.rewards.copy_(torch.randn(T, N, 1, generator=g))
    # rollout_storage.py:154-160, restated on a copy
    done = st.dones.clone()
    done[-1] = 1
    flat = done.permute(1, 0, 2).reshape(-1, 1)
    idx = torch.cat((flat.new_tensor([-1], dtype=torch.int64), flat.nonzero(as_tuple=False)[:, 0]))
    want_len, want_rew = (idx[1:] - idx[:-1]).float().mean(), st.rewards.mean()
    got_len, got_rew = st.get_statistics()
    assert torch.equal(got_len, want_len) and torch.equal(got_rew, want_rew)
    assert bool((st.dones[-1] == 1).all())               # the reference's in-place side effect
