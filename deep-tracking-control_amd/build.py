"""Build libdtc_hip.so (hand-written HIP kernels + C ABI) for gfx950, in-tree.

    python deep-tracking-control_amd/build.py [--force] [--save-temps]

hipcc cross-compiles without a GPU.  Objects land in deep-tracking-control_amd/build/, the
shared library in deep-tracking-control_amd/dtc_amd/lib/libdtc_hip.so (git-ignored; it travels
to the GPU box with the gpurun snapshot).
"""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIBDIR = os.path.join(HERE, "dtc_amd", "lib")
LIB = os.path.join(LIBDIR, "libdtc_hip.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
          "-fhip-fp32-correctly-rounded-divide-sqrt", "-I", os.path.join(ROOT, "include"), "-I", CSRC]
# bit-exact kernels: no fused multiply-add contraction (see oracle/foothold.py, oracle/gae.py)
PER_FILE = {"foothold.hip": ["-ffp-contract=off"], "gae.hip": ["-ffp-contract=off"],
            "optim.hip": ["-ffp-contract=off"], "envstep.hip": ["-ffp-contract=off"],      # Adam mirrors torch's separately rounded ops
            # split kernels: no SLP packing of the remainder subtractions into v_pk_add_f32 -- a packed fp32 op next to MFMAs
            # costs more than the two scalar ones it replaces (MI355X_MICROARCH.md, issue-slot table); 69.8 -> 68.1 ms per step
            "wgrad_s3.hip": ["-fno-slp-vectorize"], "gru_s3.hip": ["-fno-slp-vectorize"],
            # forward / data-gradient split kernels additionally with the backend's max-ILP scheduling strategy (the fragment reads and
            # the first MFMAs of a stage, outside the fenced conversion block): 67.86 -> 66.55 ms per step interleaved; the same flag on
            # wgrad_s3.hip / gru_s3.hip / the single-pass kernels is neutral
            # (and without the post-RA scheduler pass on top of it: 65.97 -> 65.59)
            "gemm_s3.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-enable-post-misched=false"]}
if os.environ.get("DTC_BK"):                       # tuning aid: K step of the GEMM kernels
    PER_FILE["gemm.hip"] = [f"-DDTC_BK={int(os.environ['DTC_BK'])}"]


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src, force, save_temps):
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(ROOT, "include", "dtc_hip.h"))
    if not force and not _stale(obj, [os.path.join(CSRC, src), __file__] + headers):
        return obj, ""
    cmd = [HIPCC, *COMMON, *PER_FILE.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
    if save_temps:
        cmd.insert(1, "-save-temps=obj")
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=OBJ)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
    return obj, r.stderr


def kernel_resources():
    """Device-side resource use of every kernel: {kernel name: dict(scratch=bytes, vgprs=n, lds=bytes)}, read from the code
    object metadata of a device-only assembly pass over csrc/*.hip (same flags as the product build).  Used by
    tests/test_abi_and_host.py: NO kernel may use scratch (a kernel with any private segment pays at every wave launch -- the
    forward GEMM lost 1.3 % of the whole step to five spilled registers of its prologue)."""
    import re
    import tempfile

    def one(src):
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "k.s")
            cmd = [HIPCC, *COMMON, *PER_FILE.get(src, []), "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", out]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc -S failed on {src}:\n{r.stderr[-1500:]}")
            text = open(out).read()
        res = {}
        for blk in text.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            if not name:
                continue
            g = lambda key: int(re.search(rf"\.{key}:\s+(\d+)", blk).group(1))
            res[name.group(1)] = dict(scratch=g("private_segment_fixed_size"), vgprs=g("vgpr_count"), lds=g("group_segment_fixed_size"),
                                      source=src)
        return res

    out = {}
    with cf.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
        for res in ex.map(one, sources()):
            out.update(res)
    return out


def build_asan(verbose=True):
    """Host-side AddressSanitizer build of the same sources (device code unsanitised): libdtc_hip_asan.so.  Used by
    tests/test_abi_and_host.py to run the argument-validation / descriptor-marshalling layer of the C ABI under ASan
    (`LD_PRELOAD=<libclang_rt.asan> DTC_LIB=<this file>`); no GPU needed, the calls under test fail validation before
    any HIP call."""
    out = os.path.join(LIBDIR, "libdtc_hip_asan.so")
    srcs = [os.path.join(CSRC, f) for f in sources()]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")] + [os.path.join(ROOT, "include", "dtc_hip.h")]
    if _stale(out, deps):
        os.makedirs(LIBDIR, exist_ok=True)
        cmd = [HIPCC, *COMMON, "-O1", "-g", "-fsanitize=address", "-fno-gpu-sanitize", "-fno-omit-frame-pointer", "-shared", *srcs, "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"ASan build failed:\n{r.stderr[-2000:]}")
    if verbose:
        print(f"built {out}")
    return out


def asan_runtime():
    r = subprocess.run([os.path.join(os.path.dirname(HIPCC), "..", "lib", "llvm", "bin", "clang"), "-print-file-name=libclang_rt.asan-x86_64.so"],
                       capture_output=True, text=True)
    return r.stdout.strip()


def build(force=False, save_temps=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = sources()
    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        results = list(ex.map(lambda s: _compile(s, force, save_temps), srcs))
    objs = [o for o, _ in results]
    for (_, warn), s in zip(results, srcs):
        if warn.strip() and verbose:
            print(f"[{s}]\n{warn}", file=sys.stderr)
    if force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
    if verbose:
        print(f"built {LIB} ({os.path.getsize(LIB) / 1024:.0f} KiB) from {len(srcs)} sources")
    return LIB


if __name__ == "__main__":
    if "--asan" in sys.argv:
        build_asan()
    else:
        build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv)
