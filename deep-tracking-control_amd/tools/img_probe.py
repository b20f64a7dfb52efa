"""Round-4 probe: the image-operand kernels (linear_i3_kernel) next to the split kernels that convert X inside their K loop, on the
bench's layer shapes; HIP-event time per launch over back-to-back launches (weight images prebuilt, as in the trainer)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dtc_amd import ops  # noqa: E402

DEV = "cuda:0"


def timeit(fn, reps=40):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    g = torch.Generator(device=DEV).manual_seed(1)
    for M, N, K in ((24576, 512, 512), (24576, 693, 512), (24576, 256, 512), (24576, 128, 256)):
        X = torch.randn(M, K, device=DEV, generator=g)
        W = torch.randn(N, K, device=DEV, generator=g) / K ** 0.5
        b = torch.randn(N, device=DEV, generator=g)
        dZ = torch.randn(M, N, device=DEV, generator=g)
        Y, dX = torch.empty(M, N, device=DEV), torch.empty(M, K, device=DEV)
        Ximg, dZimg = ops.AImage.from_tensor(X), ops.AImage.from_tensor(dZ)
        Yimg, dXimg = ops.AImage(M, N, DEV), ops.AImage(M, K, DEV)
        mask = ops.relu_mask(M, N, DEV) if N % 128 == 0 else None
        maskK = ops.relu_mask(M, K, DEV)
        maskK.fill_(-1)
        imgs = ops.WeightImages()
        res = {}
        for rnd in range(2):                      # first block learns the layer set, second runs with prebuilt images
            with imgs:
                res["fwd s3 (fp32 X, fp32 Y)"] = timeit(lambda: ops.linear_fwd(X, W, b, Y, "relu", mask=mask, split=True))
                res["fwd i3 -> fp32 Y"] = timeit(lambda: ops.linear_fwd_img(Ximg, W, b, Y, None, "relu", mask=mask))
                res["fwd i3 -> image"] = timeit(lambda: ops.linear_fwd_img(Ximg, W, b, None, Yimg, "relu", mask=mask))
                res["fwd i3 -> fp32 + image"] = timeit(lambda: ops.linear_fwd_img(Ximg, W, b, Y, Yimg, "relu", mask=mask))
                res["dgrad s3 (fp32 dZ, fp32 dX, mask)"] = timeit(lambda: ops.linear_dgrad(dZ, W, dX, None, "relu", mask=maskK, split=True))
                res["dgrad i3 -> fp32 dX"] = timeit(lambda: ops.linear_dgrad_img(dZimg, W, dX, None, mask=maskK))
                res["dgrad i3 -> image"] = timeit(lambda: ops.linear_dgrad_img(dZimg, W, None, dXimg, mask=maskK))
                res["aimage (fp32 -> image)"] = timeit(lambda: ops.AImage.from_tensor(X, Ximg))
        fl = 2.0 * M * N * K
        print(f"--- {M} x {N} x {K}")
        for k, us in res.items():
            print(f"  {k:38s} {us:8.1f} us   {fl / us / 1e6:7.1f} TFLOP/s" if "aimage" not in k else f"  {k:38s} {us:8.1f} us")


def wgrad():
    g = torch.Generator(device=DEV).manual_seed(2)
    M = 24576
    for layers in ([(512, 512)], [(512, 512), (512, 512), (693, 512)], [(512, 512), (512, 512)]):
        s3, i3 = [], []
        fl = 0.0
        for N, K in layers:
            dZ, X = torch.randn(M, N, device=DEV, generator=g), torch.randn(M, K, device=DEV, generator=g)
            dW, db = torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
            s3.append((dZ, X, dW, db))
            i3.append((ops.AImage.from_tensor(dZ), ops.AImage.from_tensor(X), dW, db))
            fl += 2.0 * M * N * K
        ws = ops.workspace(ops.wgrad_group_workspace_bytes(s3, M, split=True), DEV)
        wi = ops.workspace(ops.wgrad_group_img_workspace_bytes(i3, M), DEV)
        t3 = timeit(lambda: ops.wgrad_group(s3, M, ws, split=True))
        ti = timeit(lambda: ops.wgrad_group_img(i3, M, wi))
        print(f"--- wgrad {layers}: split (fp32 operands) {t3:7.1f} us {fl / t3 / 1e6:6.1f} TFLOP/s   images {ti:7.1f} us {fl / ti / 1e6:6.1f} TFLOP/s")


if __name__ == "__main__":
    if "wgrad" in sys.argv:
        wgrad()
    else:
        main()
