"""numpy restatement of the ACTIVATION / WEIGHT IMAGE format of the split-precision kernels (TEST INFRASTRUCTURE ONLY).

include/dtc_hip.h ("Activation images"), csrc/s3_core.hpp (split3, rslot), csrc/gemm_s3.hip (aimage_kernel, wimage_kernel):
  * every fp32 value a is the sum of three bf16 terms a1 = bf16(a), a2 = bf16(a - a1), a3 = bf16((a - a1) - a2) -- round to nearest
    even, the two remainders are exact fp32 subtractions (csrc/gemm_s3.hip:4-13);
  * a matrix [M, K] is cut into 128-row tiles and 16-column stages; chunk (tile r, stage s) sits at byte ((r * stages + s) * 12288) and
    holds [plane 3][slot 256][8 bf16]: slot = row * 2 + (half ^ ((row >> 3) & 1)) with row the row inside the tile and half the 8-column
    half of the stage; rows >= M and columns >= K are zero.
The reference has no such format (it is torch fp32 throughout, rsl_rl/rsl_rl/modules/actor_critic_decoder.py:98-188): this file pins the
layout the HIP kernels produce and consume, byte for byte."""
import numpy as np


def bf16_rne(x: np.ndarray) -> np.ndarray:
    """fp32 -> the fp32 value of its bf16 rounding (round to nearest even; v_cvt_pk_bf16_f32)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32)


def split3(x: np.ndarray):
    a1 = bf16_rne(x)
    r1 = (x.astype(np.float32) - a1).astype(np.float32)
    a2 = bf16_rne(r1)
    r2 = (r1 - a2).astype(np.float32)
    a3 = bf16_rne(r2)
    return a1, a2, a3


def encode(A: np.ndarray) -> np.ndarray:
    """fp32 [M, K] -> the image as uint16 [row tiles, stages, 3, 256, 8] (bf16 bit patterns)."""
    M, K = A.shape
    rt, st = -(-M // 128), -(-K // 16)
    pad = np.zeros((rt * 128, st * 16), dtype=np.float32)
    pad[:M, :K] = A
    planes = split3(pad)
    img = np.zeros((rt, st, 3, 256, 8), dtype=np.uint16)
    rows = np.arange(128)
    for p, pl in enumerate(planes):
        bits = (pl.view(np.uint32) >> 16).astype(np.uint16).reshape(rt, 128, st, 2, 8)       # [tile, row, stage, half, 8]
        for h in range(2):
            slot = rows * 2 + (h ^ ((rows >> 3) & 1))
            img[:, :, p, slot, :] = bits[:, :, :, h, :].transpose(0, 2, 1, 3)
    return img


def decode(img: np.ndarray, M: int, K: int) -> np.ndarray:
    """uint16 [row tiles, stages, 3, 256, 8] -> fp32 [M, K]: (plane 1 + plane 2) + plane 3."""
    rt, st = img.shape[:2]
    vals = (img.astype(np.uint32) << 16).view(np.float32)
    rows = np.arange(128)
    out = np.zeros((3, rt, 128, st, 2, 8), dtype=np.float32)
    for h in range(2):
        slot = rows * 2 + (h ^ ((rows >> 3) & 1))
        out[:, :, :, :, h, :] = vals[:, :, :, slot, :].transpose(2, 0, 3, 1, 4)
    tot = (out[0] + out[1]) + out[2]
    return tot.reshape(rt * 128, st * 16)[:M, :K]
