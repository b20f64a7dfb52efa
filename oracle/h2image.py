"""Numpy restatement of the block-scaled two-term fp16 operand-image format (TEST INFRASTRUCTURE: only tests/ import this).

Restates include/dtc_hip.h ("block-scaled two-term fp16 operand images"), csrc/h2i_core.hpp (hi_exp, hi_split8, the chunk / slot
layout) and csrc/gemm_h2i.hip (h2i_pack_kernel): the representation of the wide layers' operands -- the fp32 activations and gradients
of rsl_rl/rsl_rl/modules/actor_critic_decoder.py:98-188, 323-349 -- as the GPU kernels write it, byte for byte.

    image(M, K) = ceil(M / 128) x ceil(K / 16) chunks of [plane 2][slot 256][8 fp16]  +  int32 exps[row tile][k block][128]
    slot(row r of the tile, k half h) = 2 r + (h ^ ((r >> 3) & 1));   k block = 8 stages = 128 columns
    e(row, block) = clamp(141 - biased_exponent(max finite |x| of the row's block), <= 100), 0x7fff for a block without finite non-zero
    hi = fp16(x 2^e),  lo = fp16(x 2^e - hi)      (round to nearest even, the remainder exact in fp32)
"""
import numpy as np

EZERO = 0x7FFF


def _f32(a):
    return np.asarray(a, dtype=np.float32)


def exponents(A):
    """[row tiles, k blocks, 128] int32 exponents of A [M, K] (rows / columns behind the matrix count as zeros)."""
    M, K = A.shape
    rt, st = -(-M // 128), -(-K // 16)
    kb = -(-st // 8)
    P = np.zeros((rt * 128, kb * 128), dtype=np.float32)
    P[:M, :K] = A
    bits = P.view(np.uint32) & 0x7FFFFFFF
    bits = np.where(bits < 0x7F800000, bits, 0)                               # non-finite elements do not take part
    mx = bits.reshape(rt, 128, kb, 128).max(axis=3).transpose(0, 2, 1)        # [rt, kb, 128]
    e = 141 - (mx >> 23).astype(np.int32)
    e = np.minimum(e, 100)
    return np.where(mx == 0, EZERO, e).astype(np.int32)


def encode(A):
    """-> (chunks uint16 [rt, stages, 2, 256, 8], exps int32 [rt, kb, 128])"""
    A = _f32(A)
    M, K = A.shape
    rt, st = -(-M // 128), -(-K // 16)
    ex = exponents(A)
    P = np.zeros((rt * 128, st * 16), dtype=np.float32)
    P[:M, :K] = A
    out = np.zeros((rt, st, 2, 256, 8), dtype=np.uint16)
    r = np.arange(128)
    with np.errstate(over="ignore", invalid="ignore"):
        for t in range(rt):
            for s in range(st):
                e = ex[t, s // 8].copy()
                e[e == EZERO] = 0
                x = np.ldexp(P[t * 128:(t + 1) * 128, s * 16:(s + 1) * 16], e[:, None]).astype(np.float32)     # exact (power of two)
                hi = x.astype(np.float16)
                lo = (x - hi.astype(np.float32)).astype(np.float32).astype(np.float16)
                for h in range(2):
                    slot = 2 * r + (h ^ ((r >> 3) & 1))
                    out[t, s, 0, slot] = hi[:, 8 * h:8 * h + 8].view(np.uint16)
                    out[t, s, 1, slot] = lo[:, 8 * h:8 * h + 8].view(np.uint16)
    return out, ex


def decode(chunks, ex, M, K):
    rt, st = chunks.shape[:2]
    r = np.arange(128)
    out = np.zeros((rt * 128, st * 16), dtype=np.float32)
    for t in range(rt):
        for s in range(st):
            e = ex[t, s // 8].copy()
            e[e == EZERO] = 0
            for h in range(2):
                slot = 2 * r + (h ^ ((r >> 3) & 1))
                v = chunks[t, s, 0, slot].view(np.float16).astype(np.float32) + chunks[t, s, 1, slot].view(np.float16).astype(np.float32)
                out[t * 128:(t + 1) * 128, s * 16 + 8 * h:s * 16 + 8 * h + 8] = np.ldexp(v, -e[:, None])
    return out[:M, :K]
