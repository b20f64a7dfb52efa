"""TEST INFRASTRUCTURE (only tests/ import this): numpy restatement of csrc/rng.hip -- Philox4x32-10 (Salmon, Moraes, Dror,
Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11; the generator family behind torch's device RNG, which the
reference's `torch.randperm` / `torch.randn_like` draw from: rollout_storage.py:165, actor_critic_decoder.py:283), the
Box-Muller transform and the keyed Feistel permutation with cycle walking.  The reference itself takes these draws from
torch's global generator; parity tests of the update inject identical draws into both implementations, these functions
pin the PRODUCTION draws: integer work bit-exact, the normal variates to float rounding of log / sqrt / sincos."""
import numpy as np

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
U32 = 0xFFFFFFFF


def philox4x32_10(c, k0, k1):
    """c: uint64 array [..., 4] of 32-bit words; returns the same shape."""
    c = [c[..., i].astype(np.uint64) for i in range(4)]
    k0, k1 = np.uint64(k0), np.uint64(k1)
    for _ in range(10):
        p0, p1 = np.uint64(M0) * c[0], np.uint64(M1) * c[2]
        c = [((p1 >> np.uint64(32)) ^ c[1] ^ k0) & np.uint64(U32), p1 & np.uint64(U32),
             ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & np.uint64(U32), p0 & np.uint64(U32)]
        k0, k1 = (k0 + np.uint64(W0)) & np.uint64(U32), (k1 + np.uint64(W1)) & np.uint64(U32)
    return np.stack(c, axis=-1)


def randn(n, seed, offset=0):
    g = np.arange((n + 3) // 4, dtype=np.uint64) + np.uint64(offset)
    ctr = np.stack([g & np.uint64(U32), g >> np.uint64(32), np.full_like(g, 0x6474635f), np.full_like(g, 0x726e646e)], axis=-1)
    x = philox4x32_10(ctr, seed & U32, (seed >> 32) & U32)

    def bm(a, b):
        u1 = ((a >> np.uint64(8)) + np.uint64(1)).astype(np.float32) * np.float32(2.0 ** -24)
        u2 = (b >> np.uint64(8)).astype(np.float32) * np.float32(2.0 ** -24)
        r = np.sqrt(np.float32(-2.0) * np.log(u1))
        t = np.float32(6.283185307179586) * u2
        return r * np.cos(t), r * np.sin(t)
    n0, n1 = bm(x[..., 0], x[..., 1])
    n2, n3 = bm(x[..., 2], x[..., 3])
    return np.stack([n0, n1, n2, n3], axis=-1).reshape(-1)[:n].astype(np.float32)


def _mix32(h):
    h = h.astype(np.uint64)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & np.uint64(U32)
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & np.uint64(U32)
    h ^= h >> np.uint64(16)
    return h


def randperm(n, seed):
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    k = 1
    while (1 << k) < n:
        k += 1
    rbits = (k + 1) // 2
    lbits = k - rbits
    c = np.array([0x6474635f, 0x7065726d, n & U32, (n >> 32) & U32], dtype=np.uint64)
    a = philox4x32_10(c, seed & U32, (seed >> 32) & U32)
    c2 = a.copy()
    c2[0] ^= np.uint64(W0)
    b = philox4x32_10(c2, (seed >> 32) & U32, seed & U32)
    keys = [np.uint64(v) for v in list(a) + list(b)]
    lmask, rmask = np.uint64((1 << lbits) - 1 if lbits else 0), np.uint64((1 << rbits) - 1)
    x = np.arange(n, dtype=np.uint64)
    todo = np.ones(n, dtype=bool)
    while todo.any():
        v = x[todo]
        L, R = (v >> np.uint64(rbits)) & lmask, v & rmask
        for r in range(0, 8, 2):
            L ^= _mix32(R ^ keys[r]) & lmask
            R ^= _mix32(L ^ keys[r + 1]) & rmask
        v = (L << np.uint64(rbits)) | R
        x[todo] = v
        todo[todo] = v >= np.uint64(n)
    return x.astype(np.int64)
