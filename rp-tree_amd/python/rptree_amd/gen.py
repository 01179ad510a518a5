"""Host-side random generation: the part of the hot path that STAYS on the host.

The reference samples its hyperplanes with `sparse pnz dim stdNormal` under
`sample seed` (Batch.hs:37-39,57-61; Gen.hs:148-153,178-195) using the third-party packages
splitmix / splitmix-distributions-0.9.0.0 (stack.yaml:46), which are not part of the reference
tree.  This module restates SplitMix64 and the three distributions used on the path from the
published algorithm.  It is self-consistent but NOT verified against Hackage output; a Haskell
host keeps using the real library and hands the vectors to the C ABI.
"""
import math

import numpy as np

_M = (1 << 64) - 1
_GOLDEN = 0x9E3779B97F4A7C15


def _mix64(z):
    z = ((z ^ (z >> 33)) * 0xFF51AFD7ED558CCD) & _M
    z = ((z ^ (z >> 33)) * 0xC4CEB9FE1A85EC53) & _M
    return z ^ (z >> 33)


def _mix64v13(z):
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M
    return z ^ (z >> 31)


def _mix_gamma(z):
    g = _mix64v13(z) | 1
    n = bin(g ^ (g >> 1)).count("1")
    return g if n >= 24 else g ^ 0xAAAAAAAAAAAAAAAA


class SMGen:
    """SplitMix64 generator state (mkSMGen)."""

    def __init__(self, seed):
        seed &= _M
        self.seed = _mix64(seed)
        self.gamma = _mix_gamma((seed + _GOLDEN) & _M)

    def next_word64(self):
        self.seed = (self.seed + self.gamma) & _M
        return _mix64(self.seed)

    def next_double(self):                      # stdUniform
        return (self.next_word64() >> 11) * (2.0 ** -53)


def bernoulli(g, p):
    return g.next_double() < p


def normal(g, mu, sig):                         # Box-Muller, u1 then u2
    u1 = g.next_double()
    u2 = g.next_double()
    return math.sqrt(-2.0 * math.log(u1)) * math.cos(2.0 * math.pi * u2) * sig + mu


def std_normal(g):
    return normal(g, 0.0, 1.0)


def uniform_r(g, lo, hi):
    return g.next_double() * (hi - lo) + lo


def sparse(g, pnz, dim, rand):
    """Gen.hs:148-153,178-195: for i in [0,dim): bernoulli pnz; if hit draw x, emit (i, x)."""
    idx, val = [], []
    for i in range(dim):
        if bernoulli(g, pnz):
            idx.append(i)
            val.append(rand(g))
    return idx, val


def dense(g, dim, rand):
    """Gen.hs:156-175"""
    return [rand(g) for _ in range(dim)]


def forest_hyperplanes(seed, ntrees, maxd, pnz, dim):
    """Batch.hs:57-61: `sample seed $ replicateM ntrees $ V.replicateM maxd (sparse pnz dim
    stdNormal)` — tree outermost, level inner, one generator threaded through.
    Returns (vectors, R): vectors[t][l] = (idx, val) and the dense-ified R[T][L][dim]."""
    g = SMGen(seed)
    R = np.zeros((ntrees, maxd, dim), dtype=np.float64)
    vectors = []
    for t in range(ntrees):
        lv = []
        for l in range(maxd):
            idx, val = sparse(g, pnz, dim, std_normal)
            if idx:
                R[t, l, idx] = val
            lv.append((idx, val))
        vectors.append(lv)
    return vectors, R


# ---- vectorised draws: word j (0-based) of a generator is mix64(seed + (j + 1) * gamma) -------
def _mix64_np(z):
    z = (z ^ (z >> np.uint64(33))) * np.uint64(0xFF51AFD7ED558CCD)
    z = (z ^ (z >> np.uint64(33))) * np.uint64(0xC4CEB9FE1A85EC53)
    return z ^ (z >> np.uint64(33))


def _doubles(g, first, count):
    """next_double values number first .. first+count-1 of generator g, without advancing it"""
    with np.errstate(over="ignore"):
        j = np.arange(first + 1, first + count + 1, dtype=np.uint64)
        z = np.uint64(g.seed) + j * np.uint64(g.gamma)
        return (_mix64_np(z) >> np.uint64(11)).astype(np.float64) * (2.0 ** -53)


def normal_dense2(seed, n, dim, rows_per_chunk=32768):
    """Gen.hs:132-137 normalDense2 via dataBatch (Batch.hs:66-75), the synthetic data of SURVEY
    8(d): per vector one coin, then `dim` components N(0, 0.5) or N(2, 0.5), each a Box-Muller
    draw of two uniforms.  Every row consumes 1 + 2*dim words, so the whole matrix is computed
    in numpy chunks; the word stream is the scalar generator's (tested), libm's log / cos may
    differ from numpy's in the last bit."""
    g = SMGen(seed)
    per = 1 + 2 * dim
    X = np.empty((n, dim), dtype=np.float64)
    for r0 in range(0, n, rows_per_chunk):
        r1 = min(n, r0 + rows_per_chunk)
        u = _doubles(g, r0 * per, (r1 - r0) * per).reshape(r1 - r0, per)
        mu = np.where(u[:, 0] < 0.5, 0.0, 2.0)              # b <- bernoulli 0.5: True -> mean 0
        u1, u2 = u[:, 1::2], u[:, 2::2]
        X[r0:r1] = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2) * 0.5 + mu[:, None]
    return X


def normal_dense2_torch(seed, n, dim, device, rows_per_chunk=65536):
    """normal_dense2 evaluated with torch on `device` (the bench draws its 1M x 128 matrix in HBM:
    nothing crosses PCIe).  Same SplitMix64 word stream — int64 arithmetic wraps like uint64, the
    logical shifts are arithmetic shifts with the sign extension masked off — and the same
    Box-Muller formula; the device's log / cos may differ from libm's in the last bit."""
    import torch

    def s64(x):                      # uint64 constant as the int64 with the same bits
        x &= _M
        return x - (1 << 64) if x >= (1 << 63) else x

    def lsr(z, k):
        return (z >> k) & ((1 << (64 - k)) - 1)

    def mix64(z):
        z = (z ^ lsr(z, 33)) * s64(0xFF51AFD7ED558CCD)
        z = (z ^ lsr(z, 33)) * s64(0xC4CEB9FE1A85EC53)
        return z ^ lsr(z, 33)

    g = SMGen(seed)
    per = 1 + 2 * dim
    X = torch.empty((n, dim), dtype=torch.float64, device=device)
    for r0 in range(0, n, rows_per_chunk):
        r1 = min(n, r0 + rows_per_chunk)
        j = torch.arange(r0 * per + 1, r1 * per + 1, dtype=torch.int64, device=device)
        z = j * s64(g.gamma) + s64(g.seed)
        u = (lsr(mix64(z), 11).to(torch.float64) * (2.0 ** -53)).view(r1 - r0, per)
        mu = torch.where(u[:, 0] < 0.5, 0.0, 2.0).to(torch.float64)
        u1, u2 = u[:, 1::2], u[:, 2::2]
        X[r0:r1] = torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(2.0 * math.pi * u2) * 0.5 + mu[:, None]
    return X
