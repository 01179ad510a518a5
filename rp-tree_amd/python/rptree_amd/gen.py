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
