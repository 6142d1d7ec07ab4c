"""TEST INFRASTRUCTURE -- numpy restatement of Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as
easy as 1, 2, 3", SC'11; the Random123 library's philox4x32 with 10 rounds), the generator behind the product's
device-side noise (`ssn_philox_uniform_*`, tc_gan_amd/csrc/ssn_aux.hip).  The reference has no device noise (its z is
`rng.rand` on the host, tc_gan/networks/ssn.py:434-439), so this is pinned by the published known-answer vectors
(tests/test_noise.py), not by the reference.  Only tests/ may import this module."""
import numpy as np

M0, M1 = 0xD2511F53, 0xCD9E8D57
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF


def philox4x32_10(counter, key):
    """counter: 4 words, key: 2 words (Python ints) -> 4 output words."""
    c0, c1, c2, c3 = (int(c) & MASK for c in counter)
    k0, k1 = (int(k) & MASK for k in key)
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        hi0, lo0 = p0 >> 32, p0 & MASK
        hi1, lo1 = p1 >> 32, p1 & MASK
        c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
        k0, k1 = (k0 + W0) & MASK, (k1 + W1) & MASK
    return c0, c1, c2, c3


def uniform(seed, offset, n):
    """The stream of ssn_philox_uniform_*: element g = word g % 4 of the block with counter (g // 4, 0, 0, 0) and key
    (seed & 0xffffffff, seed >> 32); value = (word >> 8) / 2^24 as float32."""
    out = np.empty(n, dtype=np.float32)
    key = (seed & MASK, (seed >> 32) & MASK)
    cache = {}
    for i in range(n):
        g = offset + i
        blk = g // 4
        if blk not in cache:
            cache = {blk: philox4x32_10((blk & MASK, blk >> 32, 0, 0), key)}
        out[i] = np.float32(cache[blk][g % 4] >> 8) * np.float32(1.0 / 16777216.0)
    return out
