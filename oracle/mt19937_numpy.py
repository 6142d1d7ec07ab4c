"""TEST INFRASTRUCTURE (oracle) -- never imported by the product package.

numpy restatement of the generator behind the reference's noise: ``zs = rng.rand(batchsize, 2N, 2N)`` with
``rng = numpy.random.RandomState(seed)`` (tc_gan/networks/ssn.py:434-439; stream order networks/cwgan.py:438-481).
The algorithm is not in /root/reference: it is numpy's (pinned ``numpy=1.13.1`` in requirements-conda.txt:46), whose
``RandomState`` is randomkit's MT19937 (Matsumoto & Nishimura 1998):

* state = ``key[624]`` (uint32) + ``pos``; ``pos == 624`` means the block is used up and the next output first
  regenerates all 624 words: ``key[i] = key[(i+397) % 624] ^ (y >> 1) ^ (-(y & 1) & 0x9908b0df)`` with
  ``y = (key[i] & 0x80000000) | (key[(i+1) % 624] & 0x7fffffff)``, in index order (so i >= 227 reads NEW words at
  i - 227 and i = 623 reads the new ``key[0]``);
* output word = tempered ``key[pos++]``: ``y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680; y ^= (y << 15) & 0xefc60000;
  y ^= y >> 18``;
* ``random_sample`` / ``rand``: two words per double, ``a = w0 >> 5, b = w1 >> 6, (a * 2**26 + b) / 2**53``;
* integer seeding (``RandomState(seed)``): ``key[0] = seed; key[i] = 1812433253 * (key[i-1] ^ (key[i-1] >> 30)) + i``, pos = 624.

PINNED against ``numpy.random.RandomState`` itself (tests/test_mt19937.py: words, doubles and ``get_state()`` after the
draw, for seeds, positions and counts incl. odd positions and block boundaries) -- numpy keeps this stream frozen
(NEP 19), so the installed numpy IS the reference's generator.
"""
import numpy as np

N, M = 624, 397
MATRIX_A = np.uint32(0x9908b0df)
UPPER, LOWER = np.uint32(0x80000000), np.uint32(0x7fffffff)


def seed_state(seed):
    """(key, pos) of ``RandomState(seed)`` for an integer seed (randomkit ``rk_seed``)."""
    key = np.empty(N, dtype=np.uint32)
    s = int(seed) & 0xffffffff
    for i in range(N):
        key[i] = s
        s = (1812433253 * (s ^ (s >> 30)) + i + 1) & 0xffffffff
    return key, N


def _twist(u, v):
    y = (u & UPPER) | (v & LOWER)
    return (y >> np.uint32(1)) ^ (np.where(v & np.uint32(1), MATRIX_A, np.uint32(0)))


def regenerate(key):
    """The next block of 624 untempered words (randomkit ``rk_random``'s refill), as a new array."""
    old = np.asarray(key, dtype=np.uint32)
    new = np.empty(N, dtype=np.uint32)
    new[:N - M] = old[M:] ^ _twist(old[:N - M], old[1:N - M + 1])                   # i = 0 .. 226
    new[N - M:2 * (N - M)] = new[:N - M] ^ _twist(old[N - M:2 * (N - M)], old[N - M + 1:2 * (N - M) + 1])   # 227 .. 453
    lo = 2 * (N - M)
    new[lo:N - 1] = new[lo - (N - M):N - 1 - (N - M)] ^ _twist(old[lo:N - 1], old[lo + 1:N])   # 454 .. 622
    new[N - 1] = new[M - 1] ^ _twist(old[N - 1:N], new[0:1])[0]
    return new


def temper(y):
    y = np.asarray(y, dtype=np.uint32).copy()
    y ^= y >> np.uint32(11)
    y ^= (y << np.uint32(7)) & np.uint32(0x9d2c5680)
    y ^= (y << np.uint32(15)) & np.uint32(0xefc60000)
    y ^= y >> np.uint32(18)
    return y


def words(key, pos, n):
    """The next n output words and the state after them: (words, key, pos)."""
    key = np.asarray(key, dtype=np.uint32).copy()
    out = np.empty(n, dtype=np.uint32)
    done = 0
    while done < n:
        if pos == N:
            key = regenerate(key)
            pos = 0
        take = min(N - pos, n - done)
        out[done:done + take] = temper(key[pos:pos + take])
        pos += take
        done += take
    return out, key, pos


def random_sample(key, pos, n):
    """``RandomState.random_sample(n)``: (doubles, key, pos)."""
    w, key, pos = words(key, pos, 2 * n)
    a = (w[0::2] >> np.uint32(5)).astype(np.float64)
    b = (w[1::2] >> np.uint32(6)).astype(np.float64)
    return (a * 67108864.0 + b) / 9007199254740992.0, key, pos


def choice2_signs(key, pos, n):
    """``RandomState.choice(2, n) * 2 - 1`` (zs_in of the reference's heterogeneous-input models, tc_gan/networks/ssn.py:714-715):
    (signs, key, pos).  `choice(2, n)` is `randint(0, 2, n)`: numpy's masked rejection for the range [0, 1] draws one 32-bit
    output per element, keeps its low bit (mask 1) and never rejects (`_rand_int64` -> `random_bounded_uint64_fill`, rng = 1;
    randomkit's `rk_random_uint64` before numpy 1.17 does the same)."""
    w, key, pos = words(key, pos, n)
    return (w & np.uint32(1)).astype(np.int64) * 2 - 1, key, pos


def advance_blocks(key, nblocks):
    """key after nblocks refills (sequential; the jump-ahead of the device generator is checked against this)."""
    key = np.asarray(key, dtype=np.uint32)
    for _ in range(int(nblocks)):
        key = regenerate(key)
    return key


def untempered_stream(key, nwords):
    """x[0 .. nwords): x[0:624] = key, x[k + 624] = x[k + 397] ^ twist(x[k], x[k + 1]) -- the word sequence the jump
    polynomials act on (a polynomial g applied at k: XOR of x[k + i] over the set bits i of g)."""
    x = np.empty(max(nwords, N), dtype=np.uint32)
    x[:N] = key
    k = 0
    while k + N < nwords:
        step = min(N - M, nwords - N - k)
        x[k + N:k + N + step] = x[k + M:k + M + step] ^ _twist(x[k:k + step], x[k + 1:k + 1 + step])
        k += step
    return x[:nwords]


def apply_jump(key, poly_bits):
    """The state `poly` maps `key` to: word j = XOR over the set bits i of x[i + j] (bit 31 of word 0 and words
    1..623 are the 19937 state bits; the low 31 bits of word 0 are not part of the state)."""
    taps = np.flatnonzero(poly_bits)
    x = untempered_stream(key, int(taps.max()) + N + 1)
    out = np.zeros(N, dtype=np.uint32)
    for i in taps:
        out ^= x[i:i + N]
    return out
