"""CPU checks of the device generator's host half and of its oracle.

* `oracle/mt19937_numpy.py` (restatement of randomkit's MT19937 behind numpy's RandomState -- the reference's noise source,
  tc_gan/networks/ssn.py:434-439) against numpy itself: words, doubles, state after the draw.
* the jump-ahead polynomials the library computes on the host (`ssn_mt19937_jump_poly`, no device needed) applied with the
  oracle's numpy arithmetic reproduce sequential stepping of the generator.
"""
import numpy as np
import pytest

from oracle import mt19937_numpy as mt


@pytest.mark.parametrize('seed', [0, 42, 2 ** 32 - 1])
def test_oracle_seeding_and_stream_equal_numpy(seed):
    rs = np.random.RandomState(seed)
    key, pos = mt.seed_state(seed)
    st = rs.get_state()
    assert (st[1] == key).all() and st[2] == pos == 624
    for n in (1, 311, 312, 313, 5, 1000):
        want = rs.random_sample(n)
        got, key, pos = mt.random_sample(key, pos, n)
        assert (want == got).all()
        st = rs.get_state()
        assert (st[1] == key).all() and st[2] == pos


def test_oracle_odd_position_and_fp32_downcast():
    rs = np.random.RandomState(7)
    rs.randint(0, 2 ** 31)                      # one word: the position is odd from here on
    st = rs.get_state()
    assert st[2] % 2 == 1
    want = rs.random_sample(2000)
    got, key, pos = mt.random_sample(st[1], st[2], 2000)
    assert (want == got).all() and (rs.get_state()[1] == key).all() and rs.get_state()[2] == pos
    assert (want.astype('float32') == got.astype('float32')).all()


@pytest.mark.parametrize('nblocks', [1, 4, 33, 256, 1000])
def test_jump_polynomial_equals_stepping(nblocks):
    from tc_gan_amd import clib
    bits = np.zeros(313, dtype=np.uint64)
    assert clib.libssnode.ssn_mt19937_jump_poly(nblocks, bits.ctypes.data) == 0
    poly = np.unpackbits(bits.view(np.uint8), bitorder='little')
    assert poly[19937:].sum() == 0
    key = np.random.RandomState(nblocks).get_state()[1]
    want = mt.advance_blocks(key, nblocks)
    got = mt.apply_jump(key, poly)
    assert (got[1:] == want[1:]).all()
    assert (got[0] ^ want[0]) & np.uint32(0x80000000) == 0


def test_jump_polynomials_compose():
    """t^(624 a) * t^(624 b) = t^(624 (a + b)): jumping twice equals jumping once (how the level ladder is used)."""
    from tc_gan_amd import clib
    def poly(n):
        bits = np.zeros(313, dtype=np.uint64)
        assert clib.libssnode.ssn_mt19937_jump_poly(n, bits.ctypes.data) == 0
        return np.unpackbits(bits.view(np.uint8), bitorder='little')
    key = np.random.RandomState(3).get_state()[1]
    once = mt.apply_jump(key, poly(4 * 5 + 256 * 3))
    twice = mt.apply_jump(mt.apply_jump(key, poly(256 * 3)), poly(4 * 5))
    assert (once[1:] == twice[1:]).all() and (once[0] ^ twice[0]) & np.uint32(0x80000000) == 0


def test_device_continued_randomstate_resolves_a_pending_state_on_first_touch():
    """`utils.DeviceContinuedRandomState` (what `as_randomstate(seed)` builds): same stream as numpy's RandomState; a deferred
    state (the one a device draw will hand back) is installed the first time ANY public attribute is used, exactly once."""
    from tc_gan_amd.utils import DeviceContinuedRandomState, as_randomstate
    r, q = as_randomstate(5), np.random.RandomState(5)
    assert isinstance(r, DeviceContinuedRandomState) and isinstance(r, np.random.RandomState)
    assert as_randomstate(q) is q                      # an existing generator is passed through (cwgan.py:452 shares one)
    assert (r.rand(4) == q.rand(4)).all()
    calls = []
    q.rand(1000)                                       # what the device draw consumes

    def finish(rr):
        calls.append(1)
        np.random.RandomState.set_state(rr, q.get_state())
    r._defer(finish)
    assert not calls
    assert (r.choice(100, 5) == q.choice(100, 5)).all() and calls == [1]
    assert (r.rand(3) == q.rand(3)).all() and calls == [1]
    r._defer(finish)                                   # get_state (checkpoints) resolves too
    st = r.get_state()
    assert calls == [1, 1] and (st[1] == q.get_state()[1]).all() and st[2] == q.get_state()[2]
