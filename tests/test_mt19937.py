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


@pytest.mark.parametrize('start_odd', [False, True])
def test_oracle_choice2_signs_equal_numpy_and_the_stream_goes_on(start_odd):
    """zs_in of the heterogeneous-input models behind zs (ssn.py:714-715, 764-767): one output word per element, its low bit."""
    rs = np.random.RandomState(3)
    if start_odd:
        rs.randint(0, 2 ** 31)
    rs.random_sample(1000)
    st = rs.get_state()
    key, pos = st[1], st[2]
    for n in (1, 7, 624, 1249):
        want = rs.choice(2, (n,)) * 2 - 1
        got, key, pos = mt.choice2_signs(key, pos, n)
        assert (want == got).all()
        assert (rs.get_state()[1] == key).all() and rs.get_state()[2] == pos
        w2 = rs.random_sample(5)                   # (an odd number of words consumed: the doubles behind it pair up from there)
        g2, key, pos = mt.random_sample(key, pos, 5)
        assert (w2 == g2).all()


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


@pytest.mark.parametrize('world', [1, 2, 8])
@pytest.mark.parametrize('pos', [0, 1, 311, 623, 624])
def test_draw_plan_positions_and_rank_segments(world, pos):
    """`ssn_mt19937_plan` (the host arithmetic of a device draw, no GPU): the position after the draw and the number of
    regenerations equal numpy's own (`get_state()` before / after `random_sample`), and the segments the ranks of a
    data-parallel job generate cover every block that holds one of their rows' words -- SURVEY 8e: each rank its own rows of the
    GLOBAL draw, all ranks the same state afterwards."""
    from tc_gan_amd import clib
    B, M = 64, 100
    total = B * M * M
    rs = np.random.RandomState(5)
    rs.randint(0, 2 ** 31, size=1000)                     # away from the seed state
    st = rs.get_state()
    rs.set_state((st[0], st[1], pos, st[3], st[4]))
    rs.random_sample(total)
    want_pos = rs.get_state()[2]
    ends = set()
    for rank in range(world):
        per = B // world
        skip, count = rank * per * M * M, per * M * M
        out = np.zeros(7, dtype=np.int64)
        assert clib.libssnode.ssn_mt19937_plan(pos, total, skip, count, out.ctypes.data) == 0
        new_pos, b_f, seg_blocks, s_lo, s_hi, b_lo, b_hi = (int(v) for v in out)
        ends.add((new_pos, b_f))
        assert new_pos == want_pos and b_f == max(0, (pos + 2 * total - 1) // 624) * (pos + 2 * total > 624)
        # the rank's words: stream words [2 skip, 2 (skip + count)) sit at positions pos + w of the block grid
        assert b_lo == (pos + 2 * skip) // 624 and b_hi == (pos + 2 * (skip + count) - 1) // 624
        assert seg_blocks in (128, 256, 512, 1024)
        # segment s regenerates blocks s * seg_blocks + 1 .. (s + 1) * seg_blocks (segment 0 also holds block 0)
        first = 0 if s_lo == 0 else s_lo * seg_blocks + 1
        assert first <= b_lo and b_hi <= (s_hi + 1) * seg_blocks and (s_lo == 0 or b_lo > (s_lo - 1) * seg_blocks + seg_blocks)
    assert len(ends) == 1                                  # every rank hands back the same generator state


@pytest.mark.parametrize('kind', [1, 2])
@pytest.mark.parametrize('world', [1, 2, 8])
@pytest.mark.parametrize('pos', [0, 311, 624])
def test_draw_plan_with_the_input_noise_behind_zs(kind, world, pos):
    """`ssn_mt19937_plan_tail`: zs and, behind it, zs_in of the heterogeneous-input models (kind 1: `choice(2, (B, M))`, one word
    per element; kind 2: `rand(B, M)`, two) -- the position and the regenerations after BOTH draws equal numpy's; a rank whose
    rows of the two draws lie in one stretch of the stream plans one launch that covers both, every rank the same end state."""
    from tc_gan_amd import clib
    B, M = 64, 101
    total = B * M * M
    rs = np.random.RandomState(9)
    rs.randint(0, 2 ** 31, size=700)
    st = rs.get_state()
    rs.set_state((st[0], st[1], pos, st[3], st[4]))
    rs.random_sample(total)
    if kind == 1:
        rs.choice(2, (B, M))
    else:
        rs.rand(B, M)
    want_pos = rs.get_state()[2]
    words = 2 * total + (1 if kind == 1 else 2) * B * M
    ends = set()
    for rank in range(world):
        per = B // world
        out = np.zeros(7, dtype=np.int64)
        assert clib.libssnode.ssn_mt19937_plan_tail(pos, total, rank * per * M * M, per * M * M, kind, B * M, rank * per * M, per * M,
                                                    out.ctypes.data) == 0
        new_pos, b_f, seg_blocks, s_lo, s_hi, b_lo, b_hi = (int(v) for v in out)
        ends.add((new_pos, b_f))
        assert new_pos == want_pos and b_f == (0 if pos + words <= 624 else (pos + words - 1) // 624)
        last = 2 * total + (1 if kind == 1 else 2) * (rank + 1) * per * M - 1       # last word of the rank's rows of the tail
        assert b_lo == (pos + 2 * rank * per * M * M) // 624 and b_hi == (pos + last) // 624
        first = 0 if s_lo == 0 else s_lo * seg_blocks + 1
        assert first <= b_lo and b_hi <= (s_hi + 1) * seg_blocks
    assert len(ends) == 1
