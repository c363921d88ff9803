"""The arithmetic of the tagged-word hand-over (conjugate-gradient_amd/csrc/cgx_kernels.hip, "Tagged words"), modelled on the
host: what a reader accepts and what it cannot mistake.  The kernels themselves are exercised on the GPU
(tests/test_gpu_p2p.py); this file pins the invariants their comments claim."""
import struct

import numpy as np

M = 2 ** 32 - 1


def tag_of(epoch):
    """p2p_tag (cgx_kernels.hip, round 4): 1 + epoch mod (2^32 - 1) -- 1 ... 2^32 - 1, never 0."""
    return epoch % M + 1


def pack(value, tag):
    bits = struct.unpack("<Q", struct.pack("<d", value))[0]
    return (bits & 0xFFFFFFFF) | (tag << 32), (bits >> 32) | (tag << 32)


def accept(w0, w1, tag):
    """A reader's view of two 8-byte words: the double if BOTH carry the tag, else None (keep polling)."""
    if (w0 >> 32) != tag or (w1 >> 32) != tag:
        return None
    return struct.unpack("<d", struct.pack("<Q", (w0 & 0xFFFFFFFF) | ((w1 & 0xFFFFFFFF) << 32)))[0]


def test_round_trip_is_bit_exact():
    rng = np.random.default_rng(0)
    vals = list(rng.standard_normal(200) * 10.0 ** rng.integers(-300, 300, 200)) + [0.0, -0.0, np.inf, -np.inf, 5e-324, 1.7976931348623157e308]
    for e in (1, 2, 12345, 2**19 - 1, 0xFFF80000, 2**31, 2**32 - 2, 2**32 - 1, 2**32 + 5, 2 * M):
        t = tag_of(e)
        for v in vals:
            got = accept(*pack(float(v), t), t)
            assert struct.pack("<d", got) == struct.pack("<d", float(v))
    t = tag_of(77)
    nan_bits = struct.unpack("<d", struct.pack("<Q", 0x7FF8000000000123))[0]
    assert struct.pack("<d", accept(*pack(nan_bits, t), t)) == struct.pack("<Q", 0x7FF8000000000123)   # NaN payloads survive


def test_a_torn_or_stale_pair_is_never_accepted():
    """Each word validates itself: a pair made of one new and one old word (the two halves of a double arriving at different
    times, or a slot still holding the epoch of two exchanges ago) is rejected, whichever half is the old one."""
    new, old = tag_of(1000), tag_of(998)
    n0, n1 = pack(3.25, new)
    o0, o1 = pack(-7.5, old)
    assert accept(n0, n1, new) == 3.25
    assert accept(n0, o1, new) is None and accept(o0, n1, new) is None and accept(o0, o1, new) is None
    assert accept(0, 0, new) is None


def test_a_zero_filled_slot_is_never_accepted_at_any_epoch():
    """The mailbox is zero-filled at creation and the tagged region again whenever it is laid out anew: no tag is 0 -- also not
    at the epochs where round 3's tag (epoch XOR 0xFFF80000) was, nor where the 32-bit tag wraps."""
    for e in (1, 2, 2 ** 19, 0xFFF80000, M - 1, M, M + 1, 2 ** 32, 2 * M, 2 * M + 1, 3 * M - 1, 2 ** 63):
        t = tag_of(e)
        assert 1 <= t <= M
        assert accept(0, 0, t) is None and accept(*pack(1.5, t), t) == 1.5


def test_the_two_epochs_that_share_a_position_never_share_a_tag():
    """A position is rewritten in every epoch of its parity, so the newest stale word a reader can meet is that of epoch e - 2;
    the tags of e and e - 2 (and e - 1, e - 4) differ for every e, across the wrap of the tag as well."""
    for base in (2, 1000, 2 ** 19, 0xFFF80000, M, 2 ** 32, 2 * M, 7 * M):
        for e in range(base - 6, base + 7):
            if e < 4:
                continue
            assert len({tag_of(e), tag_of(e - 1), tag_of(e - 2), tag_of(e - 4)}) == 4, e
            stale = pack(-7.5, tag_of(e - 2))
            assert accept(*stale, tag_of(e)) is None


def test_tags_repeat_only_a_whole_period_apart():
    es = np.arange(1, 1 << 22, 4099, dtype=np.uint64)
    assert len({tag_of(int(e)) for e in es}) == len(es)
    assert tag_of(5) == tag_of(5 + M) and tag_of(5) != tag_of(5 + 2 ** 32) and tag_of(5) != tag_of(6)
