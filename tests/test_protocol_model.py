"""The arithmetic of the tagged-word hand-over (conjugate-gradient_amd/csrc/cgx_kernels.hip, "Tagged words"), modelled on the
host: what a reader accepts and what it cannot mistake.  The kernels themselves are exercised on the GPU
(tests/test_gpu_p2p.py); this file pins the invariants their comments claim."""
import struct

import numpy as np

NAN_HI = 0xFFF80000


def tag_of(epoch):
    return (epoch & 0xFFFFFFFF) ^ NAN_HI


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
    for e in (1, 2, 12345, 2**19 - 1, 2**31, 2**32 - 1, 2**32 + 5):
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
    assert accept(0, 0, new) is None                      # a zero-filled mailbox holds no valid word (tag 0 is epoch 0xFFF80000)
    assert tag_of(0xFFF80000) == 0


def test_tags_are_a_bijection_of_the_low_32_epoch_bits():
    es = np.arange(0, 1 << 20, 4099, dtype=np.uint64)
    tags = {tag_of(int(e)) for e in es}
    assert len(tags) == len(es)
    assert tag_of(5) == tag_of(5 + 2**32) and tag_of(5) != tag_of(6)      # collisions only 2^32 epochs apart


def test_no_finite_double_left_in_a_slot_passes_for_a_tagged_word_of_the_first_2_19_epochs():
    """The set-up phases store plain doubles into the same slots.  A plain double's upper 32 bits equal a tag of an epoch below
    2^19 only if they lie in [0xFFF80000, 0xFFFFFFFF]: sign 1, exponent all ones, top mantissa bit 1 -- a quiet NaN."""
    for e in (1, 2, 1000, 2**19 - 1):
        hi = tag_of(e)
        assert 0xFFF80000 <= hi <= 0xFFFFFFFF
        for lo in (0, 1, 0xFFFFFFFF):
            v = struct.unpack("<d", struct.pack("<Q", (hi << 32) | lo))[0]
            assert np.isnan(v)
    rng = np.random.default_rng(1)
    finite = rng.standard_normal(100000) * 10.0 ** rng.integers(-300, 300, 100000)
    his = (finite.view(np.uint64) >> np.uint64(32)).astype(np.uint64)
    assert not np.any(his >= np.uint64(0xFFF80000))
