"""The mailbox protocol of CGX_COMM_P2P under EVERY interleaving of a small configuration (an explicit-state model, CPU only).

What the GPU tests cannot do on a one-GPU box is let one rank run far ahead of another: the processes time-share the device.
On a real node the ranks run concurrently, and the safety of the things libcgx does without a launcher barrier -- re-laying
the mailbox out for a new problem size, zero-filling the tagged region, starting the next solve while a peer is still in the
last exchange of the previous one -- rests on an argument about what a fast rank can and cannot have pushed (DESIGN.md
section 6; csrc/cgx_context.cpp scrub_tagged_region).  This file states the protocol as a transition system and explores all
schedules: which rank moves next, and in which order the stores in flight arrive.

Modelled (csrc/cgx_solve.cpp, csrc/cgx_context.cpp, csrc/cgx_kernels.hip):
  * a rank's work is a sequence of kernels on one stream: a kernel starts when the previous one has finished, and a kernel
    finishes only when its waits are satisfied and its own stores have arrived;
  * plain all-gather (k_mailbox_allgather): per peer, payload words into the peer's slot [parity][me], then -- after they
    have arrived (fence) -- the flag word; wait for every peer's flag >= epoch; read the slots;
  * fused exchange, flag form (k_update_xr_p2p): per peer INCLUDING myself, payload words, then the chunk flag; wait; read;
  * fused exchange, tagged form: per peer including myself, self-validating words that arrive in ANY order; every word is
    polled until it carries the tag of the epoch; the tag is 1 + epoch mod M with a tiny M here, so that tags repeat all the time;
  * layouts: channel 2 (scalars) at a fixed place; channel 1 (segments) behind it with a slot size that depends on the problem;
    tagged contexts: channel 0 (plain segments) behind channel 1 -- addresses of different layouts overlap;
  * a problem = [tagged: zero-fill of my own channel-1 region] begin-gather, iterations, end-gather, scalar exchange;
    the self-test = [zero-fill] plain gathers, scalar exchange, fused exchanges, scalar exchange.
Checked in every reachable state: a reader only ever accepts exactly what the sender sent for that epoch and position; no
plain double is ever found where a tagged reader polls; nobody waits forever.  Mutants (no zero-fill, plain gathers on the
tagged channel, no closing exchange of the self-test with a launcher that does not synchronise) must FAIL: the model has teeth.
"""
import pytest

P = 2                   # ranks (test_three_ranks sets 3 for a smaller program)
TAG_PERIOD = 3          # tags 1, 2, 3: tag(e) == tag(e - 3) -- far more adversarial than the real 2^32 - 1


class Violation(Exception):
    pass


def tag_of(epoch):
    return 1 + epoch % TAG_PERIOD


class Layout:
    def __init__(self, s1, s0):
        self.s1, self.s0 = s1, s0                       # words per channel-1 slot, per plain slot of a tagged context

    def addr1(self, parity, sender, w):
        return (parity * P + sender) * self.s1 + w

    def addr0(self, parity, sender, w):                 # tagged contexts only: behind the tagged region
        return 2 * P * self.s1 + (parity * P + sender) * self.s0 + w

    def region1(self):
        return range(2 * P * self.s1)


def build_program(tagged, layouts, iters, mutant):
    """The kernels a rank enqueues, in order.  Each entry: (kind, layout, ...)."""
    ops = []
    plain_chan = 0 if (tagged and mutant != "shared_channel") else 1
    st, *problems = layouts
    # self-test
    if tagged and mutant != "no_scrub":
        ops.append(("scrub", st))
    ops += [("plain", st, plain_chan), ("plain", st, plain_chan), ("scalar",), ("fused", st), ("fused", st)]
    if mutant != "selftest_open_end":
        ops.append(("scalar",))
    for L in problems:
        if tagged and mutant != "no_scrub":
            ops.append(("scrub", L))
        ops.append(("plain", L, plain_chan))
        ops += [("fused", L)] * iters
        ops += [("plain", L, plain_chan), ("scalar",)]
    return ops


def explore(tagged, layouts, iters=2, mutant=None, max_states=400000):
    """Exhaustive search over schedules.  Returns the number of states; raises Violation."""
    prog = build_program(tagged, layouts, iters, mutant)
    nops = len(prog)

    # state: (pcs, phases, epochs, mems, flags, cflags, inflight)
    #   pcs[r]      index into prog
    #   phases[r]   0 = not issued, 1 = issued / waiting, 2 = waits satisfied: draining own stores
    #   epochs[r]   (e0, e1, e2) channel epoch counters of rank r
    #   mems[r]     frozenset of (addr, value); value = ('p', chan, epoch, sender, w) plain | ('t', tag, epoch, sender, w) tagged
    #               | ('d', epoch, sender, w) flag-form payload;  scalar channel: addr ('s', parity, sender)
    #   flags       dict (owner, chan, sender) -> epoch ; cflags dict (owner, sender) -> epoch
    #   inflight    frozenset of pushes: (src, dst, remaining payload stores (tuple of (addr, value)), flag or None, ordered?)
    #   accepted[r] frozenset of (q, w) already accepted by the tagged poll of the current op
    def initial():
        return (tuple([0] * P), tuple([0] * P), tuple([(0, 0, 0)] * P), tuple([frozenset()] * P), frozenset(), frozenset(),
                frozenset(), tuple([frozenset()] * P))

    def mem_get(mem, addr):
        for a, v in mem:
            if a == addr:
                return v
        return None

    def mem_set(mem, addr, value):
        return frozenset([(a, v) for a, v in mem if a != addr] + ([(addr, value)] if value is not None else []))

    seen = set()
    stack = [initial()]
    while stack:
        state = stack.pop()
        if state in seen:
            continue
        seen.add(state)
        if len(seen) > max_states:
            raise RuntimeError("state space larger than expected: %d" % len(seen))
        pcs, phases, epochs, mems, flags, cflags, inflight, accepted = state
        fl, cf = dict(flags), dict(cflags)
        succ = []

        # --- a store in flight arrives (any push, any of its unordered payload words; its flag only after all of them) ---
        for push in inflight:
            src, dst, payload, flag, unordered = push
            rest = inflight - {push}
            if payload:
                choices = range(len(payload)) if unordered else [0]
                for i in choices:
                    addr, value = payload[i]
                    left = payload[:i] + payload[i + 1:]
                    nm = list(mems)
                    nm[dst] = mem_set(mems[dst], addr, value)
                    np_ = (src, dst, left, flag, unordered)
                    ni = rest | ({np_} if (left or flag) else set())
                    succ.append((pcs, phases, epochs, tuple(nm), flags, cflags, frozenset(ni), accepted))
            elif flag:
                kind, key, e = flag
                if kind == "f":
                    nf = dict(fl)
                    nf[key] = max(nf.get(key, 0), e)
                    succ.append((pcs, phases, epochs, mems, frozenset(nf.items()), cflags, frozenset(rest), accepted))
                else:
                    nc = dict(cf)
                    nc[key] = max(nc.get(key, 0), e)
                    succ.append((pcs, phases, epochs, mems, flags, frozenset(nc.items()), frozenset(rest), accepted))

        # --- a rank moves ---
        for r in range(P):
            if pcs[r] >= nops:
                continue
            op = prog[pcs[r]]
            kind = op[0]
            e0, e1, e2 = epochs[r]

            def with_rank(pc=None, phase=None, ep=None, mem=None, acc=None, new_pushes=(), nflags=None, ncflags=None):
                npc, nph, nep, nm, nacc = list(pcs), list(phases), list(epochs), list(mems), list(accepted)
                if pc is not None:
                    npc[r] = pc
                if phase is not None:
                    nph[r] = phase
                if ep is not None:
                    nep[r] = ep
                if mem is not None:
                    nm[r] = mem
                if acc is not None:
                    nacc[r] = acc
                return (tuple(npc), tuple(nph), tuple(nep), tuple(nm), flags if nflags is None else nflags,
                        cflags if ncflags is None else ncflags, inflight | frozenset(new_pushes), tuple(nacc))

            own_in_flight = any(p[0] == r for p in inflight)
            if phases[r] == 2:                                  # the kernel ends when its own stores have arrived
                if not own_in_flight:
                    succ.append(with_rank(pc=pcs[r] + 1, phase=0, acc=frozenset()))
                continue

            if kind == "scrub":                                 # hipMemsetAsync of my own channel-1 region (one stream op)
                L = op[1]
                mem = frozenset((a, v) for a, v in mems[r] if not (isinstance(a, int) and a in L.region1()))
                succ.append(with_rank(pc=pcs[r] + 1, phase=0, mem=mem))
                continue

            if kind in ("plain", "scalar"):
                chan = 2 if kind == "scalar" else op[2]
                L = None if kind == "scalar" else op[1]
                if phases[r] == 0:
                    e = (e0, e1, e2)[chan] + 1
                    ep = (e if chan == 0 else e0, e if chan == 1 else e1, e if chan == 2 else e2)
                    pushes = []
                    for q in range(P):
                        if q == r:
                            continue
                        if kind == "scalar":
                            payload = ((("s", e & 1, r), ("p", 2, e, r, 0)),)
                        else:
                            words = L.s0 if chan == 0 else L.s1
                            addr = L.addr0 if chan == 0 else L.addr1
                            payload = tuple((addr(e & 1, r, w), ("p", chan, e, r, w)) for w in range(words))
                        pushes.append((r, q, payload, ("f", (q, chan, r), e), False))
                    succ.append(with_rank(phase=1, ep=ep, new_pushes=pushes))
                else:
                    e = (e0, e1, e2)[chan]
                    if all(fl.get((r, chan, q), 0) >= e for q in range(P) if q != r):
                        for q in range(P):                      # read every peer's slot: it must hold what q sent for e
                            if q == r:
                                continue
                            if kind == "scalar":
                                got = mem_get(mems[r], ("s", e & 1, q))
                                if got != ("p", 2, e, q, 0):
                                    raise Violation("scalar exchange: rank %d read %r for epoch %d from %d" % (r, got, e, q))
                            else:
                                words = L.s0 if chan == 0 else L.s1
                                addr = L.addr0 if chan == 0 else L.addr1
                                for w in range(words):
                                    got = mem_get(mems[r], addr(e & 1, q, w))
                                    if got != ("p", chan, e, q, w):
                                        raise Violation("plain gather ch%d: rank %d read %r, wanted epoch %d sender %d word %d" % (chan, r, got, e, q, w))
                        succ.append(with_rank(phase=2))
                continue

            if kind == "fused":
                L = op[1]
                if phases[r] == 0:
                    e = e1 + 1
                    pushes = []
                    for q in range(P):                          # own mailbox included
                        if tagged:
                            payload = tuple((L.addr1(e & 1, r, w), ("t", tag_of(e), e, r, w)) for w in range(L.s1))
                            pushes.append((r, q, payload, None, True))
                        else:
                            payload = tuple((L.addr1(e & 1, r, w), ("d", e, r, w)) for w in range(L.s1))
                            pushes.append((r, q, payload, ("c", (q, r), e), False))
                    succ.append(with_rank(phase=1, ep=(e0, e, e2), new_pushes=pushes, acc=frozenset()))
                else:
                    e = e1
                    if tagged:
                        todo = [(q, w) for q in range(P) for w in range(L.s1) if (q, w) not in accepted[r]]
                        if not todo:
                            succ.append(with_rank(phase=2))
                        # the order in which a reader accepts its words does not matter (it needs all of them), WHEN it
                        # accepts each does (late polls meet later overwrites): one canonical order, any time
                        for q, w in todo[:1]:
                            got = mem_get(mems[r], L.addr1(e & 1, q, w))
                            if got is not None and got[0] == "p":
                                raise Violation("a plain double where a tagged reader polls: rank %d, epoch %d, found %r" % (r, e, got))
                            if got is not None and got[0] == "t" and got[1] == tag_of(e):
                                if got != ("t", tag_of(e), e, q, w):
                                    raise Violation("tagged reader accepted %r for epoch %d sender %d word %d" % (got, e, q, w))
                                succ.append(with_rank(acc=accepted[r] | {(q, w)}))
                    else:
                        if all(cf.get((r, q), 0) >= e for q in range(P)):
                            for q in range(P):
                                for w in range(L.s1):
                                    got = mem_get(mems[r], L.addr1(e & 1, q, w))
                                    if got != ("d", e, q, w):
                                        raise Violation("fused exchange (flags): rank %d read %r, wanted epoch %d sender %d word %d" % (r, got, e, q, w))
                            succ.append(with_rank(phase=2))
                continue

        if not succ:
            if any(pc < nops for pc in pcs):
                where = [(r, prog[pcs[r]][0], phases[r]) for r in range(P) if pcs[r] < nops]
                raise Violation("deadlock: nobody can move, still running: %r" % where)
            continue
        stack.extend(succ)
    return len(seen)


# self-test layout, then three problems: a slot size that shrinks and grows again, so that regions of different layouts overlap
LAYOUTS = [Layout(2, 1), Layout(2, 1), Layout(1, 2), Layout(3, 1)]


@pytest.mark.parametrize("tagged", [False, True])
def test_every_interleaving_of_two_ranks_is_safe(tagged):
    n = explore(tagged, LAYOUTS, iters=2)
    print("tagged=%s: %d states explored" % (tagged, n))
    assert n > 1000


def test_without_the_zero_fill_a_stale_tagged_word_is_accepted():
    """Mutant: no scrub_tagged_region.  With tags repeating every 3 epochs a word the self-test or another layout left behind
    passes for the current epoch somewhere in the schedule space -- the model must find it."""
    with pytest.raises(Violation):
        explore(True, LAYOUTS, iters=2, mutant="no_scrub")


def test_plain_gathers_on_the_tagged_channel_are_caught():
    """Mutant: round 3's layout, plain all-gathers through the slots a tagged reader polls."""
    with pytest.raises(Violation):
        explore(True, LAYOUTS, iters=2, mutant="shared_channel")


def test_why_a_passing_self_test_ends_with_an_exchange_on_the_scalar_channel():
    """Mutant: the self-test returns right after its last fused exchange (rounds 1-3) and the launcher puts no barrier there
    (the round-3 test workers did not): with a next layout whose slots are smaller, the begin-gather of a fast rank lands in the
    self-test slots a slow peer is still reading.  With the closing exchange (round 4) the same layouts are safe in both forms."""
    lays = [Layout(2, 1), Layout(1, 2), Layout(2, 1)]
    with pytest.raises(Violation):          # (which wrong read is met first depends on the search order)
        explore(False, lays, iters=2, mutant="selftest_open_end")
    assert explore(False, lays, iters=2) > 1000 and explore(True, lays, iters=2) > 1000


def test_three_ranks():
    """Three ranks, a shorter program (the state space grows fast): self-test and one problem, both forms."""
    global P
    P = 3
    try:
        assert explore(False, [Layout(1, 1), Layout(1, 1)], iters=1) > 10000
        assert explore(True, [Layout(1, 1), Layout(1, 1)], iters=1) > 10000
    finally:
        P = 2
