"""Model check of the hand-synchronised weight ring (csrc/x3_common.hpp: WeightRing, gemm_x3_roll, gemm_x2_roll) on the CPU.

The ring is synchronised by counting: an `acquire` is an `s_waitcnt vmcnt(keep)` on the wave's own DMA queue plus a workgroup
barrier, and a buffer is refilled only behind a barrier that follows every wave's last read of it (since round 6 the waves also
wait for those reads to RETURN -- lgkmcnt(0) -- before that barrier, so the ordering this model checks is sufficient by itself).  This test restates the CONTROL FLOW of the two GEMM loops (which
section loads which fragments, where the acquires sit, which sections issue a refill chunk) as an event list in program order --
the same for all four waves, which meet at every barrier -- and checks, for sequences of matrices like the ones the engines run:

  A. counting: when an acquire returns, `issued - keep` DMA chunks have landed (in-order completion), and that covers every
     chunk of every stage the acquire hands out;
  B. reads: a fragment / record read of stage t comes after the acquire that handed out stage t;
  C. write-after-read: the refill of a buffer is ISSUED only after a barrier that follows the last read of the stage the buffer
     held (so every wave's reads of it were issued before any wave's refill of it), and nothing reads that stage afterwards;
  D. bookkeeping: every matrix acquires and refills exactly its number of stages, so matrices can be chained in any order.
"""
import pytest


class Ring:
    def __init__(self, P, kbuf, lag):
        self.P, self.kbuf, self.lag = P, kbuf, lag
        self.ev = []                       # program order: ("issue", stage, chunk) | ("acq", first_stage, n, keep) | ("read", stage)
        self.issued_stages = 0             # complete stages whose chunks were all issued
        self.partial = 0
        self.acquired = 0
        for _ in range(kbuf - 1 - lag):    # WeightRing::init
            self.issue_stage()

    def issue_chunk(self, c):
        assert c == self.partial, "chunks of a stage are issued in order"
        self.ev.append(("issue", self.issued_stages, c))
        self.partial += 1
        if self.partial == self.P:
            self.partial, self.issued_stages = 0, self.issued_stages + 1

    def issue_stage(self):
        for c in range(self.P):
            self.issue_chunk(c)

    def acquire(self):
        keep = (self.kbuf - 2 - self.lag) * self.P
        self.ev.append(("acq", self.acquired, 1, keep))
        self.acquired += 1
        return self.acquired - 1

    def acquire2(self):
        keep = (self.kbuf - 3 - self.lag) * self.P
        assert keep >= 0
        self.ev.append(("acq", self.acquired, 2, keep))
        self.acquired += 2
        return self.acquired - 2, self.acquired - 1

    def read(self, stage):
        self.ev.append(("read", stage))


def gemm_x3_roll(r, KS, L):
    P, G = r.P, KS * r.P
    st = {0: r.acquire()}
    r.issue_stage()

    def load_pair(q):
        r.read(st[(q // P) & 1])           # hi and lo fragments of the pair: the k-step's own stage
    for q in range(L):
        load_pair(q)
    for g in range(G):
        s, p = divmod(g, P)
        if p == P - L and s + 1 < KS:
            st[(s + 1) & 1] = r.acquire()
        if g + L < G:
            load_pair(g + L)
        since = g - (P - L)
        if since >= 0 and since // P + 1 < KS:
            r.issue_chunk(since % P)


def gemm_x2_roll(r, KS2, KS3, L, paired=True):
    P, KS = r.P, KS2 + KS3
    G = KS * P
    st = {}
    if KS >= 2 and paired:
        st[0], st[1] = r.acquire2()
    else:
        st[0] = r.acquire()
    r.issue_stage()

    def load_pair(q):
        s = q // P
        r.read(st[s & 1])                  # f16 hi fragments (x3 tail: hi and lo) of k-step s
        if s < KS2 and s % 2 == 1:         # the K-tile's fp6 record: halves in stages s - 1 and s
            r.read(st[(s - 1) & 1])
    for q in range(L):
        load_pair(q)
    for g in range(G):
        s, p = divmod(g, P)
        if p == P - L and s + 1 < KS:
            if not paired:
                st[(s + 1) & 1] = r.acquire()
            elif s % 2 == 1:
                if s + 2 < KS:
                    st[(s + 1) & 1], st[s & 1] = r.acquire2()
                else:
                    st[(s + 1) & 1] = r.acquire()
        if g + L < G:
            load_pair(g + L)
        since = g - (P - L)
        if since >= 0 and since // P + 1 < KS:
            r.issue_chunk(since % P)


def check(r, total_stages):
    P, kbuf = r.P, r.kbuf
    issued = 0                             # chunks issued so far (program order)
    handed = 0                             # stages handed out by acquires so far
    last_barrier = -1
    last_read = {}                         # stage -> index of its last read
    first_issue = {}                       # stage -> index of the first chunk issue
    for i, e in enumerate(r.ev):
        if e[0] == "issue":
            issued += 1
            first_issue.setdefault(e[1], i)
        elif e[0] == "acq":
            _, first, n, keep = e
            assert first == handed
            handed += n
            assert issued - keep >= handed * P, f"A: acquire of stages {first}..{first + n - 1} returns before they landed"
            last_barrier = i
        else:
            assert e[1] < handed, f"B: read of stage {e[1]} before its acquire"
            last_read[e[1]] = i
    assert handed == total_stages and r.partial == 0
    assert r.issued_stages == total_stages + kbuf - 1 - r.lag, "D: one refill per acquired stage"
    # C: the refill of the buffer that held stage t is stage t + kbuf
    barriers = [i for i, e in enumerate(r.ev) if e[0] == "acq"]
    for t, rd in last_read.items():
        if t + kbuf not in first_issue:
            continue
        fi = first_issue[t + kbuf]
        assert rd < fi, f"C: stage {t} is read after its buffer's refill was issued"
        assert any(rd < b < fi for b in barriers), f"C: no barrier between the last read of stage {t} and the refill of its buffer"


@pytest.mark.parametrize("NT,kbuf,lag,L", [(8, 5, 1, 2), (8, 8, 1, 2), (4, 5, 1, 2), (8, 4, 1, 2), (8, 8, 1, 3), (8, 7, 1, 4)])
@pytest.mark.parametrize("paired", [True, False])
def test_x2_rolls_chain_safely(NT, kbuf, lag, L, paired):
    """The x2 engines' rings (LAG = 1): the synthesis network's matrices (16- and 8-k-step x2 GEMMs), and the field network's mix
    of x3 input layers, x2 layers and the colour layer's 16 + 1 k-steps."""
    if kbuf - 3 - lag < 0 and paired:
        pytest.skip("ring too shallow for paired acquires (static_assert in acquire2)")
    L = min(L, NT // 2)
    r = Ring(NT // 2, kbuf, lag)
    total = 0
    for KS2, KS3, kind in [(16, 0, "x2"), (8, 0, "x2"), (8, 0, "x2"), (16, 0, "x2"), (2, 0, "x3"), (1, 0, "x3"), (16, 1, "x2"), (16, 0, "x2"),
                           (2, 0, "x2"), (16, 0, "x2")]:
        if kind == "x2":
            gemm_x2_roll(r, KS2, KS3, L, paired)
        else:
            gemm_x3_roll(r, KS2, L)
        total += KS2 + KS3
    check(r, total)


@pytest.mark.parametrize("NT,kbuf,L", [(8, 7, 2), (8, 4, 2), (4, 7, 2), (2, 7, 1), (8, 6, 4)])
def test_x3_rolls_chain_safely(NT, kbuf, L):
    """The three-product engines and the training convolutions (LAG = 0)."""
    r = Ring(NT // 2, kbuf, 0)
    total = 0
    for KS in (16, 8, 1, 2, 16, 4, 8):
        gemm_x3_roll(r, KS, min(L, NT // 2))
        total += KS
    check(r, total)


def test_the_checker_catches_a_broken_protocol():
    """An acquire that keeps one stage too many in flight, and a ring without the LAG buffer under an x2 roll, must both fail."""
    r = Ring(4, 5, 1)
    gemm_x2_roll(r, 16, 0, 2)
    r.ev = [(e[0], e[1], e[2], e[3] + 4) if e[0] == "acq" else e for e in r.ev]
    with pytest.raises(AssertionError, match="A:"):
        check(r, 16)
    r = Ring(4, 4, 0)                      # LAG = 0: the record half of stage s - 1 is read while its buffer is being refilled
    gemm_x2_roll(r, 16, 0, 2, paired=False)
    with pytest.raises(AssertionError, match="C:"):
        check(r, 16)
