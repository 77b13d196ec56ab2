"""CPU: the counted ``s_waitcnt lgkmcnt(N)`` of conv3_kernel's hand-written LDS -> MFMA pipeline (csrc/conv2.hip, ``C3Seq``).

The kernel issues its fragment reads (inline-asm ds_read_b128, invisible to the compiler's own wait insertion) PD sub-steps ahead and waits,
before the MFMAs of sub-step s, until at most ``wait_count(s)`` LDS operations are outstanding.  LDS operations retire in issue order, so
the wait is correct iff every read sub-step s consumes -- its own weight fragments af(s) and, on the first sub-step of a group, the group's
patch rows brow(g) -- is among the retired ones, i.e. NOT among the last ``wait_count(s)`` issued.  This transcribes ``C3Seq`` and replays the
kernel's issue order for every configuration the library instantiates (3x3: 6 groups x 3 row shifts, stride 2: 4 x 2; 1 or 2 accumulator
rows; 2 or 4 pixel rows per wave; fragments one or two sub-steps ahead); it also checks that the wait is never stricter than needed by more
than the reads of one sub-step (a wait of 0 everywhere would be correct and slow) and that the hardware's 4-bit counter is respected."""
import pytest


class Seq:
    """csrc/conv2.hip C3Seq<NG, NDY, MF, R, PD>, line for line."""

    def __init__(self, NG, NDY, MF, R, PD):
        self.NG, self.NDY, self.MF, self.R, self.PD = NG, NDY, MF, R, PD
        self.NSUB = NG * NDY

    def issued_through(self, s):
        n = self.R + self.MF * min(self.PD, self.NSUB)
        for t in range(s + 1):
            if t + self.PD < self.NSUB:
                n += self.MF
            if t % self.NDY == 0 and t // self.NDY + 1 < self.NG:
                n += self.R
        return n

    def last_af(self, s):
        if s < self.PD:
            return self.R + self.MF * (s + 1)
        return self.issued_through(s - self.PD - 1) + self.MF

    def last_brow(self, g):
        if g == 0:
            return self.R
        t = (g - 1) * self.NDY
        return self.issued_through(t - 1) + (self.MF if t + self.PD < self.NSUB else 0) + self.R

    def wait_count(self, s):
        need = self.last_af(s)
        if s % self.NDY == 0 and self.last_brow(s // self.NDY) > need:
            need = self.last_brow(s // self.NDY)
        return min(15, self.issued_through(s) - need)


def replay(NG, NDY, MF, R, PD):
    """The kernel's issue order as a list of read tags; -> per sub-step (reads issued so far, tags it consumes)."""
    NSUB = NG * NDY
    order = [("brow", 0)] * R
    for s in range(min(PD, NSUB)):
        order += [("af", s)] * MF
    steps = []
    for s in range(NSUB):
        g, dy = divmod(s, NDY)
        if s + PD < NSUB:
            order += [("af", s + PD)] * MF
        if dy == 0 and g + 1 < NG:
            order += [("brow", g + 1)] * R
        needs = {("af", s)} | ({("brow", g)} if dy == 0 else set())
        steps.append((len(order), needs, list(order)))
    return steps


CONFIGS = [(NG, NDY, MF, RPW + NDY - 1, PD) for (NG, NDY) in ((6, 3), (4, 2)) for MF in (1, 2) for RPW in (2, 4) for PD in (1, 2)]


@pytest.mark.parametrize("NG,NDY,MF,R,PD", CONFIGS)
def test_counted_waits_cover_every_operand_and_nothing_more(NG, NDY, MF, R, PD):
    sq = Seq(NG, NDY, MF, R, PD)
    slack = 0
    for s, (issued, needs, order) in enumerate(replay(NG, NDY, MF, R, PD)):
        assert sq.issued_through(s) == issued, (s, sq.issued_through(s), issued)
        k = sq.wait_count(s)
        assert 0 <= k <= 15
        retired = order[:issued - k]                       # in-order retirement: everything but the last k issued has landed
        for tag in needs:
            last = max(i for i, t in enumerate(order) if t == tag)
            assert last < len(retired), f"sub-step {s}: {tag} may still be in flight behind lgkmcnt({k})"
        # tightness: the youngest read this sub-step needs is the LAST retired one (or the 4-bit counter clamps the wait)
        youngest = max(max(i for i, t in enumerate(order) if t == tag) for tag in needs)
        if issued - youngest - 1 <= 15:
            assert youngest == len(retired) - 1, (s, youngest, len(retired))
        else:
            slack += 1
    assert slack <= 2                                       # (only the deepest configuration ever hits the clamp)
    # the last sub-step waits for everything: nothing of this stage is in flight when the K-step ends
    assert sq.wait_count(NG * NDY - 1) == 0
