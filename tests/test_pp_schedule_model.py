"""Hazard model of the ping-pong GEMM loop's schedule (csrc/gemm_pp.hip), CPU-only.

The kernel's two wave groups run half a K step apart; LDS-DMA pieces stay in flight across barriers and are retired by counted
vmcnt waits; the LDS is a ring of slots.  Its file header argues the RAW / WAR hazards by interval number.  This test replays that
argument as a discrete model over many steps, for the shipped parameters (ring of 4 slots, pieces of step k + 3 issued in M(k), wait
for step k + 1 at the end of L(k)), and shows that the neighbouring parameter choices one might be tempted by are NOT safe -- i.e.
that the model can tell the difference.

Model (intervals are separated by workgroup barriers; within an interval the two groups run concurrently, unordered):
  group 0 runs L(k) in interval 2k and M(k) in 2k + 1;  group 1 runs L(k) in 2k + 1 and M(k) in 2k + 2.
  L(k): reads slot k % R (all fragment reads retired before the barrier that ends the interval);
        ends with a wait for THIS group's pieces of step k + W (they have landed when the barrier is passed).
  M(k): issues THIS group's pieces of step k + D into slot (k + D) % R (LDS is written at any time from issue to the matching wait).
  prologue (before interval 0): both groups issue steps 0 .. D - 1 and wait for step 0.
Hazards:
  RAW: a read of step s in interval t needs BOTH groups' waits for step s in intervals < t (a wait in the same interval is not ordered
       against the other group's read).
  WAR: pieces of step s go to the slot that held step s - R: every read of step s - R must be in an interval < the issue's.
  in-flight: a wait for step s must come after its issue (same wave, program order) -- else the counted vmcnt would wait for the wrong
       pieces; and at the wait, the number of younger steps still in flight must equal what the kernel counts (D - W - 1 = 1: wait_keep1).
"""
import pytest


def schedule(nsteps, R, D, W, issue_in='M'):
    """-> (reads, issues, waits): lists of (interval, group, step); prologue events carry interval -1.
    issue_in = 'L': the pieces of step k + D leave at the TOP of L(k) instead of in M(k) (the three-slot kernel, NSL = 3)"""
    reads, issues, waits = [], [], []
    for g in (0, 1):
        for s in range(D):
            issues.append((-1, g, s))
        waits.append((-1, g, 0))
        for k in range(nsteps):
            tl, tm = 2 * k + g, 2 * k + g + 1
            reads.append((tl, g, k))
            waits.append((tl, g, k + W))
            issues.append((tm if issue_in == 'M' else tl, g, k + D))
    return reads, issues, waits


def hazards(nsteps, R, D, W, issue_in='M'):
    reads, issues, waits = schedule(nsteps, R, D, W, issue_in)
    errs = []
    wait_at = {(g, s): t for t, g, s in waits}
    issue_at = {(g, s): t for t, g, s in issues}
    last_read = {}
    for t, g, s in reads:
        last_read[s] = max(last_read.get(s, -9), t)
    for t, g, s in reads:                                   # RAW
        for gg in (0, 1):
            w = wait_at.get((gg, s))
            if w is None or not w < t:
                errs.append(f'RAW: group {g} reads step {s} in interval {t}, group {gg} waits for it in {w}')
    for t, g, s in issues:                                  # WAR
        prev = s - R
        if prev >= 0 and prev < nsteps and not last_read[prev] < t:
            errs.append(f'WAR: group {g} issues step {s} into slot {s % R} in interval {t}, step {prev} is read there until {last_read[prev]}')
    for (g, s), w in wait_at.items():                       # program order of issue and wait inside a wave, and the vmcnt count
        i = issue_at.get((g, s))
        if i is None or i > w or (i == w and i >= 0 and issue_in == 'M'):   # (an issue in M(k) is after L(k)'s wait; one at the top of L(k) before it)
            errs.append(f'order: group {g} waits for step {s} in {w} but issues it in {i}')
        in_flight = sum(1 for (gg, ss), ii in issue_at.items() if gg == g and ss > s and (ii < w or ii == -1))
        if w >= 0 and in_flight != D - W - 1:
            errs.append(f'count: group {g} at its wait for step {s}: {in_flight} younger steps in flight, the kernel counts {D - W - 1}')
    return errs


def test_shipped_schedule_is_hazard_free():
    # ring of 4 slots, distance 3, wait for the next step: what csrc/gemm_pp.hip does (wait_keep1 leaves exactly one step in flight)
    for n in (2, 3, 10, 57, 360):
        assert hazards(n, R=4, D=3, W=1) == [], hazards(n, 4, 3, 1)[:3]


def test_three_slot_schedule_is_hazard_free():
    # NSL = 3 (two blocks per CU): ring of 3 slots, pieces of step k + 2 leave at the top of L(k), wait for step k + 1 at its end with one
    # step (k + 2) still in flight -- pp_wait_vm<NPM>
    for n in (2, 3, 10, 57, 360):
        assert hazards(n, R=3, D=2, W=1, issue_in='L') == [], hazards(n, 3, 2, 1, 'L')[:3]
    # the same ring with the four-slot kernel's distance, or the issue one step further ahead, is not
    assert any(e.startswith('WAR') for e in hazards(40, 3, 3, 1, 'L'))
    assert any(e.startswith('WAR') for e in hazards(40, 2, 2, 1, 'L'))
    # issued in M(k) with distance 2 is safe as well, but nothing is in flight at the wait (D - W - 1 = 0: the first version, DMA latency exposed)
    assert hazards(40, 3, 2, 1, 'M') == []


@pytest.mark.parametrize('R,D,W,kind', [
    (4, 4, 1, 'WAR'),       # one step more of prefetch: the slot is still being read by the other group
    (3, 3, 1, 'WAR'),       # one slot fewer
    (4, 3, 2, 'RAW'),       # waiting one step later: the other group reads before the wait
    (4, 3, 0, 'RAW'),       # waiting for the step that was just read instead of the next one
    (4, 3, 3, 'order'),     # waiting for a step whose pieces leave only later in the same step
])
def test_neighbouring_schedules_are_caught(R, D, W, kind):
    errs = hazards(40, R, D, W)
    assert any(e.startswith(kind) for e in errs), errs[:3]
