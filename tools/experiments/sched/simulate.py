"""Discrete-event model of a per-draw launch on one MI355X: 8 XCDs x 128 resident waves, workgroups issued IN ORDER with workgroup b
on XCD b mod 8 (tools/experiments/xcc_id.hip), the dispatcher waiting whenever the next workgroup's XCD has no free slot.  Compares

  classic   one workgroup per chain, all N transitions (the committed kernel), chains in launch order (longest first or not)
  v3        one SEGMENT per workgroup; a workgroup with nothing to take polls until ALL chains are finished
            (tools/experiments/sched/segment_scheduler_v3_workgroup_per_segment.patch: did not finish at bench size)
  v4        the same with a per-XCD count of chains in flight: nothing to take and nothing of this XCD's in flight -> leave at once,
            otherwise poll for at most `poll_us`                      (segment_scheduler_v4_untested.patch)

on BASELINE configs[1]'s shape: 4096 chains x 1000 transitions, 41 µs per transition, one chain at `slow` x the work.

    python tools/experiments/sched/simulate.py            # prints makespans (ms) and workgroups used
"""
import heapq
import sys

XCDS, SLOTS = 8, 128


def classic(work_ms, order):
    """work_ms[c]: the chain's time; order: launch order.  In-order issue, b -> XCD b % 8."""
    free_at = [[0.0] * SLOTS for _ in range(XCDS)]
    for x in range(XCDS):
        heapq.heapify(free_at[x])
    t_issue, end = 0.0, 0.0
    for b, c in enumerate(order):
        x = b % XCDS
        t = max(t_issue, heapq.heappop(free_at[x]))         # the dispatcher waits for a slot on THIS workgroup's XCD
        t_issue = t
        heapq.heappush(free_at[x], t + work_ms[c])
        end = max(end, t + work_ms[c])
    return end


def segmented(work_ms, order, long_flag, nseg, per_xcd_count, poll_us, grid, empty_us=1.0, limit_ms=5000.0, xcd_speed=None):
    """One segment per workgroup.  Returns (makespan or None if chains were left / the time limit hit, workgroups issued).
    xcd_speed[x]: relative speed of XCD x (1.0 = nominal) — real XCDs do not run in lock step."""
    C = len(work_ms)
    xcd_speed = xcd_speed or [1.0] * XCDS
    seg_ms = [work_ms[c] if long_flag[c] else work_ms[c] / nseg for c in range(C)]
    left = [1 if long_flag[c] else nseg for c in range(C)]
    fresh = 0
    rings = [[] for _ in range(XCDS)]            # (ready time, chain)
    inflight = [0] * XCDS
    unfinished = C
    slots = [[0.0] * SLOTS for _ in range(XCDS)]
    for x in range(XCDS):
        heapq.heapify(slots[x])
    events = []                                   # (time, kind, xcd, chain): segment ends
    t_issue, end, b = 0.0, 0.0, 0

    def flush(until):
        nonlocal unfinished, end
        while events and events[0][0] <= until:
            t, x, c = heapq.heappop(events)
            left[c] -= 1
            if left[c] == 0:
                unfinished -= 1; inflight[x] -= 1
                end = max(end, t)
            else:
                rings[x].append((t, c))
    while b < grid and unfinished > 0:
        x = b % XCDS
        t = max(t_issue, heapq.heappop(slots[x]))
        if t > limit_ms:
            return None, b
        t_issue = t
        flush(t)
        took = None
        if fresh < C:
            took = order[fresh]; fresh += 1; inflight[x] += 1
        elif rings[x] and rings[x][0][0] <= t:
            took = rings[x].pop(0)[1]
        if took is not None:
            heapq.heappush(events, (t + seg_ms[took] / xcd_speed[x], x, took))
            heapq.heappush(slots[x], t + seg_ms[took] / xcd_speed[x])
        else:
            # nothing to take now
            if per_xcd_count:
                if inflight[x] == 0:
                    wait = empty_us / 1000.0
                else:                                   # poll for a chain this XCD's own workgroups will hand back
                    nxt = min((e[0] for e in events if e[1] == x), default=None)
                    if nxt is not None and nxt - t <= poll_us / 1000.0:
                        flush(nxt)
                        c = rings[x].pop(0)[1] if rings[x] else None
                        if c is not None:
                            heapq.heappush(events, (nxt + seg_ms[c] / xcd_speed[x], x, c))
                            heapq.heappush(slots[x], nxt + seg_ms[c] / xcd_speed[x])
                            b += 1
                            continue
                    wait = poll_us / 1000.0
                heapq.heappush(slots[x], t + wait)
            else:
                # v3: sits on the slot until ALL chains are finished (or ≈ 4 s)
                heapq.heappush(slots[x], t + 4000.0)
        b += 1
    flush(float("inf"))
    return (end if unfinished == 0 else None), b


def main():
    C, N, per_tr = 4096, 1000, 0.041
    for slow in (1.0, 1.175, 1.483):
        work = [N * per_tr] * C
        work[3187] *= slow
        ident = list(range(C))
        lpt = sorted(ident, key=lambda c: -work[c])
        longf = [w > 1.03 * (sum(work) / C) for w in work]
        nseg = 24
        ideal = max(sum(work) / (XCDS * SLOTS), max(work))
        print(f"one chain at {slow} x: ideal {ideal:.1f} ms | classic, launch order as is {classic(work, ident):.1f}, longest first {classic(work, lpt):.1f}", end="")
        total = sum(1 if longf[c] else nseg for c in range(C))
        m3, w3 = segmented(work, lpt, longf, nseg, False, 0, 2 * total + 2048)
        m4, w4 = segmented(work, lpt, longf, nseg, True, 30.0, 8 * total + 8192)
        print(f" | v3 {'does not finish' if m3 is None else '%.1f' % m3} ({w3} workgroups) | v4 {'chains left' if m4 is None else '%.1f' % m4} ({w4} of {8 * total + 8192} workgroups)")
        # the same with XCDs that differ by a percent or two in speed: one of them runs out of its own chains first
        speed = [1.0, 0.99, 1.01, 1.0, 0.985, 1.015, 1.0, 1.02]
        m3, w3 = segmented(work, lpt, longf, nseg, False, 0, 2 * total + 2048, xcd_speed=speed)
        m4, w4 = segmented(work, lpt, longf, nseg, True, 30.0, 8 * total + 8192, xcd_speed=speed)
        print(f"      XCD speeds within 2 %: v3 {'does not finish within 5 s' if m3 is None else '%.1f' % m3} ({w3} workgroups) | v4 {'chains left' if m4 is None else '%.1f' % m4} ({w4} of {8 * total + 8192} workgroups)")


if __name__ == "__main__":
    main()
