"""CPU model of oc_cell_kernel's pass loop (csrc/pcps_onchip.hip) and its LDS exchange buffer (csrc/fft_onchip.h): when may the barrier between two cells of a
work-group be left out?

The phased exchanges move whole complex values through TWO regions of LDS that take the phases alternately; a step is "read phase p - 1, write phase p" between two
barriers.  All waves pass a barrier together, and each has waited for its own LDS operations before it arrives (s_waitcnt lgkmcnt(0) in front of s_barrier), so what
can overlap in time is exactly what lies between the same two barriers ("epoch") in different waves.  A region may therefore never be both read and written inside
one epoch.  The model lists the exchange operations of consecutive passes for every plan of GSH_OC_PLANS that uses the phased form, cuts them into epochs, and
checks (a) that the in-pass structure is hazard-free, and (b) that the kernel's compile-time rule -- no barrier between the passes iff exchange 2 ENDS in the
region exchange 1 does not START in -- is exactly the plans for which the pass boundary is hazard-free without one."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LDS_BUDGET_BYTES = 160000


def _round_up_congruent(at_least, minus, mod):
    s = at_least
    while (s - minus) % mod != 0:
        s += 1
    return s


def _plan(r1, r2, r3):
    t1, t2, t3 = r2 * r3, r1 * r3, r1 * r2
    threads = (max(t1, t2, t3) + 63) // 64 * 64
    s1 = _round_up_congruent(t1, r3, 32)
    p2 = r1 | 1
    s2 = _round_up_congruent(r3 * p2, r1, 32)

    def phases_for(rows, row_elems):
        n = 1
        while 2 * ((rows + n - 1) // n) * row_elems * 8 > LDS_BUDGET_BYTES:
            n += 1
        return n

    np1, np2 = phases_for(r1, s1), phases_for(r2, s2)
    ex64 = max(r1 * s1, r2 * s2) * 4 > 72 * 1024 or threads == 1024
    return dict(np1=np1, np2=np2, start1=0, start2=np1 % 2, ex64=ex64)


def _plans():
    text = open(os.path.join(ROOT, "gnss-sdr_amd", "csrc", "fft_onchip.h")).read()
    body = text[text.index("#define GSH_OC_PLANS(X)"):]
    body = body[:body.index("#endif")]
    return [tuple(int(v) for v in m) for m in re.findall(r"X\((\d+),\s*(\d+),\s*(\d+)\)", body)]


def _pass_ops(p):
    """the exchange operations of one pass in program order: ('R' | 'W', region) and 'B' for a barrier"""
    ops = []
    for (n, start) in ((p["np1"], p["start1"]), (p["np2"], p["start2"])):
        for ph in range(n):
            if ph > 0:
                ops.append(("R", (ph - 1 + start) % 2))
            ops.append(("W", (ph + start) % 2))
            ops.append("B")
        ops.append(("R", (n - 1 + start) % 2))
    return ops


def _hazards(ops):
    """regions both read and written between two barriers"""
    bad, reads, writes = [], set(), set()
    for op in ops + ["B"]:
        if op == "B":
            bad += sorted(reads & writes)
            reads, writes = set(), set()
        elif op[0] == "R":
            reads.add(op[1])
        else:
            writes.add(op[1])
    return bad


def test_plan_list_is_read():
    plans = _plans()
    assert (25, 25, 40) in plans and len(plans) >= 20
    p = _plan(25, 25, 40)
    assert (p["np1"], p["np2"], p["start2"], p["ex64"]) == (3, 3, 1, True)   # fft_onchip.h: "N = 25 000: NP = 3 (9 + 9 + 7 rows)"


def test_no_region_is_read_and_written_between_two_barriers():
    phased = 0
    for r in _plans():
        p = _plan(*r)
        if not p["ex64"]:
            continue
        phased += 1
        one = _pass_ops(p)
        assert _hazards(one) == [], (r, "inside a pass")
        kernel_leaves_it_out = (p["np2"] - 1 + p["start2"]) % 2 != p["start1"]          # pcps_onchip.hip: PASS_BARRIER
        three_passes_without = one + one + one
        three_passes_with = one + ["B"] + one + ["B"] + one
        assert _hazards(three_passes_with) == [], r
        assert (_hazards(three_passes_without) == []) == kernel_leaves_it_out, (r, p, _hazards(three_passes_without))
    assert phased >= 3
