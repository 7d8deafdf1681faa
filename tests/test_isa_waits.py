"""The device code keeps its memory round trips in flight: `tools/wait_scan.py` over the gfx950
assembly of the library. A load inside a divergent `if`, a register copy of a prefetched value or a conditional store in a
pipelined loop shows up here as a load that is waited for within a few instructions of being issued -- the defect that cost
9 % of the step until round 3 and that no functional test can see. Cross-compiles (no GPU), ~1 min."""
import os
import re
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    import wait_scan
    path = str(tmp_path_factory.mktemp("asm") / "gsr.s")
    wait_scan.device_asm(path)
    return path, wait_scan.scan(path)


def _kernel(found, prefix):
    hits = [k for k in found if k.startswith(prefix)]
    assert hits, f"no kernel {prefix}* in the assembly"
    return hits


def test_hot_kernels_have_no_loads_waited_for_on_the_spot(asm):
    _, found = asm
    # (kernel symbol prefix, what may remain: dependent pairs, returning atomics, loads on rarely taken paths)
    budget = {
        "_Z21gsr_render_fwd_serialILb1E": 1,      # the work-list reservation (returning atomic) at the end
        "_Z21gsr_render_fwd_serialILb0E": 1,
        "_Z17gsr_render_bwd_q2": 5,               # later rounds of segments longer than 64 entries: list entry -> records; + round 6: the record's tile
                                                  # rectangle in the flush of the opt-in deterministic mode (GSR_VIEW_DETERMINISTIC)
        "_Z11gsr_scatterILi": 32,                        # the segment forward's work items (rare path), the tail of the eight-deep fetch of the ranges, the refill of a pinned grid's later rounds: 11;
                                                  # + round 5: K2's body inlined for the launch's ONE scan workgroup (a chain of dependent phases by nature: 17) and the
                                                  # scatter workgroups' own scan of the tile counts (the counts of the other views, the tail of the fetch: 2)
        "_Z18gsr_preprocess_fwdILb0E": 14,        # camera staging, cov3D_precomp / colors_precomp / degree-0 paths, the two polls of the "counters cleared" tag,
        "_Z18gsr_preprocess_fwdILb1E": 14,        # the last of the flush's four reserving atomics (their results are what is stored); + round 6: the tail of
                                                  # the twelve-deep fetch of the group-mates' histogram rows (the last three of twelve loads issued together) and the
                                                  # workgroup's arrival ticket (one returning atomic by one lane)
        "_Z18gsr_preprocess_bwdILb0ELb0E": 14,    # camera staging, the accumulate read-modify-write of views after the first, the
        "_Z18gsr_preprocess_bwdILb1ELb0E": 15,    # row-predicated element loads of the split / odd-row-length staging paths
    }
    for prefix, allowed in budget.items():
        for k in _kernel(found, prefix):
            n = len(found[k])
            assert n <= allowed, f"{k}: {n} loads are waited for where they are issued (allowed {allowed}):\n" + \
                "\n".join(f"  +{f[0]} {f[1]} -> +{f[2]} {f[3]}" for f in found[k])


def test_the_serial_walk_leaves_two_rounds_of_requests_outstanding(asm):
    """Inside gsr_render_fwd_serial's loop the wait in front of a round's first use of its records is a count of >= 20
    younger operations (two rounds of gathers, list entries and checkpoint stores), and the wait in front of the gather
    leaves the previous round's seven stores outstanding."""
    path, _ = asm
    text = open(path).read()
    for q in ("ILb1E", "ILb0E"):
        m = re.search(r"^_Z21gsr_render_fwd_serial%s\w*:(.*?)^\.Lfunc_end" % q, text, flags=re.S | re.M)
        assert m
        counts = [int(c) for c in re.findall(r"s_waitcnt vmcnt\((\d+)\)", m.group(1))]
        assert max(counts) >= 20, counts
        assert sum(1 for c in counts if 6 <= c <= 8) >= 3, counts      # one per unrolled round
