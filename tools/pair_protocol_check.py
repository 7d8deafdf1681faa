#!/usr/bin/env python
"""Exhaustive check of the tester / blender hand-shake of gsr_render_fwd_pair (csrc/gsr_render.hip) on a small model.

The two waves of an 8x8 block exchange two LDS buffers through four words: ready[buf] (tester -> blender: round + 1 and the
longest list, or END behind the list's end) and freed[buf] (blender -> tester: round + 1 once consumed, STOP when every pixel
has stopped). This script walks EVERY interleaving of the two waves' steps for lists of 0..7 rounds and every round at which
the pixels may stop, and asserts
  * exclusion: the tester never writes a buffer the blender is reading (or has not read yet),
  * order: the blender consumes rounds 0, 1, 2, ... each exactly once, out of the buffer the tester filled for that round,
  * termination: every reachable state can still make progress (no state in which both waves wait on words nobody will write),
    and both waves end.
It models what the kernel's code does, step by step (names of the steps in the comments of the kernel); it is a check of the
protocol, not of the kernel's memory ordering (release / acquire fences around the words are the kernel's job).

    python tools/pair_protocol_check.py
"""
import sys
from collections import deque

END, STOP = "END", "STOP"


def step_tester(s, n):
    """one atomic step of the tester; returns the list of successor states (empty: blocked), or None when it has ended"""
    pc, r = s["t"]
    buf = r & 1
    if pc == "done":
        return None
    out = dict(s)
    if pc == "top":
        if r >= n:                                                  # behind the end of the list
            if r >= 2:
                f = s["freed"][buf]
                if f == STOP:
                    out["t"] = ("done", r); return [out]
                if f != r - 1:
                    return []                                       # spins
            rd = list(s["ready"]); rd[buf] = (r + 1, END); out["ready"] = tuple(rd)
            out["t"] = ("done", r); return [out]
        out["t"] = ("tested", r); return [out]                      # loads + quad tests + the masks: nothing shared
    if pc == "tested":                                              # the buffer must be free
        f = s["freed"][buf]
        if f == STOP:
            out["t"] = ("done", r); return [out]
        if r >= 2 and f != r - 1:
            return []                                               # spins
        out["t"] = ("writing", r); out["tw"] = buf; return [out]    # begins to write stage / lists of `buf`
    if pc == "writing":
        out["tw"] = None
        fill = list(s["fill"]); fill[buf] = r; out["fill"] = tuple(fill)
        rd = list(s["ready"]); rd[buf] = (r + 1, "lists"); out["ready"] = tuple(rd)
        out["t"] = ("top", r + 1); return [out]
    raise AssertionError(pc)


def step_blender(s, stop_at):
    pc, r = s["b"]
    buf = r & 1
    if pc == "done":
        return None
    out = dict(s)
    if pc == "wait":
        rd = s["ready"][buf]
        if rd is None or rd[0] != r + 1:
            return []                                               # spins
        if rd[1] == END:
            out["b"] = ("done", r); return [out]
        if r >= stop_at:                                            # every pixel has stopped
            out["freed"] = (STOP, STOP); out["b"] = ("done", r); return [out]
        assert s["fill"][buf] == r, ("blender would read round", s["fill"][buf], "for round", r)
        out["b"] = ("reading", r); out["br"] = buf; return [out]
    if pc == "reading":
        out["br"] = None
        out["consumed"] = s["consumed"] + (r,)
        fr = list(s["freed"]); fr[buf] = r + 1; out["freed"] = tuple(fr)
        out["b"] = ("wait", r + 1); return [out]
    raise AssertionError(pc)


def freeze(s):
    return tuple(sorted(s.items()))


def check(n, stop_at):
    s0 = dict(t=("top", 0), b=("wait", 0), ready=(None, None), freed=(0, 0), tw=None, br=None, fill=(None, None), consumed=())
    seen, todo, ends = {freeze(s0)}, deque([s0]), 0
    while todo:
        s = todo.popleft()
        assert s["tw"] is None or s["tw"] != s["br"], ("tester writes the buffer the blender reads", s)
        succ, alive = [], 0
        for fn, arg in ((step_tester, n), (step_blender, stop_at)):
            r = fn(s, arg)
            if r is None:
                continue
            alive += 1
            succ += r
        if alive == 0:                                              # both have ended
            want = tuple(range(min(n, stop_at)))
            assert s["consumed"] == want, (s["consumed"], want)
            ends += 1
            continue
        assert succ, ("deadlock", n, stop_at, s)
        for x in succ:
            k = freeze(x)
            if k not in seen:
                seen.add(k); todo.append(x)
    assert ends > 0
    return len(seen)


def main():
    total = 0
    for n in range(0, 8):
        for stop_at in range(0, n + 2):
            total += check(n, stop_at)
    print(f"pair hand-shake: {total} states over lists of 0..7 rounds x every stopping round: exclusion, order and termination hold")


if __name__ == "__main__":
    main()
