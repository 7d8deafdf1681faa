"""The tester / blender hand-shake of the experimental gsr_render_fwd_pair, checked exhaustively on a model (tools/pair_protocol_check.py)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import pair_protocol_check as P


def test_every_interleaving_of_the_handshake_terminates_in_order():
    for n in range(0, 6):
        for stop_at in range(0, n + 2):
            assert P.check(n, stop_at) > 0


def test_the_model_notices_a_broken_protocol(monkeypatch):
    """the check is not vacuous: a tester that does not wait for its buffer is caught"""
    real = P.step_tester

    def hasty(s, n):
        pc, r = s["t"]
        if pc == "tested":
            out = dict(s); out["t"] = ("writing", r); out["tw"] = r & 1
            return [out]
        return real(s, n)
    monkeypatch.setattr(P, "step_tester", hasty)
    try:
        P.check(5, 9)
    except AssertionError:
        return
    raise AssertionError("a tester that never waits for a free buffer went unnoticed")
