"""Test hooks of libgsr.so (gsr_testing_override, include/gsr.h): force one of the choices the library otherwise makes from the
problem shape -- the forward's compositing mode, the depth-segment length, list kind, hints, launch geometry. Explicit calls only:
the library reads nothing from the process environment (two of these choices change where the per-pixel sums are cut, i.e. the
rounding of the results). Used by tests/ and by A/B measurements; not part of the drop-in surface.

    with _testing.override(fwd_mode="seg", seg_shift=7):
        ...
"""
from __future__ import annotations

import contextlib

from . import _lib

NAMES = ("fwd_mode", "seg_shift", "fwd_lists", "fwd_hints", "speculate", "hist_max", "k1_grid", "fwd_grid", "k6_grid", "fwd_lds_kb", "bwd_grid", "k6_compact", "scan_fold", "grad_clear", "k1_group", "sort_kernel")
_WORDS = {"fwd_mode": {"seq": 1, "seg": 2, "pair": 3}, "fwd_lists": {"block": 1, "q": 2, "quad": 2}, "fwd_hints": {"off": 1, "skipall": 2}}
_current = {}


def set(name: str, value) -> None:
    """value None / -1 = the library decides again; words as in the table above ("seq", "seg", "block", "q", "off", "skipall")."""
    if name not in NAMES:
        raise KeyError(f"unknown override {name!r}: one of {NAMES}")
    v = -1 if value is None else _WORDS.get(name, {}).get(value, value)
    rc = _lib.load().gsr_testing_override(name.encode(), int(v))
    _lib.check(rc, "gsr_testing_override")
    if int(v) == -1:
        _current.pop(name, None)
    else:
        _current[name] = int(v)


def reset() -> None:
    for name in list(_current):
        set(name, None)


@contextlib.contextmanager
def override(**kw):
    before = dict(_current)
    try:
        for k, v in kw.items():
            set(k, v)
        yield
    finally:
        for k in kw:
            set(k, before.get(k))
