"""The per-step trace that lets a machine with cargo bisect the golden-hash mismatch in one run (tests/reference_trace.py,
`bench/rapier_ref --trace`): format, differ, the committed oracle trace of the golden scene, and — the day a trace of the real crate
is dropped into tests/golden/reference/ — the comparison itself."""
import glob
import os

import pytest

from rapier_amd import scenes as S
from oracle_ffi import OracleWorld
import reference_trace as T
from test_reference_dump import SCENES

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN_TRACE = os.path.join(HERE, "golden", "reference_pile_s120.rptrace")


def small_sleeping_trace():
    sc = S.box_stack(3).enable_sleep()
    return "\n".join(T.trace_lines(OracleWorld(sc), "stack3", 200)) + "\n"


def test_trace_format_and_self_diff():
    text = small_sleeping_trace()
    head, steps = T.parse(text)
    assert head == {"scene": "stack3", "bodies": 4, "steps": 200}
    assert sum(1 for s in steps for e in s["events"] if e.startswith("S ")) == 3  # every cube falls asleep exactly once
    assert any(e.startswith("B ") for e in steps[0]["events"] | steps[1]["events"] | steps[2]["events"])
    assert steps[-1]["asleep"] == 3
    assert T.diff(text, text) is None


def test_diff_names_the_first_decision_that_differs():
    text = small_sleeping_trace()
    lines = text.splitlines()
    k = next(i for i, ln in enumerate(lines) if ln.startswith("S "))
    sleep_step = int([ln for ln in lines[:k] if ln.startswith("step")][-1].split()[1])
    moved = lines[:k] + lines[k + 1:]  # one body "sleeps" a step later on the other side
    nxt = next(i for i in range(k, len(moved)) if moved[i].startswith("step"))
    moved.insert(nxt + 1, lines[k])
    rep = T.diff(text, "\n".join(moved), "ref", "oracle")
    assert rep.startswith(f"step {sleep_step}: first structural difference") and f"only ref: {lines[k]}" in rep
    # a differing hash with identical decisions is reported as a rounding, at the step it first shows
    i5 = next(i for i, ln in enumerate(lines) if ln.startswith("step 5 "))
    f = lines[i5].split(); f[3] = "%016x" % (int(f[3], 16) ^ 1)
    rounded = lines[:i5] + [" ".join(f)] + lines[i5 + 1:]
    rep = T.diff(text, "\n".join(rounded))
    assert rep.startswith("step 5: state hashes differ") and "rounding" in rep


def test_committed_trace_of_the_golden_scene_is_the_oracles():
    """tests/golden/reference_pile_s120.rptrace = what the oracle does on simd_backend_determinism.rs's scene; its last hash is the
    value test_reference_golden.py compares with the reference's GOLDEN"""
    want = open(GOLDEN_TRACE).read()
    got = "\n".join(T.trace_lines(OracleWorld(SCENES["reference_pile"]()), "reference_pile", 120)) + "\n"
    assert T.diff(want, got, "committed", "oracle") is None
    head, steps = T.parse(want)
    assert steps[-1]["hash"] == "b4922d463afe5301"
    slept = [k for k, s in enumerate(steps, start=1) if any(e.startswith("S ") for e in s["events"])]
    assert slept == [46] and steps[45]["asleep"] == 432  # all 432 cubes, one step: no island waits on a pending split
    ends = [e.split() for s in steps[:46] for e in s["events"] if e.startswith("E ")]
    assert not [e for e in ends if 1 <= int(e[1]) <= 432 and 1 <= int(e[2]) <= 432]  # no cube-cube constraint is removed before the sleep


_REF = sorted(glob.glob(os.path.join(HERE, "golden", "reference", "*.rptrace")))


@pytest.mark.skipif(not _REF, reason="no traces of the real crate in tests/golden/reference (needs cargo: bench/rapier_ref --trace)")
@pytest.mark.parametrize("path", _REF or ["-"])
def test_oracle_trace_matches_reference_trace(path):
    text = open(path).read()
    head, _ = T.parse(text)
    got = "\n".join(T.trace_lines(OracleWorld(SCENES[head["scene"]]()), head["scene"], head["steps"])) + "\n"
    assert T.diff(text, got, "reference", "oracle") is None
