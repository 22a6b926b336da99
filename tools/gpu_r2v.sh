#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
TAG=${1:-r2v}
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_$TAG.log 2>&1; echo "rc=$?" >> $OUT/pytest_$TAG.log
tail -30 $OUT/pytest_$TAG.log | cut -c1-220
timeout 100 python tools/sleep_awake_cost.py
python - <<'PY'
import sys, time
sys.path.insert(0, '.')
from rapier_amd import PhysicsWorld, scenes as S
w = PhysicsWorld.from_scene(S.many_pyramids().enable_sleep())
w.step(4); w.sync(); c0 = w.counters()
t = time.perf_counter(); w.step(26); w.sync(); dt = time.perf_counter() - t
c = w.counters()
print("awake sleep-enabled many_pyramids: %.1f us/step, fast %d full %d replayed %d" % (dt / 26 * 1e6, c["fast_steps"] - c0["fast_steps"], c["full_steps"] - c0["full_steps"], c["replayed_steps"] - c0["replayed_steps"]))
w.step(60); w.sync(); c2 = w.counters()
print("after 90 steps: sleeping", c2["num_sleeping_bodies"], "fast", c2["fast_steps"], "full", c2["full_steps"], "replayed", c2["replayed_steps"])
PY
