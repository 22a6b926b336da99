#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
TAG=${1:-r2y}
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_$TAG.log 2>&1; echo "rc=$?" >> $OUT/pytest_$TAG.log
tail -4 $OUT/pytest_$TAG.log | cut -c1-200
python - <<'PY'
import sys, time
sys.path.insert(0, '.')
from rapier_amd import PhysicsWorld, scenes as S
sc = S.many_pyramids()
trig = sc.add_body(body_type=S.BODY_FIXED, translation=(0.0, 3.0, 0.0))
sc.add_collider(trig, half_extents=(30.0, 2.0, 30.0), sensor=1, active_events=S.ACTIVE_EVENTS_COLLISION)
w = PhysicsWorld.from_scene(sc)
w.step(60); w.sync(); c0 = w.counters(); n0 = len(w.collision_events())
t = time.perf_counter(); w.step(1000); w.sync(); dt = time.perf_counter() - t
c = w.counters()
print("many_pyramids + a trigger volume over a quarter of the pyramids: %.1f us/step, fast %d full %d replayed %d, sensor pairs %d (intersecting %d), Started events at warm-up %d" % (
    dt / 1000 * 1e6, c["fast_steps"] - c0["fast_steps"], c["full_steps"] - c0["full_steps"], c["replayed_steps"] - c0["replayed_steps"],
    len(w.intersection_pairs()), int(w.intersection_pairs()[:, 2].sum()), n0))
PY
