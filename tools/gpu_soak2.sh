#!/bin/bash
# Soak of the round's late additions: half-space ground + slanted plane on many seeds (every 4th under Coulomb: k_island_generic),
# the plain driver on fresh seeds (Coulomb islands), parameter-randomised variants of both.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python - <<'PY' > $OUT/soak2.log 2>&1
import sys, time
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import test_gpu_fuzz as F
bad = []; n = 0; t0 = time.time()
def go(tag, seed, **kw):
    global n
    n += 1
    try: F._run(seed, **kw)
    except AssertionError as e: bad.append((tag, seed, str(e)[:300]))
for seed in list(range(300, 360)) + list(range(2100, 2130)):
    go("halfspace", seed, steps=200, params=seed >= 2000, halfspace=True)
for seed in list(range(403, 520, 4)) + list(range(2203, 2260, 4)):      # seed % 4 == 3: FrictionModel::Coulomb
    go("coulomb", seed, steps=200, params=seed >= 2000)
for seed in range(603, 660, 4):
    go("coulomb+sensors+halfspace", seed, steps=200, sensors=True, halfspace=True)
print("soak2:", n, "runs in %.0f s, failures" % (time.time() - t0), bad)
PY
tail -3 $OUT/soak2.log | cut -c1-1500
