#!/bin/bash
# Wider randomised differential soak than the suite holds: seeds 16..139 of the fuzz driver (every feature + user actions), the
# parameter-randomised variant on other seeds, solve groups and sensors on more seeds.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
RP_FUZZ_FIRST=16 RP_FUZZ_LAST=140 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -x -k "test_fuzz_bit_exact" > $OUT/soak_a.log 2>&1; echo "rc=$?" >> $OUT/soak_a.log
tail -3 $OUT/soak_a.log | cut -c1-200
python - <<'PY' > $OUT/soak_b.log 2>&1
import sys
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
import test_gpu_fuzz as F
bad = []
for seed in range(2016, 2060):
    try: F._run(seed, steps=160, params=True)
    except AssertionError as e: bad.append(("params", seed, str(e)[:200]))
for seed in list(range(30, 50)) + list(range(2010, 2020)):
    try: F._run(seed, steps=160, params=seed >= 2000, extras=True)
    except AssertionError as e: bad.append(("groups", seed, str(e)[:200]))
for seed in list(range(50, 70)) + list(range(2020, 2030)):
    try: F._run(seed, steps=200, params=seed >= 2000, sensors=True)
    except AssertionError as e: bad.append(("sensors", seed, str(e)[:200]))
print("soak b: failures", bad)
PY
tail -3 $OUT/soak_b.log | cut -c1-600
