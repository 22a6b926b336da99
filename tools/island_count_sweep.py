"""Step time of the fused single-kernel step against the number of LDS-resident islands: many_pyramids grids of
r x c pyramids (one island each, 55 cuboids / 145 manifolds), 60 warm-up + 600 timed steps.  One 512-thread workgroup fills a CU's
register file (256 VGPRs per lane), so islands beyond rp_fused_grid() = 240 run in further passes of the same launch."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rapier_amd import PhysicsWorld, scenes as S  # noqa: E402

print("islands  cuboids   us/step  steps/s   us/step per pass of 240")
grids = ((7, 7), (10, 10), (14, 14), (15, 16), (16, 16), (19, 19), (20, 24), (22, 22), (27, 27), (38, 38), (54, 54))
if len(sys.argv) > 1 and sys.argv[1] == "short":
    grids = ((14, 14), (16, 16), (19, 19), (20, 24), (22, 22), (27, 27), (27, 54), (54, 54))
print("RP_NO_ISL_DENSE =", os.environ.get("RP_NO_ISL_DENSE"), " RP_ISL_DENSE =", os.environ.get("RP_ISL_DENSE"))
for r, c in grids:
    w = PhysicsWorld.from_scene(S.many_pyramids(rows=r, cols=c))
    w.step(60); w.sync()
    t = time.perf_counter(); w.step(600); w.sync(); dt = (time.perf_counter() - t) / 600
    n = r * c
    passes = -(-n // 240)
    print(f"{n:7d} {n * 55:8d} {dt * 1e6:9.1f} {1 / dt:8.0f} {dt * 1e6 / passes:9.1f}   ({passes} pass{'es' if passes > 1 else ''})", flush=True)
    del w
