"""What the routing of tiny islands (rp_islands.hip, lay_isl_number) costs or gains in a SETTLED debris field: n separate boxes at rest
on a slab, each its own island of one manifold — the fused / fast step against the global path.  Run with and without RP_NO_TINY_ROUTING=1."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rapier_amd import PhysicsWorld, scenes as S  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
side = int(n ** 0.5)
sc = S.Scene(name="debris", gravity=(0.0, -9.81, 0.0))
g = sc.add_body(body_type=S.BODY_FIXED, translation=(0.0, -0.5, 0.0)); sc.add_collider(g, half_extents=(2.0 * side, 0.5, 2.0 * side))
for i in range(side * side):
    b = sc.add_body(translation=(2.0 * (i % side) - side, 0.5, 2.0 * (i // side) - side)); sc.add_collider(b, half_extents=(0.5, 0.5, 0.5))
w = PhysicsWorld.from_scene(sc)
w.step(120); w.sync()
t = time.perf_counter(); w.step(600); w.sync(); dt = (time.perf_counter() - t) / 600
c = w.counters()
print(f"RP_NO_TINY_ROUTING={os.environ.get('RP_NO_TINY_ROUTING')}: {side * side} resting boxes, {1 / dt:.0f} steps/s ({dt * 1e3:.3f} ms/step); fast_steps {c['fast_steps']} full_steps {c['full_steps']} manifolds {c['num_manifolds']}")
