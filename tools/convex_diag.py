"""Where do the device's support-mapped shapes (rp_convex.h) leave the oracle (ro_convex.h)?  (1) two-body micro scenes — a fixed shape and
a dynamic one placed at a random pose around it, overlapping a little or a little apart, a few steps each — counted per shape pair, so
a mismatch names the generator; (2) the clutter scene in lockstep with the first mismatching body and its manifolds on both sides."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from rapier_amd import PhysicsWorld, scenes as S  # noqa: E402
from oracle_ffi import OracleWorld  # noqa: E402

NAMES = {0: "ball", 1: "cuboid", 2: "capsule", 3: "halfspace", 4: "cylinder", 5: "cone", 6: "polyhedron"}


def rand_shape(rng, kind):
    if kind == S.SHAPE_BALL:
        return (float(rng.uniform(.2, .6)), 0.0, 0.0), 0.6
    if kind == S.SHAPE_CUBOID:
        he = tuple(float(x) for x in rng.uniform(.2, .7, 3)); return he, float(np.linalg.norm(he))
    if kind == S.SHAPE_CAPSULE:
        hh, r = float(rng.uniform(.2, .6)), float(rng.uniform(.1, .3)); return (hh, r, float(rng.integers(0, 3))), hh + r
    if kind == S.SHAPE_CONVEX:
        return None, 0.6       # (a fresh point cloud, registered by the caller)
    hh, r = float(rng.uniform(.2, .7)), float(rng.uniform(.15, .6)); return (hh, r, 0.0), float(np.hypot(hh, r))


def micro(n_cases, seed):
    rng = np.random.default_rng(seed)
    kinds = [S.SHAPE_BALL, S.SHAPE_CUBOID, S.SHAPE_CAPSULE, S.SHAPE_CYLINDER, S.SHAPE_CONE, S.SHAPE_CONVEX]
    stats = {}
    shown = 0
    for case in range(n_cases):
        ka, kb = kinds[rng.integers(0, 6)], kinds[rng.integers(0, 6)]
        if ka < S.SHAPE_CYLINDER and kb < S.SHAPE_CYLINDER:
            continue
        (hea, ra), (heb, rb) = rand_shape(rng, ka), rand_shape(rng, kb)
        sc = S.Scene(name="micro", gravity=(0.0, -2.0, 0.0))
        if hea is None:
            hea = (sc.add_convex_polyhedron((rng.standard_normal((int(rng.integers(6, 30)), 3)) * 0.3).astype(np.float32) + np.float32(rng.uniform(-0.1, 0.1, 3))), 0.0, 0.0)
        if heb is None:
            heb = (sc.add_convex_polyhedron((rng.standard_normal((int(rng.integers(6, 30)), 3)) * 0.3).astype(np.float32)), 0.0, 0.0)
        qa = rng.standard_normal(4); qa /= np.linalg.norm(qa)
        qb = rng.standard_normal(4); qb /= np.linalg.norm(qb)
        a = sc.add_body(body_type=S.BODY_FIXED if case % 3 else S.BODY_DYNAMIC, translation=(0.0, 0.0, 0.0), rotation=tuple(float(x) for x in qa))
        sc.add_collider(a, shape=ka, half_extents=hea)
        d = rng.standard_normal(3); d /= np.linalg.norm(d)
        dist = float(rng.uniform(0.3, 1.0)) * (ra + rb)
        b = sc.add_body(translation=tuple(float(x) for x in d * dist), rotation=tuple(float(x) for x in qb), linvel=tuple(float(x) for x in -d * 1.5),
                        angvel=tuple(float(x) for x in rng.standard_normal(3)))
        sc.add_collider(b, shape=kb, half_extents=heb)
        g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
        key = (NAMES[ka], NAMES[kb])
        st = stats.setdefault(key, [0, 0])
        st[0] += 1
        bad_at = -1
        for step in range(40):
            g.step(1); o.step(1)
            gp, gv = g.read_bodies(); op, ov = o.read()
            if not (np.array_equal(gp, op) and np.array_equal(gv, ov)):
                bad_at = step
                break
        if bad_at >= 0:
            st[1] += 1
            if shown < 12:
                shown += 1
                gm, gn, gi = g.contacts(); om, on, oi = o.manifolds()
                print(f"MISMATCH case {case} {key} at step {bad_at}: he_a {hea} he_b {heb}")
                print("   device  :", gm.tolist(), np.asarray(gn).tolist(), np.asarray(gi).tolist())
                print("   oracle  :", om.tolist(), np.asarray(on).tolist(), np.asarray(oi).tolist())
                print("   dpos", (gp - op)[1], "dvel", (gv - ov)[1], flush=True)
        del g
    print("pair type            cases  mismatching")
    for k in sorted(stats):
        print(f"{k[0]:>9s} {k[1]:<9s} {stats[k][0]:6d} {stats[k][1]:6d}")
    return sum(v[1] for v in stats.values())


def clutter(ground, steps):
    sc = S.polyhedra_clutter(28, 2) if ground == "polyhedra" else S.convex_clutter(40, 3, ground)
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    for step in range(steps):
        g.step(1); o.step(1)
        gp, gv = g.read_bodies(); op, ov = o.read()
        if not (np.array_equal(gp, op) and np.array_equal(gv, ov)):
            badb = np.flatnonzero((gp != op).any(1) | (gv != ov).any(1))
            print(f"clutter[{ground}]: first mismatch at step {step + 1}, bodies {badb.tolist()[:10]} (max |dpos| {np.abs(gp - op).max():.3e})")
            gm, gn, gi = g.contacts(); om, on, oi = o.manifolds()
            gk = {(a, b): (c, n, tuple(i), tuple(nn)) for (a, b, c, n), i, nn in zip(gm.tolist(), np.asarray(gi).tolist(), np.asarray(gn).tolist())}
            ok = {(a, b): (c, n, tuple(i), tuple(nn)) for (a, b, c, n), i, nn in zip(om.tolist(), np.asarray(oi).tolist(), np.asarray(on).tolist())}
            par = sc.collider_parents
            for k in sorted(set(gk) | set(ok)):
                if gk.get(k) != ok.get(k):
                    sh = (NAMES[int(sc.colliders[k[0]]["shape"])], NAMES[int(sc.colliders[k[1]]["shape"])])
                    print("   pair", k, sh, "bodies", (par[k[0]], par[k[1]]), "\n      device", gk.get(k), "\n      oracle", ok.get(k))
            return 1
    print(f"clutter[{ground}]: {steps} steps bit-exact")
    return 0


if __name__ == "__main__":
    bad = micro(int(sys.argv[1]) if len(sys.argv) > 1 else 400, 1)
    for gr in ("cuboid", "cylinder", "halfspace"):
        bad += clutter(gr, 300)
    bad += clutter("polyhedra", 300)
    print("TOTAL MISMATCHES", bad)
