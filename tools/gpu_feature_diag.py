"""Step-by-step divergence report for the sleeping scenarios (GPU vs oracle).  Debug aid: prints the first
step at which poses / velocities / sleeping flags differ and which bodies are involved."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from rapier_amd import PhysicsWorld, scenes as S  # noqa: E402
from oracle_ffi import OracleWorld  # noqa: E402


def run(name, scene, steps, actions=None):
    g, o = PhysicsWorld.from_scene(scene), OracleWorld(scene)
    actions = actions or {}
    for k in range(1, steps + 1):
        if k in actions:
            actions[k](g, o)
        g.step(1); o.step(1)
        gp, gv = g.read_bodies(); op, ov = o.read()
        gs, os_ = g.sleeping(), o.sleeping()
        bad = np.where((gp != op).any(1) | (gv != ov).any(1) | (gs != os_))[0]
        if len(bad):
            print(f"[{name}] FIRST DIVERGENCE at step {k}: bodies {bad[:12].tolist()} (of {len(bad)})")
            for b in bad[:4]:
                print(f"   body {b}: sleep gpu={int(gs[b])} oracle={int(os_[b])}\n     gpu pos {gp[b]} vel {gv[b]}\n     ora pos {op[b]} vel {ov[b]}")
            print(f"   gpu counters {g.counters()}\n   oracle stats {o.stats()}")
            gm, gn, gi = g.contacts(); om, on, oi = o.manifolds()
            gk = {(a, b): (c, n, tuple(i)) for (a, b, c, n), i in zip(gm.tolist(), gi.tolist())}
            ok = {(a, b): (c, n, tuple(i)) for (a, b, c, n), i in zip(om.tolist(), oi.tolist())}
            for key in sorted(set(gk) | set(ok)):
                if gk.get(key) != ok.get(key):
                    print(f"   manifold {key}: gpu {gk.get(key)}\n                     ora {ok.get(key)}")
            print(f"   sleeping gpu {gs.astype(int).tolist()[:40]}\n   sleeping ora {os_.astype(int).tolist()[:40]}")
            return False
    print(f"[{name}] ok: {steps} steps bit-exact, asleep at the end: {int(g.sleeping().sum())}")
    return True


def kick(g, o):
    v = np.array([[1.0, 0, 0, 0, 0, 0]], np.float32)
    g.write_bodies([3], vel6=v); o.set_vel(3, v[0, :3], v[0, 3:])


def wake(g, o):
    g.wake_up([2]); o.wake_up(2)


def rm(g, o):
    g.remove_body(1); o.remove_body(1)


def tele(g, o):
    p = np.array([[3.0, 0.5, 0.0, 0, 0, 0, 1.0]], np.float32)
    g.write_bodies([2], pos7=p); o.set_pose(2, p[0])


def kin_targets(g, o, k):
    t = k / 60.0
    p = np.array([0.6 * t, 1.0 + 0.15 * t, 0.0, 0.0, np.sin(0.1 * t), 0.0, np.cos(0.1 * t)], np.float32)
    g.set_next_kinematic_position([1], p); o.set_next_kinematic_position(1, p)


def churn(steps=300, cap=60, env_sleep=1):
    """fountain churn (tests/test_gpu_parity.py::test_body_churn_bit_exact) with a per-step comparison"""
    from oracle_ffi import lib
    sc = S.Scene(name="churn", gravity=(0.0, -9.81, 0.0))
    gb = sc.add_body(body_type=S.BODY_FIXED, translation=(0.0, -2.1, 0.0))
    sc.add_collider(gb, half_extents=(40.0, 2.1, 40.0))
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    alive = []
    for k in range(1, steps):
        g.step(1); o.step(1)
        gp, gv = g.read_bodies(); op, ov = o.read()
        bad = [b for b in alive if (gp[b] != op[b]).any() or (gv[b] != ov[b]).any()]
        if bad:
            print(f"[churn] FIRST DIVERGENCE at step {k}: bodies {bad[:12]} (of {len(bad)}), alive {len(alive)}")
            for b in bad[:4]:
                print(f"   body {b}:\n     gpu pos {gp[b]} vel {gv[b]}\n     ora pos {op[b]} vel {ov[b]}")
            print(f"   gpu counters {g.counters()}\n   oracle stats {o.stats()}")
            gm, gn, gi = g.contacts(); om, on, oi = o.manifolds()
            gk = {(a, b): (c, n, tuple(i)) for (a, b, c, n), i in zip(gm.tolist(), gi.tolist())}
            ok = {(a, b): (c, n, tuple(i)) for (a, b, c, n), i in zip(om.tolist(), oi.tolist())}
            for key in sorted(set(gk) | set(ok)):
                if gk.get(key) != ok.get(key):
                    print(f"   manifold {key}: gpu {gk.get(key)}\n                     ora {ok.get(key)}")
            return False
        body = S.body_desc(translation=(0.0, 10.0, 0.0), can_sleep=env_sleep)
        col = S.collider_desc(shape=S.SHAPE_BALL, half_extents=(0.5, 0.0, 0.0)) if k % 3 == 0 else \
            S.collider_desc(half_extents=(0.5, 0.5, 0.5) if k % 3 == 2 else (0.5, 0.25, 0.5))
        hb = g.insert_body(body); g.insert_collider(col, hb)
        ob = o.add_body(translation=(0.0, 10.0, 0.0), can_sleep=env_sleep)
        lib().ro_add_collider(o._w, np.array([col], S.COLLIDER_DTYPE).ctypes.data, ob)
        alive.append(ob)
        if len(alive) > cap:
            op = o.read()[0]
            order = sorted(alive, key=lambda h: -(abs(op[h, 0]) + abs(op[h, 2])))
            for h in order[:len(alive) - cap]:
                g.remove_body(h); o.remove_body(h); alive.remove(h)
    print(f"[churn] ok: {steps} steps bit-exact")
    return True


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "churn":
        churn(int(sys.argv[2]) if len(sys.argv) > 2 else 300, int(sys.argv[3]) if len(sys.argv) > 3 else 60, int(sys.argv[4]) if len(sys.argv) > 4 else 1)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "compound":
        run("compound 12", S.compound_bodies(12), 80)
        run("compound 2", S.compound_bodies(2), 200)
        run("compound 1 (hammer)", S.compound_bodies(1), 200)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "kin":
        run("kinematic velocity platform", S.kinematic_platform(False), 150)
        acts = {k: (lambda g, o, k=k: kin_targets(g, o, k)) for k in list(range(1, 91)) + [212]}
        run("kinematic position platform", S.kinematic_platform(True).enable_sleep(), 240, acts)
        sys.exit(0)
    run("box_stack3 no-sleep", S.box_stack(3), 50)
    run("box_stack3 sleep", S.box_stack(3).enable_sleep(), 320, {111: kick, 230: wake})
    run("sleep_impact", S.sleep_impact(), 260)
    run("pyramids2x2 sleep", S.many_pyramids(rows=2, cols=2).enable_sleep(), 70)
    run("box_stack4 remove/teleport", S.box_stack(4).enable_sleep(), 400, {71: rm, 232: tele})
    run("tumble40 sleep", S.tumble(40, seed=11).enable_sleep(), 600)
