"""Persistent islands on the device (rp_sleep.hip) against the oracle's restatement of the reference's
island_manager/{persistent,local_split,global_split}.rs — the GPU twins of tests/test_persistent_islands.py: the same public
operations are applied to both worlds and, after EVERY step, body states, sleeping flags, island ids, the island table rows that
matter and the machinery's counters must be identical (ids included: both sides hand them out like alloc_island does)."""
import numpy as np
import pytest

from rapier_amd import PhysicsWorld, scenes as S
from oracle_ffi import OracleWorld, lib as olib
from test_reference_kats import world, ground, _cube
from test_persistent_islands import row_scene, pendulum_scene, COOLDOWN

pytestmark = pytest.mark.gpu

# counters both sides keep (the device does not count multi-way groups, bid ties and blocked island-steps)
STATS = ("merged", "removals", "connected", "detached", "hot", "over_budget", "sleeping_deferred", "global_splits", "global_split_pieces", "bids",
         "detach_size_ties", "split_keep_ties")


class Pair:
    """one device world and one oracle world driven by the same operations"""

    def __init__(self, scene):
        self.g = PhysicsWorld.from_scene(scene)
        self.o = OracleWorld(scene)
        self.steps = 0

    def check(self, msg=""):
        g, o = self.g, self.o
        gp, gv = g.read_bodies()
        op, ov = o.read()
        where = f"{msg} @ step {self.steps}"
        np.testing.assert_array_equal(gp, op, err_msg="poses " + where)
        np.testing.assert_array_equal(gv, ov, err_msg="velocities " + where)
        np.testing.assert_array_equal(g.sleeping(), o.sleeping(), err_msg="sleeping flags " + where)
        gl, ol = g.island_labels(), o.island_labels()
        np.testing.assert_array_equal(gl, ol, err_msg="island ids " + where)
        gs, os_ = g.island_stats(), o.island_stats()
        assert {k: gs[k] for k in STATS} == {k: os_[k] for k in STATS}, where
        assert g.island_globals() == o.island_globals(), where
        for isl in sorted(set(int(x) for x in ol if x >= 0)):
            assert g.island_state(isl) == o.island_state(isl), (isl, where)

    def step(self, n=1, msg=""):
        for _ in range(n):
            self.g.step(1); self.o.step(1); self.steps += 1
            self.check(msg)

    def run(self, n):  # unchecked stretch (checked at its end)
        self.g.step(n); self.o.step(n); self.steps += n
        self.check()

    def set_pose(self, body, pos7):
        self.g.write_bodies([body], pos7=np.asarray([pos7], np.float32)); self.o.set_pose(body, pos7)

    def set_vel(self, body, lin, ang=(0, 0, 0)):
        self.g.write_bodies([body], vel6=np.asarray([list(lin) + list(ang)], np.float32)); self.o.set_vel(body, lin, ang)

    def remove_body(self, body):
        self.g.remove_body([body]); self.o.remove_body(body)

    def remove_joint(self, j):
        self.g.remove_impulse_joint([j]); self.o.remove_joint(j)


def test_body_removal_gate_and_global_split_bit_exact():
    sc, (left, middle, right) = row_scene()
    p = Pair(sc)
    p.run(240)
    assert p.g.sleeping()[[left, middle, right]].all()
    p.remove_body(middle)
    for _ in range(80):
        p.step(1, "row without its middle box")
        if p.g.sleeping()[left]:
            break
    st = p.g.island_stats()
    assert st["global_splits"] == 1 and st["global_split_pieces"] == 1 and p.g.sleeping()[right]
    assert p.g.island_labels()[left] != p.g.island_labels()[right]


def test_cold_and_hot_separation_bit_exact():
    sc, row = row_scene(6)
    p = Pair(sc)
    p.run(240)
    for i in range(3, 6):
        p.set_pose(row[i], [30.0 + (i - 3), 0.5, 0.0, 0.0, 0.0, 0.0, 1.0])
    p.step(3, "half a row lifted away")
    assert p.g.island_stats()["detached"] == 1 and p.g.island_labels()[row[0]] != p.g.island_labels()[row[3]]
    p.run(120)
    # hot: two touching boxes shot apart stay one island until the deferred split
    sc2 = world(); ground(sc2)
    a, b = _cube(sc2, (0.0, 0.5, 0.0)), _cube(sc2, (1.0, 0.5, 0.0))
    q = Pair(sc2)
    q.run(120)
    q.set_vel(a, (-6.0, 0.0, 0.0)); q.set_vel(b, (6.0, 0.0, 0.0))
    for _ in range(400):
        q.step(1, "two boxes shot apart")
        if q.g.sleeping()[[a, b]].all():
            break
    st = q.g.island_stats()
    assert st["hot"] == 1 and st["detached"] == 0 and st["global_splits"] == 1 and q.g.sleeping()[[a, b]].all()


def test_split_retry_cooldown_bit_exact():
    sc, (r, a, b), links = pendulum_scene()
    p = Pair(sc)
    p.step(45, "pendulum")
    p.remove_joint(links[0])
    p.step(2, "first redundant link removed")
    assert p.g.island_stats()["global_splits"] == 1
    p.remove_joint(links[1])
    seq = []
    for _ in range(COOLDOWN + 2):
        p.step(1, "second redundant link removed")
        seq.append(p.g.island_stats()["global_splits"])
    assert seq == [1] * COOLDOWN + [2, 2], seq


def test_merges_free_and_reuse_ids_like_the_oracle():
    sc = world(); ground(sc)
    big = [_cube(sc, (0.0, 0.5 + i, 0.0)) for i in range(3)]
    lone = _cube(sc, (10.0, 0.5, 0.0))
    p = Pair(sc)
    p.step(5)
    lone_id = int(p.g.island_labels()[lone])
    p.set_pose(lone, [1.0, 0.5, 0.0, 0.0, 0.0, 0.0, 1.0])
    p.step(2, "lone box moved next to the stack")
    assert p.g.island_state(lone_id)["used"] == 0
    bd = S.body_desc(translation=(50.0, 0.5, 0.0), can_sleep=1)
    cd = S.collider_desc(half_extents=(0.5, 0.5, 0.5))
    h = p.g.insert_body(bd); p.g.insert_collider(cd, h)
    ho = p.o.add_body(translation=(50.0, 0.5, 0.0), can_sleep=1); p.o.add_collider(ho, half_extents=(0.5, 0.5, 0.5))
    assert int(h) == ho
    p.step(3, "a body inserted into the running world")
    assert int(p.g.island_labels()[int(h)]) == lone_id


def test_joint_links_in_insertion_order_bit_exact():
    p = Pair(S.reference_pile(2, 1, 2, chain=True))
    p.step(3, "pile + chain")
    lab = p.g.island_labels()
    assert len({int(x) for x in lab[-4:]}) == 1
    p.run(150)


def test_reference_golden_scene_islands_bit_exact():
    """the scene of the reference's bitwise golden (simd_backend_determinism.rs:61-139): states, sleeping flags and island ids every
    10 steps, the state hash at the end equals the oracle's"""
    from test_reference_golden import fnv1a_state_hash
    p = Pair(S.reference_pile(12, 3, 12, chain=True))
    for _ in range(12):
        p.run(10)
    gp, gv = p.g.read_bodies(); op, ov = p.o.read()
    assert fnv1a_state_hash(gp, gv) == fnv1a_state_hash(op, ov)
    assert int(p.g.sleeping().sum()) >= 432


def test_churn_with_sleeping_many_pyramids_islands():
    """196 pyramids with sleeping allowed: a pyramid is one island from its first step on; kicked apart, its cubes leave the island one
    by one (hot removals, deferred splits) and everything falls asleep again, ids and flags identical to the oracle throughout"""
    sc = S.many_pyramids(rows=2, cols=2).enable_sleep()
    p = Pair(sc)
    p.run(60)
    lab = p.g.island_labels()
    assert len({int(x) for x in lab if x >= 0}) == 4
    rng = np.random.default_rng(5)
    for b in rng.choice(np.arange(1, 56), 12, replace=False):
        p.set_vel(int(b), tuple(float(x) for x in rng.uniform(-8, 8, 3)), tuple(float(x) for x in rng.uniform(-6, 6, 3)))
    for _ in range(30):
        p.run(10)
    st = p.g.island_stats()
    assert st["removals"] > 0 and st["global_splits"] + st["detached"] > 0


# ---- 32-bit step stamps move back long before they can wrap (k_rebase_stamps; round 5) ------------------------------------------------
def _rebase_body():
    """sleeping, waking, island splits and solver hints across many rebases of the step stamps: every few steps the device's FL_STEP
    and everything stamped with it moves back by RP_TEST_REBASE_AT steps — the world must not notice"""
    import ctypes as C
    for sc, steps in ((S.sleep_impact(), 720), (S.many_pyramids(rows=1, cols=2).enable_sleep(), 300)):
        g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
        for k in range(steps // 6):
            g.step(6); o.step(6)
            gp, gv = g.read_bodies(); op, ov = o.read()
            np.testing.assert_array_equal(gp, op, err_msg=f"{sc.name} +{6 * (k + 1)}"); np.testing.assert_array_equal(gv, ov)
            np.testing.assert_array_equal(g.sleeping(), o.sleeping(), err_msg=f"{sc.name} sleep states +{6 * (k + 1)}")
            if sc.name.startswith("many") and k == 30:   # wake one pyramid up long after everything fell asleep
                v = np.array([[1.0, 2.0, 0.0, 0.0, 0.0, 0.0]], np.float32)
                g.write_bodies([55], vel6=v); o.set_vel(55, v[0, :3], v[0, 3:]); g.wake_up([55]); o.wake_up(55)
        assert g.sleeping().any() and g.counters()["overflow_flags"] == 0
        L = g._lib; L.rp_debug_rebases.restype = C.c_int64; L.rp_debug_rebases.argtypes = [C.c_void_p]
        assert L.rp_debug_rebases(g._ptr) > steps // 40, L.rp_debug_rebases(g._ptr)
    # events queued ACROSS rebases keep their order and their step numbers (the queue is re-stamped with the rest, the host
    # adds back what the stamps moved): drained only every 60 steps, compared with the oracle's list, steps included
    sc = S.tumble(40, seed=11).enable_events(S.ACTIVE_EVENTS_COLLISION | S.ACTIVE_EVENTS_CONTACT_FORCE, 2.0)
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    for k in range(4):
        for _ in range(10):
            g.step(6); o.step(6); g.sync()                      # (the stamps move back inside settle(): let it run between batches)
        ge = [tuple(int(x) for x in e) for e in g.collision_events()]; oe = sorted(tuple(int(x) for x in e) for e in o.collision_events())
        assert sorted(ge) == oe and len(ge) > 0, (k, ge[:4], oe[:4])
        assert [e[4] for e in ge] == sorted(e[4] for e in ge), "events are handed out oldest first"
        gm, gv = g.contact_force_events(); om, ov = o.force_events()
        go = np.lexsort((gm[:, 1], gm[:, 0], gm[:, 2])); oo = np.lexsort((om[:, 1], om[:, 0], om[:, 2]))
        np.testing.assert_array_equal(gm[go], om[oo]); np.testing.assert_array_equal(gv[go], ov[oo])
    assert L.rp_debug_rebases(g._ptr) > 4


def test_step_stamps_move_back_without_a_trace():
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "rapier_amd", "librapier_hip_testing.so")
    assert os.path.exists(lib), "build the testing library: make -C rapier_amd/csrc testing"
    code = f"import sys; sys.path[:0] = [{root!r}, {os.path.join(root, 'tests')!r}]; import test_gpu_islands as t; t._rebase_body()"
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, RP_HIP_LIB=lib, RP_TEST_REBASE_AT="16"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
