"""Batches of small worlds on the device (rp_world_begin_subworld): the sub-worlds of one device world overlap in
space, never pair, and are stepped by the same launches.  Parity: the device batch equals the ORACLE batch bit for bit (any mix of
scenes), and a batch of copies equals the scene stepped alone, bit for bit, in every copy."""
import numpy as np
import pytest

from rapier_amd import PhysicsWorld, scenes as S, step_many
from oracle_ffi import OracleWorld

pytestmark = pytest.mark.gpu


def _same(g, o, msg):
    gp, gv = g.read_bodies(); op, ov = o.read()
    assert np.isfinite(gp).all()
    np.testing.assert_array_equal(gp, op, err_msg=msg); np.testing.assert_array_equal(gv, ov, err_msg=msg)
    c, st = g.counters(), o.stats()
    assert c["overflow_flags"] == 0 and c["num_pairs"] == st["num_pairs"] and c["num_manifolds"] == st["num_active_manifolds"], (c, st)


def test_batch_of_copies_every_copy_evolves_alike_and_like_the_oracle_batch():
    """16 copies of capsules(6) in the same place: device batch = oracle batch, and all copies carry the same bits.  (Not the bits of
    the scene stepped ALONE: a batch is ONE reference world — init.rs:163-254 sweeps its colours with >= 32 chunks first, and 16
    copies lift colours over that line that a single copy leaves below it.)"""
    sc = S.capsules(6)
    n = 16
    b = S.batch([S.capsules(6) for _ in range(n)])
    g, o = PhysicsWorld.from_scene(b), OracleWorld(b)
    assert g.subworlds[3] == (3 * len(sc.bodies), 4 * len(sc.bodies))
    done = 0
    for cp in (1, 2, 30, 120, 300):
        d, done = cp - done, cp
        g.step(d); o.step(d)
        _same(g, o, f"batch of {n} capsules(6) @ {cp}")
        gp, gv = g.read_bodies()
        b0, b1 = g.subworlds[0]
        for k, (c0, c1) in enumerate(g.subworlds):
            np.testing.assert_array_equal(gp[c0:c1], gp[b0:b1], err_msg=f"copy {k} @ {cp}"); np.testing.assert_array_equal(gv[c0:c1], gv[b0:b1])


def test_small_batch_of_copies_equals_the_scene_stepped_alone():
    """4 copies: every colour stays on the same side of the >= 32-chunk line as in the single scene, so each copy is the scene alone, bit for bit"""
    sc = S.capsules(6)
    alone = PhysicsWorld.from_scene(sc)
    g = PhysicsWorld.from_scene(S.batch([S.capsules(6) for _ in range(4)]))
    done = 0
    for cp in (1, 30, 120, 300):
        d, done = cp - done, cp
        alone.step(d); g.step(d)
        ap, av = alone.read_bodies(); gp, gv = g.read_bodies()
        for k, (b0, b1) in enumerate(g.subworlds):
            np.testing.assert_array_equal(gp[b0:b1], ap, err_msg=f"copy {k} @ {cp}"); np.testing.assert_array_equal(gv[b0:b1], av)


def test_mixed_batch_bit_exact_against_the_oracle_batch():
    parts = [S.box_stack(3), S.pyramid10(), S.joint_chain(4, with_boxes=True), S.box_stack(2, gap=0.5), S.tumble(24, seed=5), S.capsules(4)]
    for p in parts:
        p.gravity, p.params = parts[0].gravity, parts[0].params.copy()
    b = S.batch(parts)
    g, o = PhysicsWorld.from_scene(b), OracleWorld(b)
    done = 0
    for cp in (1, 5, 40, 160, 320):
        g.step(cp - done); o.step(cp - done); done = cp
        _same(g, o, f"mixed batch @ {cp}")
    # no pair ever links two sub-worlds
    m, _, _ = g.contacts()
    sub_of = np.zeros(len(b.colliders), np.int64)
    for k, (_, c0, _) in enumerate(b.subworlds):
        sub_of[c0:] = k
    assert len(m) > 0 and (sub_of[m[:, 0]] == sub_of[m[:, 1]]).all()


def test_sleeping_batch_with_events_and_a_late_sub_world():
    """sub-worlds fall asleep on their own; a sub-world added to a world that has been stepped (rp_world_begin_subworld on a live world)
    starts moving while the others sleep on"""
    parts = [S.box_stack(3).enable_sleep(), S.many_pyramids(rows=1, cols=1).enable_sleep()]
    for p in parts:
        p.gravity, p.params = parts[0].gravity, parts[0].params.copy()
    b = S.batch(parts)
    g, o = PhysicsWorld.from_scene(b), OracleWorld(b)
    for k in range(8):
        g.step(40); o.step(40)
        _same(g, o, f"sleeping batch +{40 * (k + 1)}")
        np.testing.assert_array_equal(g.sleeping(), o.sleeping())
    assert g.sleeping()[1:4].all()
    # a third sub-world, in the same place as the first (its boxes fall THROUGH the sleeping stack of sub-world 0 and its ground)
    from oracle_ffi import lib
    assert g.begin_subworld() == 2 and lib().ro_begin_subworld(o._w) == 2
    late = S.box_stack(2, gap=0.5)
    hb = {}
    for i, bd in enumerate(late.bodies):
        hb[i] = g.insert_body(bd)
        ob = lib().ro_add_body(o._w, np.array([bd], S.BODY_DTYPE).ctypes.data)
        assert (int(hb[i]) & 0xffffffff) == ob
    for c, p in zip(late.colliders, late.collider_parents):
        g.insert_collider(c, hb[p])
        lib().ro_add_collider(o._w, np.array([c], S.COLLIDER_DTYPE).ctypes.data, int(hb[p]) & 0xffffffff)
    o.n = lib().ro_num_bodies(o._w)
    for k in range(6):
        g.step(30); o.step(30)
        _same(g, o, f"after the late sub-world +{30 * (k + 1)}")
        np.testing.assert_array_equal(g.sleeping(), o.sleeping())
    assert g.sleeping()[1:4].all()                         # nobody woke the first stack: the newcomers never touched it


def test_step_many_advances_separate_worlds_together():
    scenes = [S.box_stack(3), S.pyramid10(), S.capsules(4)]
    gs = [PhysicsWorld.from_scene(sc) for sc in scenes]
    os_ = [OracleWorld(sc) for sc in scenes]
    for k in range(4):
        step_many(gs, 25)
        for o in os_:
            o.step(25)
        for g, o, sc in zip(gs, os_, scenes):
            _same(g, o, f"{sc.name} via rp_step_many +{25 * (k + 1)}")


def test_batch_with_polyhedra_and_a_mesh_world_bit_exact():
    """registered shapes travel with their sub-world (scenes.batch moves the ids): hull clutter twice and a box on a triangle mesh"""
    from test_composite_oracle import _box_on_mesh
    mesh_world, _ = _box_on_mesh()
    parts = [S.polyhedra_clutter(6, 2), mesh_world, S.polyhedra_clutter(6, 2), S.capsules(4)]
    for p in parts:
        p.gravity, p.params = parts[0].gravity, parts[0].params.copy()
    b = S.batch(parts)
    g, o = PhysicsWorld.from_scene(b), OracleWorld(b)
    done = 0
    for cp in (1, 3, 40, 160):
        g.step(cp - done); o.step(cp - done); done = cp
        _same(g, o, f"batch with registered shapes @ {cp}")


def test_tiny_islands_share_bundles_bit_exact(monkeypatch):
    """rp_islands.hip, lay_isl_number (round 5): once a world holds more island candidates than one resident pass of the island kernel,
    the components of at most 8 manifolds of a world WITHOUT sleeping are packed into shared islands ("bundles": a union of components
    that share no body gives each the bits it gets alone) instead of going to the global path.  RP_ISL_MANY=4 makes a batch of 24
    capsule worlds such a world: the bundled world, the same world with RP_NO_TINY_BUNDLES=1 (tiny islands on the global path, round 4's
    routing) and the oracle must agree bit for bit — and the bundled one must leave nothing but free-flying bodies to the global path."""
    b = S.batch([S.capsules(6) for _ in range(24)] + [S.tumble(24, seed=5)])
    monkeypatch.setenv("RP_ISL_MANY", "4")
    g = PhysicsWorld.from_scene(b); g.read_bodies()             # (the device world — and with it the switches — is built by the first call that needs it)
    monkeypatch.setenv("RP_NO_TINY_BUNDLES", "1")
    h = PhysicsWorld.from_scene(b); h.read_bodies()
    monkeypatch.delenv("RP_NO_TINY_BUNDLES"); monkeypatch.delenv("RP_ISL_MANY")
    o = OracleWorld(b)
    done, fewer = 0, 0
    for cp in (1, 2, 3, 10, 40, 120, 300, 450):
        d, done = cp - done, cp
        g.step(d); h.step(d); o.step(d)
        _same(g, o, f"bundled @ {cp}")
        gp, gv = g.read_bodies(); hp, hv = h.read_bodies()
        np.testing.assert_array_equal(gp, hp, err_msg=f"bundles vs routing @ {cp}"); np.testing.assert_array_equal(gv, hv)
        cg, ch = g.counters(), h.counters()
        assert cg["num_manifolds"] == ch["num_manifolds"]
        fewer += cg["num_global_bodies"] < ch["num_global_bodies"]
        assert cg["num_global_bodies"] <= ch["num_global_bodies"]
    assert fewer >= 3, "the bundles never took a tiny island off the global path"


def test_bundled_debris_at_rest_takes_fused_steps_bit_exact(monkeypatch):
    """400 separate boxes dropped a little above a slab, sleeping off: 400 islands of one manifold = bundles of 56 (the rebuild that first
    counts them leaves the layout dirty, the next one bundles: rp_islands.hip, k_layout_rebuild); once they lie still the world runs on
    the ONE-kernel fused step, whose validators then walk the bodies and pairs of bundles — same bits as the oracle
    and as the world with every island in a workgroup of its own (RP_NO_TINY_ROUTING=1), and fused steps must have been taken."""
    side = 20
    sc = S.Scene(name="debris", gravity=(0.0, -9.81, 0.0))
    g0 = sc.add_body(body_type=S.BODY_FIXED, translation=(0.0, -0.5, 0.0)); sc.add_collider(g0, half_extents=(2.0 * side, 0.5, 2.0 * side))
    for i in range(side * side):
        b = sc.add_body(translation=(2.0 * (i % side) - side, 0.5 + 0.002 * (i % 7), 2.0 * (i // side) - side)); sc.add_collider(b, half_extents=(0.5, 0.5, 0.5))
    g = PhysicsWorld.from_scene(sc); g.read_bodies()
    monkeypatch.setenv("RP_NO_TINY_ROUTING", "1")
    h = PhysicsWorld.from_scene(sc); h.read_bodies()
    monkeypatch.delenv("RP_NO_TINY_ROUTING")
    o = OracleWorld(sc)
    done = 0
    for cp in (1, 2, 30, 120, 400):
        d, done = cp - done, cp
        g.step(d); h.step(d); o.step(d)
        _same(g, o, f"debris @ {cp}")
        gp, gv = g.read_bodies(); hp, hv = h.read_bodies()
        np.testing.assert_array_equal(gp, hp, err_msg=f"bundles vs one island each @ {cp}"); np.testing.assert_array_equal(gv, hv)
    cg, ch = g.counters(), h.counters()
    assert cg["num_global_bodies"] == 0 and 0 < cg["num_islands"] <= 10 and ch["num_islands"] == side * side, (cg, ch)
    assert cg["fused_steps"] > 100, cg
