"""Arena slot reuse with generations (data/arena.rs:28-90, 260-380) through the C ABI, against the oracle bit for bit.

Restates crates/rapier3d/tests/issue_617_broad_phase_memory_growth.rs: one body spawned per tick, at most ten alive, 400 ticks — the
reference pins that the broad phase's size plateaus under that churn; here the device world's ROW COUNT plateaus (the eleven slots of the
body and of the collider arena are handed out again, LIFO, with a fresh generation each time), the pair set holds no dead pair, and the
poses equal the oracle's every tick."""
import numpy as np
import pytest

from rapier_amd import PhysicsWorld, scenes as S
from rapier_amd.world import RapierHipError
from oracle_ffi import OracleWorld, lib

pytestmark = pytest.mark.gpu


def _lcg():
    state = 0x12345678
    while True:
        state = (state * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        yield np.float32(np.float32(state >> 33) / np.float32(1 << 31))


def _empty_scene():
    sc = S.Scene(name="issue_617")
    sc.gravity = (0.0, -9.81, 0.0)
    return sc


def test_issue_617_spawn_despawn_churn_reuses_slots_bit_exact():
    sc = _empty_scene()
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    rnd = _lcg()
    alive = []   # (device body handle, oracle body index)
    issued = []  # every body handle ever handed out
    rows_at_100 = 0
    for tick in range(400):
        p = (float(next(rnd) * np.float32(100.0)), float(next(rnd) * np.float32(100.0)), float(next(rnd) * np.float32(100.0)))
        body = S.body_desc(translation=p)
        col = S.collider_desc(half_extents=(1.0, 1.0, 1.0))
        hb = g.insert_body(body)
        hc = g.insert_collider(col, hb)
        ob = lib().ro_add_body(o._w, np.array([body], S.BODY_DTYPE).ctypes.data)
        oc = lib().ro_add_collider(o._w, np.array([col], S.COLLIDER_DTYPE).ctypes.data, ob)
        # the same slot and the same generation as the reference's arena hands out
        assert (int(hb) & 0xFFFFFFFF, int(hb) >> 32) == (ob, o.body_generation(ob)), (tick, hex(int(hb)), ob, o.body_generation(ob))
        assert (int(hc) & 0xFFFFFFFF, int(hc) >> 32) == (oc, o.collider_generation(oc)), (tick, hex(int(hc)), oc)
        alive.append((hb, ob)); issued.append(int(hb))
        while len(alive) > 10:
            hb0, ob0 = alive.pop(0)
            g.remove_body([hb0]); o.remove_body(ob0)
            with pytest.raises(RapierHipError):
                g.remove_body([hb0])            # the handle died with the body
        g.step(1); o.step(1)
        if tick % 10 == 0 or tick > 380:
            gp, gv = g.read_bodies([h for h, _ in alive]); op, ov = o.read()
            idx = [b for _, b in alive]
            np.testing.assert_array_equal(gp, op[idx], err_msg=f"tick {tick}"); np.testing.assert_array_equal(gv, ov[idx], err_msg=f"tick {tick}")
        rows = g._lib.rp_num_bodies(g._ptr)
        if tick == 100:
            rows_at_100 = rows
        elif tick > 100:
            assert rows <= rows_at_100, (tick, rows, rows_at_100)
    assert rows_at_100 == 11 and o.n == 11                       # ten alive + the slot freed last
    # a handle of an earlier occupant of a slot that is in use again is stale: same index, older generation
    assert issued[30] >> 32 > 0 and issued[30] & 0xFFFFFFFF in [int(h) & 0xFFFFFFFF for h, _ in alive]
    with pytest.raises(RapierHipError):
        g.read_bodies([issued[30]])
    with pytest.raises(RapierHipError):
        g.apply_impulse([issued[30]], impulse=(1.0, 0.0, 0.0))
    c = g.counters()
    assert c["overflow_flags"] == 0 and c["num_pairs"] == o.stats()["num_pairs"], c


def test_a_slot_reused_before_any_step_ran_keeps_no_pair_of_its_previous_occupant():
    """remove a box from the middle of a settled stack and, in the same tick, insert a new body (it takes the freed body slot and its
    collider the freed collider slot) somewhere else: the removed collider's touching pairs must be gone before the slot is handed out
    again — Stopped | REMOVED events included — not be taken for pairs of the new occupant"""
    sc = S.box_stack(5).enable_events(S.ACTIVE_EVENTS_COLLISION, 0.0)
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    g.step(60); o.step(60)
    g.collision_events(); o.collision_events()
    victim = 3
    g.remove_body([victim]); o.remove_body(victim)
    body = S.body_desc(translation=(6.0, 3.0, 0.0), linvel=(0.0, -1.0, 0.0))
    col = S.collider_desc(half_extents=(0.4, 0.4, 0.4), density=3.0, active_events=S.ACTIVE_EVENTS_COLLISION)
    hb = g.insert_body(body); hc = g.insert_collider(col, hb)
    ob = lib().ro_add_body(o._w, np.array([body], S.BODY_DTYPE).ctypes.data)
    oc = lib().ro_add_collider(o._w, np.array([col], S.COLLIDER_DTYPE).ctypes.data, ob)
    assert int(hb) & 0xFFFFFFFF == ob == victim and int(hb) >> 32 == 1 and int(hc) & 0xFFFFFFFF == oc
    for n in (1, 30, 120):
        g.step(n); o.step(n)
        gp, gv = g.read_bodies(); op, ov = o.read()
        np.testing.assert_array_equal(gp, op); np.testing.assert_array_equal(gv, ov)
    ge, oe = g.collision_events(), o.collision_events()
    key = lambda e: np.lexsort((e[:, 2], e[:, 1], e[:, 0], e[:, 4]))  # noqa: E731
    assert len(ge) == len(oe) and len(ge) > 0
    np.testing.assert_array_equal(ge[key(ge)], oe[key(oe)])
    assert g.counters()["num_pairs"] == o.stats()["num_pairs"]


@pytest.mark.parametrize("spare", [None, 1])
def test_a_body_in_a_reused_slot_has_the_mass_of_its_own_colliders(monkeypatch, spare):
    """remove a compound body, let a step go by, insert a different body with an offset, rotated collider (and additional mass) — it takes
    the freed body slot, its collider the slot freed last — and hit it with an impulse and a torque impulse: the response must be the
    oracle's, i.e. that of the new body's own mass properties (in place and with RP_SPARE_ROWS=1, where later appends move the world
    through the growth carry-over)"""
    if spare is not None:
        monkeypatch.setenv("RP_SPARE_ROWS", str(spare))
    sc = S.compound_bodies(6)
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    g.step(5); o.step(5)
    dyn = [i for i, b in enumerate(sc.bodies) if int(b["body_type"]) == S.BODY_DYNAMIC]
    victim = dyn[2]
    g.remove_body([victim]); o.remove_body(victim)
    g.step(2); o.step(2)

    def add(pos, he, dens, addm, rot):
        body = S.body_desc(translation=pos, rotation=rot, additional_mass=addm)
        col = S.collider_desc(half_extents=he, density=dens, translation=(0.1, 0.2, 0.0), rotation=(0.0, 0.3, 0.1, 0.9))
        hb = g.insert_body(body); hc = g.insert_collider(col, hb)
        ob = lib().ro_add_body(o._w, np.array([body], S.BODY_DTYPE).ctypes.data)
        oc = lib().ro_add_collider(o._w, np.array([col], S.COLLIDER_DTYPE).ctypes.data, ob)
        assert int(hb) & 0xFFFFFFFF == ob and int(hc) & 0xFFFFFFFF == oc
        return hb, ob

    hb, ob = add((-20.0, 8.0, 3.0), (0.2, 0.7, 0.4), 3.0, 0.5, (0.1, 0.2, 0.3, 0.9))
    assert ob == victim and int(hb) >> 32 == 1
    hb2, ob2 = add((20.0, 8.0, -3.0), (0.3, 0.3, 0.6), 2.0, 0.0, (0.0, 0.0, 0.0, 1.0))     # a fresh row (growth with RP_SPARE_ROWS=1)
    for h, b_ in ((hb, ob), (hb2, ob2)):
        g.apply_impulse([h], impulse=(1.0, 2.0, 3.0), torque_impulse=(0.1, 0.2, 0.3)); o.apply_impulse(b_, impulse=(1.0, 2.0, 3.0), torque_impulse=(0.1, 0.2, 0.3))
    for n in (1, 5, 30):
        g.step(n); o.step(n)
        gp, gv = g.read_bodies(); op, ov = o.read()
        alive = [i for i in range(len(op)) if i != 0 or True]
        np.testing.assert_array_equal(gp[alive], op[alive], err_msg=f"+{n} poses"); np.testing.assert_array_equal(gv[alive], ov[alive], err_msg=f"+{n} velocities")


@pytest.mark.parametrize("spare", [None, 1])
def test_ccd_clamps_a_fast_body_that_lives_in_a_reused_slot(monkeypatch, spare):
    """a body removed, another body given a second collider (with RP_SPARE_ROWS=1 that append moves the world to larger arrays), then a
    new body in the freed slot thrown at the ground at 300 m/s: the continuous-collision pass must clamp its pose like the oracle's
    (found by the growth fuzz, seed 3000: the clamp was missing for the body in the reused slot)"""
    if spare is not None:
        monkeypatch.setenv("RP_SPARE_ROWS", str(spare))
    sc = S.compound_bodies(6)
    g, o = PhysicsWorld.from_scene(sc), OracleWorld(sc)
    g.step(5); o.step(5)
    dyn = [i for i, b in enumerate(sc.bodies) if int(b["body_type"]) == S.BODY_DYNAMIC]
    victim, other = dyn[1], dyn[3]
    g.remove_body([victim]); o.remove_body(victim)
    g.step(2); o.step(2)
    extra = S.collider_desc(half_extents=(0.2, 0.2, 0.2), translation=(0.3, 0.0, 0.0))
    extra2 = S.collider_desc(half_extents=(0.25, 0.2, 0.2), translation=(-0.3, 0.1, 0.0))
    for col in (extra, extra2):                # the first takes the freed collider slot(s), the second is appended
        hc = g.insert_collider(col, other); oc = lib().ro_add_collider(o._w, np.array([col], S.COLLIDER_DTYPE).ctypes.data, other)
        assert int(hc) & 0xFFFFFFFF == oc
    g.step(2); o.step(2)
    body = S.body_desc(translation=(30.0, 6.0, 30.0), linvel=(40.0, -300.0, 0.0))
    col = S.collider_desc(half_extents=(0.3, 0.3, 0.3), density=2.0)
    hb = g.insert_body(body); hc = g.insert_collider(col, hb)
    ob = lib().ro_add_body(o._w, np.array([body], S.BODY_DTYPE).ctypes.data); oc = lib().ro_add_collider(o._w, np.array([col], S.COLLIDER_DTYPE).ctypes.data, ob)
    assert int(hb) & 0xFFFFFFFF == ob == victim and int(hc) & 0xFFFFFFFF == oc
    for n in (1, 1, 5, 40):
        g.step(n); o.step(n)
        gp, gv = g.read_bodies(); op, ov = o.read()
        np.testing.assert_array_equal(gp, op, err_msg=f"+{n} poses"); np.testing.assert_array_equal(gv, ov, err_msg=f"+{n} velocities")
    assert g.counters()["ccd_clamp_count"] > 0


def test_a_stale_generation_zero_handle_is_refused_not_retargeted():
    """a handle of the initial scene has generation 0 and equals its index; after its slot was reused it must be REFUSED by
    the Python wrapper too (Arena::get -> None, data/arena.rs), not rewritten to the slot's new occupant."""
    sc = S.Scene(name="stale0", gravity=(0.0, -9.81, 0.0))
    gb = sc.add_body(body_type=S.BODY_FIXED, translation=(0.0, -1.0, 0.0)); sc.add_collider(gb, half_extents=(10.0, 1.0, 10.0))
    a = sc.add_body(translation=(0.0, 0.5, 0.0)); sc.add_collider(a)
    g = PhysicsWorld.from_scene(sc, index_addressing=False)
    g.step(2)
    old = int(g.body_handles()[a])
    assert old == a                                              # generation 0: the handle IS the index
    g.remove_body([old])
    new = int(g.insert_body(S.body_desc(translation=(3.0, 0.5, 0.0))))
    assert new & 0xFFFFFFFF == a and new >> 32 > 0               # the slot is reused under a higher generation
    for call in (lambda: g.read_bodies([old]), lambda: g.remove_body([old]), lambda: g.write_bodies([old], vel6=np.zeros((1, 6), np.float32)),
                 lambda: g.apply_impulse([old], impulse=(1.0, 0.0, 0.0))):
        with pytest.raises(RapierHipError):
            call()
    assert int(g.body_handles_at([a])[0]) == new                 # explicit index addressing names the current occupant
    pos, _ = g.read_bodies([new])
    assert abs(float(pos[0, 0]) - 3.0) < 1e-6


def test_impulse_joint_handles_are_generational_and_bodies_are_named_by_handle():
    """ImpulseJointSet is arena-backed in the reference (joint_ids: Arena<..>, impulse_joint_set.rs:48; insert(body1: RigidBodyHandle,
    body2: RigidBodyHandle, ..) :329-375; remove :595-618): a removed joint's slot is handed out again, LIFO, under a higher generation
    and the old handle is refused; rp_joint_desc names its bodies by 64-bit handles, so a stale body handle is refused as well
    (VERDICT r5 missing #4).  Strict handles throughout (index_addressing=False)."""
    sc = S.Scene(name="joint_arena", gravity=(0.0, -9.81, 0.0))
    gb = sc.add_body(body_type=S.BODY_FIXED, translation=(0.0, 5.0, 0.0)); sc.add_collider(gb, half_extents=(0.2, 0.2, 0.2))
    balls = []
    for k in range(4):
        b = sc.add_body(translation=(1.0 + k, 5.0, 0.0)); sc.add_collider(b, shape=S.SHAPE_BALL, half_extents=(0.3, 0.0, 0.0)); balls.append(b)
    g = PhysicsWorld.from_scene(sc, index_addressing=False)
    hb = [int(h) for h in g.body_handles()]
    tmp = S.Scene(name="tmp"); tmp.add_joint(0, 0, (0.5, 0, 0), (-0.5, 0, 0)); proto = tmp.joints[0]
    j0 = g.insert_impulse_joint(hb[gb], hb[balls[0]], proto)
    j1 = g.insert_impulse_joint(hb[balls[0]], hb[balls[1]], proto)
    j2 = g.insert_impulse_joint(hb[balls[1]], hb[balls[2]], proto)
    assert (j0, j1, j2) == (0, 1, 2)                                   # generation 0: the handle is the slot index
    g.step(3)
    g.remove_impulse_joint([j1])
    with pytest.raises(RapierHipError):
        g.remove_impulse_joint([j1])                                  # already removed
    with pytest.raises(RapierHipError):
        g.set_joint_motor([j1], [3], target_vel=1.0, damping=1.0)
    j3 = g.insert_impulse_joint(hb[balls[2]], hb[balls[3]], proto)
    assert j3 & 0xFFFFFFFF == 1 and j3 >> 32 == 1                      # the freed slot, generation = the arena's removal count
    assert [int(h) for h in g.joint_handles()] == [j0, 0xFFFFFFFFFFFFFFFF, j2, j3]
    with pytest.raises(RapierHipError):
        g.remove_impulse_joint([j1])                                  # the stale handle does not reach the slot's new occupant
    g.set_joint_motor([j3], [3], target_vel=0.5, damping=1.0)          # ... the current one does
    # a joint names its bodies by handle: a stale body handle is refused
    victim = hb[balls[3]]
    g.remove_body([victim])                                            # takes j3 with it (RigidBodySet::remove removes attached joints)
    assert int(g.joint_handles()[3]) == 0xFFFFFFFFFFFFFFFF
    nb = int(g.insert_body(S.body_desc(translation=(9.0, 5.0, 0.0))))
    assert nb & 0xFFFFFFFF == balls[3] and nb >> 32 > 0
    with pytest.raises(RapierHipError):
        g.insert_impulse_joint(hb[balls[2]], victim, proto)            # generation 0 handle of a reused slot
    j4 = g.insert_impulse_joint(hb[balls[2]], nb, proto)
    assert j4 & 0xFFFFFFFF == 1 and j4 >> 32 == 2
    g.step(20)
    pos, _ = g.read_bodies()
    assert np.isfinite(pos).all()
    # the chain gb - b0, b1 - b2 - nb holds: the two anchors of every live joint coincide in world space
    def anchor(b, local):
        t, (x, y, z, w_) = pos[b, :3].astype(np.float64), pos[b, 3:].astype(np.float64)
        u = np.array([x, y, z]); v = np.asarray(local, np.float64)
        return t + v + 2.0 * np.cross(u, np.cross(u, v) + w_ * v)
    for b1, b2 in ((gb, balls[0]), (balls[1], balls[2]), (balls[2], balls[3])):
        assert np.linalg.norm(anchor(b1, (0.5, 0, 0)) - anchor(b2, (-0.5, 0, 0))) < 0.02, (b1, b2)
    assert np.linalg.norm(anchor(balls[0], (0.5, 0, 0)) - anchor(balls[1], (-0.5, 0, 0))) > 0.05   # the removed joint holds nothing
