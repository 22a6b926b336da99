"""The continuous-collision pass on the device (rp_ccd.h: k_ccd) against the oracle's twin, bit for bit: the reference's CCD scenes
(tests/test_ccd_oracle.py restates them), every fast shape against every target shape, tumbling bodies, bullets against dynamic and
kinematic targets — body states and the (activation, clamp) counters after every step."""
import numpy as np
import pytest

from rapier_amd import PhysicsWorld, scenes as S
from oracle_ffi import OracleWorld
import test_ccd_oracle as T

pytestmark = pytest.mark.gpu


def _twin(scene, steps, every=1):
    g = PhysicsWorld.from_scene(scene)
    o = OracleWorld(scene)
    for k in range(0, steps, every):
        g.step(every); o.step(every)
        gp, gv = g.read_bodies(); op, ov = o.read()
        np.testing.assert_array_equal(gp, op, err_msg=f"{scene.name} poses @ {k + every}")
        np.testing.assert_array_equal(gv, ov, err_msg=f"{scene.name} velocities @ {k + every}")
        c = g.counters()
        assert (c["ccd_active_count"], c["ccd_clamp_count"]) == o.ccd_counts(), (k + every, c["ccd_active_count"], c["ccd_clamp_count"], o.ccd_counts())
    return g, o


def test_default_tier_against_a_fixed_wall_bit_exact():
    sc, body = T.default_ccd_vs_fixed_scene()
    g, o = _twin(sc, 120)
    assert g.read_bodies()[0][body, 0] < 0.0 and g.counters()["ccd_clamp_count"] >= 1


def test_bullet_against_a_dynamic_target_bit_exact():
    sc, bullet, target = T.bullet_vs_dynamic_scene()
    g, o = _twin(sc, 60)
    pos = g.read_bodies()[0]
    assert pos[target, 0] > 0.05 and pos[bullet, 0] < pos[target, 0]


def test_large_dt_and_tiled_floor_bit_exact():
    _twin(T.large_dt_scene()[0], 10)
    sc, ball = T.tiled_floor_scene()
    g, o = _twin(sc, 40)
    assert g.read_bodies()[0][ball, 0] > 15.0


def test_ccd_off_tunnels_like_the_oracle():
    sc = T.harness(max_ccd_substeps=0)
    T.thin_fixed_wall(sc)
    body = T.fast_dynamic(sc, False)
    g, o = _twin(sc, 30)
    assert g.read_bodies()[0][body, 0] > 1.0 and g.counters()["ccd_active_count"] == 0


@pytest.mark.parametrize("shape,he", [(S.SHAPE_CUBOID, (0.1, 0.2, 0.15)), (S.SHAPE_BALL, (0.15, 0.0, 0.0)), (S.SHAPE_CAPSULE, (0.2, 0.08, 0.0))])
def test_every_shape_pair_bit_exact(shape, he):
    """one world per fast shape: four fast tumbling bodies fly at a wall, a half-space, a ball and a capsule (fixed), a fifth — a bullet —
    at a kinematic platform and a sixth bullet at a resting dynamic box"""
    sc = T.harness()
    t = sc.add_body(body_type=S.BODY_FIXED)
    sc.add_collider(t, half_extents=(0.05, 5.0, 5.0), translation=(0.0, 0.0, 0.0))
    sc.add_collider(t, shape=S.SHAPE_BALL, half_extents=(1.0, 0.0, 0.0), translation=(0.0, 0.0, 20.0))
    sc.add_collider(t, shape=S.SHAPE_CAPSULE, half_extents=(2.0, 0.5, 1.0), translation=(0.0, 0.0, 40.0))
    plane = sc.add_body(body_type=S.BODY_FIXED, translation=(0.0, 0.0, 60.0))
    sc.add_collider(plane, shape=S.SHAPE_CAPSULE, half_extents=(3.0, 0.4, 2.0))
    kin = sc.add_body(body_type=S.BODY_KINEMATIC_VELOCITY, translation=(0.0, 0.0, 80.0), linvel=(-1.0, 0.0, 0.0))
    sc.add_collider(kin, half_extents=(0.1, 3.0, 3.0))
    rest = sc.add_body(translation=(0.0, 0.0, 100.0))
    sc.add_collider(rest, half_extents=(0.3, 0.3, 0.3))
    for k, z in enumerate((0.0, 20.0, 40.0, 60.0, 80.0, 100.0)):
        b = sc.add_body(translation=(-3.0 - 0.1 * k, 0.05 * k, z), linvel=(200.0 - 7.0 * k, 0.0, 0.0), angvel=(3.0, -7.0 + k, 5.0), ccd_enabled=1 if z >= 80.0 else 0)
        sc.add_collider(b, shape=shape, half_extents=he)
    g, o = _twin(sc, 30)
    assert g.counters()["ccd_clamp_count"] >= 5


def test_scenes_with_falling_bodies_bit_exact():
    """feature scenes in which bodies hit the ground fast enough to be swept"""
    for sc, steps in ((S.tumble(40, seed=11), 90), (S.capsules(6), 100), (S.halfspace_scene(), 120), (S.compound_bodies(6), 100)):
        g, o = _twin(sc, steps, every=10)
        assert g.counters()["ccd_clamp_count"] > 0


# ---- composite targets (round 5): a mesh / compound / height-field wall, a dynamic compound hit by a bullet ------------------------------
@pytest.mark.parametrize("kind", ["trimesh", "compound", "heightfield"])
def test_composite_wall_stops_the_fast_body_bit_exact(kind):
    sc, body = T.mesh_wall_scene(kind)
    g, o = _twin(sc, 120, every=4)
    assert g.read_bodies()[0][body, 0] < 0.0 and g.counters()["ccd_clamp_count"] >= 1


def test_bullets_rain_on_a_mesh_ground_and_a_dynamic_compound_bit_exact():
    """every shape kind shot at a triangle-mesh floor at 120 m/s under gravity (tier 0 against the fixed mesh), and a bullet against a
    floating compound plate (tier 1)"""
    from test_composite_oracle import _grid_mesh
    sc = T.harness(gravity=(0.0, -9.81, 0.0))
    g0 = sc.add_body(body_type=S.BODY_FIXED)
    v, t = _grid_mesh(8, 24.0)
    sc.add_collider(g0, shape=S.SHAPE_TRIMESH, half_extents=(sc.add_trimesh(v, t), 0, 0))
    kinds = [(S.SHAPE_BALL, (0.15, 0.0, 0.0)), (S.SHAPE_CUBOID, (0.12, 0.1, 0.15)), (S.SHAPE_CAPSULE, (0.2, 0.08, 1.0)), (S.SHAPE_CYLINDER, (0.15, 0.1, 0.0)), (S.SHAPE_CONE, (0.15, 0.12, 0.0))]
    for k, (sh, he) in enumerate(kinds):
        b = sc.add_body(translation=(-4.0 + 2.0 * k, 3.0, -2.0 + 0.9 * k), linvel=(0.3, -120.0, 0.1), angvel=(2.0, 0.5, -1.0), ccd_enabled=k % 2)
        sc.add_collider(b, shape=sh, half_extents=he)
    gp, o = _twin(sc, 90, every=3)
    assert (gp.read_bodies()[0][1:, 1] > -0.5).all()        # nobody fell through the mesh (it is 24 m wide: nobody reaches its edge either)
    sc2 = T.harness()
    d = sc2.add_body(translation=(0.0, 0.0, 0.0))
    cid = sc2.add_compound([S.collider_desc(half_extents=(0.05, 1.0, 1.0)), S.collider_desc(shape=S.SHAPE_BALL, half_extents=(0.3, 0.0, 0.0), translation=(0.0, 1.2, 0.0))])
    sc2.add_collider(d, shape=S.SHAPE_COMPOUND, half_extents=(cid, 0, 0), density=50.0)
    T.fast_dynamic(sc2, True)
    _twin(sc2, 30)


def test_fast_compound_body_swept_part_by_part_bit_exact():
    sc, b = T.fast_compound_scene()
    g, o = _twin(sc, 60, every=2)
    assert g.read_bodies()[0][b, 0] < 0.0 and g.counters()["ccd_clamp_count"] >= 1
    sc2, b2 = T.fast_compound_scene(ccd=True)            # ... as a bullet too
    _twin(sc2, 30)
