"""Generates tests/golden/*.npz: body poses/velocities after N steps of small fixed scenes.

The reference itself (Rust, v0.35.2) cannot be built or run in this environment and is not a
Python package, so these vectors come from the CPU oracle after it passed the reference's
known-answer tests (tests/test_oracle_kat.py).  Run: python -m tests.golden.make_golden
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from rapier_amd import scenes as S  # noqa: E402

CASES = {
    "box_stack3_s60": lambda: (S.box_stack(3), 60),
    "pyramid10_s120": lambda: (S.pyramid10(), 120),
    "tumble40_s90": lambda: (S.tumble(40, seed=11), 90),
    "many_pyramids_2x2_s30": lambda: (S.many_pyramids(rows=2, cols=2), 30),
    "joint_chain8_s200": lambda: (S.joint_chain(8), 200),
    "joint_grid12_s100": lambda: (S.joint_grid(12), 100),
    # feature scenes (sleeping, Coulomb friction, kinematic platform, locked angular joint axes, compound bodies, locked axes)
    "sleep_impact_s200": lambda: (S.sleep_impact(), 200),
    "pyramid10_coulomb_s120": lambda: (_coulomb(S.pyramid10()), 120),
    "kinematic_platform_vel_s120": lambda: (S.kinematic_platform(False), 120),
    "jointed_pairs2_s150": lambda: (S.jointed_pairs(2), 150),
    "compound_bodies6_s150": lambda: (S.compound_bodies(6), 150),
    "locked_axes_s150": lambda: (S.locked_axes_scene(), 150),
    "overlapping_chain6_s100": lambda: (S.overlapping_chain(6, 0), 100),
    "limited_joints_s150": lambda: (S.limited_joints(), 150),
    "motorised_joints_s150": lambda: (S.motorised_joints(), 150),
    "capsules6_s150": lambda: (S.capsules(6), 150),
    "reference_pile_12x3x12_s100": lambda: (S.reference_pile(12, 3, 12, chain=True), 100),
    # round 2: substep solve-groups (additional_solver_iterations) and sensors
    "solve_groups_s150": lambda: (S.solve_groups_scene(), 150),
    "sensors_s200": lambda: (S.sensor_scene(), 200),
    # half-spaces (ground plane, slanted plane, rising kinematic plane) under cuboids, balls and capsules
    "halfspaces_s240": lambda: (S.halfspace_scene(), 240),
}


def _coulomb(scene):
    scene.params["friction_model"] = S.FRICTION_COULOMB
    return scene


if __name__ == "__main__":
    from oracle_ffi import OracleWorld
    only = set(sys.argv[1:])
    for name, mk in CASES.items():
        if only and name not in only:
            continue
        scene, steps = mk()
        w = OracleWorld(scene)
        w.step(steps)
        pos, vel = w.read()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), pos=pos, vel=vel)
        print(name, pos.shape, float(np.abs(pos).max()))
