"""CPU tests of the host side: the C-ABI library loads and exports every declared symbol, the
descriptor layouts match the header, there is no CPU fallback, sharding logic."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from rapier_amd import _ffi, scenes as S, sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "rapier_hip.h")).read()
    declared = set(re.findall(r"\b(rp_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_ffi.SYMBOLS), declared ^ set(_ffi.SYMBOLS)
    L = _ffi.lib()
    for name in declared:
        assert hasattr(L, name), name


def test_descriptor_layouts_match_header():
    # sizes implied by include/rapier_hip.h (all 4-byte fields, no padding)
    assert S.PARAMS_DTYPE.itemsize == 14 * 4 + 8 * 4 + 2 * 4  # ... + min_ccd_dt + contact_clustering
    assert S.BODY_DTYPE.itemsize == 4 + 12 + 16 + 12 + 12 + 4 * 4 + 5 * 4 + 4 + 4  # ... + additional_solver_iterations + ccd_enabled
    assert S.COLLIDER_DTYPE.itemsize == 4 + 12 + 12 + 16 + 12 + 8 + 8 + 8 + 4 + 4  # ... + sensor + border_radius
    assert S.JOINT_DTYPE.itemsize == 16 + 24 + 32 + 8 + 4 + 48 + 4 + 6 * 24 + 4 + 4  # two 64-bit body handles ... + coupled_axes + reserved
    assert C.sizeof(_ffi.Counters) == 9 * 4 + 28 * 4  # ... + num_tiles, tile_sweeps, bp_large_list, lean_steps, fused_steps, num_islands, num_global_bodies, fused_disabled, fused_launches, joint_net_steps, joint_net_disabled
    p = S.default_params()
    q = np.zeros((), S.PARAMS_DTYPE)
    _ffi.lib().rp_default_params(q.ctypes.data)
    assert p.tobytes() == q.tobytes()  # IntegrationParameters::default() agrees on both sides


def test_shape_and_body_enums_agree_across_header_python_and_oracle():
    """the numeric shape / body-type codes are spelled three times (C header, scenes.py, the oracle's header): they must agree"""
    def enum_values(path, prefix):
        txt = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
        return {m.group(1): int(m.group(2)) for m in re.finditer(prefix + r"_([A-Z_]+)\s*=\s*(\d+)", txt)}
    hdr = enum_values(os.path.join(ROOT, "include", "rapier_hip.h"), "RP_SHAPE")
    ora = enum_values(os.path.join(ROOT, "oracle", "rapier_oracle.h"), "RO_SHAPE")
    assert hdr == ora == {"BALL": S.SHAPE_BALL, "CUBOID": S.SHAPE_CUBOID, "CAPSULE": S.SHAPE_CAPSULE, "HALFSPACE": S.SHAPE_HALFSPACE,
                          "CYLINDER": S.SHAPE_CYLINDER, "CONE": S.SHAPE_CONE, "CONVEX_POLYHEDRON": S.SHAPE_CONVEX_POLYHEDRON, "ROUND_CUBOID": S.SHAPE_ROUND_CUBOID,
                          "ROUND_CYLINDER": S.SHAPE_ROUND_CYLINDER, "ROUND_CONE": S.SHAPE_ROUND_CONE, "ROUND_CONVEX_POLYHEDRON": S.SHAPE_ROUND_CONVEX_POLYHEDRON,
                          "COMPOUND": S.SHAPE_COMPOUND, "TRIMESH": S.SHAPE_TRIMESH, "TRIANGLE": S.SHAPE_TRIANGLE}
    hb = enum_values(os.path.join(ROOT, "include", "rapier_hip.h"), "RP_BODY")
    ob = enum_values(os.path.join(ROOT, "oracle", "rapier_oracle.h"), "RO_BODY")
    assert hb == ob and hb["DYNAMIC"] == S.BODY_DYNAMIC and hb["FIXED"] == S.BODY_FIXED and hb["KINEMATIC_POSITION"] == S.BODY_KINEMATIC_POSITION


def test_no_cpu_fallback_without_device():
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import rapier_amd\n"
            "try:\n    rapier_amd.PhysicsWorld(); print('CREATED')\n"
            "except rapier_amd.RapierHipError as e:\n    print('REFUSED')\n") % ROOT
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert "REFUSED" in out.stdout, out.stdout + out.stderr


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "rapier_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "rapier_oracle" not in txt and "oracle_ffi" not in txt, f


def test_bin_pack_balances_islands():
    ranks = sharding.bin_pack([55] * 196, 8)
    counts = np.bincount(ranks, minlength=8)
    assert counts.max() - counts.min() <= 1


def test_partition_scene_keeps_islands_and_replicates_fixed():
    sc = S.many_pyramids(rows=2, cols=4)
    br = sharding.many_pyramids_body_ranks(2, 4, 10, 2)
    total_dyn = 0
    for r in range(2):
        sub, gids = sharding.partition_scene(sc, br, r)
        assert int(sub.bodies[0]["body_type"]) == S.BODY_FIXED and gids[0] == 0
        total_dyn += sub.num_dynamic
        assert sub.num_dynamic % 55 == 0
        for li, gi in enumerate(gids):
            assert np.array_equal(sub.bodies[li]["translation"], sc.bodies[gi]["translation"])
    assert total_dyn == sc.num_dynamic


def test_header_documents_scope_limits():
    hdr = open(os.path.join(ROOT, "include", "rapier_hip.h")).read()
    # what the device path refuses is stated where the entry points are declared
    assert "compound bodies" in hdr and "a stale or removed handle is refused" in hdr and "a world that holds a compound / mesh / height" in hdr


def test_column_shard_global_ids_partition_the_world():
    """bench.py --gpus N: the per-rank global body ids are disjoint, cover every dynamic body once, and agree with the generator
    (rank r's local scene = columns [14 r, 14 r + 14) of the 14 x 14N world)."""
    import numpy as np
    from rapier_amd import scenes as S, sharding
    for world in (2, 4, 8):
        ids = [sharding.column_shard_global_ids(14, 14, 10, world, r) for r in range(world)]
        cat = np.concatenate([g[1:] for g in ids])
        assert len(np.unique(cat)) == len(cat) == 14 * 14 * world * 55 and cat.min() == 1 and cat.max() == len(cat)
        assert all(g[0] == 0 for g in ids)
    full = S.many_pyramids(rows=2, cols=4)
    for r in range(2):
        local = S.many_pyramids(rows=2, cols=4, col_range=(2 * r, 2 * r + 2))
        gids = sharding.column_shard_global_ids(2, 2, 10, 2, r)
        assert len(gids) == len(local.bodies)
        for li, gi in enumerate(gids):
            np.testing.assert_array_equal(local.bodies[li]["translation"], full.bodies[gi]["translation"])


def test_header_is_plain_c_and_the_c_example_links(tmp_path):
    """include/rapier_hip.h is a C header (C99, -pedantic clean) and examples/pyramid.c — a plain-C caller of the ABI — compiles and
    links against librapier_hip.so (running it needs the GPU)."""
    import subprocess
    out = tmp_path / "pyramid"
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "pyramid.c"), "-L", os.path.join(ROOT, "rapier_amd"), "-lrapier_hip",
           "-Wl,-rpath," + os.path.join(ROOT, "rapier_amd"), "-o", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert out.exists()


def test_roofline_bookkeeping_of_bench():
    """bench.py's algorithmic bytes: S * [M * 2584 + N * 224] = 303.4 MB for b3d_many_pyramids"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    assert mod.algorithmic_bytes_per_step(28420, 10780) == 4 * (28420 * 2584 + 10780 * 224) == 303408000
    assert mod.HBM_PEAK_GBS == 8000.0


def test_world_create_validates_integration_parameters():
    """IntegrationParameters are checked before any device work (NonZeroUsize num_solver_iterations, finite dt >= 0, length_unit > 0,
    a known friction model — integration_parameters.rs:181-304): a bad set is RP_ERR_INVALID (-1) even on a box without a GPU,
    where a good one gets as far as RP_ERR_DEVICE (-2)."""
    L = _ffi.lib()
    g = np.array([0.0, -9.81, 0.0], np.float32)

    def create(**kw):
        p = S.default_params()
        for k, v in kw.items():
            p[k] = v
        out = C.c_void_p()
        rc = L.rp_world_create(p.ctypes.data, g.ctypes.data, 0, C.byref(out))
        if rc == 0:
            L.rp_world_destroy(out)
        return rc
    assert create() in (0, -2)
    for bad in (dict(num_solver_iterations=0), dict(num_solver_iterations=-3), dict(dt=np.nan), dict(dt=-1.0), dict(dt=np.inf),
                dict(length_unit=0.0), dict(length_unit=-1.0), dict(friction_model=7), dict(num_internal_pgs_iterations=-1),
                dict(contact_natural_frequency=np.nan), dict(normalized_prediction_distance=-0.1)):
        assert create(**bad) == -1, bad
    out = C.c_void_p()
    bad_g = np.array([0.0, np.nan, 0.0], np.float32)
    assert L.rp_world_create(S.default_params().ctypes.data, bad_g.ctypes.data, 0, C.byref(out)) == -1
