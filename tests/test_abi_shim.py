"""The drop-in boundary is spelt three times — the C header (include/rapier_hip.h), the Rust shim a rapier maintainer would add
(INTEGRATION.md's ```rust block) and the numpy descriptor dtypes the tests and bench.py drive the library with.  These tests parse all
three and require them to agree: every #[repr(C)] struct's field list (name, scalar type, element count), its C offsets and size, and
every export's name, return type and argument types (the r5 shim had lost rp_body_desc.ccd_enabled and bound 23 of
49 exports).  Reference seam: PhysicsWorld (src/pipeline/physics_world.rs:61-157), IntegrationParameters
(src/dynamics/integration_parameters.rs:181-304), ImpulseJointSet::insert (impulse_joint_set.rs:329-375)."""
import ctypes as C
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import abi_parse  # noqa: E402

from rapier_amd import _ffi, scenes as S  # noqa: E402


def test_rust_shim_structs_and_exports_equal_the_header():
    diffs = abi_parse.compare()
    assert not diffs, "\n".join(diffs)


def test_every_header_struct_and_export_is_bound_by_the_shim():
    hs, hf = abi_parse.parse_header()
    rs, rf = abi_parse.parse_rust(abi_parse.rust_block())
    assert len(hs) == 9 and set(abi_parse.RUST_NAME[c] for c in hs) <= set(rs)   # params, body, collider, 2 events, motor, joint, counters, comm id
    assert set(hf) == set(rf) == set(_ffi.SYMBOLS) and len(hf) >= 50
    # the fields whose loss VERDICT r5 reported, by name
    assert ("ccd_enabled", "i32", 1) in rs["BodyDesc"]
    assert ("min_ccd_dt", "f32", 1) in rs["IntegrationParameters"] and ("contact_clustering", "i32", 1) in rs["IntegrationParameters"]
    assert rs["GenericJoint"][0] == ("body1", "u64", 1) and rs["GenericJoint"][1] == ("body2", "u64", 1)


def _np_fields(dt):
    out = []
    for name in dt.names:
        sub, off = dt.fields[name][0], dt.fields[name][1]
        base = sub.base
        count = int(np.prod(sub.shape)) if sub.shape else 1
        if base.names:   # a nested struct (rp_joint_motor)
            kind = "struct"
        else:
            kind = {("f", 4): "f32", ("i", 4): "i32", ("u", 4): "u32", ("u", 8): "u64"}[(base.kind, base.itemsize)]
        out.append((name, kind, count, off, sub.itemsize))
    return out


def test_numpy_descriptors_have_the_headers_layout():
    hs, _ = abi_parse.parse_header()
    for cname, dt in (("rp_integration_params", S.PARAMS_DTYPE), ("rp_body_desc", S.BODY_DTYPE), ("rp_collider_desc", S.COLLIDER_DTYPE),
                      ("rp_joint_motor", S.MOTOR_DTYPE), ("rp_joint_desc", S.JOINT_DTYPE)):
        fields, size, _ = abi_parse.struct_layout(hs, cname)
        got = _np_fields(dt)
        want = [(f, "struct" if t in abi_parse.RUST_NAME.values() else t, c, off, sz) for f, t, c, off, sz in fields]
        assert got == want, (cname, got, want)
        assert dt.itemsize == size, (cname, dt.itemsize, size)
    fields, size, _ = abi_parse.struct_layout(hs, "rp_counters")
    assert [f for f, *_ in fields] == [f for f, _ in _ffi.Counters._fields_] and C.sizeof(_ffi.Counters) == size
    # no implicit padding anywhere (a #[repr(C)] struct and a packed numpy dtype then agree without alignment rules)
    for cname in hs:
        fields, size, _ = abi_parse.struct_layout(hs, cname)
        assert sum(sz for *_, sz in fields) == size, cname


def test_oracle_descriptors_share_the_layout():
    """the oracle is driven with the same numpy descriptors: its header must spell the same structs"""
    txt = open(os.path.join(ROOT, "oracle", "rapier_oracle.h")).read()
    txt = re.sub(r"\bro_", "rp_", txt).replace("rp_params", "rp_integration_params")
    tmp = os.path.join(ROOT, "tests", "_build"); os.makedirs(tmp, exist_ok=True)
    path = os.path.join(tmp, "oracle_as_rp.h"); open(path, "w").write(txt)
    hs, _ = abi_parse.parse_header()
    os_, _ = abi_parse.parse_header(path)
    for cname in ("rp_integration_params", "rp_collider_desc", "rp_joint_motor", "rp_joint_desc"):
        assert [(f, t, c) for f, t, c, _ in os_[cname]] == [(f, t, c) for f, t, c, _ in hs[cname]], cname
    # ro_body_desc is a PREFIX of rp_body_desc (ro_add_body reads that far; the two trailing fields go through the oracle's setters)
    ob, hb = [(f, t, c) for f, t, c, _ in os_["rp_body_desc"]], [(f, t, c) for f, t, c, _ in hs["rp_body_desc"]]
    assert ob == hb[:len(ob)] and [f for f, *_ in hb[len(ob):]] == ["additional_solver_iterations", "ccd_enabled"]


