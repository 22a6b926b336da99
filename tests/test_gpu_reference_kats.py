"""The reference's outcome-level tests — restated on the oracle in tests/test_reference_kats.py and tests/test_oracle_kat.py with the
reference's scenes, step counts and thresholds — played on the DEVICE as well: every test function of those two modules runs once
more with `OracleWorld` replaced by a twin that drives the oracle and a device world side by side through the same calls and compares
the two bit for bit (poses, velocities, sleeping flags) at every read.  The test's own assertions then hold for the device because
they hold for the oracle and the two are equal.  Nothing here is a new scene: it is the device's share of ~100 reference scenes —
sleeping and waking, gyroscopic terms, joints with limits and motors, kinematic bodies, additional mass, restitution, speed caps,
body churn, persistent islands, solve groups — that the hand-written GPU tests reach only through the fuzz driver."""
import inspect

import numpy as np
import pytest

import oracle_ffi
from oracle_ffi import OracleWorld, lib
from rapier_amd import PhysicsWorld, scenes as S

import test_oracle_kat as KAT_A
import test_reference_kats as KAT_B



class _SecondOracle:
    """a second oracle behind the PhysicsWorld calls the twin makes: the CPU dry run of this harness (`-m "not gpu"`)"""

    def __init__(self, scene): self.o = OracleWorld(scene)
    def step(self, n=1): self.o.step(n)
    def read_bodies(self): return self.o.read()
    def sleeping(self): return self.o.sleeping()
    def counters(self): return {"overflow_flags": 0}
    def insert_collider(self, col, parent): return lib().ro_add_collider(self.o._w, np.array([col], S.COLLIDER_DTYPE).ctypes.data, int(parent))

    def insert_body(self, body):
        h = lib().ro_add_body(self.o._w, np.array([body], S.BODY_DTYPE).ctypes.data); self.o.n += 1
        return h
    def remove_body(self, hs): self.o.remove_body(hs[0])
    def remove_collider(self, hs): self.o.remove_collider(hs[0])
    def remove_impulse_joint(self, j): self.o.remove_joint(j)
    def set_joint_motor(self, j, axis, **kw): self.o.set_joint_motor(j, axis, **kw)
    def set_next_kinematic_position(self, hs, p): self.o.set_next_kinematic_position(hs[0], p)
    def set_additional_solver_iterations(self, hs, ns): self.o.set_additional_solver_iterations(hs[0], ns[0])
    def set_integration_parameters(self, p): self.o.set_params(p)
    def wake_up(self, hs, strong=True): self.o.wake_up(hs[0], strong)
    def apply_impulse(self, hs, impulse=None, torque_impulse=None): self.o.apply_impulse(hs[0], None if impulse is None else impulse[0], None if torque_impulse is None else torque_impulse[0])
    def add_force(self, hs, force=None, torque=None, reset=False): self.o.add_force(hs[0], None if force is None else force[0], None if torque is None else torque[0], reset)

    def write_bodies(self, hs, pos7=None, vel6=None):
        if pos7 is not None: self.o.set_pose(hs[0], pos7[0])
        if vel6 is not None: self.o.set_vel(hs[0], vel6[0][:3], vel6[0][3:])


DRY_RUN = False   # True: _SecondOracle stands in for the device (the CPU test of the harness below)


class DeviceTwin(OracleWorld):
    """OracleWorld with a device world in lockstep; reads compare the two"""

    def __init__(self, scene):
        super().__init__(scene)
        self.g = _SecondOracle(scene) if DRY_RUN else PhysicsWorld.from_scene(scene)
        self._removed = set()
        self.reads = 0

    # ---- stepping and reading ----
    def step(self, n=1):
        super().step(n)
        self.g.step(n)

    def _alive(self):
        return [b for b in range(self.n) if b not in self._removed]

    def read(self):
        p, v = super().read()
        gp, gv = self.g.read_bodies()
        rows = self._alive()
        np.testing.assert_array_equal(gp[rows], p[rows], err_msg="device poses differ from the oracle's")
        np.testing.assert_array_equal(gv[rows], v[rows], err_msg="device velocities differ from the oracle's")
        self.reads += 1
        return p, v

    def sleeping(self):
        s = super().sleeping()
        rows = self._alive()
        gs = self.g.sleeping()
        np.testing.assert_array_equal(np.asarray(gs, bool)[rows], s[rows], err_msg="device sleeping flags differ from the oracle's")
        return s

    # ---- user changes (the mappings of the randomised differential test, tests/test_gpu_fuzz.py) ----
    def add_body(self, **kw):
        b = super().add_body(**kw)
        hb = self.g.insert_body(S.body_desc(**kw))
        assert int(hb) & 0xFFFFFFFF == b
        return b

    def add_collider(self, parent, **kw):
        c = super().add_collider(parent, **kw)
        hc = self.g.insert_collider(S.collider_desc(**kw), parent)
        assert int(hc) & 0xFFFFFFFF == c
        return c

    def remove_body(self, body):
        super().remove_body(body); self.g.remove_body([int(body)]); self._removed.add(int(body))

    def remove_collider(self, collider):
        super().remove_collider(collider); self.g.remove_collider([int(collider)])

    def remove_joint(self, joint):
        super().remove_joint(joint); self.g.remove_impulse_joint(int(joint))

    def set_joint_motor(self, joint, axis, **motor):
        super().set_joint_motor(joint, axis, **motor); self.g.set_joint_motor(int(joint), int(axis), **motor)

    def set_pose(self, body, pos7):
        super().set_pose(body, pos7); self.g.write_bodies([int(body)], pos7=[np.asarray(pos7, np.float32)])

    def set_vel(self, body, linvel, angvel=(0, 0, 0)):
        super().set_vel(body, linvel, angvel)
        self.g.write_bodies([int(body)], vel6=[np.concatenate([np.asarray(linvel, np.float32), np.asarray(angvel, np.float32)])])

    def set_next_kinematic_position(self, body, pos7):
        super().set_next_kinematic_position(body, pos7); self.g.set_next_kinematic_position([int(body)], np.asarray(pos7, np.float32))

    def add_force(self, body, force=None, torque=None, reset=False):
        super().add_force(body, force=force, torque=torque, reset=reset)
        self.g.add_force([int(body)], force=None if force is None else [np.asarray(force, np.float32)],
                         torque=None if torque is None else [np.asarray(torque, np.float32)], reset=reset)

    def apply_impulse(self, body, impulse=None, torque_impulse=None):
        super().apply_impulse(body, impulse=impulse, torque_impulse=torque_impulse)
        self.g.apply_impulse([int(body)], impulse=None if impulse is None else [np.asarray(impulse, np.float32)],
                             torque_impulse=None if torque_impulse is None else [np.asarray(torque_impulse, np.float32)])

    def wake_up(self, body, strong=True):
        super().wake_up(body, strong); self.g.wake_up([int(body)], strong)

    def set_additional_solver_iterations(self, body, n):
        super().set_additional_solver_iterations(body, n); self.g.set_additional_solver_iterations([int(body)], [int(n)])

    def set_params(self, params):
        super().set_params(params); self.g.set_integration_parameters(params)


# tests that reach into the oracle through calls the twin cannot mirror on the device (raw ro_* calls on `o._w`, collider sensors flipped
# at run time, thread-count experiments): they stay oracle-only
ORACLE_ONLY = {"test_same_machine_runs_are_bitwise_identical"}


def _cases():
    out = []
    for mod in (KAT_A, KAT_B):
        for name, fn in sorted(vars(mod).items()):
            if not name.startswith("test_") or not inspect.isfunction(fn) or name in ORACLE_ONLY:
                continue
            src = inspect.getsource(fn)
            if "OracleWorld" not in src and not any(h in src for h in ("_settled_angle_deg", "_rebound", "_free_ball", "_pendulum", "_topple_world", "_spinning_box",
                                                                           "_offset_com_body", "_sw_world", "_islands_world", "_heavy_stack")):
                continue                                      # pure-math tests (no world)
            if "o._w" in src or "._w," in src or "lib()" in src or "set_sensor" in src:
                continue                                      # raw FFI calls / run-time sensor flips the twin does not mirror
            marks = [m for m in getattr(fn, "pytestmark", []) if m.name == "parametrize"]
            if not marks:
                out.append(pytest.param(mod, name, {}, id=f"{mod.__name__[5:]}::{name}"))
                continue
            # the cartesian product of the function's own parametrize marks
            combos = [{}]
            for m in marks:
                names = [n.strip() for n in m.args[0].split(",")] if isinstance(m.args[0], str) else list(m.args[0])
                nxt = []
                for c in combos:
                    for val in m.args[1]:
                        vals = val.values if hasattr(val, "values") else val
                        vals = vals if len(names) > 1 else (vals[0] if hasattr(val, "values") else vals,)
                        d = dict(c); d.update(dict(zip(names, vals))); nxt.append(d)
                combos = nxt
            for k, c in enumerate(combos):
                out.append(pytest.param(mod, name, c, id=f"{mod.__name__[5:]}::{name}[{k}]"))
    return out


def test_the_harness_itself_on_two_oracles(monkeypatch):
    """CPU: every collected case runs through the twin with a second oracle in the device's place — the harness (case collection,
    parametrize expansion, the mirrored calls) is covered without a GPU, and a case that cannot be mirrored shows up here"""
    import test_gpu_reference_kats as me
    monkeypatch.setattr(me, "DRY_RUN", True)
    cases = _cases()
    assert len(cases) >= 80, len(cases)
    for c in cases[::7]:                                      # a seventh of them: the whole set takes as long as the two KAT modules again
        _play(monkeypatch, *c.values)


@pytest.mark.gpu
@pytest.mark.parametrize("mod,name,kwargs", _cases())
def test_reference_kat_on_the_device(monkeypatch, mod, name, kwargs):
    _play(monkeypatch, mod, name, kwargs)


def _play(monkeypatch, mod, name, kwargs):
    made = []

    def twin(scene):
        t = DeviceTwin(scene)
        made.append(t)
        return t

    monkeypatch.setattr(mod, "OracleWorld", twin)
    fn = getattr(mod, name)
    params = inspect.signature(fn).parameters
    extra = {}
    if "monkeypatch" in params:
        extra["monkeypatch"] = monkeypatch
    try:
        fn(**kwargs, **extra)
    finally:
        oracle_ffi.set_threads(1)
    for t in made:                                            # every world the test built ends equal on both sides
        t.read()
        c = t.g.counters()
        assert c["overflow_flags"] == 0, c
