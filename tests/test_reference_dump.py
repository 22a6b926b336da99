"""Oracle vs body states dumped by the REAL reference (bench/rapier_ref `--dump`).

The dumps cannot be produced in this image (no cargo); the loader, the file format and the comparison are exercised on a
synthetic file so the path is known to work the day fixtures are dropped into tests/golden/reference/."""
import glob
import os
import struct

import numpy as np
import pytest

from rapier_amd import scenes as S
from oracle_ffi import OracleWorld
from test_reference_golden import fnv1a_state_hash

REF_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference")
SCENES = {
    "pyramid10": lambda: S.pyramid10(),
    "many_pyramids": lambda: S.many_pyramids(),
    "many_pyramids_c4": lambda: S.many_pyramids(54, 54),
    "large_pyramid": lambda: S.large_pyramid(),
    "joint_grid": lambda: S.joint_grid(),
    "reference_pile": lambda: S.reference_pile(12, 3, 12, chain=True),
}


def load_rpdump(path):
    raw = open(path, "rb").read()
    assert raw[:8] == b"RPDUMP1\0", "not an rpdump file"
    n, steps = struct.unpack_from("<II", raw, 8)
    body = np.frombuffer(raw, "<f4", count=13 * n, offset=16).reshape(n, 13).copy()
    (stored_hash,) = struct.unpack_from("<Q", raw, 16 + 52 * n)
    assert fnv1a_state_hash(body[:, :7], body[:, 7:]) == stored_hash, "rpdump payload does not match its own state hash"
    return steps, body[:, :7], body[:, 7:], stored_hash


def write_rpdump(path, steps, pos7, vel6):
    body = np.concatenate([pos7, vel6], axis=1).astype("<f4")
    with open(path, "wb") as f:
        f.write(b"RPDUMP1\0" + struct.pack("<II", body.shape[0], steps) + body.tobytes() + struct.pack("<Q", fnv1a_state_hash(pos7, vel6)))


def compare(pos, vel, rpos, rvel, steps):
    """BASELINE.json: positions within 1e-4 relative (to the scene extent); quaternions sign-aligned."""
    scale = max(1.0, float(np.abs(rpos[:, :3]).max()))
    assert np.abs(pos[:, :3] - rpos[:, :3]).max() <= 1.0e-4 * scale, f"positions after {steps} steps"
    sign = np.sign(np.sum(pos[:, 3:] * rpos[:, 3:], axis=1, keepdims=True))
    assert np.abs(pos[:, 3:] * sign - rpos[:, 3:]).max() <= 1.0e-3, f"rotations after {steps} steps"
    vscale = max(1.0, float(np.abs(rvel).max()))
    assert np.abs(vel - rvel).max() <= 1.0e-2 * vscale, f"velocities after {steps} steps"


def test_rpdump_round_trip(tmp_path):
    w = OracleWorld(S.pyramid10())
    w.step(3)
    pos, vel = w.read()
    p = tmp_path / "pyramid10_s3.rpdump"
    write_rpdump(p, 3, pos, vel)
    steps, rpos, rvel, h = load_rpdump(p)
    assert steps == 3 and np.array_equal(rpos, pos) and np.array_equal(rvel, vel)
    compare(pos, vel, rpos, rvel, steps)
    raw = bytearray(open(p, "rb").read()); raw[40] ^= 1
    bad = tmp_path / "bad.rpdump"; open(bad, "wb").write(bytes(raw))
    with pytest.raises(AssertionError):
        load_rpdump(bad)


_FILES = sorted(glob.glob(os.path.join(REF_DIR, "*.rpdump")))


@pytest.mark.skipif(not _FILES, reason="no reference dumps in tests/golden/reference (needs cargo: bench/rapier_ref --dump)")
@pytest.mark.parametrize("path", _FILES or ["-"])
def test_oracle_matches_reference_dump(path):
    name = os.path.basename(path)
    scene = name[: name.rindex("_s")]
    steps, rpos, rvel, _ = load_rpdump(path)
    w = OracleWorld(SCENES[scene]())
    w.step(steps)
    pos, vel = w.read()
    assert pos.shape == rpos.shape
    compare(pos, vel, rpos, rvel, steps)
