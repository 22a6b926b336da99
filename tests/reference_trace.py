"""Per-step trace of a scene, from the oracle, in the format `bench/rapier_ref --trace` writes from the real crate — and the differ.

A bitwise divergence between the reference build and the oracle cannot be bisected in this image (no cargo).  With these two files
it takes one run on any machine that has cargo:

    cd bench/rapier_ref && cargo run --release -- reference_pile --trace /tmp/ref.rptrace 120
    python tests/reference_trace.py diff /tmp/ref.rptrace tests/golden/reference_pile_s120.rptrace

The differ names the first step whose state hash, sleep set or touching set differs and lists the transitions only one side made
(the step an island fell asleep, the step a pair gained its first solver contact).

Format (text): `RPTRACE1 <scene> <bodies> <steps>`, then per step
    step <k> hash <fnv1a of all body states, simd_backend_determinism.rs:36-57> asleep <n> touching <n> contacts <n>
    S <body> | W <body>          fell asleep / woke up in step k (arena index)
    B <c1> <c2> | E <c1> <c2>    the collider pair gained its first / lost its last solver contact in step k

    python tests/reference_trace.py write <scene> <steps> <file>      (scenes: test_reference_dump.SCENES)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def trace_lines(world, scene_name, steps):
    """`world`: an OracleWorld (or anything with step / read / sleeping / manifolds of the same meaning)"""
    from test_reference_golden import fnv1a_state_hash
    pos, _ = world.read()
    lines = [f"RPTRACE1 {scene_name} {pos.shape[0]} {steps}"]
    asleep, touching = set(), set()
    for k in range(1, steps + 1):
        world.step(1)
        pos, vel = world.read()
        now_asleep = set(int(b) for b in np.nonzero(world.sleeping())[0])
        meta, _, _ = world.manifolds()
        now_touching = set((int(min(a, b)), int(max(a, b))) for a, b in meta[:, :2])
        contacts = int(meta[:, 3].sum())
        lines.append(f"step {k} hash {fnv1a_state_hash(pos, vel):016x} asleep {len(now_asleep)} touching {len(now_touching)} contacts {contacts}")
        lines += [f"S {b}" for b in sorted(now_asleep - asleep)] + [f"W {b}" for b in sorted(asleep - now_asleep)]
        lines += [f"B {a} {b}" for a, b in sorted(now_touching - touching)] + [f"E {a} {b}" for a, b in sorted(touching - now_touching)]
        asleep, touching = now_asleep, now_touching
    return lines


def parse(text):
    """-> (header dict, [per-step dict: hash, asleep, touching, contacts, events (set of line strings)])"""
    lines = [ln.strip() for ln in text.splitlines() if ln.strip()]
    tag, scene, bodies, steps = lines[0].split()
    assert tag == "RPTRACE1", "not an rptrace file"
    out = []
    for ln in lines[1:]:
        f = ln.split()
        if f[0] == "step":
            assert int(f[1]) == len(out) + 1, f"steps out of order at '{ln}'"
            out.append({"hash": f[3], "asleep": int(f[5]), "touching": int(f[7]), "contacts": int(f[9]), "events": set()})
        else:
            assert f[0] in ("S", "W", "B", "E") and out, f"bad line '{ln}'"
            out[-1]["events"].add(ln)
    assert len(out) == int(steps), "truncated trace"
    return {"scene": scene, "bodies": int(bodies), "steps": int(steps)}, out


def diff(text_a, text_b, name_a="a", name_b="b"):
    """None when the traces agree; otherwise a report of the first differing step (structural differences first: they are causes,
    a differing hash alone is a rounding)"""
    ha, a = parse(text_a)
    hb, b = parse(text_b)
    if (ha["scene"], ha["bodies"]) != (hb["scene"], hb["bodies"]):
        return f"different worlds: {ha} vs {hb}"
    first_hash = None
    for k, (x, y) in enumerate(zip(a, b), start=1):
        if x["events"] != y["events"] or (x["asleep"], x["touching"], x["contacts"]) != (y["asleep"], y["touching"], y["contacts"]):
            rep = [f"step {k}: first structural difference" + (f" (state hashes differ since step {first_hash})" if first_hash else " (state hashes equal up to the step before)")]
            rep.append(f"  {name_a}: asleep {x['asleep']} touching {x['touching']} contacts {x['contacts']}")
            rep.append(f"  {name_b}: asleep {y['asleep']} touching {y['touching']} contacts {y['contacts']}")
            rep += [f"  only {name_a}: {e}" for e in sorted(x["events"] - y["events"])[:40]]
            rep += [f"  only {name_b}: {e}" for e in sorted(y["events"] - x["events"])[:40]]
            return "\n".join(rep)
        if first_hash is None and x["hash"] != y["hash"]:
            first_hash = k
    if first_hash:
        return f"step {first_hash}: state hashes differ ({a[first_hash - 1]['hash']} vs {b[first_hash - 1]['hash']}) with identical sleep / touching sets through step {min(len(a), len(b))}: a rounding, not a decision"
    if len(a) != len(b):
        return f"traces agree over the common {min(len(a), len(b))} steps (lengths {len(a)} / {len(b)})"
    return None


def main(argv):
    if len(argv) >= 4 and argv[0] == "write":
        from oracle_ffi import OracleWorld
        from test_reference_dump import SCENES
        w = OracleWorld(SCENES[argv[1]]())
        open(argv[3], "w").write("\n".join(trace_lines(w, argv[1], int(argv[2]))) + "\n")
        return 0
    if len(argv) == 3 and argv[0] == "diff":
        rep = diff(open(argv[1]).read(), open(argv[2]).read(), os.path.basename(argv[1]), os.path.basename(argv[2]))
        print(rep or "traces identical")
        return 1 if rep else 0
    print(__doc__)
    return 2


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
