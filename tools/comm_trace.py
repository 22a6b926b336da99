"""The readback collective under rocprofv3 (kernel + memory-copy trace): one rank, its own RCCL communicator.
    rocprofv3 --kernel-trace --memory-copy-trace -d DIR -o ct -- python tools/comm_trace.py run
    python tools/comm_trace.py show DIR        # the device operations of the gather in start order
Expected order: k_fill_ids, k_pack_bodies, the RCCL all-gather kernel, then the device-to-host copies — no copy to the host before the
collective."""
import glob
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import numpy as np
    from rapier_amd import PhysicsWorld, ShardComm, scenes as S, sharding
    full = S.many_pyramids(4, 4)
    body_rank = sharding.many_pyramids_body_ranks(4, 4, 10, 2)
    sub, gids = sharding.partition_scene(full, body_rank, 0)
    w = PhysicsWorld.from_scene(sub)
    w.step(20); w.sync()
    pos, vel = w.read_bodies()
    dyn = np.array([int(b["body_type"]) == S.BODY_DYNAMIC for b in sub.bodies])
    comm = ShardComm(ShardComm.unique_id(), 1, 0, 0)
    w.step(3)                                   # pending steps: the gather settles them first
    gp, gv, per = sharding.all_gather_bodies_native(w, comm, pos, vel, gids, len(full.bodies), dyn, int(dyn.sum()))
    p2, v2 = w.read_bodies()
    assert np.array_equal(gp[gids], p2) and np.array_equal(gv[gids], v2) and per.tolist() == [int(dyn.sum())]
    print("gathered", per.tolist(), "rows through RCCL; equal to rp_bodies_read")
    comm.close()


def show(d):
    db = sqlite3.connect(sorted(glob.glob(os.path.join(d, "**", "*.db"), recursive=True))[0])
    cur = db.cursor()
    tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    ev = [(s, e, "kernel", n) for n, s, e in cur.execute("select name, start, end from kernels")]
    mc = next((t for t in ("memory_copies", "memory_copy") if t in tables), None)
    if mc:
        cols = [r[1] for r in cur.execute(f"pragma table_info({mc})")]
        name = "name" if "name" in cols else cols[0]
        size = "size" if "size" in cols else None
        q = f"select {name}, start, end{', ' + size if size else ''} from {mc}"
        for r in cur.execute(q):
            ev.append((r[1], r[2], "copy", f"{r[0]}" + (f" {r[3]} B" if size else "")))
    ev.sort()
    i = max(k for k, e in enumerate(ev) if "k_pack_bodies" in e[3])
    t0 = ev[i][0]
    for s, e, kind, n in ev[max(0, i - 3): i + 8]:
        print(f"{(s - t0) / 1e3:10.2f} us  +{(e - s) / 1e3:8.2f} us  {kind:6s} {n[:110]}")


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else show(sys.argv[2])
