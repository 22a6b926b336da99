"""The oracle's arenas against a model of data/arena.rs (Arena::insert :260-290, Arena::remove :353-380, Arena::reserve :646-663): removed
slots are handed out again LIFO before any fresh index, the generation an Index carries is the arena's removal count at insertion time.
(CPU only; the device twin is tests/test_gpu_arena.py.)"""
import numpy as np

from rapier_amd import scenes as S
from oracle_ffi import OracleWorld, lib


class ArenaModel:
    """free list + generation counter of the reference's Arena, indices only"""

    def __init__(self):
        self.free, self.n, self.generation, self.gen_of = [], 0, 0, {}

    def insert(self):
        i = self.free.pop() if self.free else self.n
        if i == self.n:
            self.n += 1
        self.gen_of[i] = self.generation
        return i, self.generation

    def remove(self, i):
        self.generation += 1
        self.free.append(i)


def test_oracle_arenas_follow_the_reference_model():
    sc = S.Scene(name="arena")
    o = OracleWorld(sc)
    bodies, cols = ArenaModel(), ArenaModel()
    rng = np.random.default_rng(617)
    alive = {}                                     # body -> its colliders in attachment order
    for tick in range(300):
        if alive and rng.random() < 0.45:
            b = int(rng.choice(sorted(alive)))
            if rng.random() < 0.3 and len(alive[b]) > 1:      # ColliderSet::remove of one collider
                c = alive[b].pop(int(rng.integers(len(alive[b]))))
                o.remove_collider(c); cols.remove(c)
            else:                                             # RigidBodySet::remove: the attached colliders go in attachment order
                o.remove_body(b)
                for c in alive.pop(b):
                    cols.remove(c)
                bodies.remove(b)
        else:
            body = S.body_desc(translation=(float(10 * tick), 5.0, 0.0))
            ob = lib().ro_add_body(o._w, np.array([body], S.BODY_DTYPE).ctypes.data)
            assert (ob, o.body_generation(ob)) == bodies.insert(), tick
            alive[ob] = []
            for k in range(int(rng.integers(1, 4))):
                col = S.collider_desc(half_extents=(0.3, 0.3, 0.3), translation=(0.4 * k, 0.0, 0.0))
                oc = lib().ro_add_collider(o._w, np.array([col], S.COLLIDER_DTYPE).ctypes.data, ob)
                assert (oc, o.collider_generation(oc)) == cols.insert(), (tick, k)
                alive[ob].append(oc)
        if tick % 3 == 0:
            o.step(1)
    assert o.n == bodies.n and o.n < 120           # rows plateau: slots are reused, not appended


def test_a_reused_slot_behaves_like_a_fresh_one():
    """the same body inserted into a reused slot and into a fresh world row: same response to the same impulses"""
    def world():
        sc = S.Scene(name="w"); sc.gravity = (0.0, 0.0, 0.0)
        return OracleWorld(sc)

    def add(o, pos, he, dens, rot=(0, 0, 0, 1)):
        b = S.body_desc(translation=pos, rotation=rot, additional_mass=0.25)
        c = S.collider_desc(half_extents=he, density=dens, translation=(0.1, 0.2, 0.0))
        ob = lib().ro_add_body(o._w, np.array([b], S.BODY_DTYPE).ctypes.data)
        lib().ro_add_collider(o._w, np.array([c], S.COLLIDER_DTYPE).ctypes.data, ob)
        return ob

    a, b = world(), world()
    va = add(a, (0, 5, 0), (0.5, 0.5, 0.5), 1.0); add(a, (10, 5, 0), (0.3, 0.3, 0.3), 2.0)
    add(b, (0, 5, 0), (0.5, 0.5, 0.5), 1.0); add(b, (10, 5, 0), (0.3, 0.3, 0.3), 2.0)
    a.step(1); a.remove_body(va); a.step(1); b.step(2)
    ra = add(a, (-10, 5, 0), (0.2, 0.7, 0.4), 3.0, rot=(0.1, 0.2, 0.3, 0.9))
    rb = add(b, (-10, 5, 0), (0.2, 0.7, 0.4), 3.0, rot=(0.1, 0.2, 0.3, 0.9))
    assert ra == va and rb == 2 and a.body_generation(ra) == 1 and b.body_generation(rb) == 0
    for o, h in ((a, ra), (b, rb)):
        o.apply_impulse(h, impulse=(1, 2, 3), torque_impulse=(0.1, 0.2, 0.3)); o.step(3)
    (pa, va_), (pb, vb_) = a.read(), b.read()
    np.testing.assert_array_equal(pa[ra], pb[rb]); np.testing.assert_array_equal(va_[ra], vb_[rb])
