import os
import sys

import pytest

# The lockstep tests drive the device world and the oracle with the SAME plain arena indices (the oracle has no generations): for them
# a value below 2^32 names a slot's current occupant (RigidBodySet::get_unknown_gen).  The product default is strict handles; the
# stale-handle tests in test_gpu_arena.py create their worlds with index_addressing=False.
os.environ.setdefault("RP_INDEX_ADDRESSING", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle_ffi
    return oracle_ffi.lib()
