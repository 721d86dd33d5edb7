import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import rt_octree_amd as R
        return R.lib().rto_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not skip: the product has no CPU fallback.
    pass


@pytest.fixture(scope="session")
def small_tree_sh9():
    from rt_octree_amd import synth
    return synth.make_tree(depth_limit=6, basis_dim=9, seed=7)


@pytest.fixture(scope="session")
def small_tree_sh16():
    from rt_octree_amd import synth
    return synth.make_tree(depth_limit=6, basis_dim=16, seed=11)
