import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'needs_reference: needs /root/reference (build container only)')


def pytest_collection_modifyitems(config, items):
    from oracle.ref_shims import reference_available
    if reference_available():
        return
    skip = pytest.mark.skip(reason='reference tree not present on this machine')
    for item in items:
        if 'needs_reference' in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope='session')
def sim():
    """The product kernel sources compiled for the CPU simulator (tests/sim), behind the same bindings."""
    from tests.sim.build_sim import build
    from fiery_amd import native
    return native.Lib(build())


@pytest.fixture(scope='session')
def hip():
    """The real library on the GPU box."""
    import torch
    from fiery_amd import native
    assert torch.cuda.is_available(), 'gpu tests need a GPU'
    return native.get()
