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


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """The achieved max-abs errors of every float comparison of the GPU tier (tests/parity_report.py): printed into
    the log and written where gpurun merges files back from."""
    from tests import parity_report
    text = parity_report.table()
    if text:
        terminalreporter.write_sep('=', 'parity: achieved errors (literal tolerance 1e-4, asserted bound 1e-4 * max(1, |ref|_inf))')
        terminalreporter.write_line(text)
        parity_report.dump(os.path.join(ROOT, 'gpurun_out', 'parity_errors.json'))
