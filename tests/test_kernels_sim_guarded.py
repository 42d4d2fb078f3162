"""No kernel reads or writes outside its operands: the CPU-simulated kernels on tensors that end (and begin) exactly at an
inaccessible page (tests/sim/guard_check.py).  Each case runs in a process of its own - an out-of-range access is a fault."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('where', ['end', 'begin'])
@pytest.mark.parametrize('case', ['conv', 'bn', 'gru', 'resample', 'inputs', 'pool', 'pool_compact_off', 'train_images'])
def test_kernels_stay_inside_their_operands(sim, case, where):
    if case == 'train_images' and where == 'begin':
        pytest.skip('the whole training step is run once, against the page after each tensor')
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'sim', 'guard_check.py'), case, where], cwd=ROOT, capture_output=True,
                         text=True, timeout=900)
    assert res.returncode == 0 and f'ok {case} {where} (' in res.stdout, res.stdout[-1500:] + res.stderr[-3000:]
