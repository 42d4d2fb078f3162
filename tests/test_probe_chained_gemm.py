"""The layout algebra of DESIGN.md section 11 item 1 / docs/DESIGN_HISTORY.md section 11 (the Bottleneck tails without a transpose), checked lane by lane on the
CPU simulator: with the first product's MFMA operands swapped, its accumulator block is the second product's A operand,
provided W2's rows are packed in the accumulator's register order.  `tools/probe/chained_gemm_swapped.hip` is a probe, not
product code - the product kernel still stages h through LDS; this test pins the claim the next kernel will be built on."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'tools', 'probe', 'chained_gemm_swapped.hip')
OUT_DIR = os.path.join(ROOT, 'tests', 'sim', '_build')
OUT = os.path.join(OUT_DIR, 'libprobe_chained_gemm.so')


def _build():
    os.makedirs(OUT_DIR, exist_ok=True)
    hdr = os.path.join(ROOT, 'tests', 'sim', 'include', 'hip', 'hip_runtime.h')
    if not os.path.exists(OUT) or max(os.path.getmtime(SRC), os.path.getmtime(hdr)) > os.path.getmtime(OUT):
        subprocess.check_call(['g++', '-std=c++20', '-O2', '-fPIC', '-pthread', '-shared', '-Wno-attributes', '-Wno-unknown-pragmas',
                               '-Wno-psabi', '-I' + os.path.join(ROOT, 'tests', 'sim', 'include'), '-x', 'c++', SRC, '-o', OUT])
    return C.CDLL(OUT)


def _cout_of(r, hi):
    return 8 * (r // 4) + 4 * hi + r % 4


def test_swapped_first_product_feeds_the_second_from_registers():
    dll = _build()
    rng = np.random.default_rng(7)
    n_pixels, K1 = 96, 64
    A1 = rng.standard_normal((n_pixels, K1)).astype(np.float32)
    W1 = (rng.standard_normal((K1, 32)) / np.sqrt(K1)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, 32).astype(np.float32)
    shift = rng.standard_normal(32).astype(np.float32) * 0.1
    W2 = (rng.standard_normal((32, 64)) / np.sqrt(32)).astype(np.float32)
    # the host-side packing the design relies on: row (r, hi) of the packed image is W2's row c(r, hi)
    W2p = np.stack([W2[_cout_of(r, hi)] for r in range(16) for hi in range(2)]).astype(np.float32)
    assert sorted(_cout_of(r, hi) for r in range(16) for hi in range(2)) == list(range(32))       # every cout exactly once
    O = np.full((n_pixels, 64), np.nan, dtype=np.float32)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = dll.probe_chained_swapped(ptr(A1), ptr(W1), ptr(scale), ptr(shift), ptr(W2p), ptr(O), n_pixels, K1)
    assert rc == 0
    h = np.maximum((A1.astype(np.float64) @ W1.astype(np.float64)) * scale + shift, 0.0)
    want = h @ W2.astype(np.float64)
    assert np.isfinite(O).all()
    assert np.abs(O - want).max() <= 2e-5 * max(1.0, np.abs(want).max())
    # and it is NOT right with W2 in its natural row order - the packing is what makes the register order a valid k order
    O2 = np.empty_like(O)
    assert dll.probe_chained_swapped(ptr(A1), ptr(W1), ptr(scale), ptr(shift), ptr(np.ascontiguousarray(W2)), ptr(O2), n_pixels, K1) == 0
    assert np.abs(O2 - want).max() > 1e-2
