"""TEST INFRASTRUCTURE: compile the product kernel sources for the CPU simulator (see include/hip/hip_runtime.h)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'fiery_amd', 'csrc')
OUT_DIR = os.path.join(HERE, '_build')
OUT = os.path.join(OUT_DIR, 'libfiery_sim.so')
sys.path.insert(0, ROOT)
from fiery_amd.build import SOURCES  # noqa: E402  (the product's own source list)


def build():
    os.makedirs(OUT_DIR, exist_ok=True)
    deps = [os.path.join(CSRC, n) for n, _ in SOURCES] + [
        os.path.join(CSRC, 'common.h'), os.path.join(CSRC, 'conv_igemm_kernel.h'), os.path.join(ROOT, 'include', 'fiery_hip.h'),
        os.path.join(HERE, 'include', 'fiery_gfx950.h'),
        os.path.join(HERE, 'include', 'hip', 'hip_runtime.h')]
    if os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    objs, jobs = [], []
    for name, extra in SOURCES:
        obj = os.path.join(OUT_DIR, name + '.o')
        cmd = ['g++', '-std=c++20', '-O2', '-fPIC', '-pthread', '-Wno-attributes', '-Wno-unknown-pragmas', '-Wno-psabi',
               '-I' + os.path.join(HERE, 'include'), '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC,
               '-DFIERY_CONV_TUNING=0', '-x', 'c++', '-c', os.path.join(CSRC, name), '-o', obj] + extra
        jobs.append((name, subprocess.Popen(cmd)))
        objs.append(obj)
    failed = [name for name, proc in jobs if proc.wait() != 0]
    if failed:
        raise RuntimeError('g++ failed for ' + ', '.join(failed))
    subprocess.check_call(['g++', '-shared', '-pthread', '-o', OUT] + objs)
    return OUT


if __name__ == '__main__':
    print(build())
