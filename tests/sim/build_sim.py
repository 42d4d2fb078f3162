"""TEST INFRASTRUCTURE: compile the product kernel sources for the CPU simulator (see include/hip/hip_runtime.h)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'fiery_amd', 'csrc')
OUT_DIR = os.path.join(HERE, '_build')
OUT = os.path.join(OUT_DIR, 'libfiery_sim.so')
SOURCES = [('runtime.cpp', []), ('lift_splat.hip', ['-ffp-contract=off']), ('warp.hip', []),
           ('conv_igemm.hip', []), ('aux_ops.hip', [])]


def build():
    os.makedirs(OUT_DIR, exist_ok=True)
    deps = [os.path.join(CSRC, n) for n, _ in SOURCES] + [
        os.path.join(CSRC, 'common.h'), os.path.join(ROOT, 'include', 'fiery_hip.h'),
        os.path.join(HERE, 'include', 'hip', 'hip_runtime.h')]
    if os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in deps):
        return OUT
    objs = []
    for name, extra in SOURCES:
        obj = os.path.join(OUT_DIR, name + '.o')
        cmd = ['g++', '-std=c++20', '-O2', '-fPIC', '-pthread', '-Wno-attributes', '-Wno-unknown-pragmas', '-Wno-psabi',
               '-I' + os.path.join(HERE, 'include'), '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC,
               '-x', 'c++', '-c', os.path.join(CSRC, name), '-o', obj] + extra
        subprocess.check_call(cmd)
        objs.append(obj)
    subprocess.check_call(['g++', '-shared', '-pthread', '-o', OUT] + objs)
    return OUT


if __name__ == '__main__':
    print(build())
