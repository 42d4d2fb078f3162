"""Builds libfiery_hip.so (gfx950) in-tree with hipcc.  `python -m fiery_amd.build [--force]`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
INCLUDE = os.path.join(os.path.dirname(HERE), 'include')
OUT = os.path.join(HERE, 'libfiery_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')

# (source, extra flags).  lift_splat.hip: the index path must round like ATen's CPU kernels -> no FMA contraction.
SOURCES = [
    ('runtime.cpp', []),
    ('lift_splat.hip', ['-ffp-contract=off']),
    ('warp.hip', ['-ffp-contract=off']),                # sampling positions round like ATen's CPU kernels (see the file)
    ('conv_igemm.hip', []),
    # one translation unit per tile shape of the convolution kernel: they compile side by side
    ('conv_tile_128x32.hip', []),
    ('conv_tile_128x64.hip', []),
    ('conv_tile_128x128.hip', []),
    ('conv_tile_128x128_rest.hip', []),
    ('conv_tile_64x64.hip', []),
    ('conv_tile_64x128.hip', []),
    ('conv_tile_bf16_a.hip', []),
    ('conv_tile_bf16_b.hip', []),
    ('conv_tile_halo_f32.hip', []),
    ('conv_tile_stream_k.hip', []),
    ('conv_tile_split.hip', []),
    ('conv_winograd.hip', []),
    ('conv_winograd_split.hip', []),
    ('aux_ops.hip', []),
    ('conv_grad.hip', []),
    ('bn_train.hip', []),
    ('gru_train.hip', []),
    ('labels.hip', []),
    ('images.hip', []),
]
# FIERY_CONV_TUNING=1: also build the convolution's clock-probe / priority variants (tools/microbench.py conv --clk)
TUNING = ['-DFIERY_CONV_TUNING=1'] if os.environ.get('FIERY_CONV_TUNING') == '1' else ['-DFIERY_CONV_TUNING=0']
COMMON = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I' + INCLUDE, '-I' + CSRC, '-x', 'hip']


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    # FIERY_BUILD_VARIANT=name (+ FIERY_BUILD_FLAGS="-D..."): a second library for A/B runs on one GPU box,
    # tools/ab/libfiery_hip_<name>.so, selected with FIERY_HIP_LIB=...; the in-tree library is left alone
    variant = os.environ.get('FIERY_BUILD_VARIANT')
    global OUT
    objdir = os.path.join(HERE, 'build' if not variant else 'build_' + variant)
    if variant:
        os.makedirs(os.path.join(os.path.dirname(HERE), 'tools', 'ab'), exist_ok=True)
        OUT = os.path.join(os.path.dirname(HERE), 'tools', 'ab', f'libfiery_hip_{variant}.so')
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, 'common.h'), os.path.join(CSRC, 'conv_igemm_kernel.h'),
               os.path.join(CSRC, 'fiery_gfx950.h'), os.path.join(INCLUDE, 'fiery_hip.h')]
    objs, jobs = [], []
    for name, extra in SOURCES:
        src = os.path.join(CSRC, name)
        obj = os.path.join(objdir, name + '.o')
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            cmd = [HIPCC] + COMMON + TUNING + os.environ.get('FIERY_BUILD_FLAGS', '').split() + extra + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            jobs.append((name, subprocess.Popen(cmd)))       # translation units compile side by side
    failed = [name for name, proc in jobs if proc.wait() != 0]
    if failed:
        raise RuntimeError('hipcc failed for ' + ', '.join(failed))
    if force or _stale(OUT, objs):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    build(force='--force' in sys.argv)
