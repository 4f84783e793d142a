"""Builds liblvae_hip.so (HIP kernels for gfx950 + the C++ host coder) in-tree with hipcc.

hipcc cross-compiles gfx950 without a GPU; the .so is git-ignored but travels to the GPU box with the
gpurun snapshot.  Usage: python lossy-vae_amd/build_native.py [--force]
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT_DIR = os.path.join(HERE, 'lvae', '_native')
OUT = os.path.join(OUT_DIR, 'liblvae_hip.so')
SOURCES = ['gemm_f32.hip', 'pointwise.hip', 'rans_host.cpp']
HEADER = os.path.join(os.path.dirname(HERE), 'include', 'lvae_hip.h')


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [HEADER, os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OUT_DIR, exist_ok=True)
    if not (force or _stale()):
        return OUT
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OUT_DIR, s + '.o')
        cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c', src, '-o', obj]
        if s.endswith('.cpp'):
            cmd.insert(1, '-x'); cmd.insert(2, 'hip')      # host-only TU still goes through hipcc for one toolchain
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs + ['-lpthread']
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
