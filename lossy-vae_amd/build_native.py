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
SOURCES = ['gemm_f32.hip', 'gemm_f32_patch2.hip', 'gemm_f32_conv3.hip', 'gemm_x3v2.hip', 'gemm_h2.hip', 'gemm_h2p.hip', 'gemm_h2n.hip', 'mlp_h2c.hip', 'mlp_sk.hip', 'gemm_lp.hip', 'gemm_q8.hip', 'pointwise.hip', 'dwconv_cl.hip', 'dwconv_cl_bf16.hip', 'dwconv_cl_h2.hip', 'dwconv_cl_q8.hip', 'rans_host.cpp', 'plan_runtime.cpp']
HEADERS = ['gemm_common.h', 'device_math.h']
INCLUDED = {'dwconv_cl_bf16.hip': 'dwconv_cl.hip', 'dwconv_cl_h2.hip': 'dwconv_cl.hip', 'dwconv_cl_q8.hip': 'dwconv_cl.hip', 'gemm_f32_patch2.hip': 'gemm_f32.hip', 'gemm_f32_conv3.hip': 'gemm_f32.hip'}   # wrapper -> the source it #includes
HEADER = os.path.join(os.path.dirname(HERE), 'include', 'lvae_hip.h')


def _stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [HEADER, os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OUT_DIR, exist_ok=True)
    if not (force or _stale()):
        return OUT
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs, procs = [], []
    hdr_t = max(os.path.getmtime(d) for d in [os.path.join(CSRC, h) for h in HEADERS] + [HEADER, os.path.abspath(__file__)])
    for s in SOURCES:                                      # translation units compile concurrently; unchanged ones are kept
        src = os.path.join(CSRC, s)
        obj = os.path.join(OUT_DIR, s + '.o')
        objs.append(obj)
        src_t = max(os.path.getmtime(src), os.path.getmtime(os.path.join(CSRC, INCLUDED[s])) if s in INCLUDED else 0)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(src_t, hdr_t):
            continue
        cmd = [hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c', src, '-o', obj]
        if s.endswith('.cpp'):
            cmd.insert(1, '-x'); cmd.insert(2, 'hip')      # host-only TU still goes through hipcc for one toolchain
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', OUT] + objs + ['-lpthread']
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
