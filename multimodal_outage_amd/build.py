"""Build the gfx950 shared library in-tree (hipcc cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libmo_hip.so')
SOURCES = ['gwnet_ops.hip', 'unet_ops.hip', 'gemm_bf16.hip']


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + \
           [os.path.join(os.path.dirname(HERE), 'include', 'mo_hip.h')]
    return any(os.path.getmtime(d) > t for d in deps if os.path.isfile(d))


def build(force=False, verbose=True):
    """hipcc --offload-arch=gfx950 -shared -fPIC -> multimodal_outage_amd/libmo_hip.so"""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [hipcc, '-O3', '--offload-arch=gfx950', '-std=c++17', '-shared', '-fPIC',
           '-I', os.path.join(os.path.dirname(HERE), 'include'), '-I', '/opt/rocm/include',
           '-o', LIB + '.tmp'] + srcs + ['-L/opt/rocm/lib', '-lrccl', '-Wl,-rpath,/opt/rocm/lib']
    if verbose:
        print(' '.join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(LIB + '.tmp', LIB)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
