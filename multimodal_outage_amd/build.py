"""Build the gfx950 shared library in-tree (hipcc cross-compiles without a GPU).

Every source is compiled to its own object (in parallel, only when it or one of the headers it includes changed) and
the objects are linked into multimodal_outage_amd/libmo_hip.so."""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, '_build')
LIB = os.path.join(HERE, 'libmo_hip.so')
SOURCES = ['gwnet_ops.hip', 'unet_ops.hip', 'gemm_bf16.hip', 'gwnet_small.hip', 'comm.hip']   # comm.hip is what needs -lrccl
_INC = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def _deps(path, seen=None):
    """The source plus every project header it includes (transitively)."""
    seen = seen if seen is not None else set()
    if path in seen or not os.path.isfile(path):
        return seen
    seen.add(path)
    for inc in _INC.findall(open(path).read()):
        for base in (os.path.dirname(path), CSRC, os.path.join(ROOT, 'include')):
            _deps(os.path.join(base, inc), seen)
    return seen


def _obj(src):
    return os.path.join(OBJ, os.path.splitext(src)[0] + '.o')


def _stale(src):
    o = _obj(src)
    if not os.path.exists(o):
        return True
    t = os.path.getmtime(o)
    return any(os.path.getmtime(d) > t for d in _deps(os.path.join(CSRC, src)))


def build(force=False, verbose=True):
    """hipcc --offload-arch=gfx950 -c per source, then -shared -> multimodal_outage_amd/libmo_hip.so"""
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if not force and os.path.exists(LIB):
        # the library is newer than every source and header: nothing to do, even where the per-source objects are absent
        # (they are build scratch and do not travel to the GPU box; the .so does)
        t = os.path.getmtime(LIB)
        deps = set()
        for s_ in srcs:
            deps |= _deps(os.path.join(CSRC, s_))
        if all(os.path.getmtime(d) <= t for d in deps):
            return LIB
    os.makedirs(OBJ, exist_ok=True)
    todo = [s for s in srcs if force or _stale(s)]
    flags = ['-O3', '--offload-arch=gfx950', '-std=c++17', '-fPIC', '-I', os.path.join(ROOT, 'include'),
             '-I', '/opt/rocm/include']

    def cc(s):
        cmd = [hipcc] + flags + ['-c', os.path.join(CSRC, s), '-o', _obj(s) + '.tmp']
        if verbose:
            print(' '.join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        os.replace(_obj(s) + '.tmp', _obj(s))

    if todo:
        with ThreadPoolExecutor(max_workers=min(4, len(todo))) as ex:
            list(ex.map(cc, todo))
    objs = [_obj(s) for s in srcs]
    if todo or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB + '.tmp'] + objs + \
              ['-L/opt/rocm/lib', '-lrccl', '-Wl,-rpath,/opt/rocm/lib']
        if verbose:
            print(' '.join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
        os.replace(LIB + '.tmp', LIB)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
