"""Builds libpastml_hip.so in-tree with hipcc for gfx950 (no JIT cache, the .so travels with the repo snapshot).

The library is several translation units -- pml_api.hip (contexts, schedules, the C-ABI) and one pml_launch_*.hip per kernel
family -- compiled in parallel into pastml_amd/csrc/build/*.o (kept out of history and off the GPU box) and linked into one
shared object.  An object is recompiled when the digest of what IT is compiled from changes; the library carries the digest
over everything (pml_build_digest)."""
import hashlib
import os
import shutil
import subprocess
import time
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, 'build')
LIB = os.path.join(HERE, 'libpastml_hip.so')
SOURCES = ['pml_launch_f81_level.hip', 'pml_launch_f81_wide.hip', 'pml_launch_f81_small.hip', 'pml_launch_f81_blocks.hip', 'pml_launch_f81_super.hip',
           'pml_launch_matrix.hip', 'pml_launch_eigen_mfma.hip', 'pml_launch_eigen_gemm.hip', 'pml_launch_eigen_gemm_wide.hip', 'pml_launch_eigen_joint.hip', 'pml_api.hip']
HEADERS = ['pml_device.h', 'pml_kernels_f81.h', 'pml_kernels_misc.h', 'pml_model.h', 'pml_kernels_matrix.h', 'pml_kernels_pij.h', 'pml_kernels_pij_wide.h',
           'pml_kernels_counts.h',
           'pml_kernels_eigen_mfma.h', 'pml_kernels_eigen_gemm.h', 'pml_kernels_eigen_joint.h', 'pml_comm.h', 'pml_host.h',
           'pml_launch.h', 'pml_launch_f81_level.h',
           os.path.join('..', '..', 'include', 'pastml_hip.h')]
CFLAGS = ['-O3', '--offload-arch=gfx950', '-std=c++17', '-ffp-contract=on', '-fPIC']
FLAGS = CFLAGS + ['-shared']
_MARKER = b'PML_BUILD_DIGEST='


def find_hipcc():
    for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found')


def source_digest():
    """sha256 (16 hex digits) over everything the library is compiled from: sources, headers, flags."""
    h = hashlib.sha256()
    h.update(' '.join(FLAGS).encode())
    for name in SOURCES + HEADERS:
        h.update(os.path.basename(name).encode())
        with open(os.path.join(CSRC, name), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def _unit_digest(source):
    """What one object file depends on: its source, every header (they are few and shared), the compile flags."""
    h = hashlib.sha256()
    h.update(' '.join(CFLAGS).encode())
    for name in [source] + HEADERS:
        h.update(os.path.basename(name).encode())
        with open(os.path.join(CSRC, name), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def library_digest(path=LIB):
    """The digest compiled into the library file (pml_build_digest), read without loading it; None if there is none."""
    try:
        with open(path, 'rb') as f:
            blob = f.read()
    except OSError:
        return None
    at = blob.find(_MARKER)
    if at < 0:
        return None
    end = blob.find(b'\0', at)
    return blob[at + len(_MARKER):end].decode('ascii', 'replace')


def is_stale():
    """The library is missing or was compiled from other sources than the tree holds (by content, not by time stamps)."""
    return library_digest() != source_digest()


def _compile(hipcc, source, digest, force, verbose):
    obj = os.path.join(OBJ, os.path.splitext(source)[0] + '.o')
    stamp = obj + '.digest'
    # pml_api.hip carries the library's digest: any change anywhere recompiles it (it is the quick one)
    want = _unit_digest(source) + (digest if source == 'pml_api.hip' else '')
    if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == want:
        return obj
    cmd = [hipcc] + CFLAGS + ['-c', '-o', obj, source]
    if source == 'pml_api.hip':
        cmd.insert(-3, '-DPML_BUILD_DIGEST="{}"'.format(digest))
    t0 = time.time()
    subprocess.check_call(cmd, cwd=CSRC)
    if verbose:
        print('{:6.1f} s  {}'.format(time.time() - t0, ' '.join(cmd)), flush=True)
    with open(stamp, 'w') as f:
        f.write(want)
    return obj


def build(force=False, verbose=False, jobs=None):
    if not force and not is_stale():
        return LIB
    hipcc = find_hipcc()
    digest = source_digest()
    os.makedirs(OBJ, exist_ok=True)
    jobs = jobs or max(1, min(len(SOURCES), len(os.sched_getaffinity(0))))
    with ThreadPoolExecutor(jobs) as pool:
        objects = list(pool.map(lambda s: _compile(hipcc, s, digest, force, verbose), SOURCES))
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objects + ['-ldl']
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == '__main__':
    import sys
    print(build(force='--incremental' not in sys.argv, verbose=True))
