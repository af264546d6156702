"""Builds libpastml_hip.so in-tree with hipcc for gfx950 (no JIT cache, the .so travels with the repo snapshot)."""
import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libpastml_hip.so')
SOURCES = ['pml_api.hip']
HEADERS = ['pml_device.h', 'pml_kernels_f81.h', 'pml_kernels_misc.h', 'pml_model.h', 'pml_kernels_matrix.h', 'pml_kernels_pij.h',
           'pml_kernels_counts.h',
           'pml_kernels_eigen_mfma.h', 'pml_kernels_eigen_gemm.h', 'pml_kernels_eigen_joint.h', 'pml_comm.h',
           os.path.join('..', '..', 'include', 'pastml_hip.h')]
FLAGS = ['-O3', '--offload-arch=gfx950', '-std=c++17', '-ffp-contract=on', '-shared', '-fPIC']
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


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    cmd = [find_hipcc()] + FLAGS + ['-DPML_BUILD_DIGEST="{}"'.format(source_digest()), '-o', LIB] + SOURCES + ['-ldl']
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == '__main__':
    print(build(force=True, verbose=True))
