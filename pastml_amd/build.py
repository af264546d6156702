"""Builds libpastml_hip.so in-tree with hipcc for gfx950 (no JIT cache, the .so travels with the repo snapshot)."""
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


def find_hipcc():
    for cand in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError('hipcc not found')


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    cmd = [find_hipcc(), '-O3', '--offload-arch=gfx950', '-std=c++17', '-ffp-contract=on', '-shared', '-fPIC', '-o', LIB] + SOURCES + ['-ldl']
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == '__main__':
    print(build(force=True, verbose=True))
