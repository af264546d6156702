"""CPU-only: the C-ABI library builds, loads and exports every symbol that include/pastml_hip.h declares."""
import ctypes
import os
import re

import pytest

from conftest import REPO
from pastml_amd import hip


def header_functions():
    text = open(os.path.join(REPO, 'include', 'pastml_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(pml_[a-z0-9_]+)\s*\(', text)))


def test_library_is_built():
    assert os.path.exists(hip.library_path()), 'run __graft_entry__.build() first'


def test_exports_match_header():
    lib = ctypes.CDLL(hip.library_path())
    declared = header_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), '{} is declared in pastml_hip.h but not exported'.format(name)
    # and the Python binding knows every one of them
    assert sorted(hip.SIGNATURES) == declared


def test_prototypes_load_without_gpu():
    lib = hip.load_library()
    assert lib.pml_version() >= 100
    assert isinstance(lib.pml_last_error(), bytes)


def test_no_cpu_fallback_without_device():
    """On a box without a GPU the product path must fail loudly, not compute on the host."""
    if hip.device_count() > 0:
        pytest.skip('a GPU is present')
    from pastml_amd.tree import FlatForest
    with pytest.raises(hip.HipUnavailableError):
        hip.Engine(FlatForest.balanced(3), 1, 4)
