"""The C-ABI library loads and exports every symbol include/sparsefusion_hip.h declares
(no compute calls: runs without a GPU), and the ctypes table matches the header."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'sparsefusion_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    out = {}
    for m in re.finditer(r'SF_API\s+[\w\s\*]+?\b(sf_\w+)\s*\(([^;]*?)\)\s*;', text, flags=re.S):
        args = [a.strip() for a in m.group(2).split(',') if a.strip() and a.strip() != 'void']
        out[m.group(1)] = len(args)
    return out


def test_library_exports_every_declared_symbol():
    from sparsefusion_amd import _lib
    syms = declared_symbols()
    assert len(syms) >= 18
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in syms:
        assert hasattr(L, name), f'{name} declared in the header but not exported'


def test_ctypes_table_matches_header():
    from sparsefusion_amd import _lib
    syms = declared_symbols()
    assert set(_lib.SIGNATURES) == set(syms)
    for name, nargs in syms.items():
        assert len(_lib.SIGNATURES[name][1]) == nargs, name


def test_host_side_queries_and_errors():
    from sparsefusion_amd import _lib
    L = _lib.lib()
    assert L.sf_version() >= 100
    assert L.sf_hash_capacity(1000) == 2048 and L.sf_hash_capacity(0) == 1024
    assert [L.sf_conv_padded_cin(c) for c in (1, 5, 8, 9, 16, 64, 65, 128, 129)] == \
        [8, 8, 8, 16, 16, 64, 128, 128, -1]
    assert L.sf_hard_voxelize_workspace(300000, 10) > 300000 * 4
    # argument validation happens before anything touches a device
    rc = L.sf_dynamic_voxelize(None, 10, 2, _lib.f3([1, 1, 1]), _lib.f6([0, 0, 0, 1, 1, 1]), None,
                               None)
    assert rc == -1 and b'f >= 3' in L.sf_last_error()
    rc = L.sf_sparse_conv_f32(None, 0, 7, None, 0, 27, -1, None, 16, None, None, None, 0, None, None, 0,
                              None)
    assert rc == 0  # empty problem is a no-op
    rc = L.sf_sparse_conv_f32(None, 4, 16, None, 4, 64, -1, None, 16, None, None, None, 0, None, None,
                              4, None)
    assert rc == -1


def test_product_path_has_no_cpu_fallback():
    import torch
    from sparsefusion_amd.voxelize import Voxelization
    layer = Voxelization([0.1, 0.1, 0.1], [0, 0, 0, 1, 1, 1], 5, 100)
    with pytest.raises(RuntimeError, match='HIP device'):
        layer(torch.zeros(10, 4))
    # and nothing in the package imports the oracle
    pkg = os.path.join(ROOT, 'sparsefusion_amd')
    for fn in os.listdir(pkg):
        if fn.endswith('.py'):
            assert 'oracle' not in open(os.path.join(pkg, fn)).read().replace('oracle-sized', ''), fn
