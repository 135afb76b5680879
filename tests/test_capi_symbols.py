"""CPU: the C-ABI library loads and exports exactly what include/drp.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, 'include', 'drp.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(drp_[a-z0-9_]+)\s*\(', src)))


@pytest.fixture(scope='module')
def built():
    import __graft_entry__ as g
    g.build()
    from dyn_res_pile_manip_amd import _lib
    return _lib


def test_header_and_binding_agree(built):
    names = declared_functions()
    assert len(names) >= 25
    assert sorted(built.SIGNATURES.keys()) == names


def test_library_exports_every_declared_symbol(built):
    lib = ctypes.CDLL(built.LIB_PATH)
    for name in declared_functions():
        assert hasattr(lib, name), name


def test_create_without_gpu_fails_loudly(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from dyn_res_pile_manip_amd.engine import Engine
    with pytest.raises(built.DrpError):
        Engine(0)


def test_weight_blob_roundtrip(golden):
    import numpy as np
    from dyn_res_pile_manip_amd import weights
    blob = weights.blob_from_state_dict(golden.weights_seed0)
    assert blob.shape == (38403,)
    sd = weights.state_dict_from_blob(blob)
    for k, shape in weights.STATE_DICT_KEYS:
        np.testing.assert_array_equal(sd[k], golden.weights_seed0['w/' + k])
    with pytest.raises(KeyError):
        weights.blob_from_state_dict({}, strict=True)
    assert weights.blob_from_state_dict({}, strict=False).sum() == 0
