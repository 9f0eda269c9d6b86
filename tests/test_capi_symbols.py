"""The C-ABI libraries load on a machine without a GPU and export every symbol the headers
declare (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(rmhd?_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g

    if not os.path.exists(os.path.join(ROOT, "remhos_amd", "librmh.so")):
        g.build()
    return True


def test_rmh_h_symbols(built):
    from remhos_amd.capi import SYMBOLS, load_library

    names = declared("rmh.h")
    assert names == sorted(SYMBOLS)
    lib = load_library()
    for n in names:
        assert getattr(lib, n) is not None
    assert b"gfx950" in lib.rmh_version()


def test_rmh_driver_h_symbols(built):
    from remhos_amd.case import DRIVER_SYMBOLS

    names = declared("rmh_driver.h")
    assert names == sorted(DRIVER_SYMBOLS)
    lib = ctypes.CDLL(os.path.join(ROOT, "remhos_amd", "librmh.so"))
    for n in names:
        assert getattr(lib, n) is not None
    host = ctypes.CDLL(os.path.join(ROOT, "remhos_amd", "librmh_host.so"))
    for n in names:
        if n not in ("rmhd_run", "rmhd_run_state", "rmhd_run_rank", "rmhd_run_partitioned", "rmhd_id_file_exchange", "rmhd_axpby"):  # (the time loops and the vector kernel need the GPU library)
            assert getattr(host, n) is not None


def test_no_device_is_an_error_not_a_fallback(built):
    """Without a GPU the product path must refuse to run (no CPU fallback)."""
    import numpy as np
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from remhos_amd.capi import Context, RmhError, load_library

    lib = load_library()
    x0 = np.zeros((1, 3, 27))
    st = -np.ones((1, 27), np.int32)
    st[0, 13] = 0
    with pytest.raises(RmhError, match="no HIP device|hip"):
        Context(lib, order=2, exec_mode=1, x0=x0, vel=x0, face_nbr=-np.ones((1, 6), np.int32), stencil27=st)


def test_layout_tables_are_validated(built):
    """rmh_create rejects neighbour tables that would index outside u / the ghost block (checked before any
    device work, so this runs without a GPU)."""
    import numpy as np

    from remhos_amd.capi import Context, RmhError, load_library

    lib = load_library()
    x0 = np.zeros((1, 3, 27))
    nbr = -np.ones((1, 6), np.int32)
    st = -np.ones((1, 27), np.int32)
    st[0, 13] = 0
    bad = nbr.copy()
    bad[0, 2] = 1  # only element 0 exists, no ghosts
    with pytest.raises(RmhError, match="face_nbr entry out of range"):
        Context(lib, order=2, exec_mode=1, x0=x0, vel=x0, face_nbr=bad, stencil27=st)
    bad = st.copy()
    bad[0, 5] = 7
    with pytest.raises(RmhError, match="stencil27 entry out of range"):
        Context(lib, order=2, exec_mode=1, x0=x0, vel=x0, face_nbr=nbr, stencil27=bad)
    bad = st.copy()
    bad[0, 13] = -1
    with pytest.raises(RmhError, match="must be the element itself"):
        Context(lib, order=2, exec_mode=1, x0=x0, vel=x0, face_nbr=nbr, stencil27=bad)
    with pytest.raises(RmhError, match="order must be in 1..6"):
        Context(lib, order=7, exec_mode=1, x0=x0, vel=x0, face_nbr=nbr, stencil27=st)
