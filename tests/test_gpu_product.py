"""GPU twin of tests/test_product_remap.py: the product-remap kernels (rmh_product_ratio, rmh_elem_minmax_masked,
rmh_fct_product) on the MI355X against the oracle's restatement of remhos_sync.cpp / remhos_fct.cpp:26-153, 543-566."""
import numpy as np
import pytest

from tests.test_product_remap import check_product, product_case, run_product

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("p,mesh,rs", [(2, "cube01_hex", 1), (3, "cube01_hex", 2), (4, "cube01_hex", 1), (3, "periodic-cube", 1)])
def test_product_remap_gpu(p, mesh, rs):
    import torch

    from remhos_amd.capi import load_library

    lib = load_library()
    r, u, us = product_case(p, mesh, rs)
    dev = torch.device("cuda:0")
    o = run_product(r, lib, lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev), lambda t: t.cpu().numpy(), p, u, us, 0.01, 0.3)
    assert o["el"].sum() not in (0, len(o["el"]))
    check_product(r, o, u, us, 0.01, 1e-12)
