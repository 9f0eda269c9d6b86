"""rmh_stream_create_reserving: the flags of the two kinds of stream it hands out (include/rmh.h; advisor, round 5) and a stage
on a CU-masked stream giving the same bits as on the default stream."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_stream_flags_and_masked_stream_results():
    import torch

    from oracle.remhos_oracle import Config, Remhos
    from remhos_amd.capi import Context, load_library
    from tests.helpers import layout_from_oracle, perturbed

    assert torch.cuda.is_available()
    lib = load_library()
    hip = C.CDLL("libamdhip64.so")
    hip.hipStreamGetFlags.argtypes = [C.c_void_p, C.POINTER(C.c_uint)]
    plain, masked = C.c_void_p(), C.c_void_p()
    assert lib.rmh_stream_create_reserving(0, 0, C.byref(plain)) == 0
    assert lib.rmh_stream_create_reserving(0, 8, C.byref(masked)) == 0
    fl = C.c_uint(99)
    assert hip.hipStreamGetFlags(plain, C.byref(fl)) == 0 and fl.value == 1  # hipStreamNonBlocking
    assert hip.hipStreamGetFlags(masked, C.byref(fl)) == 0 and fl.value == 0  # hipStreamDefault: blocking (documented)
    assert lib.rmh_stream_create_reserving(0, 100000, C.byref(C.c_void_p())) != 0  # cannot reserve every compute unit

    r = Remhos(Config(mesh="periodic-cube", rs=1, order=3, problem=10, dt=0.01, t_final=0.7, lo=5))
    x0, vel, nbr, st = layout_from_oracle(r)
    ctx = Context(lib, order=3, exec_mode=1, x0=x0, vel=vel, face_nbr=nbr, stencil27=st)
    u = torch.from_numpy(perturbed(r.u)).to("cuda:0")
    y0, y1, y2 = (torch.empty_like(u) for _ in range(3))
    ctx.setup(0.3)
    ctx.stage_fused(u, 0.01, y0)
    torch.cuda.synchronize()
    for s, y in ((plain, y1), (masked, y2)):
        ctx.set_stream(s)
        ctx.stage_fused(u, 0.01, y)
        assert hip.hipStreamSynchronize(s) == 0
    assert torch.equal(y0, y1) and torch.equal(y0, y2)
    ctx.set_stream(None)
    ctx.close()
    assert lib.rmh_stream_destroy(plain) == 0 and lib.rmh_stream_destroy(masked) == 0
