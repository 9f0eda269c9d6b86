"""remhos_amd -- MI355X-native (gfx950) implementation of the Remhos DG remap RK stage.

The product is the HIP library `librmh.so` behind the C ABI of include/rmh.h; this package is
the thin Python plumbing (ctypes binding, torch device buffers, torch.distributed halo
exchange) used by tests/ and bench.py.
"""
from .capi import Context, RmhError, load_library  # noqa: F401
