"""The ncclUniqueId rendezvous of the C++ stage loops (rmhd_id_file_exchange = read_or_write_id of rmh_driver.hip) across real
processes, without a GPU: what `bench.py --gpus N` / `remhos_amd_run -comm-file` do before ncclCommInitRank (the reference gets its
communicator from MPI_Init, remhos.cpp:217-220).  Readers that start BEFORE the writer, a stale file of an earlier launch on the same
path, a launch without a nonce."""
import ctypes as C
import multiprocessing as mp
import os
import time

import pytest


def _lib():
    from remhos_amd.capi import load_library
    from remhos_amd.case import bind_driver
    from tests.helpers import emu_library_path

    return bind_driver(load_library(emu_library_path()))


def _rank(path, writer, nonce, delay, payload, q):
    if nonce is None:
        os.environ.pop("RMH_COMM_NONCE", None)
    else:
        os.environ["RMH_COMM_NONCE"] = str(nonce)
    lib = _lib()
    time.sleep(delay)
    buf = C.create_string_buffer(bytes(payload) if writer else b"\0" * 128, 128)
    rc = lib.rmhd_id_file_exchange(path.encode(), 1 if writer else 0, buf)
    q.put((writer, rc, bytes(buf.raw)))


def run_ranks(path, nonce, n_readers, writer_delay, payload):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_rank, args=(path, False, nonce, 0.0, payload, q)) for _ in range(n_readers)]
    ps.append(ctx.Process(target=_rank, args=(path, True, nonce, writer_delay, payload, q)))
    for p in ps:
        p.start()
    out = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=30)
    return out


def test_readers_started_before_the_writer_get_this_launchs_id(tmp_path):
    path = str(tmp_path / "rmh.id")
    payload = bytes(range(128))
    out = run_ranks(path, 424242, 3, 1.0, payload)
    assert len(out) == 4 and all(rc == 0 for _, rc, _ in out)
    assert all(raw == payload for w, _, raw in out if not w)


def test_a_stale_file_of_another_launch_is_not_taken(tmp_path):
    """the same path carries the record of an EARLIER launch (other nonce): the readers wait for this launch's record"""
    path = str(tmp_path / "rmh.id")
    old, new = bytes([7] * 128), bytes([9] * 128)
    assert run_ranks(path, 111, 0, 0.0, old)[0][1] == 0 and os.path.exists(path)  # (left behind: a crashed launch)
    out = run_ranks(path, 222, 2, 1.5, new)
    assert all(rc == 0 for _, rc, _ in out) and all(raw == new for w, _, raw in out if not w)


def test_without_a_nonce_the_parent_pid_is_the_tag(tmp_path):
    """ranks started by one parent (a shell loop, one torchrun agent) share its pid: works without RMH_COMM_NONCE"""
    path = str(tmp_path / "rmh.id")
    payload = bytes([3] * 128)
    out = run_ranks(path, None, 2, 0.5, payload)
    assert all(rc == 0 for _, rc, _ in out) and all(raw == payload for w, _, raw in out if not w)
