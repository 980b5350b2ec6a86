"""CPU: the C-ABI shared library loads and exports every symbol include/mi355conv.h declares
(no compute calls — there is no GPU here), and the ctypes binding agrees with the header."""
import ctypes
import os

import pytest

from mi355 import lib as L


def test_header_parses():
    protos = L.parse_header()
    assert len(protos) >= 40
    assert "mi355_conv2d_igemm" in protos and len(protos["mi355_conv2d_igemm"][1]) == 24
    for name, (ret, args) in protos.items():
        assert name.startswith("mi355_")


@pytest.mark.skipif(not os.path.exists(L.SO_PATH), reason="libmi355conv.so not built (run __graft_entry__.build())")
def test_library_exports_every_declared_symbol():
    dll = ctypes.CDLL(L.SO_PATH)
    missing = [n for n in L.parse_header() if not hasattr(dll, n)]
    assert not missing, f"declared in include/mi355conv.h but not exported: {missing}"
    dll.mi355_version.restype = ctypes.c_int
    assert dll.mi355_version() >= 100
    dll.mi355_rowreduce_blocks.restype = ctypes.c_int
    dll.mi355_rowreduce_blocks.argtypes = [ctypes.c_longlong]
    assert dll.mi355_rowreduce_blocks(1) == 1 and dll.mi355_rowreduce_blocks(10**9) == 1024
    dll.mi355_conv2d_wgrad_splits.restype = ctypes.c_int
    dll.mi355_conv2d_wgrad_splits.argtypes = [ctypes.c_int] * 7
    assert dll.mi355_conv2d_wgrad_splits(32, 256, 256, 64, 64, 3, 3) >= 32


@pytest.mark.skipif(not os.path.exists(L.SO_PATH), reason="libmi355conv.so not built")
def test_argument_errors_are_reported_without_a_gpu():
    """Launchers validate arguments before touching the device, so bad calls fail cleanly."""
    dll = ctypes.CDLL(L.SO_PATH)
    dll.mi355_last_error.restype = ctypes.c_char_p
    fn = dll.mi355_conv2d_igemm
    fn.restype = ctypes.c_int
    fn.argtypes = [t for t, _ in L.parse_header()["mi355_conv2d_igemm"][1]]
    rc = fn(None, None, None, None, 1, 4, 4, 32, 32, 4, 4, 32, 32, 3, 3, 1, 1, -1, 1, 0, 0, None, 0, None)
    assert rc == -1 and b"null" in dll.mi355_last_error()


@pytest.mark.skipif(not os.path.exists(L.SO_PATH), reason="libmi355conv.so not built")
def test_plan_replay_knows_every_launcher_with_its_arity():
    """csrc/plan.cpp reaches a launcher through a trampoline typed by its prototype: every `int mi355_*` entry point of the
    header has one (with the header's arity), a table is rejected on a wrong name / arity, and launch arguments are
    validated by the launcher itself when the plan runs (here: without a GPU, a null-pointer launch fails cleanly)."""
    lib = L.lib
    arity = lib.raw("mi355_plan_arity")
    for name, (ret, args) in L.parse_header().items():
        if ret is ctypes.c_int and not name.startswith("mi355_plan_"):
            assert arity(name.encode()) == len(args), name
    assert arity(b"mi355_no_such_launcher") == -1
    create, pset, run = lib.raw("mi355_plan_create"), lib.raw("mi355_plan_set"), lib.raw("mi355_plan_run")
    cp = create(2)
    assert cp
    n = len(L.parse_header()["mi355_conv2d_igemm"][1])
    slots = (ctypes.c_uint64 * n)()
    assert pset(cp, 0, b"mi355_conv2d_igemm", slots, n - 1, 0) == -1 and b"takes 24" in lib.raw("mi355_last_error")()
    assert pset(cp, 0, b"mi355_bogus", slots, n, 0) == -1
    assert pset(cp, 5, b"mi355_conv2d_igemm", slots, n, 0) == -1
    assert pset(cp, 0, b"mi355_conv2d_igemm", slots, n, 0) == 0
    assert run(cp, 0, 2, None, None) == -1 and lib.raw("mi355_plan_last_index")(cp) == 0       # null pointers: rejected by the launcher
    assert b"null" in lib.raw("mi355_last_error")()
    assert lib.raw("mi355_plan_destroy")(cp) == 0
