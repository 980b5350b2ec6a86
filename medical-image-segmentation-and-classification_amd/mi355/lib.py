"""ctypes binding of libmi355conv.so (the C ABI declared in include/mi355conv.h).

The prototypes are parsed from the header itself, so the Python side can never drift from
the ABI: every function declared there must be exported by the library (checked by
tests/test_abi.py) and is callable as ``lib.mi355_xxx(...)`` with torch tensors, ints and
floats.  There is no CPU fallback: if the shared object is missing or a launcher returns
non-zero, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes
import os
import re

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(os.path.dirname(_HERE))
HEADER = os.path.join(REPO_ROOT, "include", "mi355conv.h")
SO_PATH = os.environ.get("MI355_LIB") or os.path.join(_HERE, "libmi355conv.so")   # MI355_LIB: A/B another build

F32, BF16, F16 = 0, 1, 2
DTYPE_CODE = {torch.float32: F32, torch.bfloat16: BF16, torch.float16: F16}

_CTYPES = {
    "int": ctypes.c_int, "float": ctypes.c_float, "long long": ctypes.c_longlong,
    "uint64_t": ctypes.c_uint64, "int64_t": ctypes.c_int64, "mi355_stream_t": ctypes.c_void_p,
}


def parse_header(path=HEADER):
    """-> {name: (restype, [(ctype, argname), ...])} for every prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int|const char\*|void\*)\s+(mi355_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        arglist = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    arglist.append((ctypes.c_void_p, a.split("*")[-1].strip()))
                else:
                    ty, nm = a.rsplit(" ", 1)
                    ty = ty.replace("const ", "").strip()
                    arglist.append((_CTYPES[ty], nm))
        protos[name] = ({"int": ctypes.c_int, "const char*": ctypes.c_char_p, "void*": ctypes.c_void_p}[ret], arglist)
    return protos


class _Lib:
    def __init__(self):
        self._dll = None
        self._protos = None

    def load(self):
        if self._dll is not None:
            return self
        if not os.path.exists(SO_PATH):
            raise RuntimeError(
                f"libmi355conv.so not found at {SO_PATH}: build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
                "There is no CPU fallback for the MI355X path.")
        self._dll = ctypes.CDLL(SO_PATH)
        self._protos = parse_header()
        for name, (ret, args) in self._protos.items():
            try:
                fn = getattr(self._dll, name)
            except AttributeError:             # reported by tests/test_abi.py; raises when called
                continue
            fn.restype = ret
            fn.argtypes = [t for t, _ in args]
        return self

    @property
    def protos(self):
        self.load()
        return self._protos

    def raw(self, name):
        self.load()
        return getattr(self._dll, name)

    def __getattr__(self, name):
        if not name.startswith("mi355_"):
            raise AttributeError(name)
        self.load()
        fn = getattr(self._dll, name)
        ret, args = self._protos[name]
        stream_last = bool(args) and args[-1][1] == "s"

        def call(*a):
            conv = []
            for v in a:
                if isinstance(v, torch.Tensor):
                    conv.append(v.data_ptr())
                elif v is None:
                    conv.append(None)
                else:
                    conv.append(v)
            if stream_last and len(conv) == len(args) - 1:
                conv.append(torch.cuda.current_stream().cuda_stream)
            if len(conv) != len(args):
                raise TypeError(f"{name}: expected {len(args)} args ({[n for _, n in args]}), got {len(conv)}")
            rc = fn(*conv)
            if stream_last and ret is ctypes.c_int and rc != 0:
                raise RuntimeError(f"{name} failed (rc={rc}): {self._dll.mi355_last_error().decode()}")
            return rc

        call.__name__ = name
        self.__dict__[name] = call
        return call


lib = _Lib()


def available():
    return os.path.exists(SO_PATH)
