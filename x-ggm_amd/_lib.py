"""ctypes binding of libxggm_hip.so (the C ABI declared in include/xggm.h).

There is deliberately NO fallback: if the shared library is missing the import raises,
and if a call returns non-zero a RuntimeError carries ``xggm_last_error()``.
"""
import ctypes
import os
import re

import torch  # noqa: F401  (loads PyTorch's HIP runtime first so both share it)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("XGGM_LIB") or os.path.join(_HERE, "csrc", "libxggm_hip.so")  # XGGM_LIB: instrumented builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "xggm.h")

_CT = {
    "int": ctypes.c_int, "float": ctypes.c_float, "int64_t": ctypes.c_int64,
    "uint32_t": ctypes.c_uint32, "uint64_t": ctypes.c_uint64, "xggm_stream_t": ctypes.c_void_p,
    "size_t": ctypes.c_size_t,
}


RESTYPE = {}


def parse_header(path=HEADER_PATH):
    """``{symbol: [ctypes argtypes]}`` for every ``int xggm_*(...)`` declared in xggm.h."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(int|size_t|const char\*)\s+(xggm_\w+)\s*\(([^)]*)\)\s*;", src):
        name, args = m.group(2), m.group(3).strip()
        types = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    types.append(ctypes.c_void_p)
                else:
                    base = a.replace("const ", "").split()[0]
                    types.append(_CT[base])
        out[name] = types
        RESTYPE[name] = {"int": ctypes.c_int, "size_t": ctypes.c_size_t, "const char*": ctypes.c_char_p}[m.group(1)]
    return out


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "xggm_amd: %s is missing -- build it with `python __graft_entry__.py` (hipcc, gfx950). "
            "There is no CPU fallback." % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, types in parse_header().items():
        fn = getattr(lib, name)  # AttributeError if the header declares a symbol the .so lacks
        fn.argtypes = types
        fn.restype = RESTYPE[name]
    return lib


lib = _load()


def last_error():
    return lib.xggm_last_error().decode()


def check(rc, name):
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (name, rc, last_error()))


def ptr(t):
    """device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    check(getattr(lib, name)(*args), name)
