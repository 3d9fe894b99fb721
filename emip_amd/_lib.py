"""ctypes binding of libemip_hip.so (the C ABI declared in include/emip_hip.h).

The prototypes are read from the header itself, so the Python side can never
drift from the declared ABI.  There is no fallback: if the shared library is
missing or a symbol is absent, importing the compute path raises."""
import ctypes
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(_HERE), "include", "emip_hip.h")
LIB_PATH = os.environ.get("EMIP_HIP_LIB") or os.path.join(_HERE, "libemip_hip.so")   # override: A/B calibration of two builds
TUNING = os.path.basename(LIB_PATH).startswith("libemip_hip_tuning")

_CTYPES = {
    "const void*": ctypes.c_void_p, "void*": ctypes.c_void_p,
    "const float*": ctypes.c_void_p, "float*": ctypes.c_void_p,
    "const double*": ctypes.c_void_p, "double*": ctypes.c_void_p,
    "const int*": ctypes.c_void_p, "int*": ctypes.c_void_p,
    "long long*": ctypes.c_void_p, "unsigned char*": ctypes.c_void_p, "const unsigned char*": ctypes.c_void_p,
    "int": ctypes.c_int, "long": ctypes.c_long, "float": ctypes.c_float, "double": ctypes.c_double,
}

EMIP_F32, EMIP_BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
ERRORS = {-1: "EMIP_E_INVALID (argument check failed, nothing launched)", -2: "EMIP_E_LAUNCH (HIP launch error)"}


def parse_header(path=HEADER):
    """-> {name: [(ctype_name, arg_name), ...]} for every `int emip_*(...)` prototype."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    if not TUNING:       # the calibration switches exist in libemip_hip_tuning.so only
        src = re.sub(r"#ifdef EMIP_TUNING.*?#endif", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\bint\s+(emip_\w+)\s*\(([^)]*)\)\s*;", src):
        name, args = m.group(1), m.group(2).strip()
        lst = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.*?)(\w+)$", a)
                ty = mm.group(1).strip().replace(" *", "*")
                lst.append((ty, mm.group(2)))
        protos[name] = lst
    return protos


class EmipLibraryError(RuntimeError):
    pass


_lib = None
_protos = None


def load():
    """Load libemip_hip.so and attach prototypes.  Raises if it is not built."""
    global _lib, _protos
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EmipLibraryError(
            f"{LIB_PATH} not found: build it with `make -C emip_amd/csrc` (or __graft_entry__.build()). "
            "emip_amd has no non-HIP fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    _protos = parse_header()
    for name, args in _protos.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise EmipLibraryError(f"libemip_hip.so does not export {name} declared in include/emip_hip.h") from e
        fn.restype = ctypes.c_int
        fn.argtypes = [_CTYPES[t] for t, _ in args]
    _lib = lib
    return lib


_profile = None  # when a list: (name, args, start_event, end_event) of every launch is appended


def profile(records):
    """Enable (list) / disable (None) per-call HIP-event timing on the current stream."""
    global _profile
    _profile = records


def call(name, *args):
    lib = load()
    if _profile is not None:
        import torch
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = getattr(lib, name)(*args)
        e.record()
        _profile.append((name, args, s, e))
    else:
        rc = getattr(lib, name)(*args)
    if rc != 0:
        raise EmipLibraryError(f"{name} failed: {ERRORS.get(rc, rc)}")
    return rc
