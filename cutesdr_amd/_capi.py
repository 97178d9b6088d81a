"""ctypes binding of libcutesdr_mi.so (the C ABI declared in include/cutesdr_mi.h).

There is no CPU fallback: if the library is missing or no GPU is usable, calls raise.
"""
import ctypes as C
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
# CSDR_LIB_PATH: a diagnostic build of the same library (tools/k1_stamps.py), never a different implementation
LIB_PATH = os.environ.get("CSDR_LIB_PATH") or os.path.join(HERE, "libcutesdr_mi.so")
HEADER = os.path.join(HERE, "..", "include", "cutesdr_mi.h")

CSDR_OK, CSDR_EINVAL, CSDR_EHIP, CSDR_ENOMEM, CSDR_ESTATE = 0, -1, -2, -3, -4


class CsdrError(RuntimeError):
    pass


_lib = None


def declared_symbols():
    """Every function name the public header declares."""
    return sorted(prototypes())


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CsdrError("libcutesdr_mi.so is not built (run `python -m cutesdr_amd._build` or "
                            "__graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        _declare(L)
        _lib = L
    return _lib


P, D, I, LL, U64 = C.c_void_p, C.c_double, C.c_int, C.c_longlong, C.c_ulonglong


def _ctype(t):
    t = t.replace("const", "").strip()
    if "*" in t:
        return C.c_char_p if t.replace(" ", "") == "char*" else P
    return {"int": I, "double": D, "void": None, "long long": LL, "unsigned long long": U64, "float": C.c_float, "short": C.c_short, "csdr_tap_fn": P}[t]


def prototypes():
    """Parse include/cutesdr_mi.h: {name: (restype, [argtypes])}."""
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"typedef struct csdr_demod_info \{.*?\} csdr_demod_info;", "", txt, flags=re.S)
    txt = re.sub(r"typedef void \(\*csdr_tap_fn\)\([^)]*\);", "", txt)
    out = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ ]*?[ \*]+)(csdr_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", txt):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("typedef"):
            continue
        argt = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                argt.append(_ctype(a if a.endswith("*") else re.sub(r"\s*[A-Za-z_][A-Za-z0-9_]*$", "", a)))
        out[name] = (_ctype(ret), argt)
    return out


def _declare(L):
    for name, (res, args) in prototypes().items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args


def last_error():
    return lib().csdr_last_error().decode(errors="replace")


def check(rc, what=""):
    if rc is None or (isinstance(rc, int) and rc < 0):
        raise CsdrError("%s failed (%s): %s" % (what or "csdr call", rc, last_error()))
    return rc


def check_ptr(p, what=""):
    if not p:
        raise CsdrError("%s failed: %s" % (what or "csdr call", last_error()))
    return p
