"""ctypes binding of libcutesdr_mi.so (the C ABI declared in include/cutesdr_mi.h).

There is no CPU fallback: if the library is missing or no GPU is usable, calls raise.
"""
import ctypes as C
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libcutesdr_mi.so")
HEADER = os.path.join(HERE, "..", "include", "cutesdr_mi.h")

CSDR_OK, CSDR_EINVAL, CSDR_EHIP, CSDR_ENOMEM, CSDR_ESTATE = 0, -1, -2, -3, -4


class CsdrError(RuntimeError):
    pass


_lib = None


def declared_symbols():
    """Every function name the public header declares."""
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(csdr_[a-z0-9_]+)\s*\(", txt)))


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


def _declare(L):
    def f(name, res, *args):
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = list(args)
    f("csdr_version", I)
    f("csdr_last_error", C.c_char_p)
    f("csdr_device_count", I)
    f("csdr_dev_alloc", P, I, U64)
    f("csdr_dev_free", I, I, P)
    f("csdr_dev_upload", I, I, P, P, U64)
    f("csdr_dev_download", I, I, P, P, U64)
    f("csdr_dev_sync", I, I)
    f("csdr_fastfir_create", P, I, I)
    f("csdr_fastfir_destroy", None, P)
    f("csdr_fastfir_setup", I, P, D, D, D, D)
    f("csdr_fastfir_process", I, P, I, P, P)
    f("csdr_fastfir_batch_create", P, I, I, I)
    f("csdr_fastfir_batch_destroy", None, P)
    f("csdr_fastfir_batch_setup", I, P, I, D, D, D, D)
    f("csdr_fastfir_batch_reset", I, P)
    f("csdr_fastfir_batch_process", I, P, P, LL, I, P, LL, P, I)
    f("csdr_fastfir_batch_get_response", I, P, I, P)
    f("csdr_downconvert_create", P, I)
    f("csdr_downconvert_destroy", None, P)
    f("csdr_downconvert_set_cw_offset", I, P, D)
    f("csdr_downconvert_set_frequency", I, P, D)
    f("csdr_downconvert_set_data_rate", D, P, D, D)
    f("csdr_downconvert_process", I, P, I, P, P)
    f("csdr_downconvert_get_stages", I, P, P, I)
    f("csdr_downconvert_get_nco_freq", D, P)
    f("csdr_downconvert_batch_create", P, I, I)
    f("csdr_downconvert_batch_destroy", None, P)
    f("csdr_downconvert_batch_set_cw_offset", I, P, I, D)
    f("csdr_downconvert_batch_set_frequency", I, P, I, D)
    f("csdr_downconvert_batch_set_data_rate", D, P, I, D, D)
    f("csdr_downconvert_batch_get_stages", I, P, I, P, I)
    f("csdr_downconvert_batch_get_nco_freq", D, P, I)
    f("csdr_downconvert_batch_out_count", I, P, I, I)
    f("csdr_downconvert_batch_process", I, P, P, LL, I, P, LL, P)


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
