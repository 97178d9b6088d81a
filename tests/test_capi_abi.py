"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads, exports every
symbol include/cutesdr_mi.h declares, and fails loudly (no CPU fallback) without a GPU."""
import ctypes
import os
import pytest


@pytest.fixture(scope="module")
def built():
    from cutesdr_amd import _build
    return _build.build()


def test_library_exports_every_declared_symbol(built):
    from cutesdr_amd import _capi
    L = ctypes.CDLL(built)
    names = _capi.declared_symbols()
    assert len(names) >= 18
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_python_binding_covers_header(built):
    from cutesdr_amd import _capi
    L = _capi.lib()
    for n in _capi.declared_symbols():
        fn = getattr(L, n)
        assert fn.argtypes is not None, "binding for %s not declared in _capi._declare" % n


def test_no_cpu_fallback_without_gpu(built):
    from cutesdr_amd import _capi
    L = _capi.lib()
    if L.csdr_device_count() > 0:
        pytest.skip("GPU present")
    assert not L.csdr_fastfir_batch_create(0, 4, 16384)
    assert b"no HIP device" in L.csdr_last_error()
    import cutesdr_amd
    with pytest.raises(_capi.CsdrError):
        cutesdr_amd.CFastFIR(2048)


def test_product_does_not_touch_oracle():
    root = os.path.join(os.path.dirname(__file__), "..", "cutesdr_amd")
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "oracle" not in txt.replace("no oracle", ""), os.path.join(dp, f)
