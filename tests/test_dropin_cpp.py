"""The drop-in boundary as a C++ consumer sees it: tests/cpp/dropin_host.cpp uses the classes the
way the reference host does (by-value members, 256-sample calls, same method names/signatures) and
is compiled with plain g++ against cutesdr_amd/dropin + libcutesdr_mi.so."""
import os
import subprocess
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
EXE = os.path.join(ROOT, "tests", "cpp", "dropin_host")


def build_exe():
    from cutesdr_amd import _build
    _build.build()
    src = os.path.join(ROOT, "tests", "cpp", "dropin_host.cpp")
    hdrs = [os.path.join(ROOT, "cutesdr_amd", "dropin", "dsp", f) for f in os.listdir(os.path.join(ROOT, "cutesdr_amd", "dropin", "dsp"))]
    if not os.path.exists(EXE) or any(os.path.getmtime(p) > os.path.getmtime(EXE) for p in [src] + hdrs):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "cutesdr_amd", "dropin"),
                               "-I", os.path.join(ROOT, "include"), src, "-o", EXE,
                               "-L", os.path.join(ROOT, "cutesdr_amd"), "-lcutesdr_mi",
                               "-Wl,-rpath," + os.path.join(ROOT, "cutesdr_amd")])
    return EXE


def test_dropin_headers_compile_and_link_with_gpp():
    assert os.path.exists(build_exe())


def test_dropin_degrades_to_zero_samples_without_gpu(tmp_path):
    from cutesdr_amd import _capi
    if _capi.lib().csdr_device_count() > 0:
        pytest.skip("GPU present")
    exe = build_exe()
    x = np.zeros(1024, dtype=np.complex128)
    x.tofile(tmp_path / "in.bin")
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "o"), "2", "2000000", "-100000"],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 0
    assert "no HIP device" in r.stderr                  # fails loudly, returns 0 samples like the reference's int API
    assert (tmp_path / "o.meta").read_text().split()[0] == "0"


@pytest.mark.gpu
def test_dropin_host_matches_oracle(oracle, tmp_path):
    from util_signals import fm_carrier, FULL_SCALE
    exe = build_exe()
    fs, n = 2e6, 19968 * 20
    x = fm_carrier(n, fs, 100e3, dbfs=-20.0)
    x[np.random.default_rng(8).random(n) < 4e-5] += 28000.0     # impulses for the blanker
    x = x.astype(np.complex64).astype(np.complex128)            # the device sees fp32: identical trigger decisions
    x.tofile(tmp_path / "in.bin")
    r = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "o"), "2", str(fs), "-100000"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    total, rtotal, rate, smeter, ov = (tmp_path / "o.meta").read_text().split()
    audio = np.fromfile(tmp_path / "o.audio")
    pix = np.fromfile(tmp_path / "o.spec", dtype=np.int32)
    d, f, rs = oracle.CDemodulator(2048), oracle.CFft(), oracle.CFractResampler()
    d.SetInputSampleRate(fs); d.SetDemod(oracle.DEMOD_FM, oracle.fm_defaults()); d.SetDemodFreq(-100e3)
    f.SetFFTParams(4096, False, 0.0, fs); f.SetFFTAve(1); rs.Init(8192)
    nb = oracle.CNoiseProc(); nb.SetupBlanker(True, 40.0, 10.0, fs)
    x = x.copy()
    want, wtotal, wr, fftpos = [], 0, 0, 0
    for i in range(0, n - 255, 256):
        x[i:i + 256] = nb.ProcessBlanker(x[i:i + 256])             # in place in front of FFT and chain
        if i + 256 - fftpos >= 4096:
            f.PutInDisplayFFT(x[fftpos:fftpos + 4096]); fftpos += 4096
        k, o = d.ProcessData(x[i:i + 256])
        if k:
            wtotal += k
            want.append(o[:k].copy())
            wr += len(rs.Resample(o[:k], d.GetOutputRate() / 48000.0))
    want = np.concatenate(want)
    assert int(total) == wtotal == len(audio) and int(rtotal) == wr
    assert float(rate) == d.GetOutputRate()
    assert np.abs(audio[6144:] - want[6144:]).max() <= 1e-3 * FULL_SCALE
    assert float(smeter) == pytest.approx(d.GetSMeterAve(), abs=0.02)
    _, wpix = f.GetScreenIntegerFFTData(255, 700, 0.0, -160.0, -900000, 900000)
    assert np.abs(pix - wpix).max() <= 1
