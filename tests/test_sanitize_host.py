"""The CPU-side sanitizer sweep of the product's host code (tools/sanitize_host.sh: AddressSanitizer + UBSan over the
library's host halves under the host-logic and ABI tests; ThreadSanitizer and ASan over the sound sink's two-thread
queue).  Opt-in -- CSDR_RUN_SANITIZERS=1 -- because the instrumented build of every source takes a minute on eight
cores; the log of the last run is committed as profiles/r04_sanitize_host.txt."""
import os
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.environ.get("CSDR_RUN_SANITIZERS"), reason="set CSDR_RUN_SANITIZERS=1 (builds an instrumented library)")
def test_host_code_is_clean_under_asan_ubsan_tsan():
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "sanitize_host.sh")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       env={k: v for k, v in os.environ.items() if k != "CSDR_RUN_SANITIZERS"}, timeout=1800)
    text = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and "sanitize_host: clean" in text, text[-4000:]


def test_the_committed_sanitizer_log_says_clean():
    text = open(os.path.join(ROOT, "profiles", "r04_sanitize_host.txt")).read()
    assert "sanitize_host: clean" in text and "soundsink threads ok" in text
