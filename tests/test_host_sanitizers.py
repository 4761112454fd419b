"""The host side of the library (index builder + dictionary, read pipeline threads, normalizeAlphas' lock-free
union-find, file writers / readers) under AddressSanitizer + UBSan and under ThreadSanitizer: tools/host_sanitize.cpp,
built with g++ from the same sources the product compiles (the HIP translation units are not part of it).  No GPU."""
import os, shutil, subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_is_clean_under_asan_ubsan_and_tsan(tmp_path):
    if shutil.which("g++") is None or shutil.which("make") is None:
        pytest.skip("needs g++ and make")
    probe = subprocess.run("echo 'int main(){return 0;}' | g++ -x c++ - -fsanitize=thread -o %s/p && %s/p" % (tmp_path, tmp_path), shell=True,
        capture_output=True)
    if probe.returncode != 0:
        pytest.skip("this g++ has no sanitizer runtimes")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "tools"), "sanitize"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("host sanitize run ok") == 2
