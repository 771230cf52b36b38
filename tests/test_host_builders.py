"""The host-side acceleration builders (SAH BVH2 image, 4-wide BVH image, uniform grid with fat lists) compiled
WITHOUT HIP under AddressSanitizer + UBSan and run on the suzanne fixture and on cover-like sphere scenes — the
sanitizer leg for the native code that has no GPU in it (GPU sanitizers are not available on the pool)."""
import shutil
import subprocess

import pytest

from conftest import GOLDEN, REPO


def test_builders_are_clean_under_asan_and_ubsan(tmp_path):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    exe = tmp_path / "builders_sanitize"
    subprocess.run([gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                    "-fno-omit-frame-pointer", str(REPO / "tests" / "tools" / "builders_sanitize.cpp"), "-o", str(exe)],
                   check=True, capture_output=True)
    r = subprocess.run([str(exe), str(GOLDEN / "suzanne.obj")], capture_output=True, text=True,
                       env={"ASAN_OPTIONS": "detect_leaks=1"})
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
    assert "thread-start-failure builds give one image" in r.stdout
    assert "binary16 directed rounding: all 63488 finite values" in r.stdout
    assert "half nodes" in r.stdout and "planes moved outwards by at most" in r.stdout
    assert "968 triangles" in r.stdout and "fat entries of 80 bytes" in r.stdout and "fat entries of 48 bytes" in r.stdout
