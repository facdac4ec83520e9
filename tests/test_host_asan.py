"""The host-side entry points of the C ABI (graph preparation: they allocate and index a lot) under
AddressSanitizer + UBSan + LeakSanitizer: tests/asan_host_driver.cpp runs dl_host_csr_from_edges and
dl_host_plan_build (plain, sliced, with a keep mask) over 300 random graphs.  CPU only — GPU sanitizers are not
available on the pool; the device code is exercised by the poisoned-buffer GPU tests instead."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_host_graph_preparation_is_sanitizer_clean(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    import glob
    srcs = sorted(glob.glob(os.path.join(ROOT, "disenlink_amd", "csrc", "*.hip")))
    exe = str(tmp_path / "asan_host_driver")
    cmd = [hipcc, "-O1", "-g", "-std=c++17", "--offload-arch=gfx950", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "disenlink_amd", "csrc"), *srcs,
           os.path.join(ROOT, "tests", "asan_host_driver.cpp"), "-o", exe]
    b = subprocess.run(cmd, capture_output=True, text=True)
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, env={**os.environ, "ASAN_OPTIONS": "detect_leaks=1"})
    assert r.returncode == 0 and "300 random graphs clean" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
