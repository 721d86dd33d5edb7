"""The host-side loaders (own npz reader, N3Tree host loader, blender / LLFF pose loaders, PNG writer) and the CPU oracle
under AddressSanitizer + UBSan (tools/sanitize/): valid files load, mutated ones load or are refused with
std::runtime_error, nothing else.  CPU only (GPU ASan is not available on this pool).  Found in round 3: a central-directory
name length past the end of the file was read, an empty array was read at index 0, a mapped zip member was read through
a misaligned typed pointer, and the JSON parser recursed without bound."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++ with libasan")
def test_host_loaders_and_oracle_are_clean_under_asan_ubsan():
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "sanitize", "run.sh"), "150"], capture_output=True, text=True,
                       timeout=900, env=env, cwd=ROOT)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert "host loaders: clean under ASan + UBSan" in out, out[-3000:]
    assert "runtime error" not in out and "AddressSanitizer" not in out, out[-3000:]
    last = [l for l in r.stdout.splitlines() if l.strip()][-1]  # the oracle's known-answer tests under the instrumented build
    assert " passed" in last and "failed" not in last and "error" not in last, last
