"""The drop-in boundary is a C ABI: a PURE C99 program (examples/c_host/plangen_c_smoke.c -- no Python, no torch, no C++) includes
include/plangen_hip.h, links libplangen_hip.so and drives create -> load_tensor -> finalize -> prefill -> the whole CFG decode loop
-> VQ decode.  CPU: the header is valid C and every entry point the program uses links.  GPU: the program runs and its tokens are
deterministic, in range and non-trivial."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "examples", "c_host", "plangen_c_smoke.c")
EXE = os.path.join(ROOT, "examples", "c_host", "plangen_c_smoke")


def build_c_host():
    lib = os.path.join(ROOT, "plangen_amd", "lib")
    cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include", SRC,
           "-L", lib, "-lplangen_hip", "-L", "/opt/rocm/lib", "-lamdhip64", "-lm", f"-Wl,-rpath,{lib}", "-Wl,-rpath,/opt/rocm/lib", "-o", EXE]
    return subprocess.run(cmd, capture_output=True, text=True)


@pytest.mark.skipif(not shutil.which("gcc") or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"), reason="gcc / ROCm headers not available")
def test_header_is_valid_c_and_the_c_host_links():
    hdr = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", "plangen_hip.h")],
                         capture_output=True, text=True)
    assert hdr.returncode == 0, hdr.stderr
    p = build_c_host()
    assert p.returncode == 0, p.stderr
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_c_host_generates_deterministic_tokens():
    if not os.path.exists(EXE):
        p = build_c_host()
        assert p.returncode == 0, p.stderr
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)
    p = subprocess.run([EXE], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, (p.stdout[-500:], p.stderr[-1500:])
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    print(out)
    assert out["c_host"] == "ok" and out["mismatch_or_out_of_range"] == 0 and out["nonfinite_pixels"] == 0 and out["token_changes"] > 4
