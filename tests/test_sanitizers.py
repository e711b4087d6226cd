"""The HOST runtime under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY.md section 5 planned it; round-2 verdict: the loader-robustness
tests ran unsanitized).  `make -C parakeet.cpp_amd/csrc asan` compiles every host translation unit (C ABI, engine, safetensors / WAV / text
parsers, streaming, Sortformer, transformer) with -fsanitize=address,undefined into libparakeet_amd_asan.so (the kernel objects are the regular
ones); the suites that feed the library hostile bytes -- truncated / corrupt safetensors, malformed WAV headers, bad arguments -- then run in
a child interpreter with that library (PK_LIB) and the ASan runtime preloaded.  Any report aborts the child: the test fails with its output."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

CSRC = os.path.join(ROOT, "parakeet.cpp_amd", "csrc")
CLANG = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin", "clang++")
SUITES = ["tests/test_loader_robustness.py", "tests/test_capi_exports.py", "tests/test_audio_io_api.py", "tests/test_text_vs_reference.py",
          "tests/test_audio_vs_reference.py"]


@pytest.mark.skipif(not os.path.exists(CLANG), reason="no clang++ with sanitizer runtimes on this host")
def test_host_runtime_under_asan_ubsan():
    subprocess.run(["make", "-C", CSRC, "-j", "8", "asan"], check=True, capture_output=True, timeout=1200)
    lib = os.path.join(ROOT, "parakeet.cpp_amd", "libparakeet_amd_asan.so")
    rt = subprocess.run([CLANG, "-print-file-name=libclang_rt.asan-x86_64.so"], check=True, capture_output=True, text=True).stdout.strip()
    assert os.path.exists(lib) and os.path.exists(rt)
    env = dict(os.environ, PK_LIB=lib, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    # the instrumented library really is the one the child loads
    probe = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import pkload; pkload.load(); from parakeet_cpp_amd import capi; "
                            "capi.lib(); print([l.split()[-1] for l in open('/proc/self/maps') if 'libparakeet_amd' in l][0])" % ROOT],
                           env=env, capture_output=True, text=True, timeout=300)
    assert probe.returncode == 0 and probe.stdout.strip().endswith("libparakeet_amd_asan.so"), probe.stdout + probe.stderr
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + SUITES, cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=1500)
    assert out.returncode == 0, (out.stdout + out.stderr)[-4000:]
    assert "ERROR: AddressSanitizer" not in out.stdout + out.stderr and "runtime error:" not in out.stdout + out.stderr
