"""CPU container: the C ABI's HOST side under AddressSanitizer + UndefinedBehaviorSanitizer ("Race detection /
sanitizers").  tests/asan/build_and_run.sh rebuilds libadfp with the host translation instrumented (-fsanitize=address,undefined
-fno-gpu-sanitize; device ASan needs XNACK, which the pool refuses) and runs tests/asan/host_driver.cpp, which walks every entry
point of include/adfp.h through its argument-error paths (each must return the negative code adfp.h promises, before any
launch) and through its accepting host logic with opaque device pointers (workspace carve-up, job tables, kernel-argument
structs).  No GPU: a launch then fails with a positive hipError_t, a legal return.  A sanitizer report aborts the driver."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_host_side_is_clean_under_asan_and_ubsan(tmp_path):
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    clang = '/opt/rocm/lib/llvm/bin/clang++'
    if not (os.path.exists(hipcc) and os.path.exists(clang)):
        pytest.skip('no hipcc / clang++ on this box')
    if os.path.exists('/dev/kfd'):
        pytest.skip('a GPU is present: the driver hands the launches opaque (invalid) device pointers and is meant for the CPU container')
    r = subprocess.run(['bash', os.path.join(ROOT, 'tests', 'asan', 'build_and_run.sh'), str(tmp_path)], capture_output=True, text=True,
                       env=dict(os.environ, HIPCC=hipcc, CLANGXX=clang), timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert 'ERROR: AddressSanitizer' not in tail and 'runtime error' not in tail, tail
    assert r.returncode == 0 and 'host sanitizer driver: 0 failed expectations' in r.stdout, tail
