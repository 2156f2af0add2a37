"""AddressSanitizer + UBSan build of the host-only C-ABI entry points, and a ThreadSanitizer run of their thread pools (the one optional aux item).

``cf_load_npy_int16`` parses file headers nobody vouches for, ``cf_stat_files`` fills a size table from several threads and ``cf_chunks_from_spans`` / ``cf_chunks_json`` write into
caller-sized buffers (catfish_amd/csrc/loader_host.hpp, chunks_host.hpp: plain host code, no device).  They are compiled here
on their own (tests/native/host_entry_shim.cpp) with ``g++ -fsanitize=address,undefined`` and fuzzed with hypothesis in a child
process that preloads the sanitizer runtime (tests/native/fuzz_host_entries.py).  CPU only: GPU sanitizers are not available on
this pool.
"""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    path = subprocess.run(["gcc", "-print-file-name=" + name], stdout=subprocess.PIPE, universal_newlines=True).stdout.strip()
    return path if os.path.isabs(path) and os.path.exists(path) else None


@pytest.mark.timeout(900)
def test_host_entry_points_under_asan_and_ubsan(tmp_path):
    if shutil.which("g++") is None or _runtime("libasan.so") is None:
        pytest.skip("g++ with the AddressSanitizer runtime is not installed")
    lib = str(tmp_path / "libcatfish_host_asan.so")
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                            "-shared", "-fPIC", "-o", lib, os.path.join(ROOT, "tests", "native", "host_entry_shim.cpp"), "-lpthread"],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    assert build.returncode == 0, build.stdout
    scratch = tmp_path / "scratch"
    scratch.mkdir()
    env = dict(os.environ, LD_PRELOAD=_runtime("libasan.so"), ASAN_OPTIONS="detect_leaks=0:abort_on_error=0",
               UBSAN_OPTIONS="print_stacktrace=1", FUZZ_EXAMPLES=os.environ.get("FUZZ_EXAMPLES", "150"))
    run = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "native", "fuzz_host_entries.py"), lib, str(scratch)],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, env=env, timeout=800)
    assert run.returncode == 0 and run.stdout.strip().endswith("fuzz ok"), run.stdout[-4000:]


@pytest.mark.timeout(600)
def test_host_thread_pools_under_thread_sanitizer(tmp_path):
    """SURVEY section 5 lists race detection among the aux items: the host entry points that run thread pools (``cf_load_npy_int16``,
    ``cf_stat_files``, ``cf_listing_sizes``, ``cf_listing_load_npy_int16``) and two independent listings used from two caller threads
    at once, built with ``g++ -fsanitize=thread`` as a standalone program (tests/native/tsan_driver.cpp: 2 x 240 files, 1 .. 8 pool
    threads, the error path with several offending files) -- no report, right answers."""
    if shutil.which("g++") is None or _runtime("libtsan.so") is None:
        pytest.skip("g++ with the ThreadSanitizer runtime is not installed")
    exe = str(tmp_path / "tsan_driver")
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-o", exe,
                            os.path.join(ROOT, "tests", "native", "tsan_driver.cpp"), "-lpthread"],
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    assert build.returncode == 0, build.stdout
    scratch = tmp_path / "scratch"
    scratch.mkdir()
    run = subprocess.run([exe, str(scratch)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True, timeout=500,
                         env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0:report_signal_unsafe=0"))
    if "FATAL: ThreadSanitizer: unexpected memory mapping" in run.stdout:
        pytest.skip("ThreadSanitizer cannot map its shadow memory in this container")
    assert run.returncode == 0 and run.stdout.strip().endswith("tsan ok"), run.stdout[-4000:]
    assert "ThreadSanitizer" not in run.stdout, run.stdout[-4000:]
