"""AddressSanitizer + UndefinedBehaviorSanitizer builds of the CPU-side code (SURVEY.md 5 'sanitizers'; VERDICT r02
item 7): the oracle (`make -C oracle asan`, oracle/asan_driver.c) and the C++ host layer (host_cpu_test.cpp).  CPU
only - sanitizers never run on the GPU box.  The class of bug they prove absent is the reference's
`new float(n)` for `new float[n]` (Evolutionary_Strategy.hpp:236-244) and its loops over populationSize."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "survival_of_the_synthesis-gpu_accelerated_frequency_modulation_parameter_matcher_amd", "host")
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


def test_oracle_is_clean_under_asan_and_ubsan_and_unchanged_by_them():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    san = subprocess.run([os.path.join(ROOT, "oracle", "_san", "oracle_asan")], capture_output=True, text=True, env=ENV, timeout=600)
    assert san.returncode == 0, san.stderr[-4000:]
    assert "runtime error" not in san.stderr and "AddressSanitizer" not in san.stderr
    plain = subprocess.run([os.path.join(ROOT, "oracle", "_san", "oracle_plain")], capture_output=True, text=True, timeout=600)
    assert plain.returncode == 0
    assert san.stdout == plain.stdout  # -O1 with sanitizers and -O2 without: the same bits (-ffp-contract=off)
    lines = san.stdout.strip().splitlines()
    assert len(lines) == 8 and lines[-1].split()[3] == "1" and float(lines[-1].split()[5]) == 0.0


def test_host_layer_is_clean_under_asan_and_ubsan(tmp_path):
    exe = tmp_path / "host_cpu_test_san"
    subprocess.check_call(["g++", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wextra", "-Wno-unused-parameter", *SAN,
                           "-o", str(exe), os.path.join(HOST, "host_cpu_test.cpp")])
    out = subprocess.run([str(exe), str(tmp_path)], capture_output=True, text=True, env=ENV, timeout=300)
    assert out.returncode == 0, out.stderr[-4000:]
    assert "runtime error" not in out.stderr and "AddressSanitizer" not in out.stderr
    assert "counts 3 0 total 9.000" in out.stdout


def test_island_group_host_synchronisation_is_clean_under_thread_sanitizer(tmp_path):
    """csrc/sots_host_sync.h (the job gate and the spinning barrier of the island group's persistent threads) driven
    the way sots_group.hip drives it - 2, 3 and 8 "islands", 300 jobs each, plain data handed across the primitives -
    under ThreadSanitizer (SURVEY.md 5 'race detection'; the GPU side of the group cannot run sanitised)."""
    exe = tmp_path / "host_sync_tsan"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread", "-Wall", "-Wextra",
                           "-o", str(exe), os.path.join(ROOT, "tests", "host_sync_tsan.cpp")])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1:second_deadlock_stack=1"))
    assert out.returncode == 0, (out.stdout + out.stderr)[-4000:]
    assert "WARNING: ThreadSanitizer" not in out.stderr
    assert out.stdout.count(" ok: 900 generations") == 3
