"""The checker itself under AddressSanitizer + UndefinedBehaviorSanitizer (the reference has no race
or memory tooling -- it relies on Rust; the C oracle and the host builds of the device headers do not have that luxury).
  * oracle/dfx_oracle.c built with -fsanitize=address,undefined runs its own golden-vector and CSV tests;
  * the host builds of csrc/dfx_numparse.hpp and csrc/dfx_csv_walk.hpp run their fuzzers under the same sanitizers."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]


def test_oracle_golden_and_csv_tests_under_asan_ubsan(tmp_path):
    so = str(tmp_path / "libdfx_oracle_san.so")
    subprocess.check_call(["gcc"] + SAN + ["-shared", "-fPIC", "-o", so, os.path.join(ROOT, "oracle", "dfx_oracle.c"), "-lm"])
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0", DFX_ORACLE_SO=so)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_oracle_golden.py"),
                        os.path.join(ROOT, "tests", "test_csv_oracle.py") + "::test_oracle_csv_matches_independent_reader",
                        os.path.join(ROOT, "tests", "test_csv_oracle.py") + "::test_oracle_csv_quoting_rules",
                        os.path.join(ROOT, "tests", "test_csv_oracle.py") + "::test_oracle_csv_errors"],
                       env=env, capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "passed" in r.stdout


def test_host_builds_of_device_headers_under_asan_ubsan(tmp_path):
    obj = str(tmp_path / "oracle_san.o")
    subprocess.check_call(["gcc"] + SAN + ["-c", os.path.join(ROOT, "oracle", "dfx_oracle.c"), "-o", obj])
    walk = str(tmp_path / "csv_walk_fuzz_san")
    subprocess.check_call(["g++"] + SAN + ["-std=c++17", "-o", walk, os.path.join(ROOT, "tests", "native", "csv_walk_fuzz.cpp"), obj, "-lm"])
    r = subprocess.run([walk, "600", "21", str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("ok:"), r.stdout + r.stderr
    num = str(tmp_path / "numparse_fuzz_san")
    subprocess.check_call(["g++"] + SAN + ["-std=c++17", "-o", num, os.path.join(ROOT, "tests", "native", "numparse_fuzz.cpp")])
    r = subprocess.run([num, "150000", "22"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("ok:"), r.stdout + r.stderr
    # the whole-file emulation of the device CSV source over the fixture with every column type
    chk = str(tmp_path / "csv_fixture_check_san")
    subprocess.check_call(["g++"] + SAN + ["-std=c++17", "-o", chk, os.path.join(ROOT, "tests", "native", "csv_fixture_check.cpp"), obj, "-lm"])
    r = subprocess.run([chk, os.path.join(ROOT, "tests", "data", "all_types_gen.csv"), "1,6,7,8,9,2,3,4,5,10,11,12"],
                       capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
    assert r.returncode == 0 and r.stdout.startswith("ok:"), r.stdout + r.stderr
