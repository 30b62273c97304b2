"""Sanitizer builds of the HOST code (CPU only - GPU sanitizers are not available on this pool).

`make -C ferromic_amd/csrc asan` builds run_vcf (text ingest, packing, writers: the restatement of process.rs:4471-4768 and the
writers) and the host packer of the upload path with -fsanitize=address,undefined -fno-sanitize-recover=all; `make -C oracle asan`
does the same for the C restatement of the oracle.  These tests build them (half a minute, once) and run the GPU-free ingest / format
/ fuzz suites against the instrumented binary, the packer's own check, and the oracle's pin tests against the instrumented library:
any out-of-bounds access, use after free, signed overflow or misaligned load ends the process and fails the test."""

import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN_DIR = os.path.join(ROOT, "ferromic_amd", "bin")
SAN_ENV = {"ASAN_OPTIONS": "detect_leaks=0:abort_on_error=0:halt_on_error=1", "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"}


@pytest.fixture(scope="module")
def asan_build():
    if not os.path.exists(os.path.join(ROOT, "ferromic_amd", "lib", "libferromic_hip.so")):
        pytest.skip("libferromic_hip.so is not built (run __graft_entry__.build())")
    res = subprocess.run(["make", "-C", os.path.join(ROOT, "ferromic_amd", "csrc"), "asan"], capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    res = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    return {"run_vcf": os.path.join(BIN_DIR, "run_vcf.asan"), "pack": os.path.join(BIN_DIR, "host_pack_check.asan"),
            "oracle": os.path.join(ROOT, "oracle", "_asan", "liboracle_dense.so")}


def test_host_packer_under_sanitizers(asan_build):
    res = subprocess.run([asan_build["pack"], "600"], capture_output=True, text=True, timeout=600, env=dict(os.environ, **SAN_ENV))
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "600 cases" in res.stdout and "ok" in res.stdout


def test_run_vcf_host_code_under_sanitizers(asan_build):
    """The ingest, format and adversarial-text suites (all GPU-free: --ingest_only / --print_formats / --bench_tracks) with the
    instrumented binary in place of bin/run_vcf: same expectations, and no sanitizer report."""
    env = dict(os.environ, FERROMIC_RUN_VCF_BIN=asan_build["run_vcf"], FERROMIC_FUZZ_INGEST_CASES="6", **SAN_ENV)
    res = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
                          "tests/test_run_vcf_ingest_cpu.py", "tests/test_run_vcf_ingest_fuzz_cpu.py", "tests/test_output_formats_cpu.py"],
                         capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stdout[-4000:] + res.stderr[-2000:]
    assert " passed" in res.stdout and "AddressSanitizer" not in res.stdout + res.stderr and "runtime error" not in res.stdout + res.stderr


def test_oracle_c_restatement_under_sanitizers(asan_build):
    """tests/test_oracle_dense_c.py against the instrumented liboracle_dense.so (threads, missing bitsets, ragged shapes)."""
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan.so not found")
    env = dict(os.environ, FERROMIC_ORACLE_LIB=asan_build["oracle"], LD_PRELOAD=libasan, **SAN_ENV)
    res = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "tests/test_oracle_dense_c.py"],
                         capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stdout[-4000:] + res.stderr[-3000:]
    assert " passed" in res.stdout and "AddressSanitizer" not in res.stdout + res.stderr and "runtime error" not in res.stdout + res.stderr
