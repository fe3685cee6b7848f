"""Host-side code of libkdbhip (FASTQ/FASTA splitter, .kdb row writer) under AddressSanitizer + UBSan on the CPU
(the GPU pool offers no sanitizer runs): random and malformed text with exactly sized output buffers."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_parser_and_writer_are_clean_under_asan_ubsan(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "host_sanitize")
    build = subprocess.run([gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-o", exe,
                            os.path.join(ROOT, "tests/c/host_sanitize.cpp"), "-lz", "-lpthread"], capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("this g++ has no sanitizer runtime")
    assert build.returncode == 0, build.stderr
    run = subprocess.run([exe, "8000"], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "host sanitize ok" in run.stdout and "ERROR" not in run.stderr


def test_deflate_decoder_and_gzip_stream_agree_with_zlib_under_asan_ubsan(tmp_path):
    """The host feed's own DEFLATE decoder (kdb_inflate.cpp.h) and the gzip stream reader over it (what takes over gzip.open,
    kmerdb/parse.py:64-72): same bytes as zlib for every block type, resumable at any symbol, and an error exactly where zlib
    reports one on corrupted or truncated input -- all inside its arena (AddressSanitizer)."""
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "inflate_check")
    build = subprocess.run([gxx, "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-o", exe,
                            os.path.join(ROOT, "tests/c/inflate_check.cpp"), "-lz", "-lpthread"], capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("this g++ has no sanitizer runtime")
    assert build.returncode == 0, build.stderr
    run = subprocess.run([exe, "150"], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    assert " 0 bad" in run.stdout and "ERROR" not in run.stderr


@pytest.mark.parametrize("sanitizer", ["address,undefined", "thread"])
def test_kdb_row_writer_pipeline_under_sanitizers(tmp_path, sanitizer):
    """The .kdb row writer's pipeline (kdb_kdbwriter.cpp.h: chunk lengths -> member ownership -> format + row-aware deflate ->
    pwrite in parallel once the offsets are committed) against zlib's inflate on random vectors, 1..9 threads, both encoders;
    its carry-less-multiplication CRC-32 against zlib's.  Once under ASan + UBSan, once under ThreadSanitizer."""
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "writer_check")
    build = subprocess.run([gxx, "-std=c++17", "-O1", "-g", "-fsanitize=" + sanitizer, "-fno-omit-frame-pointer", "-o", exe,
                            os.path.join(ROOT, "tests/c/writer_check.cpp"), "-lz", "-lpthread"], capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("this g++ has no %s sanitizer runtime" % sanitizer)
    assert build.returncode == 0, build.stderr
    run = subprocess.run([exe, "check", "24" if sanitizer == "thread" else "50"], capture_output=True, text=True, timeout=600)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "writer check ok" in run.stdout and "ERROR" not in run.stderr and "WARNING: ThreadSanitizer" not in run.stderr
