"""The host side (C++ above the C-ABI) under AddressSanitizer + UndefinedBehaviorSanitizer: `make asan` builds
libgffx_host_asan.so, a child interpreter with the sanitizer runtime preloaded drives the index builder, the loaders and the
word-at-a-time BED parser through it on files with every quirk the parser has a path for; any report fails the test.
(GPU AddressSanitizer is not available on the pool: the device side is covered by the parity tests.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes as C, os, random, shutil, sys
root, tmp = sys.argv[1], sys.argv[2]
L = C.CDLL(os.path.join(root, "gffx_amd", "lib", "libgffx_host_asan.so"))
u32p, u64p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
L.gffx_host_build_index.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_size_t]
L.gffx_host_parse_bed_file.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(u32p), u64p, C.c_char_p, C.c_size_t]
L.gffx_host_free.argtypes = [C.c_void_p]
gff = os.path.join(tmp, "a.gff")
shutil.copy(os.path.join(root, "tests", "golden", "appendix_e.gff"), gff)
err = C.create_string_buffer(2048)
assert L.gffx_host_build_index(gff.encode(), b"gene_name", b"", 0, err, len(err)) == 0, err.value
rng = random.Random(5)
names = [ln.split("\t")[0] for ln in open(gff) if ln.strip() and not ln.startswith("#")]
rows = []
for i in range(20000):
    n = rng.choice(names)
    a, b = rng.randrange(0, 10 ** rng.randrange(1, 10)), rng.randrange(0, 10 ** rng.randrange(1, 10))
    sep = rng.choice(["\t", " ", "\t\t", " \t"])
    tail = rng.choice(["", "\tx", "\t+\t9", " # c", "\r"])
    rows.append("%s%s%s%s%d%s%d%s" % (rng.choice(["", "", " "]), n, sep, rng.choice(["", "", "+"]), a, sep, b, tail))
rows += ["# comment", "", "track name=x", "browser position chr1:1-2"]
bed = os.path.join(tmp, "q.bed")
open(bed, "w").write("\n".join(rows) + rng.choice(["", "\n"]))
out, n = u32p(), C.c_uint64()
rc = L.gffx_host_parse_bed_file(gff.encode(), bed.encode(), C.byref(out), C.byref(n), err, len(err))
if rc == 0:
    L.gffx_host_free(out)
print("ok", rc, n.value)
'''


def test_host_side_under_address_and_ub_sanitizers(tmp_path):
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "gffx_amd", "csrc"), "-j8", "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    ubsan = subprocess.run(["gcc", "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan.so next to gcc")
    env = dict(os.environ, LD_PRELOAD=asan + (":" + ubsan if os.path.exists(ubsan) else ""),
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=86", UBSAN_OPTIONS="halt_on_error=1:exitcode=87")
    r = subprocess.run([sys.executable, "-c", CHILD, ROOT, str(tmp_path)], env=env, capture_output=True, text=True, timeout=600)
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0 and r.stdout.startswith("ok"), (r.returncode, r.stdout[-500:], r.stderr[-2000:])


TSAN_CHILD = r'''
import ctypes as C, os, random, shutil, sys
root, tmp = sys.argv[1], sys.argv[2]
L = C.CDLL(os.path.join(root, "gffx_amd", "lib", "libgffx_host_tsan.so"))
u32p, u64p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)
L.gffx_host_build_index.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_char_p, C.c_size_t]
L.gffx_host_shard_bed_file.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(u32p), u64p, C.c_char_p, C.c_size_t]
L.gffx_host_parse_bed_file_chunked.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint64, C.POINTER(u32p), u64p, C.c_char_p, C.c_size_t]
L.gffx_host_free.argtypes = [C.c_void_p]
gff = os.path.join(tmp, "a.gff")
shutil.copy(os.path.join(root, "tests", "golden", "appendix_e.gff"), gff)
err = C.create_string_buffer(2048)
assert L.gffx_host_build_index(gff.encode(), b"gene_name", b"", 0, err, len(err)) == 0, err.value
rng = random.Random(7)
names = sorted(set(ln.split("\t")[0] for ln in open(gff) if ln.strip() and not ln.startswith("#")))
bed = os.path.join(tmp, "q.bed")
with open(bed, "w") as f:
    for i in range(400000):
        a = rng.randrange(0, 5000000)
        f.write("%s\t%d\t%d\n" % (rng.choice(names), a, a + rng.randrange(1, 9000)))
total = 0
for threads, chunk, n_dev, keep in ((8, 1 << 20, 2, 0), (6, 3 << 19, 3, 1), (16, 1 << 21, 8, 0)):
    out, n = u32p(), (C.c_uint64 * n_dev)()
    rc = L.gffx_host_shard_bed_file(gff.encode(), bed.encode(), threads, chunk, n_dev, keep, C.byref(out), n, err, len(err))
    assert rc == 0, err.value
    total += sum(n)
    L.gffx_host_free(out)
out, nr = u32p(), C.c_uint64()
assert L.gffx_host_parse_bed_file_chunked(gff.encode(), bed.encode(), 12, 1 << 20, C.byref(out), C.byref(nr), err, len(err)) == 0
L.gffx_host_free(out)
print("ok", total, nr.value)
'''


def test_parser_pool_and_bucket_scatter_under_thread_sanitizer(tmp_path):
    """`make tsan` builds libgffx_host_tsan.so; a child interpreter with the ThreadSanitizer runtime preloaded drives what
    `gffx intersect --gpus N` runs on the host per BED chunk -- the parser thread's worker pool (persistent workers, recycled row
    buffers) and the scatter of the rows to N devices' staging buffers by chromosome bucket (16 writers into shared buffers at
    offsets the plan makes disjoint) -- without a device.  The reference's only shared state on this path is one relaxed atomic
    (commands/intersect.rs:264); any data-race report fails the test."""
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "gffx_amd", "csrc"), "-j8", "tsan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    tsan = subprocess.run(["gcc", "-print-file-name=libtsan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(tsan) or not os.path.exists(tsan):
        pytest.skip("no libtsan.so next to gcc")
    env = dict(os.environ, LD_PRELOAD=tsan, TSAN_OPTIONS="exitcode=88:report_signal_unsafe=0:history_size=4")
    r = subprocess.run([sys.executable, "-c", TSAN_CHILD, ROOT, str(tmp_path)], env=env, capture_output=True, text=True, timeout=900)
    if "unexpected memory mapping" in r.stderr or "FATAL: ThreadSanitizer" in r.stderr:
        pytest.skip("ThreadSanitizer cannot start in this container: " + r.stderr.strip().splitlines()[0][:200])
    assert "WARNING: ThreadSanitizer" not in r.stderr, r.stderr[-4000:]
    assert r.returncode == 0 and r.stdout.startswith("ok"), (r.returncode, r.stdout[-500:], r.stderr[-2000:])
