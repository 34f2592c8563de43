"""The CPU checker under AddressSanitizer + UndefinedBehaviorSanitizer (gcc; sanitizers exist for the CPU build only):
`oracle/lsqr_oracle.c` and `lstp_oracle.c` compiled with -fsanitize=address,undefined -fno-sanitize-recover=all
(oracle/Makefile `sanitize`) and the whole of tests/test_oracle_golden.py -- every golden vector, the live reference bit
for bit where oracle/_ref holds it -- run through that build in a child interpreter with libasan preloaded.  A read past
an array or a signed overflow in the restatement aborts the child."""
import os
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def test_oracle_goldens_under_asan_ubsan():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "sanitize"], check=True)
    lib = os.path.join(ROOT, "oracle", "_ref", "liblsqr_oracle_san.so")
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], check=True, capture_output=True, text=True).stdout.strip()
    assert os.path.isabs(asan) and os.path.exists(asan), "gcc has no libasan here"
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", LSQR_ORACLE_LIB=lib)
    # the child really runs the instrumented library
    r = subprocess.run([sys.executable, "-c", "import oracle; oracle.port(); "
                        "print(sorted({l.split()[-1] for l in open('/proc/self/maps') if 'liblsqr_oracle' in l}))"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.strip() == repr([os.path.realpath(lib)]), r.stdout
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join("tests", "test_oracle_golden.py"), "-x", "-q",
                        "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    assert " passed" in r.stdout and "failed" not in r.stdout
