#!/bin/bash
# CPU sanitizer leg (`make test-asan`): AddressSanitizer + UndefinedBehaviorSanitizer over everything of this
# repository that is plain C / host C++ -- the oracle (oracle/tcsfm_oracle.c, both precisions) under its own pinning tests, the
# host SE(3) routines of the C ABI (csrc/se3_math.h via tests/asan/se3_host.cpp), and examples/c_caller.c linked against the
# real library (without a GPU it exercises the error path).  GPU-side sanitizers are not available on this pool.
#   bash scripts/run_asan.sh [report-file]
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$ROOT/profiles/r05_asan.txt}
cd "$ROOT"
LIBASAN=$(gcc -print-file-name=libasan.so)
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
{
  echo "# CPU sanitizer leg: $(date -u +%Y-%m-%dT%H:%MZ)  gcc $(gcc -dumpversion)  flags: $SAN"
  fail=0
  echo "## 1. oracle/tcsfm_oracle.c (float64 + float32) under its pinning and replay tests"
  make -C oracle -s asan || fail=1
  LD_PRELOAD=$LIBASAN TCSFM_ORACLE_BUILD_DIR=$ROOT/oracle/_build_asan python -m pytest -q -p no:cacheprovider \
      tests/test_oracle_vs_golden.py tests/test_oracle_replay_cpu.py tests/test_posenet_cpu.py tests/test_solver_independent_cpu.py 2>&1 | tail -4
  [ ${PIPESTATUS[0]} -eq 0 ] || fail=1
  echo "## 2. host SE(3) routines of the C ABI (se3_math.h, host path) -- scipy pin + edge cases"
  g++ $SAN -std=c++17 -fPIC -shared tests/asan/se3_host.cpp -o oracle/_build_asan/libtcsfm_se3_host.so -lm || fail=1
  LD_PRELOAD=$LIBASAN TCSFM_SE3_HOST_LIB=$ROOT/oracle/_build_asan/libtcsfm_se3_host.so python - <<'PY' || fail=1
import ctypes as C, numpy as np
lib = C.CDLL(__import__("os").environ["TCSFM_SE3_HOST_LIB"])
P = lambda a: a.ctypes.data_as(C.c_void_p)
rng = np.random.default_rng(0)
cases = list(rng.normal(size=(1000, 6))) + [np.zeros(6), np.array([0, 0, 0, np.pi, 0, 0.0]), np.array([1, 2, 3, 0, 0, np.pi - 1e-9]),
                                             np.full(6, np.nan), np.full(6, 1e300), np.array([0, 0, 0, 1e-300, 0, 0.0])]
n = 0
for xi in cases:
    xi = np.ascontiguousarray(xi, np.float64); T = np.zeros(12); U = np.zeros(12); V = np.zeros(12); back = np.zeros(6); pose = np.zeros(6)
    lib.tcsfm_se3_exp(P(xi), P(T)); lib.tcsfm_se3_log(P(T), P(back)); lib.tcsfm_se3_inv(P(T), P(U)); lib.tcsfm_se3_mul(P(T), P(U), P(V))
    lib.tcsfm_pose_to_matrix(P(xi), P(T)); lib.tcsfm_matrix_to_pose(P(T), P(pose))
    if np.all(np.isfinite(xi)) and np.abs(xi).max() < 10:
        assert np.allclose(V.reshape(3, 4), np.eye(3, 4), atol=1e-9) and np.allclose(pose, xi, atol=1e-9) or np.abs(xi[3:]).max() > 1.5
    n += 1
print(f"{n} twists / poses through all six entry points: clean")
PY
  echo "## 3. examples/c_caller.c (C99, -fsanitize) against the real library; no GPU here -> the library's own error path"
  gcc -std=c99 $SAN -Wall -Wextra -pedantic -Iinclude examples/c_caller.c -Ltightly_coupled_sfm_amd -ltcsfm_hip -Wl,-rpath,$ROOT/tightly_coupled_sfm_amd -lm -o /tmp/c_caller_asan || fail=1
  /tmp/c_caller_asan > /tmp/c_caller_asan.out 2>&1; rc=$?
  tail -2 /tmp/c_caller_asan.out
  if grep -q "ERROR: AddressSanitizer\|runtime error" /tmp/c_caller_asan.out; then fail=1; fi
  echo "c_caller exit code $rc (1 = tcsfm_create refused: no GPU; 0 = ran on a GPU)"
  echo "## result: $([ $fail -eq 0 ] && echo CLEAN || echo FINDINGS)"
  exit $fail
} 2>&1 | tee "$OUT"
exit ${PIPESTATUS[0]}
