#!/bin/bash
# CPU sanitizer runs of the oracle and of the host layer (SURVEY.md s.5 "race detection"; no GPU involved).  Usage: bash scripts/run_sanitizers.sh [logfile]
# oracle: san_check.c over every stage + the pipeline entry point on four threads.  host: test_png (PNG decoder), test_pnp (PnPSolver / pnp_core.h on the
# committed golden case files), test_threads (PoseGraph + Mapper::viewer + a polling thread; device calls -> san_stub_device.cpp).
# Every step's OWN exit status decides (no status of a `tail` / `grep` behind a pipe): a failed build, a missing binary, a non-zero exit or a sanitizer
# report anywhere in the full, un-tailed output makes the run fail; the log keeps the complete output of failing steps and the tail of passing ones.
set -u
set -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
LOG=${1:-$ROOT/profiles/r06_sanitizers.log}
HOST=$ROOT/semantic_slam_mapping_amd/host
TMP=$(mktemp -d)
export SSM_ROOT=$ROOT
python3 - "$TMP" <<'PY' || { echo "run_sanitizers: could not prepare the test inputs"; exit 2; }
import sys, os, struct, numpy as np
d = sys.argv[1]
root = os.environ.get("SSM_ROOT")
g = np.load(os.path.join(root, "tests", "golden", "pnp.npz"))
cam = (318.6, 255.3, 517.3, 516.5, 1000.0)
for name in ("outliers", "nodepth", "lanes"):
    img, obj, T0 = g[name + "_img"], g[name + "_obj"], g[name + "_T0"]
    with open(os.path.join(d, name + ".bin"), "wb") as f:
        f.write(struct.pack("<ii", len(img), 0) + np.asarray(cam, "<f8").tobytes() + np.ascontiguousarray(np.asarray(T0, "<f8").T).tobytes()
                + np.ascontiguousarray(img, "<f4").tobytes() + np.ascontiguousarray(obj, "<f4").tobytes())
sys.path.insert(0, os.path.join(root, "tests"))
from test_host_cpp import _write_png_set
os.makedirs(os.path.join(d, "png"), exist_ok=True)
_write_png_set(os.path.join(d, "png"))
PY
fail=0
MARK="ThreadSanitizer|ERROR: AddressSanitizer|runtime error:|LeakSanitizer|AddressSanitizer:|UndefinedBehaviorSanitizer"
# step <label> <command...>: runs the command, judges it by its own status and by sanitizer markers in its whole output
step() {
  local label=$1; shift
  local out rc
  out=$("$@" 2>&1); rc=$?
  if [ $rc -ne 0 ] || echo "$out" | grep -qE "$MARK"; then
    echo "### $label: FAILED (exit $rc)"; echo "$out"; fail=1
  else
    echo "### $label: ok"; echo "$out" | grep -v "points in global map\|Mapping cost time\|Map saved" | tail -3
  fi
}
{
  echo "# sanitizer runs $(date -u +%Y-%m-%dT%H:%MZ)  gcc $(gcc -dumpversion)"
  for s in asan ubsan tsan; do
    echo "## oracle SAN=$s"
    step "oracle build + run ($s)" make -s -C "$ROOT/oracle" SAN=$s san
    echo "## host SAN=$s"
    step "host build ($s)" make -s -C "$HOST" SAN=$s san
    for b in test_png_$s test_pnp_$s test_threads_$s; do [ -x "$HOST/$b" ] || { echo "### $b: MISSING BINARY"; fail=1; }; done
    [ -x "$HOST/test_png_$s" ] && step "test_png_$s" "$HOST/test_png_$s" "$TMP/png"
    for c in outliers nodepth lanes; do
      [ -x "$HOST/test_pnp_$s" ] && step "test_pnp_$s $c" "$HOST/test_pnp_$s" "$HOST/parameters_test.txt" "$TMP/$c.bin" "$TMP/$c.out"
    done
    [ -x "$HOST/test_threads_$s" ] && step "test_threads_$s" "$HOST/test_threads_$s" "$HOST/parameters_test.txt"
  done
  echo "## result: $([ $fail = 0 ] && echo CLEAN || echo FINDINGS)"
} > "$LOG" 2>&1
rm -rf "$TMP"
tail -3 "$LOG"
grep -q "^## result: CLEAN" "$LOG"
