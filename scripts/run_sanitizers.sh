#!/bin/bash
# CPU sanitizer runs of the oracle and of the host layer (SURVEY.md s.5 "race detection"; no GPU involved).  Usage: bash scripts/run_sanitizers.sh [logfile]
# oracle: san_check.c over every stage + the pipeline entry point on four threads.  host: test_png (PNG decoder), test_pnp (PnPSolver / pnp_core.h on the
# committed golden case files), test_threads (PoseGraph + Mapper::viewer + a polling thread; device calls -> san_stub_device.cpp).
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
LOG=${1:-$ROOT/profiles/r03_sanitizers.log}
HOST=$ROOT/semantic_slam_mapping_amd/host
TMP=$(mktemp -d)
export SSM_ROOT=$ROOT
python3 - "$TMP" <<'PY'
import sys, os, struct, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(sys.argv[0])), "."))
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
{
  echo "# sanitizer runs $(date -u +%Y-%m-%dT%H:%MZ)  gcc $(gcc -dumpversion)"
  for s in asan ubsan tsan; do
    echo "## oracle SAN=$s"; make -s -C "$ROOT/oracle" SAN=$s san 2>&1 | tail -3 || fail=1
    echo "## host SAN=$s"; make -s -C "$HOST" SAN=$s san 2>&1 | grep -E "error|warning" | head -5
    "$HOST/test_png_$s" "$TMP/png" 2>&1 | tail -2 || fail=1
    for c in outliers nodepth lanes; do "$HOST/test_pnp_$s" "$HOST/parameters_test.txt" "$TMP/$c.bin" "$TMP/$c.out" 2>&1 | tail -3 && echo "test_pnp_$s $c: exit 0" || { echo "test_pnp_$s $c: FAILED"; fail=1; }; done
    "$HOST/test_threads_$s" "$HOST/parameters_test.txt" 2>&1 | grep -v "points in global map\|Mapping cost time\|Map saved" | tail -6 || fail=1
  done
} > "$LOG" 2>&1
grep -qE "ThreadSanitizer: reported|ERROR: AddressSanitizer|runtime error:|LeakSanitizer|FAILED" "$LOG" && fail=1
echo "## result: $([ $fail = 0 ] && echo CLEAN || echo FINDINGS)" >> "$LOG"
rm -rf "$TMP"
tail -3 "$LOG"
exit $fail
