#!/usr/bin/env python3
"""One line per kernel: the order of its vector-memory operations and the waits the compiler put between them, from `hipcc -S` output.
L = load, D = buffer_load ... lds, S = store, A = atomic, SCR = scratch access (a spill), wN = s_waitcnt vmcnt(N), |B| = s_barrier, {loop dK} = a loop header of
depth K, Mn = n MFMAs in a row.  What to look for: `SCR w0` inside a loop (a spill's reload waits for everything in flight), `L ... L w0` right
behind the prefetch of a double buffer (conditional loads, or loads pending on the loop's entry edge), stores between a load and the w0 that uses it.
Usage:  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only csrc/kernels_sgbm.hip -o /tmp/k.s
        python3 scripts/vm_listing.py /tmp/k.s sgbm_sweep8ILi5ELb1 sgbm_rows8 ..."""
import re
import sys


def main():
    src = open(sys.argv[1]).read().split("\n")
    want = sys.argv[2:]
    cur, out = None, []
    for ln in src:
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur, out = m.group(1), []
            continue
        if cur is None:
            continue
        if "s_endpgm" in ln:
            if not want or any(w in cur for w in want):
                print(cur[:72], " ".join(out))
            cur = None
            continue
        t = ln.strip()
        if t.startswith("buffer_load") and " lds" in t:
            out.append("D")
        elif t.startswith(("global_load", "buffer_load", "flat_load")):
            out.append("L")
        elif t.startswith(("global_store", "buffer_store", "flat_store")):
            out.append("S")
        elif t.startswith(("global_atomic", "buffer_atomic", "flat_atomic")):
            out.append("A")
        elif t.startswith("scratch_"):
            out.append("SCR")
        elif t.startswith("s_waitcnt") and "vmcnt" in t:
            out.append("w" + re.search(r"vmcnt\((\d+)\)", t).group(1))
        elif t.startswith("s_barrier"):
            out.append("|B|")
        elif "Loop Header" in ln:
            out.append("{loop d" + re.search(r"Depth=(\d+)", ln).group(1) + "}")
        elif t.startswith("v_mfma"):
            if out and re.match(r"M\d+$", out[-1]):
                out[-1] = "M%d" % (int(out[-1][1:]) + 1)
            else:
                out.append("M1")


if __name__ == "__main__":
    main()
