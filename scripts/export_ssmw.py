#!/usr/bin/env python3
"""Writes a SegNet weight file for include/ssm/segnet.h (Classifier) from a list of (weight[Cout][Cin][3][3], scale[Cout],
shift[Cout]) numpy triples.  `python scripts/export_ssmw.py out.ssmw [seed]` writes the seeded test weights."""
import struct
import sys
import numpy as np


def write_ssmw(path, layers):
    with open(path, "wb") as f:
        f.write(b"SSMW" + struct.pack("<II", 1, len(layers)))
        for w, sc, sh in layers:
            cout, cin = w.shape[:2]
            f.write(struct.pack("<II", cin, cout))
            f.write(np.ascontiguousarray(w, "<f4").tobytes() + np.ascontiguousarray(sc, "<f4").tobytes() + np.ascontiguousarray(sh, "<f4").tobytes())


if __name__ == "__main__":
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from semantic_slam_mapping_amd.segnet_model import make_weights
    write_ssmw(sys.argv[1], make_weights(int(sys.argv[2]) if len(sys.argv) > 2 else 1234))
