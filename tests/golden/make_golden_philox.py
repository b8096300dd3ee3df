"""Golden vectors of the engine's Gaussian stream (tests/golden/philox_stream.npz), generated from oracle/philox.py.

Run in the build container:  python tests/golden/make_golden_philox.py
Data only: for two (seed, offset) pairs the first 4096 uint32 words of the Philox4x32-10 stream and the first 4096
N(0,1) numbers (f64), plus Random123's three published known-answer vectors for philox4x32-10.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import philox as ph  # noqa: E402

PAIRS = [(7, 0), (0xDEADBEEF12345678, 123457)]  # (seed, offset): offset odd-ish on purpose (mid-block start)


def main():
    out = {"pairs": np.array(PAIRS, dtype=np.uint64)}
    for i, (seed, off) in enumerate(PAIRS):
        out[f"words_{i}"] = ph.words(seed, off, 4096)
        out[f"normals_{i}"] = ph.normals(seed, off, 4096)
    out["kat_ctr"] = np.array([c for c, _, _ in ph.KAT], dtype=np.uint32)
    out["kat_key"] = np.array([k for _, k, _ in ph.KAT], dtype=np.uint32)
    out["kat_out"] = np.array([o for _, _, o in ph.KAT], dtype=np.uint32)
    path = os.path.join(HERE, "philox_stream.npz")
    np.savez_compressed(path, **out)
    print(f"philox_stream.npz: {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
