"""Finds GLOBAL ROWS of the in-kernel uniform stream (csrc/aux.hpp noise_vals, restated in oracle.noise) whose K = 10 uniforms
contain the stream's smallest value (bits >> 8 == 0 -> u = tiny after the clamp) or its largest (u = 1 - 2^-24), for a given
(seed, step).  The one-launch kernels draw their noise themselves, so the saturated-regime parity cases
(tests/test_saturated.py) reach those two values by choosing the batch's row offset (GmvaeDims.row0) -- the Philox counter
holds the global row -- instead of injecting noise.  Each kind occurs once per 2^24 uniforms; 2^22 rows x 10 give ~2.5 of each.

    python tools/find_extreme_u.py [seed] [step] [log2 rows]
"""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import oracle as O  # noqa: E402


def search(seed: int, step: int, rows: int, K: int = 10, chunk: int = 1 << 18):
    qpr = (K + 3) // 4
    lo, hi = [], []
    for r0 in range(0, rows, chunk):
        row = (np.arange(chunk, dtype=np.uint64) + np.uint64(r0))[:, None].repeat(qpr, 1)
        quad = np.arange(qpr, dtype=np.uint64)[None, :].repeat(chunk, 0)
        c1 = (quad & np.uint64(0x00FFFFFF)) | (((row >> np.uint64(32)) & np.uint64(0x3F)) << np.uint64(24)) | np.uint64(0x80000000)
        c = np.stack([row & np.uint64(0xFFFFFFFF), c1, np.full_like(row, step & 0xFFFFFFFF),
                      np.full_like(row, (step >> 32) & 0xFFFFFFFF)], axis=-1).astype(np.uint32)
        b = (O.philox4x32_10(c, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF) >> np.uint32(8)).reshape(chunk, -1)[:, :K]
        for r, k in zip(*np.nonzero(b == 0)):
            lo.append((r0 + int(r), int(k)))
        for r, k in zip(*np.nonzero(b == 0xFFFFFF)):
            hi.append((r0 + int(r), int(k)))
    return lo, hi


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 11
    step = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    lg = int(sys.argv[3]) if len(sys.argv) > 3 else 22
    lo, hi = search(seed, step, 1 << lg)
    print(f"seed {seed} step {step}: u = tiny at (row, k) {lo}; u = 1 - 2^-24 at (row, k) {hi}")
