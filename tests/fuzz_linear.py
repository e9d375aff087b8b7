"""One-off fuzz of the dense entry points (test infrastructure, not collected by pytest): random shapes, with an emphasis on
extents that are multiples of 4 but not of the tile (the clamped fast path of M-contiguous operands), both precisions.
    python tests/fuzz_linear.py [n_shapes] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gdmcf_amd import _lib  # noqa: E402
from tests.test_gpu_parity import _linear_entry_points_random_shapes  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
shapes = []
for _ in range(n):
    M = int(rng.choice([rng.integers(1, 40), rng.integers(40, 520)]))
    N = int(rng.choice([rng.integers(1, 300), rng.integers(300, 2600), 4 * rng.integers(1, 400)]))
    K = int(rng.choice([rng.integers(1, 200), rng.integers(200, 6000), 4 * rng.integers(1, 900)]))
    shapes.append((M, N, K))
lib = _lib.load()
for prec in ("f32", "bf16"):
    prev = lib.gdmcf_gemm_precision(0 if prec == "f32" else 1)
    try:
        for i in range(0, n, 10):
            _linear_entry_points_random_shapes(lib, prec, shapes=shapes[i:i + 10], seed=seed + i)
            print(prec, "ok", i + 10, flush=True)
    finally:
        lib.gdmcf_gemm_precision(prev)
print("fuzz ok:", n, "shapes x 2 paddings x 2 precisions x 4 entry points")
