"""bench.py's arithmetic (no GPU): the roofline constants are the ones DESIGN.md and the judge use."""
import importlib.util
import os

import numpy as np

from conftest import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_roofline_constants():
    b = _bench()
    assert b.algorithmic_bytes_per_pair(5, 64, 2) == 71688          # SURVEY 8(d), cfg 2/3 core/acc
    assert b.algorithmic_bytes_per_pair(5, 32, 2) == 35848          # cfg 4/5
    assert b.issue_slots_per_pair(5, 64) == 10240
    assert abs(b.VALU_PEAK_LANE_OPS - 7.8643e13) < 1e10             # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz
    blk = b.valu_block(499500, 0.16e-3, 5, 64, "cfg2")
    assert abs(blk["frac"] - (10240 * 499500 / 0.16e-3) / b.VALU_PEAK_LANE_OPS) < 1e-12
    assert abs(blk["peak_pairs_per_s"] - 7.68e9) < 1e7
    assert b.n_for_pairs(499500) == 1000


def test_condensed_index_matches_the_reference_formula():
    b = _bench()
    n = 37
    flat = [b.cond_index(i, j, n) for i in range(n) for j in range(i + 1, n)]
    assert flat == list(range(n * (n - 1) // 2))
    ii, jj = np.array([0, 5, 35]), np.array([1, 9, 36])
    assert b.cond_index(ii, jj, n).tolist() == [0, b.cond_index(5, 9, n), n * (n - 1) // 2 - 1]
