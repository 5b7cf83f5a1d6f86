"""The device evaluates ln J with the host libm's own algorithm (csrc/glibc_log.hpp): same bits as
math.log (= libm log, what Rust's f64::ln calls) for every argument, on the GPU."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_device_log_is_host_libm_log(skl, gpu_ctx):
    assert skl.log_variant() in (0, 1)
    rng = np.random.default_rng(7)
    x = np.concatenate([
        rng.random(200_000),                                   # Jaccard values
        0.9375 + 0.13 * rng.random(200_000),                   # the |x - 1| < 1/16 branch
        np.arange(0, 4097) / 4096.0,                           # J(samebits) at sketchsize64 = 64
        rng.integers(0, 2**63, 100_000).astype(np.uint64).view(np.float64),
        np.array([0.0, 1.0, 5e-324, 2.2250738585072014e-308, np.inf, 0.9375, 1.064697265625]),
    ])
    x = x[~np.isnan(x) & (x >= 0)]
    got = skl.device_log(gpu_ctx, x)
    exp = np.array([math.log(v) if v > 0 and v != np.inf else (-np.inf if v == 0 else np.inf) for v in x])
    assert np.array_equal(got.view(np.uint64), exp.view(np.uint64))
