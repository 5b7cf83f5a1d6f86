"""Shared helpers for the test-suite."""
import os

import numpy as np

from conftest import REF_FIXTURES

FIXTURE_DBS = {
    # name: (n_samples, kmers, sketchsize64)  -- decoded from the .skm files
    "sketches1": (4, [31], 16),
    "sketches2": (4, [31], 157),
    "sketches3": (4, [21], 16),
    "legacy_db": (2, [17, 21, 25], 2),
}
FIXTURE_NAMES = ["14412_3#82.contigs_velvet.fa.gz", "14412_3#84.contigs_velvet.fa.gz", "R6.fa.gz",
                 "TIGR4.fa.gz"]


def load_fixture_bins(name):
    n, kmers, ss64 = FIXTURE_DBS[name]
    bins = np.fromfile(os.path.join(REF_FIXTURES, name + ".skd"), dtype="<u8")
    assert bins.size == n * len(kmers) * ss64 * 14
    return bins, n, kmers, ss64


def rust_f32(x):
    """Rust's `{}` for f32: shortest round-trip digits, positional."""
    return np.format_float_positional(np.float32(x), unique=True, trim="-")
