"""oracle/sketcher.py against the reference's committed sketches (needs the reference's
FASTA files, i.e. the build container; skipped elsewhere -- the generated fixture it
produced is committed and covered by test_oracle_golden.py)."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN, REF_FIXTURES

REF_IN = "/root/reference/tests/test_files_in"
GENOMES = ["14412_3#82.contigs_velvet.fa.gz", "14412_3#84.contigs_velvet.fa.gz", "R6.fa.gz", "TIGR4.fa.gz"]

pytestmark = pytest.mark.skipif(not os.path.isdir(REF_IN), reason="reference genomes not available here")


def test_sketcher_reproduces_committed_skd():
    from oracle import sketcher

    paths = [os.path.join(REF_IN, g) for g in GENOMES]
    got = sketcher.sketch_files(paths, [21], 1000).astype("<u8").tobytes()
    assert got == open(os.path.join(REF_FIXTURES, "sketches3.skd"), "rb").read()


def test_generated_fixture_is_current():
    from oracle import sketcher

    paths = [os.path.join(REF_IN, g) for g in GENOMES[2:]]
    got = sketcher.sketch_files(paths, [17, 21, 25, 29], 10000).astype("<u8")
    committed = np.fromfile(os.path.join(GOLDEN, "generated", "sketch_db_4k.skd"), dtype="<u8").reshape(4, -1)
    assert np.array_equal(got, committed[2:])
