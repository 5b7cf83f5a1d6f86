#!/usr/bin/env python3
"""Generates tests/golden/generated/* from the reference's test genomes.

Run in the build container (needs /root/reference; the GPU box never runs this):
    python tests/golden/make_generated_fixtures.py

1. Pins oracle/sketcher.py: re-sketches the four genomes with the parameters of the
   committed sketches{1,2,3}.skd and requires byte identity.
2. Writes sketch_db_4k.skd -- the database the reference's tests build with
   `sketchlib sketch -o sketch_db --k-seq 17,31,4 -s 10000 -f rfile.txt`
   (tests/distance.rs:270-290) and never commit; its `dist` outputs are the exact-text
   goldens dists_knn_{ca,jaccard,ani}.stdout and dists_subset.stdout.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import sketcher  # noqa: E402

REF_IN = "/root/reference/tests/test_files_in"
GENOMES = ["14412_3#82.contigs_velvet.fa.gz", "14412_3#84.contigs_velvet.fa.gz", "R6.fa.gz", "TIGR4.fa.gz"]
OUT = os.path.join(ROOT, "tests", "golden", "generated")
FIX = os.path.join(ROOT, "tests", "golden", "reference_fixtures")


def main():
    paths = [os.path.join(REF_IN, g) for g in GENOMES]
    for name, kmers, s in [("sketches1", [31], 1000), ("sketches3", [21], 1000), ("sketches2", [31], 10000)]:
        got = sketcher.sketch_files(paths, kmers, s).astype("<u8").tobytes()
        want = open(os.path.join(FIX, name + ".skd"), "rb").read()
        assert got == want, f"sketcher does not reproduce {name}.skd"
        print(f"sketcher reproduces {name}.skd bit-exactly ({len(want)} bytes)")
    os.makedirs(OUT, exist_ok=True)
    db = sketcher.sketch_files(paths, [17, 21, 25, 29], 10000).astype("<u8")
    db.tofile(os.path.join(OUT, "sketch_db_4k.skd"))
    print("wrote sketch_db_4k.skd", db.shape, hashlib.sha256(db.tobytes()).hexdigest()[:16])


if __name__ == "__main__":
    main()
