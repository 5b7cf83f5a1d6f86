"""The reference's tests/completeness.rs (`dist` / `inverted precluster` with --ref-completeness-file,
:19-465) replayed through the GPU CLI, each output compared line by line with the oracle run on the
same sketches with the completeness vector the reference would build (src/io.rs:240-324: listed
genomes get their value, genomes missing from the file default to 1.0 with a warning, names the
database does not know are ignored with a warning, values outside [0, 1] are an error)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import REF_FIXTURES, ROOT
from helpers import rust_f32

pytestmark = pytest.mark.gpu
CLI = os.path.join(ROOT, "sketchlib.rust_amd", "csrc", "_build", "sketchlib")
G82, G84, R6, TIGR4 = ("14412_3#82.contigs_velvet.fa.gz", "14412_3#84.contigs_velvet.fa.gz", "R6.fa.gz", "TIGR4.fa.gz")


def cli(wd, *args, ok=True):
    res = subprocess.run([CLI, *args], cwd=wd, capture_output=True, text=True)
    if ok:
        assert res.returncode == 0, res.stderr
    return res


@pytest.fixture()
def wd(tmp_path):
    for f in (G82, G84, R6, TIGR4):
        shutil.copy(os.path.join(REF_FIXTURES, f), tmp_path / f)
    return tmp_path


def write_completeness(wd, name, rows):
    (wd / name).write_text("".join(f"{g}\t{c}\n" for g, c in rows))


def sketch(wd, out, genomes, *kflags, size="1000"):
    cli(wd, "sketch", "-o", out, "-v", *kflags, "-s", size, *genomes)
    n = len(genomes)
    bins = np.fromfile(wd / (out + ".skd"), dtype="<u8")
    return bins, n


def dense_text(names, d):
    lines, x = [], 0
    for i in range(len(names)):
        for j in range(i + 1, len(names)):
            lines.append("\t".join([names[i], names[j]] + [rust_f32(v) for v in d[x]]))
            x += 1
    return "\n".join(lines) + "\n"


def test_completeness_ordering_and_cutoff(oracle, gpu_ctx, wd):
    """tests/completeness.rs:19-240: k = 31, completeness 0.8 / 0.85 / 0.9; the default cutoff 0.64
    corrects every pair with matching bins, cutoff 0.8 corrects none."""
    genomes = [G82, G84, R6]
    bins, n = sketch(wd, "test_genomes", genomes, "-k", "31")
    ss64 = bins.size // (n * 14)
    write_completeness(wd, "c.txt", [(G82, 0.8), (G84, 0.85), (R6, 0.9)])
    plain = cli(wd, "dist", "test_genomes", "-k", "31", "-o", "d0", "-v")
    cli(wd, "dist", "test_genomes", "-k", "31", "-o", "d1", "--ref-completeness-file", "c.txt", "-v")
    cli(wd, "dist", "test_genomes", "-k", "31", "-o", "d2", "--ref-completeness-file", "c.txt", "--completeness-cutoff", "0.8")
    comp = np.array([0.8, 0.85, 0.9])
    o0 = oracle.Sketches(bins, n, [31], ss64)
    oc = oracle.Sketches(bins, n, [31], ss64, completeness=comp)
    assert (wd / "d0").read_text() == dense_text(genomes, oracle.self_dists_all(o0, oracle.JACCARD, 0))
    assert (wd / "d1").read_text() == dense_text(genomes, oracle.self_dists_all(oc, oracle.JACCARD, 0, cutoff=0.64))
    assert (wd / "d2").read_text() == dense_text(genomes, oracle.self_dists_all(oc, oracle.JACCARD, 0, cutoff=0.8))
    # ... and the reference test's own assertions
    val = lambda f: [float(l.split("\t")[2]) for l in (wd / f).read_text().splitlines()]
    d0, d1, d2 = val("d0"), val("d1"), val("d2")
    meaningful = sum(d < 0.99 for d in d0)
    assert meaningful > 0
    assert sum(abs(a - b) > 0.001 for a, b in zip(d0, d1)) == meaningful
    assert sum(abs(a - b) > 0.001 for a, b in zip(d0, d2)) == 0
    assert "warn" not in plain.stderr.lower()


def test_missing_genomes_default_to_one(oracle, gpu_ctx, wd):
    """tests/completeness.rs:243-310 + io.rs:309-322."""
    genomes = [G82, G84, R6]
    bins, n = sketch(wd, "test_missing", genomes, "-k", "21")
    ss64 = bins.size // (n * 14)
    write_completeness(wd, "c.txt", [(G82, 0.8), (G84, 0.9)])
    res = cli(wd, "dist", "test_missing", "-k", "21", "-o", "d", "--ref-completeness-file", "c.txt", "-v")
    assert "1 genome(s) not found in completeness file, using default 1.0: R6.fa.gz" in res.stderr
    oc = oracle.Sketches(bins, n, [21], ss64, completeness=np.array([0.8, 0.9, 1.0]))
    text = (wd / "d").read_text()
    assert text == dense_text(genomes, oracle.self_dists_all(oc, oracle.JACCARD, 0))
    assert all(0.0 <= float(l.split("\t")[2]) <= 1.0 for l in text.splitlines())


def test_extra_genomes_are_ignored(oracle, gpu_ctx, wd):
    """tests/completeness.rs:312-378 + io.rs:300-307."""
    genomes = [G82, G84]
    bins, n = sketch(wd, "test_extra", genomes, "-k", "21")
    ss64 = bins.size // (n * 14)
    write_completeness(wd, "c.txt", [(G82, 0.8), (G84, 0.9), ("NonExistentGenome1", 0.5), ("NonExistentGenome2", 0.6),
                                     ("AnotherFakeGenome", 0.7)])
    res = cli(wd, "dist", "test_extra", "-k", "21", "-o", "d", "--ref-completeness-file", "c.txt", "-v")
    assert "3 genome(s) in completeness file not found in sketch database (ignored)" in res.stderr
    assert all(g in res.stderr for g in ("NonExistentGenome1", "NonExistentGenome2", "AnotherFakeGenome"))
    oc = oracle.Sketches(bins, n, [21], ss64, completeness=np.array([0.8, 0.9]))
    assert (wd / "d").read_text() == dense_text(genomes, oracle.self_dists_all(oc, oracle.JACCARD, 0))


def test_precluster_with_completeness(oracle, gpu_ctx, wd):
    """tests/completeness.rs:381-465: inverted precluster --knn 2 --ref-completeness-file."""
    genomes = [G82, G84, R6]
    cli(wd, "inverted", "build", "-o", "precluster_index", "-v", "-k", "21", "-s", "10", "--write-skq", *genomes)
    bins, n = sketch(wd, "precluster_sketches", genomes, "-k", "21")
    ss64 = bins.size // (n * 14)
    write_completeness(wd, "c.txt", [(G82, 0.8), (G84, 0.9), (R6, 0.7)])
    cli(wd, "inverted", "precluster", "precluster_index.ski", "--skd", "precluster_sketches", "-v", "--knn", "2",
        "--ref-completeness-file", "c.txt", "-o", "pre")
    rows = [l.split("\t") for l in (wd / "pre").read_text().splitlines()]
    assert rows and all(0.0 <= float(r[2]) <= 1.0 for r in rows)
    skq = np.fromfile(wd / "precluster_index.skq", dtype="<u2").reshape(n, 10)
    oc = oracle.Sketches(bins, n, [21], ss64, completeness=np.array([0.8, 0.9, 0.7]))
    exp = oracle.self_dists_knn_precluster(oc, skq, 2, ties=oracle.TIES_RUST_HEAP)
    want = sorted(f"{genomes[i]}\t{genomes[int(e['idx'])]}\t{rust_f32(e['d0'])}" for i in range(n) for e in exp[i]
                  if not (int(e["idx"]) == i and e["d0"] >= 1.0))          # padding is not printed (distance_matrix.rs:379-381)
    assert sorted("\t".join(r) for r in rows) == want
    # without the file the corrected pair (0.8 * 0.9 >= 0.64) is further away
    cli(wd, "inverted", "precluster", "precluster_index.ski", "--skd", "precluster_sketches", "--knn", "2", "-o", "pre0")
    plain = {tuple(l.split("\t")[:2]): float(l.split("\t")[2]) for l in (wd / "pre0").read_text().splitlines()}
    corrected = {tuple(r[:2]): float(r[2]) for r in rows}
    assert corrected[(G82, G84)] < plain[(G82, G84)]


def test_core_accessory_and_cross_query_with_completeness(oracle, gpu_ctx, wd):
    """Both completeness vectors (lib.rs:334-339,401-406) in core/accessory mode, dense and kNN."""
    refs, queries = [G82, G84, R6, TIGR4], [TIGR4, G84]
    rb, nr = sketch(wd, "refs", refs, "--k-seq", "17,31,4")
    qb, nq = sketch(wd, "queries", queries, "--k-seq", "17,31,4")
    kmers = [17, 21, 25, 29]
    ss64 = rb.size // (nr * len(kmers) * 14)
    write_completeness(wd, "rc.txt", [(TIGR4, 0.95), (G82, 0.8), (R6, 0.85), (G84, 0.9)])      # any order
    write_completeness(wd, "qc.txt", [(G84, 0.75)])                                            # TIGR4 defaults to 1.0
    o_r = oracle.Sketches(rb, nr, kmers, ss64, completeness=np.array([0.8, 0.9, 0.85, 0.95]))
    o_q = oracle.Sketches(qb, nq, kmers, ss64, completeness=np.array([1.0, 0.75]))
    out = cli(wd, "dist", "refs", "--ref-completeness-file", "rc.txt").stdout
    assert out == dense_text(refs, oracle.self_dists_all(o_r))
    out = cli(wd, "dist", "refs", "queries", "--ref-completeness-file", "rc.txt", "--query-completeness-file", "qc.txt").stdout
    d = oracle.cross_dists_all(o_r, o_q)
    want = "".join("\t".join([refs[i], queries[j]] + [rust_f32(v) for v in d[i, j]]) + "\n" for i in range(nr) for j in range(nq))
    assert out == want
    # only one side given: no correction at all (jaccard.rs:36: both must be Some)
    out = cli(wd, "dist", "refs", "queries", "--ref-completeness-file", "rc.txt").stdout
    d = oracle.cross_dists_all(oracle.Sketches(rb, nr, kmers, ss64), oracle.Sketches(qb, nq, kmers, ss64))
    assert out == "".join("\t".join([refs[i], queries[j]] + [rust_f32(v) for v in d[i, j]]) + "\n"
                          for i in range(nr) for j in range(nq))
    # self kNN with completeness
    out = cli(wd, "dist", "refs", "--knn", "2", "--ref-completeness-file", "rc.txt").stdout
    exp = oracle.self_dists_knn(o_r, 2, ties=oracle.TIES_RUST_HEAP)      # the CLI's default: the reference binary's tie order
    want = "".join(f"{refs[i]}\t{refs[int(e['idx'])]}\t{rust_f32(e['d0'])}\t{rust_f32(e['d1'])}\n" for i in range(nr) for e in exp[i])
    assert out == want
    out = cli(wd, "dist", "refs", "--knn", "2", "--ref-completeness-file", "rc.txt", "--knn-ties", "canonical").stdout
    exp = oracle.self_dists_knn(o_r, 2, ties=oracle.TIES_CANONICAL)
    assert out == "".join(f"{refs[i]}\t{refs[int(e['idx'])]}\t{rust_f32(e['d0'])}\t{rust_f32(e['d1'])}\n" for i in range(nr) for e in exp[i])


def test_completeness_file_errors(gpu_ctx, wd):
    """io.rs:250-254 (missing file: Err from main, exit 1), :262-266 (unparsable value: warning, line
    skipped), :285-291 (percentages: error naming the offending lines)."""
    genomes = [G82, G84]
    sketch(wd, "db", genomes, "-k", "21")
    res = cli(wd, "dist", "db", "-k", "21", "--ref-completeness-file", "nope.txt", ok=False)
    assert res.returncode == 1 and "Failed to open completeness file: nope.txt" in res.stderr
    write_completeness(wd, "pct.txt", [(G82, 95), (G84, 0.9)])
    res = cli(wd, "dist", "db", "-k", "21", "--ref-completeness-file", "pct.txt", ok=False)
    assert res.returncode == 1 and "[0.0, 1.0]" in res.stderr and f"{G82}: 95" in res.stderr
    write_completeness(wd, "bad.txt", [(G82, "high"), (G84, 0.9)])
    res = cli(wd, "dist", "db", "-k", "21", "--ref-completeness-file", "bad.txt", "-v")
    assert f"Could not parse completeness value for '{G82}': 'high'" in res.stderr
    assert f"1 genome(s) not found in completeness file, using default 1.0: {G82}" in res.stderr
