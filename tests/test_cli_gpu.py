"""`sketchlib dist` (C++ host + HIP engine) against the reference's CLI goldens and the
oracle, in the style of the reference's own tests/distance.rs / tests/inverted.rs."""
import os
import subprocess

import numpy as np
import pytest

from conftest import REF_FIXTURES, ROOT
from helpers import FIXTURE_NAMES, load_fixture_bins, rust_f32

pytestmark = pytest.mark.gpu
CLI = os.path.join(ROOT, "sketchlib.rust_amd", "csrc", "_build", "sketchlib")


def run(*args, env=None):
    res = subprocess.run([CLI, "dist", *args], capture_output=True, text=True,
                         env=None if env is None else {**os.environ, **env})
    assert res.returncode == 0, res.stderr
    return res.stdout


@pytest.mark.parametrize("flags,golden", [((), "inverted_precluster.stdout"),
                                          (("--ani",), "inverted_precluster_ani.stdout")])
def test_sketches3_knn1_reference_stdout(gpu_ctx, flags, golden):
    out = run(os.path.join(REF_FIXTURES, "sketches3"), "-k", "21", "--knn", "1", *flags)
    expected = open(os.path.join(REF_FIXTURES, golden)).read()
    assert sorted(out.splitlines()) == sorted(expected.splitlines())


def test_dense_self_text_matches_oracle(oracle, gpu_ctx):
    bins, n, kmers, ss64 = load_fixture_bins("sketches2")
    o = oracle.Sketches(bins, n, kmers, ss64)
    d = oracle.self_dists_all(o, oracle.JACCARD, 0).ravel()
    lines, x = [], 0
    for i in range(n):
        for j in range(i + 1, n):
            lines.append(f"{FIXTURE_NAMES[i]}\t{FIXTURE_NAMES[j]}\t{rust_f32(d[x])}")
            x += 1
    assert run(os.path.join(REF_FIXTURES, "sketches2.skm"), "-k", "31") == "\n".join(lines) + "\n"


def test_legacy_db_coreacc_text(gpu_ctx):
    out = run(os.path.join(REF_FIXTURES, "legacy_db"))
    assert out == "R6.fa.gz\tTIGR4.fa.gz\t0.02203464\t0\n"


def test_subset_order_follows_file(gpu_ctx, oracle, tmp_path):
    """--subset: rows/names follow the subset file order (tests/distance.rs:690-721)."""
    sub = tmp_path / "subset.txt"
    sub.write_text("TIGR4.fa.gz\n14412_3#82.contigs_velvet.fa.gz\nR6.fa.gz\n")
    out = run(os.path.join(REF_FIXTURES, "sketches1"), "-k", "31", "--subset", str(sub))
    names = [l.split("\t")[:2] for l in out.splitlines()]
    assert names == [["TIGR4.fa.gz", "14412_3#82.contigs_velvet.fa.gz"], ["TIGR4.fa.gz", "R6.fa.gz"],
                     ["14412_3#82.contigs_velvet.fa.gz", "R6.fa.gz"]]
    vals = [l.split("\t")[2] for l in out.splitlines()]
    assert vals == ["1", "0.33789062", "1"]   # App. A: (0,3)=1, (2,3)=0.33789062, (0,2)=1


def test_cross_query_and_output_file(gpu_ctx, oracle, tmp_path):
    out_file = tmp_path / "d.txt"
    db = os.path.join(REF_FIXTURES, "sketches1")
    subprocess.check_call([CLI, "dist", db, db, "-k", "31", "-o", str(out_file)])
    lines = out_file.read_text().splitlines()
    assert len(lines) == 16 and lines[0].split("\t")[2] == "0"      # ref vs itself: 1 - J = 0
    assert lines[1] == "14412_3#82.contigs_velvet.fa.gz\t14412_3#84.contigs_velvet.fa.gz\t0.4169922"
    # cross kNN, 4 query rows x knn=2 in ANI mode.  The genome's own ref entry has ANI 1.0
    # and the same name, which the reference's Display treats as padding and suppresses
    # (distance_matrix.rs:379-381) -- so only the second neighbour of each row prints.
    out = run(db, db, "-k", "31", "--knn", "2", "--ani")
    rows = [l.split("\t") for l in out.splitlines()]
    assert [r[:2] for r in rows] == [[FIXTURE_NAMES[0], FIXTURE_NAMES[1]], [FIXTURE_NAMES[1], FIXTURE_NAMES[0]],
                                     [FIXTURE_NAMES[2], FIXTURE_NAMES[3]], [FIXTURE_NAMES[3], FIXTURE_NAMES[2]]]
    assert [r[2] for r in rows] == ["0.9901376", "0.9901376", "0.99266887", "0.99266887"]
    # Jaccard distances: the self entry (distance 0 < 1) is printed
    out = run(db, db, "-k", "31", "--knn", "2")
    rows = [l.split("\t") for l in out.splitlines()]
    assert len(rows) == 8 and rows[0] == [FIXTURE_NAMES[0], FIXTURE_NAMES[0], "0"]
    assert rows[1] == [FIXTURE_NAMES[0], FIXTURE_NAMES[1], "0.4169922"]


def test_knn_clamped_like_reference(gpu_ctx):
    """lib.rs:379-382: self kNN >= n is clamped to n - 1."""
    out = run(os.path.join(REF_FIXTURES, "sketches1"), "-k", "31", "--knn", "50")
    rows = out.splitlines()
    # 4 rows x 3 neighbours, minus padding-suppressed lines (none here: col != row always)
    assert len(rows) == 12


# ---- the reference's own CLI tests over the 4-genome, 4-k database ----
GEN_DB = os.path.join(os.path.dirname(REF_FIXTURES), "generated", "sketch_db_4k")


@pytest.mark.parametrize("flags,golden", [
    (("--knn", "1"), "dists_knn_ca.stdout"),                         # tests/distance.rs:293-302
    (("--knn", "1", "-k", "21"), "dists_knn_jaccard.stdout"),        # :304-315
    (("--knn", "1", "-k", "21", "--ani"), "dists_knn_ani.stdout"),   # :317-328
])
def test_knn_dists_exact_stdout(gpu_ctx, flags, golden):
    assert run(GEN_DB, "-v", *flags) == open(os.path.join(REF_FIXTURES, golden)).read()


def test_subset_dists_exact_stdout(gpu_ctx):
    """tests/distance.rs:690-721"""
    out = run(GEN_DB, "--subset", os.path.join(REF_FIXTURES, "subset.txt"))
    assert out == open(os.path.join(REF_FIXTURES, "dists_subset.stdout")).read()


def test_completeness_file_all_ones_is_identity(gpu_ctx):
    """completeness.txt lists 1.0 for every genome: the corrected output equals the plain one."""
    plain = run(GEN_DB)
    comp = run(GEN_DB, "--ref-completeness-file", os.path.join(REF_FIXTURES, "completeness.txt"))
    assert plain == comp


def _write_db(tmp_path, name, bins, kmers, ss64):
    """A database in the reference's on-disk layout: .skd = raw LE u64, .skm via skl_dbtool."""
    dbtool = os.path.join(ROOT, "sketchlib.rust_amd", "csrc", "_build", "skl_dbtool")
    prefix = str(tmp_path / name)
    bins.astype("<u8").tofile(prefix + ".skd")
    names = [f"sample_{i:04d}" for i in range(bins.shape[0])]
    subprocess.check_call([dbtool, "make", prefix, str(ss64 * 64), ",".join(map(str, kmers)), *names])
    return prefix, names


def test_synthetic_db_dense_text_and_parallel_formatting(gpu_ctx, oracle, tmp_path):
    from sketchlib.rust_amd import synth

    kmers, ss64, n = [15, 19, 23, 27, 31], 64, 300
    bins = synth.set_r(n, kmers, ss64, n_clusters=12)
    prefix, names = _write_db(tmp_path, "synth", bins, kmers, ss64)
    d = oracle.self_dists_all(oracle.Sketches(bins, n, kmers, ss64), threads=8)
    lines, x = [], 0
    for i in range(n):
        for j in range(i + 1, n):
            lines.append(f"{names[i]}\t{names[j]}\t{rust_f32(d[x][0])}\t{rust_f32(d[x][1])}")
            x += 1
    expected = "\n".join(lines) + "\n"
    assert run(prefix) == expected
    assert run(prefix, "--threads", "7") == expected          # row blocks formatted concurrently
    # cross mode, parallel formatting
    one = run(prefix, prefix, "-k", "23", "--threads", "1")
    assert run(prefix, prefix, "-k", "23", "--threads", "5") == one
    assert len(one.splitlines()) == n * n
    # streamed output: many small row bands (compute of band i+1 overlaps the write of band i)
    for band in ("1", "4096", "100000"):
        env = {"SKL_DIST_BAND_BYTES": band}
        assert run(prefix, "--threads", "3", env=env) == expected
        assert run(prefix, prefix, "-k", "23", "--threads", "3", env=env) == one
    assert run(prefix, "-k", "19", env={"SKL_DIST_BAND_BYTES": "777"}) == run(prefix, "-k", "19")
    assert run(prefix, "--band-mb", "1") == expected
    # --npy: the same numbers as a NumPy array (streamed in bands, and from the multi-context path)
    npy = tmp_path / "dense.npy"
    for extra, env in (((), None), ((), {"SKL_DIST_BAND_BYTES": "40000"}), (("--devices", "0,0"), None)):
        assert run(prefix, "-o", str(npy), "--npy", *extra, env=env) == ""
        arr = np.load(npy)
        assert arr.dtype == np.float32 and arr.shape == d.shape and np.array_equal(arr, d)
    assert run(prefix, prefix, "-k", "23", "-o", str(npy), "--npy", env={"SKL_DIST_BAND_BYTES": "100000"}) == ""
    arr = np.load(npy)
    assert arr.shape == (n * n, 1)
    assert [rust_f32(x) for x in arr[:50, 0]] == [l.split("\t")[2] for l in one.splitlines()[:50]]
    res = subprocess.run([CLI, "dist", prefix, "--npy"], capture_output=True, text=True)
    assert res.returncode == 2 and "--npy needs -o" in res.stderr
    # -o <file>: blocks are written at offsets from all formatting threads
    out_file = tmp_path / "dense.txt"
    for env in (None, {"SKL_DIST_BAND_BYTES": "50000"}):
        assert run(prefix, "-o", str(out_file), "--threads", "6", env=env) == ""
        assert out_file.read_text() == expected


def test_sketch_then_dist_like_reference_knn_dists(gpu_ctx, tmp_path):
    """tests/distance.rs:270-328 end to end: sketch the four genomes, then the three kNN runs."""
    db = str(tmp_path / "sketch_db")
    subprocess.check_call([CLI, "sketch", "-o", db, "-v", "--k-seq", "17,31,4", "-s", "10000", "-f", "rfile.txt"],
                          cwd=REF_FIXTURES, stderr=subprocess.DEVNULL)
    for flags, golden in [(("--knn", "1"), "dists_knn_ca.stdout"),
                          (("--knn", "1", "-k", "21"), "dists_knn_jaccard.stdout"),
                          (("--knn", "1", "-k", "21", "--ani"), "dists_knn_ani.stdout")]:
        assert run(db, "-v", *flags) == open(os.path.join(REF_FIXTURES, golden)).read()


def test_dense_distances_like_reference(gpu_ctx, tmp_path):
    """tests/distance.rs:80-266 `dense_distances`: sketch -> cross dist, compared with the
    pp-sketchlib numbers of sketchlib_output_true.txt under the reference's tolerance
    (round to 3 dp, |diff| <= 0.05)."""
    truth = {}
    for line in open(os.path.join(REF_FIXTURES, "sketchlib_output_true.txt")):
        key, val = line.split(": ")
        truth[key] = [float(x) for x in val.strip().strip("[]").split(",")]

    def close(a, b):
        return abs(round(a, 3) - round(b, 3)) <= 0.05

    def sk(name, *args):
        out = str(tmp_path / name)
        subprocess.check_call([CLI, "sketch", "-o", out, *args], cwd=REF_FIXTURES)
        return out

    # test 1: one short sequence vs the same with one SNP, k = 3
    a = sk("t1a", "--k-vals", "3", "short_sequence.fa")
    b = sk("t1b", "--k-vals", "3", "short_sequence_SNP.fa")
    assert close(float(run(a, b, "-k", "3").split()[-1]), truth["short_sequence_jaccard_dists_3"][0])
    # test 2: whole genome vs the same with one 3.6 kb contig removed, k = 17
    a = sk("t2a", "--k-vals", "17", "14412_3#82.contigs_velvet.fa.gz")
    b = sk("t2b", "--k-vals", "17", "14412_3#82.contigs_velvet_removed_block.fa.gz")
    assert close(float(run(a, b, "-k", "17").split()[-1]), truth["whole_genome_block_removed"][0])
    # test 3: four genomes, k = 31, s = 10000, self mode
    c = sk("t3", "--k-vals", "31", "-s", "10000", *FIXTURE_NAMES)
    got = [float(l.split()[-1]) for l in run(c, "-k", "31").splitlines()]
    assert len(got) == 6 and all(close(x, y) for x, y in zip(got, truth["multiple_genomes"]))


@pytest.mark.parametrize("flags", [(), ("-k", "21"), ("--knn", "2"), ("--knn", "3", "-k", "25", "--ani")])
def test_multi_context_row_bands_equal_single(gpu_ctx, flags):
    """--devices a,b,c: one host thread + one context per entry, contiguous row bands copied
    straight into the host output.  On this 1-GPU box the same device is listed three times,
    which exercises the partition / assembly code; results must equal the single-context run."""
    one = run(GEN_DB, *flags)
    assert run(GEN_DB, *flags, "--devices", "0,0,0") == one
    assert run(GEN_DB, GEN_DB, *flags, "--devices", "0,0") == run(GEN_DB, GEN_DB, *flags)


def test_multi_context_synthetic(gpu_ctx, tmp_path):
    from sketchlib.rust_amd import synth

    kmers, ss64, n = [15, 19, 23, 27, 31], 16, 257
    prefix, _ = _write_db(tmp_path, "m", synth.set_r(n, kmers, ss64, n_clusters=9), kmers, ss64)
    one = run(prefix)
    assert run(prefix, "--devices", "0,0,0,0,0", "--threads", "3") == one
    assert run(prefix, "--gpus", "1") == one
    knn = run(prefix, "--knn", "10")
    assert run(prefix, "--knn", "10", "--devices", "0,0,0") == knn
    # several bands per device: every pair is evaluated once across the devices, the partial
    # top-k states are merged shard by shard (single-k and core/accessory keys)
    # (--knn-ties canonical: row bands dealt over the devices + merged partial states; the default, reference rule: column
    # windows per device -- round 6: every device against heaps it cleared itself, accept logs replayed in window order;
    # SKL_KNN_DECOUPLED=0: heaps that travel from device to device band by band -- either way every pair once and the same
    # text whatever the partition)
    for flags in (("--knn", "10"), ("--knn", "7", "-k", "23"), ("--knn", "7", "-k", "23", "--ani")):
        for ties in ((), ("--knn-ties", "canonical")):
            want = run(prefix, *flags, *ties)
            for devices in ("0,0", "0,0,0,0,0"):
                assert run(prefix, *flags, *ties, "--devices", devices, env={"SKL_KNN_BAND_ROWS": "16"}) == want
                if not ties:
                    assert run(prefix, *flags, "--devices", devices, env={"SKL_KNN_BAND_ROWS": "16", "SKL_KNN_DECOUPLED": "0"}) == want


# ---- `sketchlib inverted precluster`, as tests/inverted.rs drives it ----

def run_cli(wd, *args, ok=True):
    res = subprocess.run([CLI, *args], cwd=wd, capture_output=True, text=True)
    if ok:
        assert res.returncode == 0, res.stderr
    return res


@pytest.fixture()
def precluster_wd(tmp_path):
    import shutil
    for f in FIXTURE_NAMES + ["rfile.txt"]:
        shutil.copy(os.path.join(REF_FIXTURES, f), tmp_path / f)
    run_cli(tmp_path, "inverted", "build", "-o", "inverted", "-v", "-k", "21", "-s", "10", "-f", "rfile.txt", "--write-skq")
    run_cli(tmp_path, "sketch", "-o", "standard", "-v", "--k-vals", "21", "-s", "1000", "-f", "rfile.txt")
    return tmp_path


def test_verbose_runs_end_with_the_complete_line(gpu_ctx, precluster_wd):
    """lib.rs:949-957: a verbose run's last line.  The fast exit of a successful run (no teardown of the HIP runtime) leaves
    from main() AFTER that line, for `dist` and `inverted precluster` alike."""
    wd = precluster_wd
    out = run_cli(wd, "-v", "dist", "standard", "--knn", "1", "-k", "21")
    assert out.stdout.strip() and out.stderr.rstrip().splitlines()[-1].startswith("INFO  [sketchlib] Complete in "), out.stderr
    out = run_cli(wd, "-v", "inverted", "precluster", "--knn", "1", "--skd", "standard", "inverted.ski")
    assert out.stdout.strip() and out.stderr.rstrip().splitlines()[-1].startswith("INFO  [sketchlib] Complete in "), out.stderr
    out = run_cli(wd, "dist", "standard", "--knn", "1", "-k", "21")
    assert "Complete in" not in out.stderr


def test_inverted_precluster_like_reference(gpu_ctx, precluster_wd):
    """tests/inverted.rs:244-349: knn 1, --ani, and knn 50 (clamped, padding not printed)."""
    wd = precluster_wd
    golden = sorted(open(os.path.join(REF_FIXTURES, "inverted_precluster.stdout")).read().splitlines())
    golden_ani = sorted(open(os.path.join(REF_FIXTURES, "inverted_precluster_ani.stdout")).read().splitlines())
    out = run_cli(wd, "inverted", "precluster", "-v", "--knn", "1", "--skd", "standard", "inverted.ski")
    assert sorted(out.stdout.splitlines()) == golden
    out = run_cli(wd, "inverted", "precluster", "-v", "--knn", "1", "--ani", "--skd", "standard", "inverted.ski")
    assert sorted(out.stdout.splitlines()) == golden_ani
    out = run_cli(wd, "inverted", "precluster", "-v", "--knn", "50", "--skd", "standard", "inverted.ski")
    assert sorted(out.stdout.splitlines()) == golden and "knn=50 is higher than number of samples=4" in out.stderr
    run_cli(wd, "inverted", "precluster", "--knn", "1", "--skd", "standard.skm", "-o", "pre.txt", "inverted.ski")
    assert sorted((wd / "pre.txt").read_text().splitlines()) == golden
    # candidate lists from the .ski bitmaps on host threads instead of on the device
    out = run_cli(wd, "inverted", "precluster", "--knn", "1", "--skd", "standard", "inverted.ski", "--host-candidates",
                  "--threads", "3")
    assert sorted(out.stdout.splitlines()) == golden


def test_inverted_precluster_reordered_index_and_retain(gpu_ctx, precluster_wd):
    """An index in a different sample order gives the same rows (tests/inverted.rs:352-452); a
    .skd with a k the index was not built for panics; unmatched genomes follow --retain-unmatched."""
    wd = precluster_wd
    golden = sorted(open(os.path.join(REF_FIXTURES, "inverted_precluster.stdout")).read().splitlines())
    (wd / "species.txt").write_text("TIGR4.fa.gz\ta\n14412_3#82.contigs_velvet.fa.gz\tb\nR6.fa.gz\ta\n")
    run_cli(wd, "inverted", "build", "-o", "reordered", "-k", "21", "-s", "10", "-f", "rfile.txt", "--write-skq",
            "--species-names", "species.txt")
    for extra in ((), ("--host-candidates",)):
        out = run_cli(wd, "inverted", "precluster", "--knn", "3", "--skd", "standard", "reordered.ski", *extra)
        assert sorted(out.stdout.splitlines()) == golden
    run_cli(wd, "sketch", "-o", "other_k", "--k-vals", "17", "-s", "1000", "-f", "rfile.txt")
    res = run_cli(wd, "inverted", "precluster", "--skd", "other_k", "inverted.ski", ok=False)
    assert res.returncode == 101 and "K-mer size 21 used for .ski not found in .skd" in res.stderr
    # an index over fewer samples than the .skd
    run_cli(wd, "inverted", "build", "-o", "two", "-k", "21", "-s", "10", "--write-skq", "R6.fa.gz", "TIGR4.fa.gz")
    res = run_cli(wd, "inverted", "precluster", "--skd", "standard", "two.ski", ok=False)
    assert res.returncode == 101 and "could not be found in the .ski" in res.stderr
    # three genomes: R6 shares bins only with TIGR4 (the golden pairs), so without it R6 is unmatched
    three = ["14412_3#82.contigs_velvet.fa.gz", "14412_3#84.contigs_velvet.fa.gz", "R6.fa.gz"]
    run_cli(wd, "inverted", "build", "-o", "sparse", "-k", "21", "-s", "10", "--write-skq", *three)
    run_cli(wd, "sketch", "-o", "standard3", "--k-vals", "21", "-s", "1000", *three)
    skq = np.fromfile(wd / "sparse.skq", dtype="<u2").reshape(3, 10)
    lonely = [i for i in range(3) if not any((skq[i] == skq[j]).any() for j in range(3) if j != i)]
    assert lonely == [2]
    plain = run_cli(wd, "inverted", "precluster", "--knn", "2", "--skd", "standard3", "sparse.ski").stdout
    names_in = {l.split("\t")[0] for l in plain.splitlines()}
    assert all(three[i] not in names_in for i in lonely)                   # only padding: nothing printed
    single = run_cli(wd, "inverted", "precluster", "--knn", "2", "--skd", "standard3", "sparse.ski",
                     "--retain-unmatched", "singleton").stdout
    for i in lonely:                                                       # tests/inverted.rs:702-748
        assert f"{three[i]}\t{three[i]}\t0\n" in single
    assert single == run_cli(wd, "inverted", "precluster", "--knn", "2", "--skd", "standard3", "sparse.ski",
                             "--retain-unmatched", "singleton", "--host-candidates").stdout
    brute = run_cli(wd, "inverted", "precluster", "--knn", "2", "--skd", "standard3", "sparse.ski",
                    "--retain-unmatched", "bruteforce").stdout
    full = run_cli(wd, "dist", "standard3", "-k", "21", "--knn", "2").stdout
    for i in lonely:                                                       # tests/inverted.rs:751-802
        rows = sorted(l for l in brute.splitlines() if l.startswith(three[i] + "\t"))
        assert rows and rows == sorted(l for l in full.splitlines() if l.startswith(three[i] + "\t"))


def _sparse_text(names, exp, suppress_padding):
    """SparseDistanceMatrix's Display (distance_matrix.rs:362-401): `row\tneighbour\tdist`; Jaccard rows with
    dist >= 1.0 pointing at themselves (padding) are not printed (:379-381)."""
    lines = []
    for r in range(exp.shape[0]):
        for item in exp[r]:
            if suppress_padding and item["d0"] >= 1.0 and int(item["idx"]) == r:
                continue
            lines.append(f"{names[r]}\t{names[int(item['idx'])]}\t{rust_f32(item['d0'])}")
    return "\n".join(lines) + "\n"


def test_knn_3000_and_reference_tie_order_through_the_cli(gpu_ctx, oracle, tmp_path):
    """`dist --knn 3000` on a 5 000-sample database: the reference only clamps knn to n - 1 (lib.rs:379-382); the default
    (`--knn-ties reference`) prints the ids the reference binary prints (its BinaryHeap replayed, mod.rs:41-48),
    `--knn-ties canonical` the lowest-index-first lists."""
    from sketchlib.rust_amd import synth

    kmers, ss64, n = [21], 2, 5000
    bins = synth.set_r(n, kmers, ss64, n_clusters=9)
    prefix, names = _write_db(tmp_path, "big", bins, kmers, ss64)
    o = oracle.Sketches(bins, n, kmers, ss64)
    for knn in (3000, 40):
        canon = oracle.self_dists_knn(o, knn, oracle.JACCARD, 0, False, ties=oracle.TIES_CANONICAL, threads=8)
        heap = oracle.self_dists_knn(o, knn, oracle.JACCARD, 0, False, ties=oracle.TIES_RUST_HEAP, threads=8)
        assert not np.array_equal(canon["idx"], heap["idx"])          # the two rules differ on this database
        # (bool() around the comparisons: pytest's diff of two multi-megabyte strings takes longer than the test)
        assert bool(run(prefix, "-k", "21", "--knn", str(knn)) == _sparse_text(names, heap, True)), "default = the reference's tie order"
        assert bool(run(prefix, "-k", "21", "--knn", str(knn), "--knn-ties", "reference") == _sparse_text(names, heap, True))
        assert bool(run(prefix, "-k", "21", "--knn", str(knn), "--knn-ties", "canonical") == _sparse_text(names, canon, True))
    res = subprocess.run([CLI, "dist", prefix, "--knn", "5", "--knn-ties", "fifo"], capture_output=True, text=True)
    assert res.returncode == 2 and "possible values: canonical, reference" in res.stderr


def test_single_k_runs_read_one_slice_of_a_multi_k_database(gpu_ctx, tmp_path):
    """`dist -k K` on a database with several k-mer lengths reads only that slice of each sample (MultiSketch::select_kmer,
    round 4): every listing must be byte for byte the listing of a ONE-k database holding just those slices -- self, cross,
    --subset, kNN, ANI; a query database that lists OTHER k-mer lengths is still refused."""
    from sketchlib.rust_amd import synth

    kmers, ss64, n, nq = [15, 19, 23, 27], 8, 37, 11
    bins = synth.set_r(n, kmers, ss64, n_clusters=5).reshape(n, len(kmers), ss64 * 14)
    qbins = synth.set_r(nq, kmers, ss64, n_clusters=5, first_sample=500).reshape(nq, len(kmers), ss64 * 14)
    db4, names = _write_db(tmp_path, "db4", bins, kmers, ss64)
    q4, _ = _write_db(tmp_path, "q4", qbins, kmers, ss64)
    subset = tmp_path / "subset.txt"
    subset.write_text("\n".join(names[i] for i in (30, 2, 17, 5, 9)) + "\n")
    for ki, k in ((0, 15), (2, 23), (3, 27)):
        db1, _ = _write_db(tmp_path, f"db1_{k}", bins[:, ki], [k], ss64)
        q1, _ = _write_db(tmp_path, f"q1_{k}", qbins[:, ki], [k], ss64)
        for flags in ((), ("--ani",), ("--knn", "4"), ("--knn", "3", "--ani"), ("--subset", str(subset)), ("--subset", str(subset), "--knn", "2")):
            assert bool(run(db4, "-k", str(k), *flags) == run(db1, "-k", str(k), *flags)), (k, flags)
        for flags in ((), ("--knn", "5"), ("--ani",)):
            assert bool(run(db4, q4, "-k", str(k), *flags) == run(db1, q1, "-k", str(k), *flags)), (k, flags)
    # a query database that lists other lengths is refused as before (whole databases read, then the compatibility check)
    q_other, _ = _write_db(tmp_path, "q_other", qbins[:, [1, 0, 3, 2]], [19, 15, 27, 23], ss64)
    res = subprocess.run([CLI, "dist", db4, q_other, "-k", "23"], capture_output=True, text=True)
    assert res.returncode != 0 and "not compatible" in res.stderr, res.stderr
