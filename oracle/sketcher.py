"""CPU restatement of the reference's assembly sketcher (TEST INFRASTRUCTURE ONLY).

Used to regenerate the sketch database behind the reference's exact-text goldens
(tests/distance.rs:270-328,690-721: `sketch --k-seq 17,31,4 -s 10000` over the four
test genomes), whose `.skd` the reference does not commit.  It is pinned bit-exactly by
regenerating the committed sketches{1,2,3}.skd from the same FASTA files
(tests/golden/make_generated_fixtures.py).

Follows (reference file:line):
  * 2-bit base encoding, valid bases, N / record-boundary offsets
        src/hashing/mod.rs:82-97, src/hashing/nthash_iterator.rs:205-251
  * canonical ntHash of a k-mer: h = srol(h) ^ seed[base] over the window, reverse strand
    with the complement seeds, hash = min(fwd, rev)
        src/hashing/nthash_iterator.rs:325-392,62-68, nthash_tables.rs:4-16,
        swapbits033 src/hashing/mod.rs:99-103
    (restated non-rolling: srol is a bit permutation, hence XOR-linear, so the window hash
    is the XOR of srol^(k-1-i)(seed[b_i]); identical values to the rolling form)
  * bin minima of hash % SIGN_MOD, densification, 14-plane transpose
        src/sketch/mod.rs:34-36,132-153,196-258
"""
import gzip

import numpy as np

BBITS = 14
SIGN_MOD = (1 << 61) - 1
HASH_LOOKUP = np.array([0x3C8BFBB395C60474, 0x3193C18562A02B4C, 0x295549F54BE24456,
                        0x20323ED082572324], dtype=np.uint64)
RC_HASH_LOOKUP = np.array([0x295549F54BE24456, 0x20323ED082572324, 0x3C8BFBB395C60474,
                           0x3193C18562A02B4C], dtype=np.uint64)
U64_MAX = np.uint64(0xFFFFFFFFFFFFFFFF)


def _srol(v):
    """rotate_left(1) then swap bits 0 and 33 (nthash_iterator.rs:368-370)."""
    v = (v << np.uint64(1)) | (v >> np.uint64(63))
    x = (v ^ (v >> np.uint64(33))) & np.uint64(1)
    return v ^ (x | (x << np.uint64(33)))


def read_fasta_bases(path):
    """Returns (codes uint8 array of valid bases, sorted offsets of N's / record ends) in
    valid-base coordinates, as NtHashIterator::add_dna_seq builds them."""
    opener = gzip.open if path.endswith(".gz") else open
    codes, offsets = [], []
    n_valid = 0
    with opener(path, "rb") as f:
        data = f.read()
    records = data.split(b">")[1:]
    for rec in records:
        nl = rec.find(b"\n")
        seq = np.frombuffer(rec[nl + 1:].replace(b"\n", b"").replace(b"\r", b""), dtype=np.uint8)
        lower = seq | 0x20
        valid = (lower == ord("a")) | (lower == ord("c")) | (lower == ord("g")) | (lower == ord("t")) | \
                (lower == ord("u"))
        # offset of an invalid base = number of valid bases before it
        before = np.cumsum(valid) - valid
        offsets.append(n_valid + before[~valid])
        enc = (seq[valid] >> 1) & 0x3     # encode_base
        codes.append(enc)
        n_valid += int(valid.sum())
        offsets.append(np.array([n_valid]))  # record boundary
    return np.concatenate(codes).astype(np.uint8), np.concatenate(offsets).astype(np.int64)


def kmer_hashes(codes, offsets, k, rc=True):
    """Canonical ntHash of every valid k-mer window."""
    n = len(codes)
    if n < k:
        return np.zeros(0, dtype=np.uint64)
    n_win = n - k + 1
    # a window [s, s+k) is invalid iff some offset o has s < o < s+k
    bad = np.zeros(n + 1, dtype=np.int32)
    for o in np.unique(offsets):
        lo, hi = max(o - k + 1, 0), min(o - 1, n_win - 1)   # s in [o-k+1, o-1]
        if lo <= hi:
            bad[lo] += 1
            bad[hi + 1] -= 1
    valid = np.cumsum(bad[:n_win]) == 0
    # srol^m of each seed
    tab_f = np.zeros((k, 4), dtype=np.uint64)
    tab_r = np.zeros((k, 4), dtype=np.uint64)
    tab_f[0], tab_r[0] = HASH_LOOKUP, RC_HASH_LOOKUP
    for m in range(1, k):
        tab_f[m] = _srol(tab_f[m - 1])
        tab_r[m] = _srol(tab_r[m - 1])
    fh = np.zeros(n_win, dtype=np.uint64)
    rh = np.zeros(n_win, dtype=np.uint64)
    for i in range(k):
        b = codes[i:i + n_win]
        fh ^= tab_f[k - 1 - i][b]
        if rc:
            rh ^= tab_r[i][b]
    h = np.minimum(fh, rh) if rc else fh
    return h[valid]


def _universal_hash(s, t):
    m = (1 << 64) - 1
    x = (s * 1009 + t * (1000 * 1000 + 3)) & m
    return ((x * 48271 + 11) & m) % ((1 << 31) - 1)


def densify_bin(signs):
    """src/sketch/mod.rs:237-258 (sequential, in place)."""
    if signs.max() != U64_MAX:
        return False
    n = len(signs)
    s = [int(v) for v in signs]
    umax = int(U64_MAX)
    for i in range(n):
        j, attempts = i, 0
        while s[j] == umax:
            j = _universal_hash(i, attempts) % n
            attempts += 1
        s[i] = s[j]
    signs[:] = np.array(s, dtype=np.uint64)
    return True


def fill_usigs(signs):
    """src/sketch/mod.rs:215-223: bit (b % 64) of word (b/64)*14 + plane = bit `plane` of signs[b]."""
    nb = len(signs)
    v = signs.reshape(nb // 64, 64)
    out = np.zeros((nb // 64, BBITS), dtype=np.uint64)
    for p in range(BBITS):
        bits = ((v >> np.uint64(p)) & np.uint64(1)).astype(np.uint8)
        out[:, p] = np.packbits(bits, axis=1, bitorder="little").view("<u8")[:, 0]
    return out.reshape(-1)


def sketch_sequence(codes, offsets, kmers, sketch_size, rc=True):
    """Sketch::new (src/sketch/mod.rs:74-129): u64 words [k][chunk][plane] of one sample."""
    ss64 = -(-sketch_size // 64)
    num_bins = ss64 * 64
    bin_size = -(-SIGN_MOD // num_bins)
    words = []
    for k in kmers:
        h = kmer_hashes(codes, offsets, k, rc) % np.uint64(SIGN_MOD)
        signs = np.full(num_bins, U64_MAX, dtype=np.uint64)
        np.minimum.at(signs, (h // np.uint64(bin_size)).astype(np.int64), h)
        densify_bin(signs)
        words.append(fill_usigs(signs))
    return np.concatenate(words)


def sketch_files(paths, kmers, sketch_size, rc=True):
    """[n_samples, nk*ss64*14] u64 in .skd order (one sample per file)."""
    rows = []
    for p in paths:
        codes, offsets = read_fasta_bases(p)
        rows.append(sketch_sequence(codes, offsets, sorted(kmers), sketch_size, rc))
    return np.stack(rows)


def inverted_sketch_files(paths, k, sketch_size, rc=True):
    """Inverted::sketch_files_inverted (src/inverted.rs:303-395) for single-file samples:
    get_signs_no_densify with num_bins = sketch_size (NOT rounded up to 64), densify_bin, then
    `as u16` (the low 16 bits of the sign).  [n_samples, sketch_size] uint16 = the .skq rows."""
    bin_size = -(-SIGN_MOD // sketch_size)
    rows = []
    for p in paths:
        codes, offsets = read_fasta_bases(p)
        h = kmer_hashes(codes, offsets, k, rc) % np.uint64(SIGN_MOD)
        signs = np.full(sketch_size, U64_MAX, dtype=np.uint64)
        np.minimum.at(signs, (h // np.uint64(bin_size)).astype(np.int64), h)
        densify_bin(signs)
        rows.append((signs & np.uint64(0xFFFF)).astype(np.uint16))
    return np.stack(rows)
