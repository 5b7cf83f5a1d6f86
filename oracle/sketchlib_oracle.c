/*
 * sketchlib_oracle.c -- CPU restatement of the reference's distance path.
 *
 * TEST INFRASTRUCTURE ONLY (see sketchlib_oracle.h).  Plain C, f64 exactly
 * where the reference uses f64 and cast to f32 at the same points.  Build with
 * -ffp-contract=off: rustc never fuses a*b+c, gcc would under -march=native.
 *
 * Citations: reference file:line (bacpop/sketchlib.rust v0.3.0).
 */
#include "sketchlib_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ */
/* src/distances/jaccard.rs                                            */
/* ------------------------------------------------------------------ */

/* jaccard.rs:15-25 -- for each chunk of BBITS words AND together !(a^b), popcount,
 * sum over chunks.  A bin (one bit position of one chunk) matches iff all 14
 * bit-planes agree. */
uint32_t sko_samebits(const uint64_t *s1, const uint64_t *s2, uint64_t sketchsize64)
{
    uint32_t samebits = 0;
    for (uint64_t c = 0; c < sketchsize64; ++c) {
        uint64_t bits = ~(uint64_t)0;
        for (int p = 0; p < SKO_BBITS; ++p) {
            bits &= ~(s1[c * SKO_BBITS + p] ^ s2[c * SKO_BBITS + p]);
        }
        samebits += (uint32_t)__builtin_popcountll(bits);
    }
    return samebits;
}

/* jaccard.rs:55-57 */
double sko_completeness_correction(double jaccard, double c1, double c2)
{
    return jaccard / (c1 * c2 / (c1 + c2 - c1 * c2));
}

/* jaccard.rs:14,26-44 -- everything after the popcount */
double sko_jaccard_from_samebits(uint32_t samebits, uint64_t sketchsize64, int has_c, double c1,
                                 double c2, double completeness_cutoff)
{
    double unionsize = (double)(64u * sketchsize64);                  /* :14 */
    uint32_t maxnbits = (uint32_t)sketchsize64 * 64u;                 /* :26 */
    uint32_t expected_samebits = maxnbits >> SKO_BBITS;               /* :27 */
    uint32_t diff = samebits > expected_samebits ? samebits - expected_samebits : 0; /* :30 saturating_sub */
    double intersize =
        ((double)diff * (double)maxnbits) / (double)(maxnbits - expected_samebits); /* :31 */
    double jaccard_index = intersize / unionsize;                     /* :33 */
    if (has_c) {                                                      /* :36 both Some */
        if (c1 * c2 >= completeness_cutoff) {                         /* :37 */
            jaccard_index = sko_completeness_correction(jaccard_index, c1, c2);
            jaccard_index = fmin(jaccard_index, 1.0);                 /* :40 */
        }
    }
    return jaccard_index;
}

/* jaccard.rs:6-45 */
double sko_jaccard_index(const uint64_t *s1, const uint64_t *s2, uint64_t sketchsize64, int has_c,
                         double c1, double c2, double completeness_cutoff)
{
    return sko_jaccard_from_samebits(sko_samebits(s1, s2, sketchsize64), sketchsize64, has_c, c1,
                                     c2, completeness_cutoff);
}

/* jaccard.rs:49-51.  f64::max returns the non-NaN operand, like fmax. */
double sko_ani_pois(double jaccard, double k)
{
    return fmax(0.0, 1.0 + 1.0 / k * log((2.0 * jaccard) / (1.0 + jaccard)));
}

/* jaccard.rs:105-142 */
void sko_simple_linear_regression(double xsum, double ysum, double xysum, double xsquaresum,
                                  double ysquaresum, double n, float *core_out, float *acc_out)
{
    if (isnan(ysum) || ysum == -INFINITY || n < 3.0) { /* :117 */
        *core_out = 1.0f;
        *acc_out = 1.0f;
        return;
    }
    double xbar = xsum / n;
    double ybar = ysum / n;
    double x_diff = xsquaresum - xsum * xsum / n;
    double y_diff = ysquaresum - ysum * ysum / n;
    double xstddev = sqrt((xsquaresum - xsum * xsum / n) / n);
    double ystddev = sqrt((ysquaresum - ysum * ysum / n) / n);
    double r = (xysum - xsum * ysum / n) / sqrt(x_diff * y_diff);
    double beta = r * ystddev / xstddev;
    double alpha = -beta * xbar + ybar;

    double core = 0.0, acc = 0.0;
    if (beta < 0.0) { /* :133-137; NaN compares false on both arms */
        core = 1.0 - exp(beta);
    } else if (r > 0.0) {
        core = 1.0;
    }
    if (alpha < 0.0) { /* :138-140 */
        acc = 1.0 - exp(alpha);
    }
    *core_out = (float)core;
    *acc_out = (float)acc;
}

/* multisketch.rs:213-219 */
static inline const uint64_t *get_sketch_slice(const sko_sketches *s, size_t idx, size_t k_idx)
{
    size_t kmer_stride = (size_t)s->sketchsize64 * SKO_BBITS;
    size_t sample_stride = kmer_stride * s->nk;
    return s->bins + idx * sample_stride + k_idx * kmer_stride;
}

/* jaccard.rs:61-101 */
void sko_core_acc_dist(const sko_sketches *ref, const sko_sketches *query, size_t ref_idx,
                       size_t query_idx, double completeness_cutoff, float *core, float *acc)
{
    double xsum = 0, ysum = 0, xysum = 0, xsquaresum = 0, ysquaresum = 0, n = 0;
    /* :75 -- sketch_size is in bins (= sketchsize64*64), times u64::BITS again */
    double tolerance = log(2.0 / (double)((ref->sketchsize64 * 64u) * 64u));
    int has_c = ref->completeness != NULL && query->completeness != NULL;
    for (size_t k_idx = 0; k_idx < ref->nk; ++k_idx) {
        double c1 = has_c ? ref->completeness[ref_idx] : 0.0;
        double c2 = has_c ? query->completeness[query_idx] : 0.0;
        double y = log(sko_jaccard_index(get_sketch_slice(ref, ref_idx, k_idx),
                                         get_sketch_slice(query, query_idx, k_idx),
                                         ref->sketchsize64, has_c, c1, c2, completeness_cutoff));
        if (y < tolerance) { /* :89-91 -- break, not continue */
            break;
        }
        double k_fl = (double)ref->kmers[k_idx];
        xsum += k_fl;
        ysum += y;
        xysum += k_fl * y;
        xsquaresum += k_fl * k_fl;
        ysquaresum += y * y;
        n += 1.0;
    }
    sko_simple_linear_regression(xsum, ysum, xysum, xsquaresum, ysquaresum, n, core, acc);
}

/* ------------------------------------------------------------------ */
/* src/distances/distance_matrix.rs:11-51                              */
/* ------------------------------------------------------------------ */

size_t sko_square_to_condensed(size_t i, size_t j, size_t n)
{
    return n * i - ((i * (i + 1)) >> 1) + j - 1 - i;
}

size_t sko_calc_row_idx(size_t k, size_t n)
{
    int64_t k_i = (int64_t)k, n_i = (int64_t)n;
    return n - 2 - (size_t)floor(sqrt((double)(-8 * k_i + 4 * n_i * (n_i - 1) - 7)) / 2.0 - 0.5);
}

size_t sko_calc_col_idx(size_t k, size_t i, size_t n)
{
    int64_t k_i = (int64_t)k, i_i = (int64_t)i, n_i = (int64_t)n;
    return (size_t)(k_i + i_i + 1 - n_i * (n_i - 1) / 2 + (n_i - i_i) * ((n_i - i_i) - 1) / 2);
}

/* ------------------------------------------------------------------ */
/* A tiny work-sharing pool standing in for rayon's par_chunks_mut     */
/* ------------------------------------------------------------------ */

typedef void (*chunk_fn)(void *ctx, size_t chunk_idx);
typedef struct {
    chunk_fn fn;
    void *ctx;
    size_t n_chunks;
    atomic_size_t next;
} pool_job;

static void *pool_worker(void *arg)
{
    pool_job *job = (pool_job *)arg;
    for (;;) {
        size_t c = atomic_fetch_add(&job->next, 1);
        if (c >= job->n_chunks) break;
        job->fn(job->ctx, c);
    }
    return NULL;
}

static void run_chunks(chunk_fn fn, void *ctx, size_t n_chunks, int threads)
{
    pool_job job;
    job.fn = fn;
    job.ctx = ctx;
    job.n_chunks = n_chunks;
    atomic_init(&job.next, 0);
    if (threads <= 1 || n_chunks <= 1) {
        pool_worker(&job);
        return;
    }
    pthread_t *tids = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)threads);
    int started = 0;
    for (int t = 0; t < threads - 1; ++t) {
        if (pthread_create(&tids[started], NULL, pool_worker, &job) == 0) ++started;
    }
    pool_worker(&job);
    for (int t = 0; t < started; ++t) pthread_join(tids[t], NULL);
    free(tids);
}

/* ------------------------------------------------------------------ */
/* src/distances/mod.rs -- dense drivers                               */
/* ------------------------------------------------------------------ */

#define CHUNK_SIZE 1000 /* mod.rs:20 */

typedef struct {
    const sko_sketches *ref;
    const sko_sketches *query; /* NULL for self */
    int dist_type;
    size_t k_idx;
    int ani;
    double cutoff;
    float *out;
    uint32_t *out_bits; /* binmatch mode when non-NULL */
    size_t n_dist;
} dense_job;

/* One pair, written the way mod.rs:83-115 / :253-283 do. */
static inline void dense_pair(const dense_job *job, const sko_sketches *a, const sko_sketches *b,
                              size_t i, size_t j, size_t dist_idx)
{
    if (job->out_bits) {
        for (size_t k = 0; k < a->nk; ++k) {
            job->out_bits[dist_idx * a->nk + k] =
                sko_samebits(get_sketch_slice(a, i, k), get_sketch_slice(b, j, k), a->sketchsize64);
        }
        return;
    }
    if (job->dist_type == SKO_JACCARD) {
        int has_c = a->completeness != NULL && b->completeness != NULL;
        double c1 = has_c ? a->completeness[i] : 0.0;
        double c2 = has_c ? b->completeness[j] : 0.0;
        double j_index = sko_jaccard_index(get_sketch_slice(a, i, job->k_idx),
                                           get_sketch_slice(b, j, job->k_idx), a->sketchsize64,
                                           has_c, c1, c2, job->cutoff);
        double k_f64 = (double)a->kmers[job->k_idx];
        job->out[dist_idx] =
            job->ani ? (float)sko_ani_pois(j_index, k_f64) : (float)(1.0 - j_index);
    } else {
        sko_core_acc_dist(a, b, i, j, job->cutoff, &job->out[dist_idx * 2],
                          &job->out[dist_idx * 2 + 1]);
    }
}

/* mod.rs:77-127 */
static void self_chunk(void *ctx, size_t chunk_idx)
{
    const dense_job *job = (const dense_job *)ctx;
    size_t n = job->ref->n_samples;
    size_t start = chunk_idx * CHUNK_SIZE;
    size_t i = sko_calc_row_idx(start, n);
    size_t j = sko_calc_col_idx(start, i, n);
    for (size_t d = 0; d < CHUNK_SIZE && start + d < job->n_dist; ++d) {
        dense_pair(job, job->ref, job->ref, i, j, start + d);
        j += 1;
        if (j >= n) {
            i += 1;
            j = i + 1;
            if (i >= n - 1) break;
        }
    }
}

/* mod.rs:248-295 */
static void cross_chunk(void *ctx, size_t chunk_idx)
{
    const dense_job *job = (const dense_job *)ctx;
    size_t n = job->ref->n_samples, nq = job->query->n_samples;
    size_t start = chunk_idx * CHUNK_SIZE;
    size_t i = start / nq, j = start % nq; /* calc_query_indices, distance_matrix.rs:25 */
    for (size_t d = 0; d < CHUNK_SIZE && start + d < job->n_dist; ++d) {
        dense_pair(job, job->ref, job->query, i, j, start + d);
        j += 1;
        if (j >= nq) {
            i += 1;
            j = 0;
            if (i >= n) break;
        }
    }
}

static int check_coreacc(const sko_sketches *s, int dist_type, size_t k_idx)
{
    if (dist_type == SKO_COREACC && s->nk < 2) return -1; /* jaccard.rs:70-72 panics */
    if (dist_type == SKO_JACCARD && k_idx >= s->nk) return -2;
    return 0;
}

/* Benchmark helper: the same computation `repeat` times inside one thread pool (chunk
 * indices wrap), so that a short workload can be timed without paying thread start-up
 * per pass.  Results are simply overwritten each pass. */
typedef struct {
    const dense_job *job;
    size_t n_chunks;
} repeat_job;

static void self_chunk_wrapped(void *ctx, size_t chunk_idx)
{
    const repeat_job *r = (const repeat_job *)ctx;
    self_chunk((void *)r->job, chunk_idx % r->n_chunks);
}

int sko_self_dists_all_repeat(const sko_sketches *s, int dist_type, size_t k_idx, int ani,
                              double cutoff, int threads, int repeat, float *out)
{
    int rc = check_coreacc(s, dist_type, k_idx);
    if (rc) return rc;
    if (s->n_samples < 2 || repeat < 1) return 0;
    dense_job job = {s, NULL, dist_type, k_idx, ani, cutoff, out, NULL,
                     s->n_samples * (s->n_samples - 1) / 2};
    repeat_job r = {&job, (job.n_dist + CHUNK_SIZE - 1) / CHUNK_SIZE};
    run_chunks(self_chunk_wrapped, &r, r.n_chunks * (size_t)repeat, threads);
    return 0;
}

int sko_self_dists_all(const sko_sketches *s, int dist_type, size_t k_idx, int ani, double cutoff,
                       int threads, float *out)
{
    int rc = check_coreacc(s, dist_type, k_idx);
    if (rc) return rc;
    if (s->n_samples < 2) return 0;
    dense_job job = {s, NULL, dist_type, k_idx, ani, cutoff, out, NULL,
                     s->n_samples * (s->n_samples - 1) / 2};
    run_chunks(self_chunk, &job, (job.n_dist + CHUNK_SIZE - 1) / CHUNK_SIZE, threads);
    return 0;
}

int sko_cross_dists_all(const sko_sketches *ref, const sko_sketches *query, int dist_type,
                        size_t k_idx, int ani, double cutoff, int threads, float *out)
{
    int rc = check_coreacc(ref, dist_type, k_idx);
    if (rc) return rc;
    dense_job job = {ref, query, dist_type, k_idx, ani, cutoff, out, NULL,
                     ref->n_samples * query->n_samples};
    if (job.n_dist == 0) return 0;
    run_chunks(cross_chunk, &job, (job.n_dist + CHUNK_SIZE - 1) / CHUNK_SIZE, threads);
    return 0;
}

int sko_self_binmatch(const sko_sketches *s, int threads, uint32_t *out)
{
    if (s->n_samples < 2) return 0;
    dense_job job = {s, NULL, SKO_COREACC, 0, 0, 0.0, NULL, out,
                     s->n_samples * (s->n_samples - 1) / 2};
    run_chunks(self_chunk, &job, (job.n_dist + CHUNK_SIZE - 1) / CHUNK_SIZE, threads);
    return 0;
}

int sko_cross_binmatch(const sko_sketches *ref, const sko_sketches *query, int threads,
                       uint32_t *out)
{
    dense_job job = {ref, query, SKO_COREACC, 0, 0, 0.0, NULL, out,
                     ref->n_samples * query->n_samples};
    if (job.n_dist == 0) return 0;
    run_chunks(cross_chunk, &job, (job.n_dist + CHUNK_SIZE - 1) / CHUNK_SIZE, threads);
    return 0;
}

/* ------------------------------------------------------------------ */
/* Bounded max-heap: std::collections::BinaryHeap as driven by          */
/* push_heap (mod.rs:41-48) and into_sorted_vec (mod.rs:183-191,219).   */
/* Rust's std is not part of the reference tree; this restates the      */
/* published algorithm of alloc::collections::binary_heap (sift_up,     */
/* sift_down_to_bottom on pop, sift_down_range in into_sorted_vec).     */
/* Keys compare on d0 only (distance_matrix.rs:215-218,245-248).        */
/* ------------------------------------------------------------------ */

typedef struct {
    sko_sparse *data;
    size_t len;
} heap_t;

static size_t heap_sift_up(heap_t *h, size_t start, size_t pos)
{
    sko_sparse elt = h->data[pos];
    while (pos > start) {
        size_t parent = (pos - 1) / 2;
        if (elt.d0 <= h->data[parent].d0) break;
        h->data[pos] = h->data[parent];
        pos = parent;
    }
    h->data[pos] = elt;
    return pos;
}

static void heap_sift_down_range(heap_t *h, size_t pos, size_t end)
{
    sko_sparse elt = h->data[pos];
    size_t child = 2 * pos + 1;
    while (child <= (end >= 2 ? end - 2 : 0)) { /* end.saturating_sub(2) */
        child += (h->data[child].d0 <= h->data[child + 1].d0) ? 1 : 0;
        if (elt.d0 >= h->data[child].d0) {
            h->data[pos] = elt;
            return;
        }
        h->data[pos] = h->data[child];
        pos = child;
        child = 2 * pos + 1;
    }
    if (child == end - 1 && elt.d0 < h->data[child].d0) {
        h->data[pos] = h->data[child];
        pos = child;
    }
    h->data[pos] = elt;
}

static void heap_sift_down_to_bottom(heap_t *h, size_t pos)
{
    size_t end = h->len;
    size_t start = pos;
    sko_sparse elt = h->data[pos];
    size_t child = 2 * pos + 1;
    while (child <= (end >= 2 ? end - 2 : 0)) {
        child += (h->data[child].d0 <= h->data[child + 1].d0) ? 1 : 0;
        h->data[pos] = h->data[child];
        pos = child;
        child = 2 * pos + 1;
    }
    if (child == end - 1) {
        h->data[pos] = h->data[child];
        pos = child;
    }
    h->data[pos] = elt;
    heap_sift_up(h, start, pos);
}

static void heap_push(heap_t *h, sko_sparse item)
{
    size_t old_len = h->len;
    h->data[h->len++] = item;
    heap_sift_up(h, 0, old_len);
}

static void heap_pop(heap_t *h)
{
    sko_sparse item = h->data[--h->len];
    if (h->len > 0) {
        sko_sparse top = h->data[0];
        h->data[0] = item;
        (void)top;
        heap_sift_down_to_bottom(h, 0);
    }
}

/* mod.rs:41-48 */
static inline void push_heap(heap_t *h, sko_sparse item, size_t knn)
{
    if (h->len < knn || item.d0 < h->data[0].d0) {
        heap_push(h, item);
        if (h->len > knn) heap_pop(h);
    }
}

static void heap_into_sorted(heap_t *h)
{
    size_t end = h->len;
    while (end > 1) {
        end -= 1;
        sko_sparse t = h->data[0];
        h->data[0] = h->data[end];
        h->data[end] = t;
        heap_sift_down_range(h, 0, end);
    }
}

/* The heap alone, for tests that pin this restatement on hand-worked cases: candidates (ids[c], keys[c]), c = 0 .. n - 1, go
 * through push_heap (mod.rs:41-48) in that order into an empty BinaryHeap bounded at knn, then into_sorted_vec.
 * out: min(n, knn) items; returns their number. */
size_t sko_heap_replay(const uint64_t *ids, const float *keys, size_t n, size_t knn, sko_sparse *out)
{
    heap_t heap;
    heap.len = 0;
    heap.data = (sko_sparse *)malloc(sizeof(sko_sparse) * (knn + 1));
    for (size_t c = 0; c < n; ++c) {
        sko_sparse item;
        item.idx = ids[c];
        item.d0 = keys[c];
        item.d1 = 0.0f;
        push_heap(&heap, item, knn);
    }
    heap_into_sorted(&heap);
    for (size_t t = 0; t < heap.len; ++t) out[t] = heap.data[t];
    free(heap.data);
    return heap.len;
}

/* The same heap kept between calls (the column-window pipeline of the multi-device kNN hands heaps from device to device):
 * heap[0 .. *len) is a BinaryHeap's array (capacity knn + 1 items); the candidates are pushed through push_heap in the order
 * given.  sko_heap_sorted: into_sorted_vec in place. */
void sko_heap_feed(sko_sparse *heap, size_t *len, const uint64_t *ids, const float *keys, const float *d1, size_t n, size_t knn)
{
    heap_t h;
    h.data = heap;
    h.len = *len;
    for (size_t c = 0; c < n; ++c) {
        sko_sparse item;
        item.idx = ids[c];
        item.d0 = keys[c];
        item.d1 = d1 ? d1[c] : 0.0f;
        push_heap(&h, item, knn);
    }
    *len = h.len;
}

/* ... and which of the candidates the heap TOOK (accepted[c] = 1: push_heap's condition held, mod.rs:42): the accept log of the
 * decoupled column windows -- a heap that arrives later with more in it takes a subset of these. */
void sko_heap_feed_logged(sko_sparse *heap, size_t *len, const uint64_t *ids, const float *keys, const float *d1, size_t n, size_t knn,
                          uint8_t *accepted)
{
    heap_t h;
    h.data = heap;
    h.len = *len;
    for (size_t c = 0; c < n; ++c) {
        sko_sparse item;
        item.idx = ids[c];
        item.d0 = keys[c];
        item.d1 = d1 ? d1[c] : 0.0f;
        accepted[c] = (uint8_t)(h.len < knn || item.d0 < h.data[0].d0);
        push_heap(&h, item, knn);
    }
    *len = h.len;
}

void sko_heap_sorted(sko_sparse *heap, size_t len)
{
    heap_t h;
    h.data = heap;
    h.len = len;
    heap_into_sorted(&h);
}

/* Canonical rule: strict weak order on (d0, idx). */
static int canon_cmp(const void *a, const void *b)
{
    const sko_sparse *x = (const sko_sparse *)a, *y = (const sko_sparse *)b;
    if (x->d0 < y->d0) return -1;
    if (x->d0 > y->d0) return 1;
    if (x->idx < y->idx) return -1;
    if (x->idx > y->idx) return 1;
    return 0;
}

typedef struct {
    const sko_sketches *rows;  /* sample set iterated as rows */
    const sko_sketches *cands; /* sample set iterated as candidates */
    int self_mode;
    size_t knn;
    int dist_type;
    size_t k_idx;
    int ani;
    double cutoff;
    int tie_mode;
    sko_sparse *out;
} knn_job;

/* One output row: mod.rs:152-192 / :198-219 (self), :335-369 / :376-391 (cross). */
static void knn_row(void *ctx, size_t row)
{
    const knn_job *job = (const knn_job *)ctx;
    size_t n_cand = job->cands->n_samples;
    size_t knn = job->knn;
    sko_sparse *all = NULL;
    heap_t heap;
    heap.len = 0;
    if (job->tie_mode == SKO_TIES_CANONICAL) {
        all = (sko_sparse *)malloc(sizeof(sko_sparse) * (n_cand ? n_cand : 1));
        heap.data = NULL;
    } else {
        heap.data = (sko_sparse *)malloc(sizeof(sko_sparse) * (knn + 1));
    }
    size_t n_all = 0;
    for (size_t c = 0; c < n_cand; ++c) {
        if (job->self_mode && c == row) continue; /* mod.rs:157,203 */
        sko_sparse item;
        item.idx = c;
        item.d1 = 0.0f;
        if (job->dist_type == SKO_JACCARD) {
            /* completeness: c1 = rows[row], c2 = cands[c] (mod.rs:163-164,343-344) */
            int has_c = job->rows->completeness != NULL && job->cands->completeness != NULL;
            double c1 = has_c ? job->rows->completeness[row] : 0.0;
            double c2 = has_c ? job->cands->completeness[c] : 0.0;
            double jac = sko_jaccard_index(get_sketch_slice(job->rows, row, job->k_idx),
                                           get_sketch_slice(job->cands, c, job->k_idx),
                                           job->cands->sketchsize64, has_c, c1, c2, job->cutoff);
            double k_f64 = (double)job->cands->kmers[job->k_idx];
            item.d0 = job->ani ? (float)(1.0 - sko_ani_pois(jac, k_f64)) /* mod.rs:173-176 */
                               : (float)(1.0 - jac);
        } else if (job->self_mode) {
            sko_core_acc_dist(job->rows, job->rows, row, c, job->cutoff, &item.d0, &item.d1);
        } else {
            /* cross: core_acc_dist(ref, query, ri, qi, ..) mod.rs:377-385 */
            sko_core_acc_dist(job->cands, job->rows, c, row, job->cutoff, &item.d0, &item.d1);
        }
        if (all) {
            all[n_all++] = item;
        } else {
            push_heap(&heap, item, knn);
        }
    }
    sko_sparse *dst = job->out + row * knn;
    if (all) {
        qsort(all, n_all, sizeof(sko_sparse), canon_cmp);
        for (size_t t = 0; t < knn && t < n_all; ++t) dst[t] = all[t];
        free(all);
    } else {
        heap_into_sorted(&heap);
        for (size_t t = 0; t < heap.len; ++t) dst[t] = heap.data[t];
        free(heap.data);
    }
    if (job->dist_type == SKO_JACCARD && job->ani) { /* mod.rs:183-189 undo transform in f32 */
        for (size_t t = 0; t < knn; ++t) dst[t].d0 = 1.0f - dst[t].d0;
    }
}

int sko_self_dists_knn(const sko_sketches *s, size_t knn, int dist_type, size_t k_idx, int ani,
                       double cutoff, int tie_mode, int threads, sko_sparse *out)
{
    int rc = check_coreacc(s, dist_type, k_idx);
    if (rc) return rc;
    if (knn == 0 || knn >= s->n_samples) return -3; /* caller clamps, lib.rs:379-382 */
    memset(out, 0, sizeof(sko_sparse) * s->n_samples * knn);
    knn_job job = {s, s, 1, knn, dist_type, k_idx, ani, cutoff, tie_mode, out};
    run_chunks(knn_row, &job, s->n_samples, threads);
    return 0;
}

long sko_cross_dists_knn(const sko_sketches *ref, const sko_sketches *query, size_t knn,
                         int dist_type, size_t k_idx, int ani, double cutoff, int tie_mode,
                         int threads, sko_sparse *out)
{
    int rc = check_coreacc(ref, dist_type, k_idx);
    if (rc) return rc;
    if (ref->n_samples == 0 || query->n_samples == 0) return -4; /* mod.rs:318-323 panics */
    if (knn > ref->n_samples) knn = ref->n_samples;               /* mod.rs:325 */
    if (knn == 0) return -3;
    memset(out, 0, sizeof(sko_sparse) * query->n_samples * knn);
    knn_job job = {query, ref, 0, knn, dist_type, k_idx, ani, cutoff, tie_mode, out};
    run_chunks(knn_row, &job, query->n_samples, threads);
    return (long)knn;
}

/* ---- mod.rs:399-553: self_dists_knn_precluster ----
 * Inverted::any_shared_bins (src/inverted.rs:259-268) restated without the index: sample j
 * (ski order) is a candidate of the query sketch iff some bin position holds the same u16
 * value in both -- exactly the union of index[bin][value] over the bins.  Candidates are
 * visited in ascending ski order, as RoaringBitmap::iter yields them. */
typedef struct {
    const sko_sketches *s;
    const uint16_t *skq;          /* [n][skq_stride], ski order */
    size_t skq_stride;
    const size_t *ski_of_skd;     /* skq_index_lookup: skd index -> ski index */
    const size_t *skd_of_ski;     /* skd_index_from_ski */
    size_t knn, k_idx;
    int ani;
    double cutoff;
    int retain_mode, tie_mode;
    sko_sparse *out;
} pre_job;

static float pre_key(const pre_job *job, size_t i, size_t j)
{
    const sko_sketches *s = job->s;
    int has_c = s->completeness != NULL;
    double c1 = has_c ? s->completeness[i] : 0.0, c2 = has_c ? s->completeness[j] : 0.0;
    double jac = sko_jaccard_index(get_sketch_slice(s, i, job->k_idx), get_sketch_slice(s, j, job->k_idx),
                                   s->sketchsize64, has_c, c1, c2, job->cutoff);
    double k_f64 = (double)s->kmers[job->k_idx];
    return job->ani ? (float)(1.0 - sko_ani_pois(jac, k_f64)) : (float)(1.0 - jac);
}

static size_t pre_collect(const pre_job *job, size_t i, int brute, sko_sparse *all, heap_t *heap)
{
    const size_t n = job->s->n_samples;
    const size_t ski_i = job->ski_of_skd[i];
    const uint16_t *qi = job->skq + ski_i * job->skq_stride;
    size_t n_all = 0;
    for (size_t c = 0; c < n; ++c) {
        size_t skd_j;
        if (brute) {                       /* mod.rs:500-523: j over skd order, skip i */
            if (c == i) continue;
            skd_j = c;
        } else {                           /* mod.rs:455-485: j over ski order, skip self */
            if (c == ski_i) continue;
            const uint16_t *qj = job->skq + c * job->skq_stride;
            int shared = 0;
            for (size_t b = 0; b < job->skq_stride && !shared; ++b) shared = qi[b] == qj[b];
            if (!shared) continue;
            skd_j = job->skd_of_ski[c];
        }
        sko_sparse item;
        item.idx = skd_j;
        item.d0 = pre_key(job, i, skd_j);
        item.d1 = 0.0f;
        if (all) all[n_all] = item; else push_heap(heap, item, job->knn);
        ++n_all;
    }
    return n_all;
}

static void pre_row(void *ctx, size_t i)
{
    const pre_job *job = (const pre_job *)ctx;
    const size_t n = job->s->n_samples, knn = job->knn;
    sko_sparse *dst = job->out + i * knn;
    sko_sparse *all = NULL;
    heap_t heap;
    heap.len = 0;
    heap.data = NULL;
    if (job->tie_mode == SKO_TIES_CANONICAL) all = (sko_sparse *)malloc(sizeof(sko_sparse) * (n ? n : 1));
    else heap.data = (sko_sparse *)malloc(sizeof(sko_sparse) * (knn + 1));
    size_t found = pre_collect(job, i, 0, all, &heap);
    if (found == 0 && job->retain_mode == 1) {          /* Singleton, mod.rs:489-497 */
        dst[0].idx = i; dst[0].d0 = 0.0f; dst[0].d1 = 0.0f;
        for (size_t t = 1; t < knn; ++t) { dst[t].idx = i; dst[t].d0 = 1.0f; dst[t].d1 = 0.0f; }
        free(all); free(heap.data);
        return;
    }
    if (found == 0 && job->retain_mode == 2) {          /* Bruteforce, mod.rs:498-525 */
        heap.len = 0;
        found = pre_collect(job, i, 1, all, &heap);
    }
    size_t got;
    if (all) {
        qsort(all, found, sizeof(sko_sparse), canon_cmp);
        got = found < knn ? found : knn;
        for (size_t t = 0; t < got; ++t) dst[t] = all[t];
        free(all);
    } else {
        heap_into_sorted(&heap);
        got = heap.len;
        for (size_t t = 0; t < got; ++t) dst[t] = heap.data[t];
        free(heap.data);
    }
    if (job->ani) for (size_t t = 0; t < got; ++t) dst[t].d0 = 1.0f - dst[t].d0;   /* mod.rs:529-534 */
    for (size_t t = got; t < knn; ++t) { dst[t].idx = i; dst[t].d0 = 1.0f; dst[t].d1 = 0.0f; }  /* :535-546 */
}

int sko_self_dists_knn_precluster(const sko_sketches *s, const uint16_t *skq, size_t skq_stride,
                                  const size_t *ski_of_skd, size_t knn, size_t k_idx, int ani,
                                  double cutoff, int retain_mode, int tie_mode, int threads,
                                  sko_sparse *out)
{
    if (k_idx >= s->nk) return -2;
    if (knn == 0 || knn >= s->n_samples) return -3; /* caller clamps, lib.rs:737-740 */
    size_t n = s->n_samples;
    size_t *skd_of_ski = (size_t *)calloc(n ? n : 1, sizeof(size_t));
    for (size_t i = 0; i < n; ++i) skd_of_ski[ski_of_skd[i]] = i;   /* mod.rs:436-439 */
    pre_job job = {s, skq, skq_stride, ski_of_skd, skd_of_ski, knn, k_idx, ani, cutoff, retain_mode, tie_mode, out};
    run_chunks(pre_row, &job, n, threads);
    free(skd_of_ski);
    return 0;
}

/* `sketchlib inverted precluster --count` (src/lib.rs:700-712, inverted.rs:271-300): pairs that
 * share at least one bin. */
uint64_t sko_prefilter_pair_count(const uint16_t *skq, size_t n, size_t skq_stride)
{
    uint64_t count = 0;
    for (size_t i = 0; i < n; ++i) {
        for (size_t j = i + 1; j < n; ++j) {
            const uint16_t *a = skq + i * skq_stride, *b = skq + j * skq_stride;
            for (size_t x = 0; x < skq_stride; ++x) {
                if (a[x] == b[x]) { ++count; break; }
            }
        }
    }
    return count;
}
