/*
 * locrec_oracle.c -- CPU restatement of the reference's two hot paths.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library links, loads or
 * calls this file: it may be used only by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg, and there only
 * as the checker / the timed CPU port -- never as a product code path.
 *
 * What it restates (all paths relative to /root/reference/recommender/src/
 * main/scala/com/github/tashoyan/recommender/):
 *   knn/Distance.scala:7-16                  cosineSimilarity, vectorLength
 *   knn/KnnRecommender.scala:17-20,27-49,51-70,76-96
 *   stochastic/StochasticRecommender.scala:33-54,66-141
 * and one third-party routine that is NOT in /root/reference:
 *   org.apache.spark:spark-mllib-local_2.12:3.1.2  BLAS.dot(sparse, sparse)
 *   (called at Distance.scala:8 through SparseVector.dot) -- the published
 *   two-pointer merge, restated in oracle_sparse_dot() below.
 *
 * Parity status:
 *   - oracle_vector_length / oracle_cosine are PINNED by the 8 known-answer
 *     tests of knn/DistanceTest.scala:10-60 (tests/golden/distance_kats.json).
 *   - oracle_sg_recommend is PINNED by stochastic/StochasticRecommenderTest
 *     .scala:11-21,53-58,76-81,85-94 (tests/golden/sg_kats.json), bit-exact.
 *   - oracle_knn_similar / oracle_knn_recommend: the reference has no test of
 *     KnnRecommender at all, so top-K / outer-join fill / aggregation are
 *     "parity unpinned" by the reference; they are pinned only by hand-derived
 *     fixtures (tests/golden/knn_handmade.json).
 *
 * The reference cannot be run in the build container (no JVM, Scala, Maven
 * or Spark), so there is no oracle/_ref build.
 *
 * Build: gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC (see Makefile).
 * -ffp-contract=off matters: the JVM never fuses a*b+c.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_OK 0
#define ORACLE_E_INVALID_ARG 1
#define ORACLE_E_NOT_FOUND 2
#define ORACLE_E_ARITHMETIC 5   /* ArithmeticException (RatingVectorsBuilder.checkedCast) */

/* ------------------------------------------------------------------ */
/* a1: Distance.vectorLength, knn/Distance.scala:11-16                 */
/* values.map(v => v * v).sum is a left fold from 0.0.                 */
double oracle_vector_length(const double *values, int64_t nnz)
{
    double sum = 0.0;
    for (int64_t i = 0; i < nnz; ++i) {
        double sq = values[i] * values[i];
        sum = sum + sq;
    }
    return sqrt(sum);
}

/* Spark 3.1.2 BLAS.dot(x: SparseVector, y: SparseVector): two-pointer merge
 * over ascending indices, "y catching x", sum += x(kx) * y(ky) in x order.  */
double oracle_sparse_dot(int64_t nnzx, const int32_t *xi, const double *xv,
                         int64_t nnzy, const int32_t *yi, const double *yv)
{
    int64_t kx = 0, ky = 0;
    double sum = 0.0;
    while (kx < nnzx && ky < nnzy) {
        int32_t ix = xi[kx];
        while (ky < nnzy && yi[ky] < ix)
            ++ky;
        if (ky < nnzy && yi[ky] == ix) {
            double prod = xv[kx] * yv[ky];
            sum = sum + prod;
            ++ky;
        }
        ++kx;
    }
    return sum;
}

/* a2: Distance.cosineSimilarity, knn/Distance.scala:7-9:
 * (v1 dot v2) / (len(v1) * len(v2)); one multiply, one divide.          */
double oracle_cosine(int64_t n1, const int32_t *i1, const double *v1,
                     int64_t n2, const int32_t *i2, const double *v2)
{
    double dot = oracle_sparse_dot(n1, i1, v1, n2, i2, v2);
    double l1 = oracle_vector_length(v1, n1);
    double l2 = oracle_vector_length(v2, n2);
    double den = l1 * l2;
    return dot / den;
}

/* ------------------------------------------------------------------ */
/* KNN                                                                  */

typedef struct {
    double sim;
    int64_t id;
    int64_t row;
} nb_t;

/* Our definition of the order Spark leaves undefined:
 * similarity descending, then person_id ascending.                     */
static int nb_cmp(const void *a, const void *b)
{
    const nb_t *x = (const nb_t *)a, *y = (const nb_t *)b;
    if (x->sim > y->sim) return -1;
    if (x->sim < y->sim) return 1;
    if (x->id < y->id) return -1;
    if (x->id > y->id) return 1;
    return 0;
}

static int64_t find_row(int64_t n, const int64_t *ids, int64_t id)
{
    for (int64_t i = 0; i < n; ++i)
        if (ids[i] == id)
            return i;
    return -1;
}

/* KnnRecommender.scala:17-20 (constructor require()s). */
int32_t oracle_knn_check_params(double pw, double cw, int64_t k)
{
    if (!(pw > 0 && pw < 1.0)) return ORACLE_E_INVALID_ARG;
    if (!(cw > 0 && cw < 1.0)) return ORACLE_E_INVALID_ARG;
    if (!(pw + cw == 1.0)) return ORACLE_E_INVALID_ARG;
    if (!(k > 0)) return ORACLE_E_INVALID_ARG;
    return ORACLE_OK;
}

/* a3 + a4: findSimilarPersons0 (KnnRecommender.scala:76-96) for both
 * families, outer join with fill 0.0 (:39-40), ps*pw + cs*cw (:43-45),
 * orderBy desc limit K (:47-48).
 *
 * A person is "present" in a family iff its row in that family is
 * non-empty (the builder never emits empty vectors,
 * RatingVectorsBuilder.scala:71-79); the query must be present in both
 * (:83 throws otherwise).
 *
 * Returns all neighbours (<= min(k, n-1)) into a freshly sorted nb_t array. */
static int32_t knn_similar_core(
    int64_t n, const int64_t *ids,
    const int64_t *prp, const int32_t *pidx, const double *pval,
    const int64_t *crp, const int32_t *cidx, const double *cval,
    int64_t qrow, double pw, double cw, int64_t k,
    nb_t *out, int64_t *out_count)
{
    const int64_t qpn = prp[qrow + 1] - prp[qrow];
    const int64_t qcn = crp[qrow + 1] - crp[qrow];
    const int32_t *qpi = pidx + prp[qrow];
    const double *qpv = pval + prp[qrow];
    const int32_t *qci = cidx + crp[qrow];
    const double *qcv = cval + crp[qrow];
    int64_t m = 0;
    for (int64_t r = 0; r < n; ++r) {
        if (ids[r] == ids[qrow]) /* where person_id =!= personId (:89) */
            continue;
        double ps = 0.0, cs = 0.0;
        int have = 0;
        int64_t pn = prp[r + 1] - prp[r];
        if (pn > 0) {
            /* cosineSimilarity(vector, personRatingVector): candidate is x (:86) */
            double s = oracle_cosine(pn, pidx + prp[r], pval + prp[r], qpn, qpi, qpv);
            if (s > 0) { ps = s; have = 1; } /* :91 */
        }
        int64_t cn = crp[r + 1] - crp[r];
        if (cn > 0) {
            double s = oracle_cosine(cn, cidx + crp[r], cval + crp[r], qcn, qci, qcv);
            if (s > 0) { cs = s; have = 1; }
        }
        if (!have)
            continue;
        double a = ps * pw;
        double b = cs * cw;
        out[m].sim = a + b;
        out[m].id = ids[r];
        out[m].row = r;
        ++m;
    }
    qsort(out, (size_t)m, sizeof(nb_t), nb_cmp);
    if (m > k) m = k;
    *out_count = m;
    return ORACLE_OK;
}

int32_t oracle_knn_similar(
    int64_t n, const int64_t *ids,
    const int64_t *prp, const int32_t *pidx, const double *pval,
    const int64_t *crp, const int32_t *cidx, const double *cval,
    int64_t person_id, double pw, double cw, int64_t k,
    int64_t *out_ids, double *out_sims, int64_t *inout_count)
{
    int32_t st = oracle_knn_check_params(pw, cw, k);
    if (st) return st;
    int64_t qrow = find_row(n, ids, person_id);
    if (qrow < 0 || prp[qrow + 1] == prp[qrow] || crp[qrow + 1] == crp[qrow])
        return ORACLE_E_NOT_FOUND; /* "No such person" :83 */
    nb_t *nb = (nb_t *)malloc(sizeof(nb_t) * (size_t)(n > 0 ? n : 1));
    int64_t m = 0;
    knn_similar_core(n, ids, prp, pidx, pval, crp, cidx, cval, qrow, pw, cw, k, nb, &m);
    int64_t cap = *inout_count;
    int64_t w = m < cap ? m : cap;
    for (int64_t i = 0; i < w; ++i) {
        out_ids[i] = nb[i].id;
        out_sims[i] = nb[i].sim;
    }
    *inout_count = m;
    free(nb);
    return ORACLE_OK;
}

typedef struct {
    int64_t place;
    double wr;   /* rating * similarity */
    double s;    /* similarity */
    int64_t seq; /* position in neighbour-rank order: makes the sort stable */
} trip_t;

static int trip_cmp(const void *a, const void *b)
{
    const trip_t *x = (const trip_t *)a, *y = (const trip_t *)b;
    if (x->place != y->place) return x->place < y->place ? -1 : 1;
    return x->seq < y->seq ? -1 : (x->seq > y->seq ? 1 : 0);
}

/* a5: makeRecommendations0, KnnRecommender.scala:51-70.
 * ratings are given as CSR aligned with ids: rrp[n+1], rplace[], rrating[]
 * (the COO frame (person_id, place_id, rating: Long) grouped by person).
 * Both sums run only over neighbours that rated the place (:58-68).
 * Output sorted by place_id ascending (the reference leaves it unordered).
 * Sums are taken in neighbour-rank order (our fixed order; Spark's is
 * undefined, so parity on estimated_rating is 1e-6 relative).            */
int32_t oracle_knn_recommend(
    int64_t n, const int64_t *ids,
    const int64_t *prp, const int32_t *pidx, const double *pval,
    const int64_t *crp, const int32_t *cidx, const double *cval,
    const int64_t *rrp, const int64_t *rplace, const int64_t *rrating,
    int64_t person_id, double pw, double cw, int64_t k,
    int64_t *out_places, double *out_ratings, int64_t *inout_count)
{
    int32_t st = oracle_knn_check_params(pw, cw, k);
    if (st) return st;
    int64_t qrow = find_row(n, ids, person_id);
    if (qrow < 0 || prp[qrow + 1] == prp[qrow] || crp[qrow + 1] == crp[qrow])
        return ORACLE_E_NOT_FOUND;
    nb_t *nb = (nb_t *)malloc(sizeof(nb_t) * (size_t)(n > 0 ? n : 1));
    int64_t m = 0;
    knn_similar_core(n, ids, prp, pidx, pval, crp, cidx, cval, qrow, pw, cw, k, nb, &m);
    int64_t nt = 0;
    for (int64_t i = 0; i < m; ++i)
        nt += rrp[nb[i].row + 1] - rrp[nb[i].row];
    trip_t *tr = (trip_t *)malloc(sizeof(trip_t) * (size_t)(nt > 0 ? nt : 1));
    int64_t t = 0;
    for (int64_t i = 0; i < m; ++i) {
        int64_t r = nb[i].row;
        for (int64_t e = rrp[r]; e < rrp[r + 1]; ++e) {
            tr[t].place = rplace[e];
            tr[t].wr = (double)rrating[e] * nb[i].sim; /* :59 */
            tr[t].s = nb[i].sim;
            tr[t].seq = t;
            ++t;
        }
    }
    qsort(tr, (size_t)nt, sizeof(trip_t), trip_cmp);
    int64_t cap = *inout_count, outn = 0;
    for (int64_t i = 0; i < nt;) {
        int64_t j = i;
        double ws = 0.0, ss = 0.0;
        while (j < nt && tr[j].place == tr[i].place) {
            ws = ws + tr[j].wr;
            ss = ss + tr[j].s;
            ++j;
        }
        if (outn < cap) {
            out_places[outn] = tr[i].place;
            out_ratings[outn] = ws / ss; /* :67 */
        }
        ++outn;
        i = j;
    }
    *inout_count = outn;
    free(tr);
    free(nb);
    return ORACLE_OK;
}

/* Batched form used as the multi-threaded CPU baseline (bench.py
 * cpu_baseline leg) and by the batch parity tests: every query row in
 * qrows[] against all persons, top-k each.  OpenMP over queries.
 * out_ids/out_sims are [nq * k], padded with id -1 / sim 0.              */
int32_t oracle_knn_similar_batch(
    int64_t n, const int64_t *ids,
    const int64_t *prp, const int32_t *pidx, const double *pval,
    const int64_t *crp, const int32_t *cidx, const double *cval,
    int64_t nq, const int64_t *qrows, double pw, double cw, int64_t k,
    int64_t *out_ids, double *out_sims, int64_t *out_counts, int32_t nthreads)
{
    int32_t st = oracle_knn_check_params(pw, cw, k);
    if (st) return st;
    int bad = 0;
#pragma omp parallel num_threads(nthreads > 0 ? nthreads : 1)
    {
        nb_t *nb = (nb_t *)malloc(sizeof(nb_t) * (size_t)(n > 0 ? n : 1));
#pragma omp for schedule(dynamic, 1)
        for (int64_t q = 0; q < nq; ++q) {
            int64_t qrow = qrows[q];
            if (qrow < 0 || qrow >= n || prp[qrow + 1] == prp[qrow] ||
                crp[qrow + 1] == crp[qrow]) {
                bad = 1;
                out_counts[q] = -1;
                continue;
            }
            int64_t m = 0;
            knn_similar_core(n, ids, prp, pidx, pval, crp, cidx, cval, qrow, pw, cw, k, nb, &m);
            out_counts[q] = m;
            for (int64_t i = 0; i < k; ++i) {
                out_ids[q * k + i] = i < m ? nb[i].id : -1;
                out_sims[q * k + i] = i < m ? nb[i].sim : 0.0;
            }
        }
        free(nb);
    }
    return bad ? ORACLE_E_NOT_FOUND : ORACLE_OK;
}

/* ------------------------------------------------------------------ */
/* SG: StochasticRecommender.scala                                      */

static int i64_cmp(const void *a, const void *b)
{
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

static int64_t lower_bound_i64(const int64_t *a, int64_t n, int64_t key)
{
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t mid = lo + (hi - lo) / 2;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/* a7-a11.  vertexes = distinct(source_id U target_id) (:42-49), kept in
 * ascending id order; x0 = 1.0 / vertexCount (:51-54);
 * calcNextX (:108-128): sigma[t] = sum x[s]*w accumulated in EDGE-LIST
 * order (the order that reproduces the reference KATs bit for bit),
 * x' = u*alpha + sigma*(1-alpha);
 * isConverged (:130-141): sum over vertices (ascending id) of diff^2
 * <= epsilon^2;  step (:92-106).
 * Output: rows with id != vertex and probability > 0 (:84-88), ascending id.
 * out_iterations = the 0-based counter the reference prints; out_converged
 * = 1 if it stopped on convergence, 0 if on maxIterations.               */
int32_t oracle_sg_recommend(
    int64_t ne, const int64_t *src, const int64_t *dst, const double *w,
    int64_t vertex_id, double alpha, double epsilon, int64_t max_iterations,
    int64_t *out_ids, double *out_probs, int64_t *inout_count,
    int64_t *out_iterations, int32_t *out_converged)
{
    if (!(epsilon >= 0)) return ORACLE_E_INVALID_ARG;   /* :33 */
    if (max_iterations < 0) return ORACLE_E_INVALID_ARG; /* :34 */
    int64_t *v = (int64_t *)malloc(sizeof(int64_t) * (size_t)(2 * ne + 1));
    for (int64_t e = 0; e < ne; ++e) { v[2 * e] = src[e]; v[2 * e + 1] = dst[e]; }
    qsort(v, (size_t)(2 * ne), sizeof(int64_t), i64_cmp);
    int64_t nv = 0;
    for (int64_t i = 0; i < 2 * ne; ++i)
        if (i == 0 || v[i] != v[i - 1]) v[nv++] = v[i];
    int64_t target = lower_bound_i64(v, nv, vertex_id);
    if (target >= nv || v[target] != vertex_id) { free(v); return ORACLE_E_NOT_FOUND; } /* :70 */

    int32_t *cs = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ne + 1));
    int32_t *ct = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ne + 1));
    for (int64_t e = 0; e < ne; ++e) {
        cs[e] = (int32_t)lower_bound_i64(v, nv, src[e]);
        ct[e] = (int32_t)lower_bound_i64(v, nv, dst[e]);
    }
    double *x = (double *)malloc(sizeof(double) * (size_t)nv);
    double *nx = (double *)malloc(sizeof(double) * (size_t)nv);
    double *sigma = (double *)malloc(sizeof(double) * (size_t)nv);
    const double x0 = 1.0 / (double)nv;
    for (int64_t i = 0; i < nv; ++i) x[i] = x0;
    const double eps2 = epsilon * epsilon;   /* :40 */
    const double one_minus_alpha = 1 - alpha; /* :121 */
    int64_t it = 0;
    int32_t converged = 0;
    for (;;) {
        if (it >= max_iterations) break; /* :93-95 */
        for (int64_t i = 0; i < nv; ++i) sigma[i] = 0.0;
        for (int64_t e = 0; e < ne; ++e) {
            double acc = x[cs[e]] * w[e]; /* :112 */
            sigma[ct[e]] = sigma[ct[e]] + acc;
        }
        double d2 = 0.0;
        for (int64_t i = 0; i < nv; ++i) {
            double u = (i == target) ? 1.0 : 0.0;
            double a = u * alpha;
            double b = sigma[i] * one_minus_alpha;
            nx[i] = a + b; /* :118-122 */
            double diff = nx[i] - x[i];
            double sq = diff * diff;
            d2 = d2 + sq;
        }
        double *tmp = x; x = nx; nx = tmp;
        if (d2 <= eps2) { converged = 1; break; } /* :99-101 */
        ++it;
    }
    int64_t cap = *inout_count, outn = 0;
    for (int64_t i = 0; i < nv; ++i) {
        if (i != target && x[i] > 0) {
            if (outn < cap) { out_ids[outn] = v[i]; out_probs[outn] = x[i]; }
            ++outn;
        }
    }
    *inout_count = outn;
    if (out_iterations) *out_iterations = it;
    if (out_converged) *out_converged = converged;
    free(x); free(nx); free(sigma); free(cs); free(ct); free(v);
    return ORACLE_OK;
}

/* Fixed-sweep variant for the CPU baseline timing: run exactly `sweeps`
 * applications of calcNextX on a pre-compacted graph, multi-threaded over
 * targets (CSR by target, in-row order = edge-list order so each sigma[t] is
 * bitwise what the scalar loop above gives).  Returns sum(x) as a checksum. */
double oracle_sg_sweeps_csr(
    int64_t nv, const int64_t *rowptr, const int32_t *col, const double *w,
    int64_t target, double alpha, int64_t sweeps, double *x_inout, int32_t nthreads)
{
    double *nx = (double *)malloc(sizeof(double) * (size_t)nv);
    double *x = x_inout;
    const double oma = 1 - alpha;
    /* rows that have edges are spread over the threads in small chunks (a few rows carry most of
     * the edges); rows without edges cost one multiply each and are done in a plain loop */
    int64_t nlive = 0;
    int64_t *live = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nv > 0 ? nv : 1));
    for (int64_t t = 0; t < nv; ++t)
        if (rowptr[t + 1] > rowptr[t]) live[nlive++] = t;
    for (int64_t s = 0; s < sweeps; ++s) {
        for (int64_t t = 0; t < nv; ++t) {
            double u = (t == target) ? 1.0 : 0.0;
            double a = u * alpha;
            double b = 0.0 * oma;
            nx[t] = a + b;
        }
#pragma omp parallel for schedule(dynamic, 8) num_threads(nthreads > 0 ? nthreads : 1)
        for (int64_t i = 0; i < nlive; ++i) {
            const int64_t t = live[i];
            double sg = 0.0;
            for (int64_t e = rowptr[t]; e < rowptr[t + 1]; ++e) {
                double acc = x[col[e]] * w[e];
                sg = sg + acc;
            }
            double u = (t == target) ? 1.0 : 0.0;
            double a = u * alpha;
            double b = sg * oma;
            nx[t] = a + b;
        }
        double *tmp = x; x = nx; nx = tmp;
    }
    if (x != x_inout) { memcpy(x_inout, x, sizeof(double) * (size_t)nv); nx = x; }
    double cks = 0.0;
    for (int64_t i = 0; i < nv; ++i) cks += x_inout[i];
    free(nx);
    free(live);
    return cks;
}

/* ================================================================== */
/* SURVEY 8f "next" rows: the producers of the two hot paths' inputs.  */
/* (Still test infrastructure; same rules as above.)                   */
/* ================================================================== */

typedef struct { int64_t p, e, at; } visit_key;

static int visit_cmp(const void *a, const void *b)
{
    const visit_key *x = (const visit_key *)a, *y = (const visit_key *)b;
    if (x->p != y->p) return x->p < y->p ? -1 : 1;
    if (x->e != y->e) return x->e < y->e ? -1 : 1;
    return x->at < y->at ? -1 : (x->at > y->at ? 1 : 0);
}

/* f-2: RatingsBuilder.calcRatings, knn/RatingsBuilder.scala:32-48.
 * groupBy(person_id, entity).agg(count("*")) (:38-40); rank() over
 * (partition by person_id order by rating desc) (:42-45) - SQL rank(): 1 +
 * the number of rows of the partition that sort strictly before, so ties
 * share a rank and a tie straddling topN is kept whole; where(rank <= topN)
 * (:46).  The reference has no test of it: "parity unpinned"; restated from
 * the SQL definition (an O(m^2) count per person - deliberately not the
 * sort-based method of the shipped code).  Rows come out ordered by (person,
 * entity); Spark's order is undefined.  Outputs need room for n rows.       */
int32_t oracle_calc_ratings(int64_t n, const int64_t *person, const int64_t *entity, int64_t top_n,
                            int64_t *out_person, int64_t *out_entity, int64_t *out_rating, int64_t *out_count)
{
    *out_count = 0;
    if (n <= 0) return ORACLE_OK;
    visit_key *k = (visit_key *)malloc(sizeof(visit_key) * (size_t)n);
    int64_t *cnt = (int64_t *)malloc(sizeof(int64_t) * (size_t)n);
    for (int64_t i = 0; i < n; ++i) { k[i].p = person[i]; k[i].e = entity[i]; k[i].at = i; }
    qsort(k, (size_t)n, sizeof(visit_key), visit_cmp);
    int64_t g = 0;                       /* groups, compacted to the front of k */
    for (int64_t i = 0; i < n; ) {
        int64_t j = i;
        while (j < n && k[j].p == k[i].p && k[j].e == k[i].e) ++j;
        k[g] = k[i]; cnt[g] = j - i; ++g;
        i = j;
    }
    int64_t m = 0;
    for (int64_t a = 0; a < g; ) {
        int64_t b = a;
        while (b < g && k[b].p == k[a].p) ++b;
        for (int64_t i = a; i < b; ++i) {
            int64_t before = 0;
            for (int64_t j = a; j < b; ++j) if (cnt[j] > cnt[i]) ++before;
            if (1 + before <= top_n) {
                out_person[m] = k[i].p; out_entity[m] = k[i].e; out_rating[m] = cnt[i]; ++m;
            }
        }
        a = b;
    }
    *out_count = m;
    free(k); free(cnt);
    return ORACLE_OK;
}

/* f-2: RatingVectorsBuilder.calcRatingVectors, knn/RatingVectorsBuilder.scala:10-25,52-84.
 * size = checkedCast(max entity id) + 1 (:27-41: an id outside Int range is an
 * ArithmeticException -> ORACLE_E_ARITHMETIC, *out_size = the offending id);
 * per person a TreeSet ordered by index (:43-50, 68-72): ascending indices, an
 * index that is already present is NOT replaced (first one wins; here "first"
 * = input order, Spark's is partition order); value = rating.toDouble (:69).
 * Persons ascending.  out_ids / out_rowptr need n (+1) entries, out_idx / out_val n. */
int32_t oracle_calc_rating_vectors(int64_t n, const int64_t *person, const int64_t *entity, const int64_t *rating,
                                   int64_t *out_ids, int64_t *out_rowptr, int32_t *out_idx, double *out_val,
                                   int64_t *out_npersons, int64_t *out_nnz, int64_t *out_size)
{
    *out_npersons = 0; *out_nnz = 0; *out_size = 0; out_rowptr[0] = 0;
    if (n <= 0) return ORACLE_OK;
    int64_t max_id = entity[0];
    for (int64_t i = 0; i < n; ++i) {
        /* the map over every row (:69) casts each id; max(id) (:31-34) is cast first */
        if (entity[i] > max_id) max_id = entity[i];
    }
    if (max_id > 2147483647LL || max_id < -2147483648LL) { *out_size = max_id; return ORACLE_E_ARITHMETIC; }
    for (int64_t i = 0; i < n; ++i)
        if (entity[i] < -2147483648LL) { *out_size = entity[i]; return ORACLE_E_ARITHMETIC; }
    /* new SparseVector(size, indices, values) (:74-77) then runs Spark's own require()s (third party,
     * spark-mllib-local_2.12 3.1.2): size >= 0 - checkedCast(max) + 1 wraps for Int.MaxValue - and
     * no negative index */
    if (max_id == 2147483647LL) { *out_size = max_id; return ORACLE_E_INVALID_ARG; }
    for (int64_t i = 0; i < n; ++i)
        if (entity[i] < 0) { *out_size = entity[i]; return ORACLE_E_INVALID_ARG; }
    visit_key *k = (visit_key *)malloc(sizeof(visit_key) * (size_t)n);
    for (int64_t i = 0; i < n; ++i) { k[i].p = person[i]; k[i].e = entity[i]; k[i].at = i; }
    qsort(k, (size_t)n, sizeof(visit_key), visit_cmp);
    int64_t np = 0, nz = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (i == 0 || k[i].p != k[i - 1].p) { out_ids[np] = k[i].p; out_rowptr[np] = nz; ++np; }
        else if (k[i].e == k[i - 1].e) continue;
        out_idx[nz] = (int32_t)k[i].e;
        out_val[nz] = (double)rating[k[i].at];
        ++nz;
    }
    out_rowptr[np] = nz;
    *out_npersons = np; *out_nnz = nz; *out_size = max_id + 1;
    free(k);
    return ORACLE_OK;
}

/* f-2: StochasticGraphBuilder.buildWithBalancedWeights, stochastic/StochasticGraphBuilder.scala:8-28:
 * every family's weight times its beta (:14,23), families concatenated in the
 * given order (union, :18-19).  Pinned by StochasticGraphBuilderTest.scala
 * (tests/golden/graph_builder_kat.json).  Outputs need sum(counts) entries.   */
int32_t oracle_balanced_edges(int32_t nfam, const double *betas, const int64_t *counts,
                              const int64_t *const *src, const int64_t *const *dst, const double *const *w,
                              int64_t *out_src, int64_t *out_dst, double *out_w)
{
    if (nfam <= 0) return ORACLE_E_INVALID_ARG;
    int64_t at = 0;
    for (int32_t f = 0; f < nfam; ++f)
        for (int64_t i = 0; i < counts[f]; ++i, ++at) {
            out_src[at] = src[f][i]; out_dst[at] = dst[f][i];
            double bw = w[f][i] * betas[f];
            out_w[at] = bw;
        }
    return ORACLE_OK;
}

/* f-4: Location.distanceMeters, Location.scala:30-43 (haversine).  sin / cos / asin /
 * sqrt / toRadians are org.apache.commons:commons-math3 FastMath there - third
 * party, not under /root/reference, documented as accurate to < 1 ulp; libm here.
 * Pinned by LocationTest.scala:8-27: 0 for the same location, 745 +- 5 m for the
 * Moscow pair, bitwise commutative.  Differences of a few ulp change a
 * `<= 100 m` decision only for pairs within ~1e-10 m of the threshold.          */
static const double kEarthRadiusMeters = 6371.0 * 1000.0;   /* Location.scala:28 */

static double to_radians(double deg) { return deg / 180.0 * M_PI; }   /* java.lang.Math.toRadians' formula */

static double haversine(double theta)                                  /* Location.scala:40-43 */
{
    double s = sin(theta / 2);
    return s * s;
}

double oracle_distance_meters(double lat1d, double lon1d, double lat2d, double lon2d)
{
    double lat1 = to_radians(lat1d), lat2 = to_radians(lat2d);
    double lon1 = to_radians(lon1d), lon2 = to_radians(lon2d);
    double h1 = haversine(lat2 - lat1);
    double cc = cos(lat1) * cos(lat2);
    double h2 = cc * haversine(lon2 - lon1);
    double hav = h1 + h2;
    double r2 = kEarthRadiusMeters * 2;
    return r2 * asin(sqrt(hav));
}

static int location_ok(double lat, double lon)   /* Location.scala:7-8 require()s; NaN fails them */
{
    return lat >= -90.0 && lat <= 90.0 && lon >= -180.0 && lon <= 180.0;
}

/* f-4: PlaceVisits.calcPlaceVisits, PlaceVisits.scala:11-46: visits with timestamp >=
 * visits_from (:24) joined with the places of the same region (:31), kept where
 * distanceMeters <= 100 (:15-21, :127) - the plain cross join the reference runs.
 * visits_from (:48-58: max timestamp minus lastDaysCount days in the session's local
 * time) is the caller's to compute.  Output: index pairs (visit, place) ordered by
 * (visit, place); Spark's order is undefined.  A Location outside its range fails
 * the whole job there (the UDF throws): ORACLE_E_INVALID_ARG, *out_count = -(1 +
 * index of the first bad visit) or -(1 + nv + index of the first bad place), counted
 * only over rows that take part in at least one joined pair, as in the reference.
 * Room for `cap` pairs; *out_count = all matches (may exceed cap).                */
int32_t oracle_place_visits(int64_t nv, const int64_t *v_ts, const double *v_lat, const double *v_lon, const int64_t *v_region,
                            int64_t np, const double *p_lat, const double *p_lon, const int64_t *p_region,
                            int64_t visits_from, double max_meters, int64_t cap,
                            int64_t *out_visit, int64_t *out_place, int64_t *out_count)
{
    int64_t m = 0;
    for (int64_t i = 0; i < nv; ++i) {
        if (v_ts[i] < visits_from) continue;
        for (int64_t j = 0; j < np; ++j) {
            if (p_region[j] != v_region[i]) continue;
            if (!location_ok(v_lat[i], v_lon[i])) { *out_count = -(1 + i); return ORACLE_E_INVALID_ARG; }
            if (!location_ok(p_lat[j], p_lon[j])) { *out_count = -(1 + nv + j); return ORACLE_E_INVALID_ARG; }
            if (oracle_distance_meters(v_lat[i], v_lon[i], p_lat[j], p_lon[j]) <= max_meters) {
                if (m < cap) { out_visit[m] = i; out_place[m] = j; }
                ++m;
            }
        }
    }
    *out_count = m;
    return ORACLE_OK;
}

/* f-3: printRecommendations of both mains (knn/KnnRecommenderMain.scala:90-101,
 * stochastic/StochasticRecommenderMain.scala:64-75): the target region's places JOIN the
 * recommendations ON id (inner join: a row that is not a place of the region drops out; a place
 * listed twice is taken once here - the reference's places table has unique ids), ORDER BY score
 * DESC (ties, undefined in Spark: id ascending), LIMIT max_recommendations.  No reference test.
 * Plain O(n * places) membership + insertion sort: deliberately not the sort-based product code. */
int32_t oracle_rank_recommendations(int64_t n, const int64_t *ids, const double *scores, int64_t np,
                                    const int64_t *place_ids, const int64_t *place_regions, int64_t target_region,
                                    int64_t max_recommendations, int64_t *out_ids, double *out_scores, int64_t *out_count)
{
    int64_t m = 0;
    for (int64_t i = 0; i < n; ++i) {
        int member = 0;
        for (int64_t j = 0; j < np && !member; ++j) member = place_ids[j] == ids[i] && place_regions[j] == target_region;
        if (!member) continue;
        int64_t at = m++;                                  /* insertion: score desc, id asc */
        while (at > 0 && (out_scores[at - 1] < scores[i] || (out_scores[at - 1] == scores[i] && out_ids[at - 1] > ids[i]))) {
            out_scores[at] = out_scores[at - 1]; out_ids[at] = out_ids[at - 1]; --at;
        }
        out_scores[at] = scores[i]; out_ids[at] = ids[i];
    }
    if (max_recommendations < 0) max_recommendations = 0;
    *out_count = m < max_recommendations ? m : max_recommendations;
    return ORACLE_OK;
}
