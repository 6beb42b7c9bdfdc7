/*
 * locrec_jni.c -- JNI shim between the reference's two Scala operators and liblocrec.so
 * (include/locrec.h).  One native method per C-ABI entry point the classes in scala/ use; every
 * status other than LOCREC_OK becomes a Java exception of the type the reference throws:
 *   LOCREC_E_INVALID_ARG / LOCREC_E_NOT_FOUND -> IllegalArgumentException (require() failures and
 *       "No such person / vertex": KnnRecommender.scala:17-20,83; StochasticRecommender.scala:33-34,70)
 *   LOCREC_E_ARITHMETIC                       -> ArithmeticException (RatingVectorsBuilder.scala:36-41)
 *   LOCREC_E_OOM                              -> OutOfMemoryError
 *   anything else                             -> RuntimeException
 *
 * How arrays cross the boundary:
 *   - EVERY array length is checked against the row counts the library reads or writes before anything is
 *     touched; a short or mismatched array is an IllegalArgumentException, never a write past a Java array.
 *   - the long calls (index / graph creation, the producers, batched requests, the SG request: 1 ms ... 1 s,
 *     device allocation and stream synchronisation inside) take COPIES: Get<Type>ArrayRegion into malloc'd
 *     buffers, results back with Set<Type>ArrayRegion.  Nothing is pinned while the device works, so the
 *     garbage collector is never locked out (the JNI specification forbids blocking inside a critical region).
 *   - only the two sub-millisecond single-request calls (knnRecommend, knnQuery) pin their two small output
 *     arrays with Get/ReleasePrimitiveArrayCritical.
 * The library copies its inputs to the device inside the call and never retains a caller pointer.
 *
 * Built by jni/Makefile on a host where JAVA_HOME is set.  The build container of this repository has no JDK:
 * there the file is compiled and EXERCISED against tests/jni_stub/ (a declaration-level stand-in for <jni.h>
 * plus a JNIEnv backed by malloc, tests/test_jni_shim.py) - every native method, every length check, every
 * exception mapping runs on the GPU box through that harness; what stays unverified is the JVM's own ABI.
 * Class: com.github.tashoyan.recommender.locrec.LocrecNative (scala/com/github/tashoyan/recommender/locrec/LocrecNative.scala).
 */
#include <jni.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "locrec.h"
#ifdef LOCREC_WITH_PARQUET
#include "locrec_parquet.h" /* liblocrec_parquet.so: make -C jni WITH_PARQUET=1 */
#endif

#define JNI_FN(name) Java_com_github_tashoyan_recommender_locrec_LocrecNative_00024_##name
/* (LocrecNative is a Scala `object`: its @native methods live in the class LocrecNative$, hence _00024_) */

static void throw_new(JNIEnv *env, const char *cls, const char *msg)
{
    jclass c = (*env)->FindClass(env, cls);
    if (c) (*env)->ThrowNew(env, c, msg);
}

static void throw_status(JNIEnv *env, int32_t status)
{
    const char *cls = "java/lang/RuntimeException";
    if (status == LOCREC_E_INVALID_ARG || status == LOCREC_E_NOT_FOUND)
        cls = "java/lang/IllegalArgumentException";
    else if (status == LOCREC_E_OOM)
        cls = "java/lang/OutOfMemoryError";
    else if (status == LOCREC_E_ARITHMETIC) /* RatingVectorsBuilder.checkedCast, RatingVectorsBuilder.scala:36-41 */
        cls = "java/lang/ArithmeticException";
    throw_new(env, cls, locrec_last_error());
}

/* IllegalArgumentException with a formatted message; returns 0 so that `return iae(...)` reads well */
static int iae(JNIEnv *env, const char *fmt, ...)
{
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    throw_new(env, "java/lang/IllegalArgumentException", buf);
    return 0;
}

static int64_t alen(JNIEnv *env, jarray a) { return a ? (int64_t)(*env)->GetArrayLength(env, a) : -1; }

/* ---- copies of Java arrays (no critical region) ---- */

typedef struct {
    void *p[20];
    int n;
} bufs; /* every malloc of one native call, freed together */

static void bufs_free(bufs *b)
{
    for (int i = 0; i < b->n; ++i) free(b->p[i]);
    b->n = 0;
}

/* room for `count` elements (at least one byte, so that an empty array is not mistaken for a failure) */
static void *buf_new(JNIEnv *env, bufs *b, int64_t count, size_t elem)
{
    void *p = malloc((size_t)(count > 0 ? count : 1) * elem);
    if (!p) {
        throw_new(env, "java/lang/OutOfMemoryError", "locrec_jni: host buffer allocation failed");
        return NULL;
    }
    b->p[b->n++] = p;
    return p;
}

static int64_t *in_longs(JNIEnv *env, bufs *b, jlongArray a, int64_t n)
{
    int64_t *p = (int64_t *)buf_new(env, b, n, sizeof *p);
    if (p && n > 0) (*env)->GetLongArrayRegion(env, a, 0, (jsize)n, (jlong *)p);
    return p;
}
static int32_t *in_ints(JNIEnv *env, bufs *b, jintArray a, int64_t n)
{
    int32_t *p = (int32_t *)buf_new(env, b, n, sizeof *p);
    if (p && n > 0) (*env)->GetIntArrayRegion(env, a, 0, (jsize)n, (jint *)p);
    return p;
}
static double *in_doubles(JNIEnv *env, bufs *b, jdoubleArray a, int64_t n)
{
    double *p = (double *)buf_new(env, b, n, sizeof *p);
    if (p && n > 0) (*env)->GetDoubleArrayRegion(env, a, 0, (jsize)n, (jdouble *)p);
    return p;
}
static void out_longs(JNIEnv *env, jlongArray a, const int64_t *p, int64_t n)
{
    if (a && n > 0) (*env)->SetLongArrayRegion(env, a, 0, (jsize)n, (const jlong *)p);
}
static void out_ints(JNIEnv *env, jintArray a, const int32_t *p, int64_t n)
{
    if (a && n > 0) (*env)->SetIntArrayRegion(env, a, 0, (jsize)n, (const jint *)p);
}
static void out_doubles(JNIEnv *env, jdoubleArray a, const double *p, int64_t n)
{
    if (a && n > 0) (*env)->SetDoubleArrayRegion(env, a, 0, (jsize)n, (const jdouble *)p);
}

/* a pinned primitive array, only for the two sub-millisecond request calls */
typedef struct {
    jarray arr;
    void *ptr;
} pinned;

static int pin(JNIEnv *env, jarray a, pinned *p)
{
    p->arr = a;
    p->ptr = NULL;
    if (!a) return 1;
    p->ptr = (*env)->GetPrimitiveArrayCritical(env, a, NULL);
    return p->ptr != NULL;
}

static void unpin(JNIEnv *env, pinned *p, jint mode)
{
    if (p->arr && p->ptr) (*env)->ReleasePrimitiveArrayCritical(env, p->arr, p->ptr, mode);
    p->ptr = NULL;
}

/* ------------------------------------------------------------------ KNN */

/* knnCreate(personIds, pRowPtr, pIdx, pVal, pDim, cRowPtr, cIdx, cVal, cDim, rRowPtr, rPlace, rRating): Long
 * rRowPtr == null: the ratings are the place vectors themselves (include/locrec.h). */
static jlong knn_create_common(JNIEnv *env, int replicas, jintArray deviceIds, jlongArray personIds, jlongArray pRowPtr,
                               jintArray pIdx, jdoubleArray pVal, jint pDim, jlongArray cRowPtr, jintArray cIdx,
                               jdoubleArray cVal, jint cDim, jlongArray rRowPtr, jlongArray rPlace, jlongArray rRating)
{
    if (!personIds || !pRowPtr || !cRowPtr) return iae(env, "knnCreate: null array");
    const int64_t n = alen(env, personIds);
    if (alen(env, pRowPtr) != n + 1 || alen(env, cRowPtr) != n + 1 || (rRowPtr && alen(env, rRowPtr) != n + 1))
        return iae(env, "knnCreate: row pointers need personIds.length + 1 = %lld entries", (long long)(n + 1));
    bufs b = {{0}, 0};
    void *h = NULL;
    int32_t st = LOCREC_OK;
    const int64_t nd = deviceIds ? alen(env, deviceIds) : 0;
    int32_t *devs = nd > 0 ? in_ints(env, &b, deviceIds, nd) : NULL;
    if (nd > 0 && !devs) goto done;
    int64_t *ids = in_longs(env, &b, personIds, n), *prp = ids ? in_longs(env, &b, pRowPtr, n + 1) : NULL,
            *crp = prp ? in_longs(env, &b, cRowPtr, n + 1) : NULL, *rrp = NULL;
    if (crp && rRowPtr) rrp = in_longs(env, &b, rRowPtr, n + 1);
    if (!crp || (rRowPtr && !rrp)) goto done; /* OutOfMemoryError pending */
    {
        /* the row pointers say how many elements the library will read from the element arrays */
        const int64_t pn = prp[n], cn = crp[n], rn = rrp ? rrp[n] : 0;
        if (pn < 0 || cn < 0 || rn < 0 || alen(env, pIdx) < pn || alen(env, pVal) < pn || alen(env, cIdx) < cn ||
            alen(env, cVal) < cn || (rrp && (alen(env, rPlace) < rn || alen(env, rRating) < rn))) {
            iae(env, "knnCreate: an element array is shorter than its row pointer's last entry (%lld / %lld / %lld)",
                (long long)pn, (long long)cn, (long long)rn);
            goto done;
        }
        int32_t *pi = in_ints(env, &b, pIdx, pn);
        double *pv = pi ? in_doubles(env, &b, pVal, pn) : NULL;
        int32_t *ci = pv ? in_ints(env, &b, cIdx, cn) : NULL;
        double *cv = ci ? in_doubles(env, &b, cVal, cn) : NULL;
        int64_t *rp = NULL, *rr = NULL;
        if (cv && rrp) {
            rp = in_longs(env, &b, rPlace, rn);
            rr = rp ? in_longs(env, &b, rRating, rn) : NULL;
        }
        if (!cv || (rrp && !rr)) goto done;
        if (replicas)
            st = locrec_knn_replicas_create((int32_t)nd, devs, n, ids, prp, pi, pv, (int32_t)pDim, crp, ci, cv, (int32_t)cDim, rrp, rp,
                                            rr, (locrec_knn_replicas **)&h);
        else
            st = locrec_knn_create(n, ids, prp, pi, pv, (int32_t)pDim, crp, ci, cv, (int32_t)cDim, rrp, rp, rr, (locrec_knn_index **)&h);
        if (st != LOCREC_OK) throw_status(env, st);
    }
done:
    bufs_free(&b);
    return st == LOCREC_OK ? (jlong)(intptr_t)h : 0;
}

JNIEXPORT jlong JNICALL JNI_FN(knnCreate)(JNIEnv *env, jobject self, jlongArray personIds, jlongArray pRowPtr,
                                          jintArray pIdx, jdoubleArray pVal, jint pDim, jlongArray cRowPtr,
                                          jintArray cIdx, jdoubleArray cVal, jint cDim, jlongArray rRowPtr,
                                          jlongArray rPlace, jlongArray rRating)
{
    (void)self;
    return knn_create_common(env, 0, NULL, personIds, pRowPtr, pIdx, pVal, pDim, cRowPtr, cIdx, cVal, cDim, rRowPtr, rPlace, rRating);
}

/* knnReplicasCreate(deviceIds (null or empty: the list of setDevices), ... as knnCreate): Long - one replica of the index per
 * device, set up by the tile-wise block all-gather of csrc/multi.hip (include/locrec.h "Several devices in one process") */
JNIEXPORT jlong JNICALL JNI_FN(knnReplicasCreate)(JNIEnv *env, jobject self, jintArray deviceIds, jlongArray personIds,
                                                  jlongArray pRowPtr, jintArray pIdx, jdoubleArray pVal, jint pDim,
                                                  jlongArray cRowPtr, jintArray cIdx, jdoubleArray cVal, jint cDim,
                                                  jlongArray rRowPtr, jlongArray rPlace, jlongArray rRating)
{
    (void)self;
    return knn_create_common(env, 1, deviceIds, personIds, pRowPtr, pIdx, pVal, pDim, cRowPtr, cIdx, cVal, cDim, rRowPtr, rPlace,
                             rRating);
}

JNIEXPORT void JNICALL JNI_FN(knnReplicasDestroy)(JNIEnv *env, jobject self, jlong handle)
{
    (void)env;
    (void)self;
    if (handle) locrec_knn_replicas_destroy((locrec_knn_replicas *)(intptr_t)handle);
}

JNIEXPORT void JNICALL JNI_FN(knnDestroy)(JNIEnv *env, jobject self, jlong handle)
{
    (void)env;
    (void)self;
    if (handle) (void)locrec_knn_destroy((locrec_knn_index *)(intptr_t)handle);
}

/* shared body of knnRecommend / knnQuery: (ids, values) rows into two caller arrays; returns the row count the
 * result HAS (larger than the arrays' length = call again with bigger arrays) */
typedef int32_t (*knn_rows_fn)(locrec_knn_index *, int64_t, double, double, int64_t, int64_t *, double *, int64_t *);

static jlong knn_rows(JNIEnv *env, knn_rows_fn fn, jlong handle, jlong personId, jdouble pw, jdouble cw, jlong k,
                      jlongArray outIds, jdoubleArray outValues)
{
    if (!handle || !outIds || !outValues) return iae(env, "null handle or output array");
    const int64_t cap_i = alen(env, outIds), cap_v = alen(env, outValues);
    int64_t count = cap_i < cap_v ? cap_i : cap_v; /* the library writes min(capacity, rows) entries */
    pinned a = {0}, b = {0};
    if (!pin(env, outIds, &a) || !pin(env, outValues, &b)) {
        unpin(env, &a, JNI_ABORT);
        throw_new(env, "java/lang/OutOfMemoryError", "could not pin an output array");
        return 0;
    }
    const int32_t st = fn((locrec_knn_index *)(intptr_t)handle, (int64_t)personId, pw, cw, (int64_t)k, (int64_t *)a.ptr,
                          (double *)b.ptr, &count);
    unpin(env, &b, 0); /* outputs: copy back */
    unpin(env, &a, 0);
    if (st != LOCREC_OK) {
        throw_status(env, st);
        return 0;
    }
    return (jlong)count;
}

/* knnRecommend(handle, personId, placeWeight, categoryWeight, kNearest, outPlaceIds, outEstimatedRatings): Long */
JNIEXPORT jlong JNICALL JNI_FN(knnRecommend)(JNIEnv *env, jobject self, jlong handle, jlong personId, jdouble pw,
                                             jdouble cw, jlong k, jlongArray outPlaceIds, jdoubleArray outRatings)
{
    (void)self;
    return knn_rows(env, locrec_knn_recommend, handle, personId, pw, cw, k, outPlaceIds, outRatings);
}

/* knnQuery(handle, personId, placeWeight, categoryWeight, kNearest, outPersonIds, outSimilarities): Long */
JNIEXPORT jlong JNICALL JNI_FN(knnQuery)(JNIEnv *env, jobject self, jlong handle, jlong personId, jdouble pw, jdouble cw,
                                         jlong k, jlongArray outPersonIds, jdoubleArray outSimilarities)
{
    (void)self;
    return knn_rows(env, locrec_knn_query, handle, personId, pw, cw, k, outPersonIds, outSimilarities);
}

/* knnRecommendBatch(handle, personIds, pw, cw, k, outOffsets[nq + 1], outPlaceIds, outRatings): Long = rows needed
 * (outPlaceIds / outRatings null or too short: only outOffsets is filled - call again with room for the return value) */
static jlong knn_recommend_batch_common(JNIEnv *env, int replicas, jlong handle, jlongArray personIds, jdouble pw, jdouble cw, jlong k,
                                        jlongArray outOffsets, jlongArray outPlaceIds, jdoubleArray outRatings)
{
    if (!handle || !personIds || !outOffsets) return iae(env, "null handle or array");
    const int64_t nq = alen(env, personIds);
    if (alen(env, outOffsets) < nq + 1) return iae(env, "outOffsets needs personIds.length + 1 entries");
    int64_t cap = 0;
    if (outPlaceIds && outRatings) {
        const int64_t c1 = alen(env, outPlaceIds), c2 = alen(env, outRatings);
        cap = c1 < c2 ? c1 : c2;
    }
    const int64_t room = cap;
    bufs b = {{0}, 0};
    int32_t st = LOCREC_E_OOM;
    int64_t *q = in_longs(env, &b, personIds, nq);
    int64_t *off = q ? (int64_t *)buf_new(env, &b, nq + 1, sizeof *off) : NULL;
    int64_t *pl = off && room ? (int64_t *)buf_new(env, &b, room, sizeof *pl) : NULL;
    double *ra = pl ? (double *)buf_new(env, &b, room, sizeof *ra) : NULL;
    if (off && (!room || ra)) {
        st = replicas ? locrec_knn_replicas_recommend_batch((locrec_knn_replicas *)(intptr_t)handle, nq, q, pw, cw, (int64_t)k, off, pl, ra, &cap)
                      : locrec_knn_recommend_batch((locrec_knn_index *)(intptr_t)handle, nq, q, pw, cw, (int64_t)k, off, pl, ra, &cap);
        if (st != LOCREC_OK) {
            throw_status(env, st);
        } else {
            out_longs(env, outOffsets, off, nq + 1);
            if (room && cap <= room) { /* the rows were written */
                out_longs(env, outPlaceIds, pl, cap);
                out_doubles(env, outRatings, ra, cap);
            }
        }
    }
    bufs_free(&b);
    return st == LOCREC_OK ? (jlong)cap : 0;
}

JNIEXPORT jlong JNICALL JNI_FN(knnRecommendBatch)(JNIEnv *env, jobject self, jlong handle, jlongArray personIds, jdouble pw,
                                                  jdouble cw, jlong k, jlongArray outOffsets, jlongArray outPlaceIds,
                                                  jdoubleArray outRatings)
{
    (void)self;
    return knn_recommend_batch_common(env, 0, handle, personIds, pw, cw, k, outOffsets, outPlaceIds, outRatings);
}

/* the same over the replicas of knnReplicasCreate: the queries are sharded over the devices */
JNIEXPORT jlong JNICALL JNI_FN(knnReplicasRecommendBatch)(JNIEnv *env, jobject self, jlong handle, jlongArray personIds, jdouble pw,
                                                          jdouble cw, jlong k, jlongArray outOffsets, jlongArray outPlaceIds,
                                                          jdoubleArray outRatings)
{
    (void)self;
    return knn_recommend_batch_common(env, 1, handle, personIds, pw, cw, k, outOffsets, outPlaceIds, outRatings);
}

/* ------------------------------------------------------------------- SG */

/* sgCreate(sourceIds, targetIds, balancedWeights): Long */
JNIEXPORT jlong JNICALL JNI_FN(sgCreate)(JNIEnv *env, jobject self, jlongArray sourceIds, jlongArray targetIds,
                                         jdoubleArray weights)
{
    (void)self;
    if (!sourceIds || !targetIds || !weights) return iae(env, "sgCreate: null array");
    const int64_t ne = alen(env, sourceIds);
    if (alen(env, targetIds) != ne || alen(env, weights) != ne) return iae(env, "sgCreate: edge columns of different lengths");
    bufs b = {{0}, 0};
    locrec_sg_graph *g = NULL;
    int32_t st = LOCREC_E_OOM;
    int64_t *s = in_longs(env, &b, sourceIds, ne), *t = s ? in_longs(env, &b, targetIds, ne) : NULL;
    double *w = t ? in_doubles(env, &b, weights, ne) : NULL;
    if (w) {
        st = locrec_sg_create(ne, s, t, w, &g);
        if (st != LOCREC_OK) throw_status(env, st);
    }
    bufs_free(&b);
    return st == LOCREC_OK ? (jlong)(intptr_t)g : 0;
}

JNIEXPORT void JNICALL JNI_FN(sgDestroy)(JNIEnv *env, jobject self, jlong handle)
{
    (void)env;
    (void)self;
    if (handle) (void)locrec_sg_destroy((locrec_sg_graph *)(intptr_t)handle);
}

/* sgVertexCount(handle): Long -- vertexCount (StochasticRecommender.scala:51), to size the output arrays */
JNIEXPORT jlong JNICALL JNI_FN(sgVertexCount)(JNIEnv *env, jobject self, jlong handle)
{
    (void)self;
    if (!handle) return iae(env, "null handle");
    int64_t v = 0;
    const int32_t st = locrec_sg_info((const locrec_sg_graph *)(intptr_t)handle, &v, NULL, NULL);
    if (st != LOCREC_OK) {
        throw_status(env, st);
        return 0;
    }
    return (jlong)v;
}

/* sgRecommend(handle, vertexId, alpha, epsilon, maxIterations, outIds, outProbabilities, outIterationsConverged[2]): Long */
JNIEXPORT jlong JNICALL JNI_FN(sgRecommend)(JNIEnv *env, jobject self, jlong handle, jlong vertexId, jdouble alpha,
                                            jdouble epsilon, jlong maxIterations, jlongArray outIds,
                                            jdoubleArray outProbabilities, jlongArray outIterationsConverged)
{
    (void)self;
    if (!handle || !outIds || !outProbabilities || !outIterationsConverged || alen(env, outIterationsConverged) < 2)
        return iae(env, "null handle / output array, or outIterationsConverged shorter than 2");
    const int64_t c1 = alen(env, outIds), c2 = alen(env, outProbabilities);
    const int64_t room = c1 < c2 ? c1 : c2;
    int64_t count = room, iterations = 0;
    int32_t converged = 0;
    bufs b = {{0}, 0};
    int32_t st = LOCREC_E_OOM;
    int64_t *ids = (int64_t *)buf_new(env, &b, room, sizeof *ids);
    double *pr = ids ? (double *)buf_new(env, &b, room, sizeof *pr) : NULL;
    if (pr) {
        st = locrec_sg_recommend((locrec_sg_graph *)(intptr_t)handle, (int64_t)vertexId, alpha, epsilon, (int64_t)maxIterations,
                                 ids, pr, &count, &iterations, &converged);
        if (st != LOCREC_OK) {
            throw_status(env, st);
        } else {
            const int64_t wrote = count < room ? count : room;
            out_longs(env, outIds, ids, wrote);
            out_doubles(env, outProbabilities, pr, wrote);
            const jlong ic[2] = {(jlong)iterations, (jlong)converged};
            (*env)->SetLongArrayRegion(env, outIterationsConverged, 0, 2, ic);
        }
    }
    bufs_free(&b);
    return st == LOCREC_OK ? (jlong)count : 0;
}

/* ----------------------------------------------------------------- misc */

JNIEXPORT jstring JNICALL JNI_FN(version)(JNIEnv *env, jobject self)
{
    (void)self;
    return (*env)->NewStringUTF(env, locrec_version());
}

JNIEXPORT jint JNICALL JNI_FN(deviceCount)(JNIEnv *env, jobject self)
{
    (void)self;
    int32_t n = 0;
    const int32_t st = locrec_device_count(&n);
    if (st != LOCREC_OK) {
        throw_status(env, st);
        return 0;
    }
    return (jint)n;
}

JNIEXPORT void JNICALL JNI_FN(setDevice)(JNIEnv *env, jobject self, jint ordinal)
{
    (void)self;
    const int32_t st = locrec_set_device((int32_t)ordinal);
    if (st != LOCREC_OK) throw_status(env, st);
}

/* setDevices(deviceIds): the default device list of the multi-device natives (an empty array forgets it) */
JNIEXPORT void JNICALL JNI_FN(setDevices)(JNIEnv *env, jobject self, jintArray deviceIds)
{
    (void)self;
    const int64_t nd = deviceIds ? alen(env, deviceIds) : 0;
    bufs b = {{0}, 0};
    int32_t *devs = nd > 0 ? in_ints(env, &b, deviceIds, nd) : NULL;
    if (nd == 0 || devs) {
        const int32_t st = locrec_set_devices((int32_t)nd, devs);
        if (st != LOCREC_OK) throw_status(env, st);
    }
    bufs_free(&b);
}

JNIEXPORT jlong JNICALL JNI_FN(deviceBytesInUse)(JNIEnv *env, jobject self)
{
    (void)self;
    int64_t v = 0;
    const int32_t st = locrec_device_bytes_in_use(&v);
    if (st != LOCREC_OK) throw_status(env, st);
    return (jlong)v;
}

/* ------------------------------------------------------- handle cache (include/locrec.h "Handle cache")
 * kind: 0 = KNN index, 1 = SG graph.  cacheAcquire returns 0 on a miss. */

JNIEXPORT jlong JNICALL JNI_FN(cacheAcquire)(JNIEnv *env, jobject self, jint kind, jstring key)
{
    (void)self;
    if (!key) return iae(env, "cacheAcquire: null key");
    const char *k = (*env)->GetStringUTFChars(env, key, NULL);
    if (!k) return 0; /* OutOfMemoryError pending */
    void *h = NULL;
    const int32_t st = locrec_cache_acquire((int32_t)kind, k, &h);
    (*env)->ReleaseStringUTFChars(env, key, k);
    if (st != LOCREC_OK) {
        throw_status(env, st);
        return 0;
    }
    return (jlong)(intptr_t)h;
}

/* cachePublish(kind, key, handle, deviceBytes): Long = the handle to use (the cache owns `handle` from here on) */
JNIEXPORT jlong JNICALL JNI_FN(cachePublish)(JNIEnv *env, jobject self, jint kind, jstring key, jlong handle, jlong deviceBytes)
{
    (void)self;
    if (!key || !handle) return iae(env, "cachePublish: null key or handle");
    const char *k = (*env)->GetStringUTFChars(env, key, NULL);
    if (!k) return 0;
    void *h = NULL;
    const int32_t st = locrec_cache_publish((int32_t)kind, k, (void *)(intptr_t)handle, (int64_t)deviceBytes, &h);
    (*env)->ReleaseStringUTFChars(env, key, k);
    if (st != LOCREC_OK) {
        throw_status(env, st);
        return 0;
    }
    return (jlong)(intptr_t)h;
}

JNIEXPORT void JNICALL JNI_FN(cacheRelease)(JNIEnv *env, jobject self, jint kind, jlong handle)
{
    (void)self;
    const int32_t st = locrec_cache_release((int32_t)kind, (void *)(intptr_t)handle);
    if (st != LOCREC_OK) throw_status(env, st);
}

JNIEXPORT void JNICALL JNI_FN(cacheSetLimits)(JNIEnv *env, jobject self, jlong maxDeviceBytes, jlong maxEntries)
{
    (void)self;
    const int32_t st = locrec_cache_set_limits((int64_t)maxDeviceBytes, (int64_t)maxEntries);
    if (st != LOCREC_OK) throw_status(env, st);
}

/* cacheStats(out[5] = entries, entry bytes, hits, misses, evictions) */
JNIEXPORT void JNICALL JNI_FN(cacheStats)(JNIEnv *env, jobject self, jlongArray out)
{
    (void)self;
    if (alen(env, out) < 5) {
        iae(env, "cacheStats: the output array needs 5 entries");
        return;
    }
    int64_t v[5] = {0, 0, 0, 0, 0};
    const int32_t st = locrec_cache_stats(&v[0], &v[1], &v[2], &v[3], &v[4]);
    if (st != LOCREC_OK) {
        throw_status(env, st);
        return;
    }
    out_longs(env, out, v, 5);
}

/* ------------------------------------------------ producers, final ranking, graph groups (host arrays) */

/* calcRatings(personIds, entityIds, topN, outPersonIds, outEntityIds, outRatings): Long = rows written
 * (RatingsBuilder.calcRatings, knn/RatingsBuilder.scala:32-48; the outputs need personIds.length entries) */
JNIEXPORT jlong JNICALL JNI_FN(calcRatings)(JNIEnv *env, jobject self, jlongArray personIds, jlongArray entityIds, jlong topN,
                                            jlongArray outPersonIds, jlongArray outEntityIds, jlongArray outRatings)
{
    (void)self;
    if (!personIds || !entityIds || !outPersonIds || !outEntityIds || !outRatings) return iae(env, "calcRatings: null array");
    const int64_t n = alen(env, personIds);
    if (alen(env, entityIds) != n) return iae(env, "calcRatings: visit columns of different lengths");
    if (alen(env, outPersonIds) < n || alen(env, outEntityIds) < n || alen(env, outRatings) < n)
        return iae(env, "calcRatings: every output array needs personIds.length = %lld entries", (long long)n);
    bufs b = {{0}, 0};
    int64_t count = 0;
    int32_t st = LOCREC_E_OOM;
    int64_t *p = in_longs(env, &b, personIds, n), *e = p ? in_longs(env, &b, entityIds, n) : NULL;
    int64_t *op = e ? (int64_t *)buf_new(env, &b, n, 8) : NULL, *oe = op ? (int64_t *)buf_new(env, &b, n, 8) : NULL,
            *orr = oe ? (int64_t *)buf_new(env, &b, n, 8) : NULL;
    if (orr) {
        st = locrec_calc_ratings(n, p, e, (int64_t)topN, LOCREC_MEM_HOST, op, oe, orr, &count);
        if (st != LOCREC_OK) {
            throw_status(env, st);
        } else {
            out_longs(env, outPersonIds, op, count);
            out_longs(env, outEntityIds, oe, count);
            out_longs(env, outRatings, orr, count);
        }
    }
    bufs_free(&b);
    return st == LOCREC_OK ? (jlong)count : 0;
}

/* calcRatingVectors(personIds, entityIds, ratings, outPersonIds[n], outRowPtr[n + 1], outIdx[n], outVal[n],
 * outCounts[3] = persons, non-zeros, vector size): Unit (RatingVectorsBuilder.calcRatingVectors, :10-25,52-84) */
JNIEXPORT void JNICALL JNI_FN(calcRatingVectors)(JNIEnv *env, jobject self, jlongArray personIds, jlongArray entityIds,
                                                 jlongArray ratings, jlongArray outPersonIds, jlongArray outRowPtr,
                                                 jintArray outIdx, jdoubleArray outVal, jlongArray outCounts)
{
    (void)self;
    if (!personIds || !entityIds || !ratings || !outPersonIds || !outRowPtr || !outIdx || !outVal || !outCounts) {
        iae(env, "calcRatingVectors: null array");
        return;
    }
    const int64_t n = alen(env, personIds);
    if (alen(env, entityIds) != n || alen(env, ratings) != n) {
        iae(env, "calcRatingVectors: rating columns of different lengths");
        return;
    }
    if (alen(env, outPersonIds) < n || alen(env, outRowPtr) < n + 1 || alen(env, outIdx) < n || alen(env, outVal) < n ||
        alen(env, outCounts) < 3) {
        iae(env, "calcRatingVectors: outputs need n = %lld entries (outRowPtr n + 1, outCounts 3)", (long long)n);
        return;
    }
    bufs b = {{0}, 0};
    int64_t c[3] = {0, 0, 0};
    int64_t *p = in_longs(env, &b, personIds, n), *e = p ? in_longs(env, &b, entityIds, n) : NULL,
            *r = e ? in_longs(env, &b, ratings, n) : NULL;
    int64_t *op = r ? (int64_t *)buf_new(env, &b, n, 8) : NULL, *orp = op ? (int64_t *)buf_new(env, &b, n + 1, 8) : NULL;
    int32_t *oi = orp ? (int32_t *)buf_new(env, &b, n, 4) : NULL;
    double *ov = oi ? (double *)buf_new(env, &b, n, 8) : NULL;
    if (ov) {
        const int32_t st = locrec_calc_rating_vectors(n, p, e, r, LOCREC_MEM_HOST, op, orp, oi, ov, &c[0], &c[1], &c[2]);
        if (st != LOCREC_OK) {
            throw_status(env, st);
        } else {
            out_longs(env, outPersonIds, op, c[0]);
            out_longs(env, outRowPtr, orp, c[0] + 1);
            out_ints(env, outIdx, oi, c[1]);
            out_doubles(env, outVal, ov, c[1]);
            out_longs(env, outCounts, c, 3);
        }
    }
    bufs_free(&b);
}

/* calcPlaceVisits(visit columns x5, place columns x5, visitsFrom, maxMeters, out columns x5): Long = matches (may exceed
 * the output arrays' length: call again with larger ones; pass arrays of length 0 to size them).
 * PlaceVisits.calcPlaceVisits, PlaceVisits.scala:11-46 */
JNIEXPORT jlong JNICALL JNI_FN(calcPlaceVisits)(JNIEnv *env, jobject self, jlongArray vPerson, jlongArray vTimestamp,
                                                jdoubleArray vLat, jdoubleArray vLon, jlongArray vRegion, jlongArray pId,
                                                jdoubleArray pLat, jdoubleArray pLon, jlongArray pRegion, jlongArray pCategory,
                                                jlong visitsFrom, jdouble maxMeters, jlongArray outPerson,
                                                jlongArray outTimestamp, jlongArray outPlace, jlongArray outRegion,
                                                jlongArray outCategory)
{
    (void)self;
    if (!vPerson || !vTimestamp || !vLat || !vLon || !vRegion || !pId || !pLat || !pLon || !pRegion || !pCategory ||
        !outPerson || !outTimestamp || !outPlace || !outRegion || !outCategory)
        return iae(env, "calcPlaceVisits: null array");
    const int64_t nv = alen(env, vPerson), np = alen(env, pId);
    if (alen(env, vTimestamp) != nv || alen(env, vLat) != nv || alen(env, vLon) != nv || alen(env, vRegion) != nv)
        return iae(env, "calcPlaceVisits: visit columns of different lengths");
    if (alen(env, pLat) != np || alen(env, pLon) != np || alen(env, pRegion) != np || alen(env, pCategory) != np)
        return iae(env, "calcPlaceVisits: place columns of different lengths");
    /* the capacity is what ALL five output columns can hold */
    jlongArray outs[5] = {outPerson, outTimestamp, outPlace, outRegion, outCategory};
    int64_t room = alen(env, outs[0]);
    for (int i = 1; i < 5; ++i)
        if (alen(env, outs[i]) < room) room = alen(env, outs[i]);
    bufs b = {{0}, 0};
    int64_t count = room;
    int32_t st = LOCREC_E_OOM;
    int64_t *a0 = in_longs(env, &b, vPerson, nv), *a1 = a0 ? in_longs(env, &b, vTimestamp, nv) : NULL;
    double *a2 = a1 ? in_doubles(env, &b, vLat, nv) : NULL, *a3 = a2 ? in_doubles(env, &b, vLon, nv) : NULL;
    int64_t *a4 = a3 ? in_longs(env, &b, vRegion, nv) : NULL, *b0 = a4 ? in_longs(env, &b, pId, np) : NULL;
    double *b1 = b0 ? in_doubles(env, &b, pLat, np) : NULL, *b2 = b1 ? in_doubles(env, &b, pLon, np) : NULL;
    int64_t *b3 = b2 ? in_longs(env, &b, pRegion, np) : NULL, *b4 = b3 ? in_longs(env, &b, pCategory, np) : NULL;
    int64_t *o[5] = {NULL, NULL, NULL, NULL, NULL};
    int ok = b4 != NULL;
    for (int i = 0; ok && i < 5; ++i) ok = (o[i] = (int64_t *)buf_new(env, &b, room, 8)) != NULL;
    if (ok) {
        st = locrec_calc_place_visits(nv, a0, a1, a2, a3, a4, np, b0, b1, b2, b3, b4, (int64_t)visitsFrom, maxMeters,
                                      LOCREC_MEM_HOST, o[0], o[1], o[2], o[3], o[4], &count);
        if (st != LOCREC_OK) {
            throw_status(env, st);
        } else {
            const int64_t wrote = count < room ? count : room;
            for (int i = 0; i < 5; ++i) out_longs(env, outs[i], o[i], wrote);
        }
    }
    bufs_free(&b);
    return st == LOCREC_OK ? (jlong)count : 0;
}

/* rankRecommendations(ids, scores, placeIds, placeRegionIds, targetRegionId, maxRecommendations, outIds, outScores): Long
 * (printRecommendations of both mains, knn/KnnRecommenderMain.scala:90-101); the outputs need
 * min(ids.length, maxRecommendations) entries */
JNIEXPORT jlong JNICALL JNI_FN(rankRecommendations)(JNIEnv *env, jobject self, jlongArray ids, jdoubleArray scores,
                                                    jlongArray placeIds, jlongArray placeRegionIds, jlong targetRegionId,
                                                    jlong maxRecommendations, jlongArray outIds, jdoubleArray outScores)
{
    (void)self;
    if (!ids || !scores || !placeIds || !placeRegionIds || !outIds || !outScores) return iae(env, "rankRecommendations: null array");
    const int64_t n = alen(env, ids), np = alen(env, placeIds);
    if (alen(env, scores) != n || alen(env, placeRegionIds) != np)
        return iae(env, "rankRecommendations: columns of different lengths");
    const int64_t limit = maxRecommendations < 0 ? 0 : (int64_t)maxRecommendations;
    const int64_t need = n < limit ? n : limit;
    if (alen(env, outIds) < need || alen(env, outScores) < need)
        return iae(env, "rankRecommendations: the outputs need min(ids.length, maxRecommendations) = %lld entries", (long long)need);
    bufs b = {{0}, 0};
    int64_t count = 0;
    int32_t st = LOCREC_E_OOM;
    int64_t *i0 = in_longs(env, &b, ids, n);
    double *s0 = i0 ? in_doubles(env, &b, scores, n) : NULL;
    int64_t *p0 = s0 ? in_longs(env, &b, placeIds, np) : NULL, *p1 = p0 ? in_longs(env, &b, placeRegionIds, np) : NULL;
    int64_t *oi = p1 ? (int64_t *)buf_new(env, &b, need, 8) : NULL;
    double *os = oi ? (double *)buf_new(env, &b, need, 8) : NULL;
    if (os) {
        st = locrec_rank_recommendations(n, i0, s0, np, p0, p1, (int64_t)targetRegionId, (int64_t)maxRecommendations,
                                         LOCREC_MEM_HOST, oi, os, &count);
        if (st != LOCREC_OK) {
            throw_status(env, st);
        } else {
            out_longs(env, outIds, oi, count);
            out_doubles(env, outScores, os, count);
        }
    }
    bufs_free(&b);
    return st == LOCREC_OK ? (jlong)count : 0;
}

/* sgGroupCreate(graphHandles): Long; sgGroupSweeps(group, vertexIds, alpha, sweeps); sgGroupSynchronize; sgGroupDestroy -
 * the per-region / per-region-pair graphs of StochasticRecommenderMain iterated together (include/locrec.h).
 * The group remembers its size so that the vertexIds arrays of later calls can be checked. */
typedef struct {
    locrec_sg_group *group;
    int64_t n_graphs;
} jni_group;

JNIEXPORT jlong JNICALL JNI_FN(sgGroupCreate)(JNIEnv *env, jobject self, jlongArray graphHandles)
{
    (void)self;
    const int64_t n = alen(env, graphHandles);
    if (n <= 0 || n > 65535) return iae(env, "sgGroupCreate: between 1 and 65535 graph handles expected");
    bufs b = {{0}, 0};
    jni_group *jg = NULL;
    int32_t st = LOCREC_E_OOM;
    int64_t *hs = in_longs(env, &b, graphHandles, n);
    locrec_sg_graph **g = hs ? (locrec_sg_graph **)buf_new(env, &b, n, sizeof *g) : NULL;
    if (g) {
        for (int64_t i = 0; i < n; ++i) g[i] = (locrec_sg_graph *)(intptr_t)hs[i];
        jg = (jni_group *)malloc(sizeof *jg);
        if (!jg) {
            throw_new(env, "java/lang/OutOfMemoryError", "sgGroupCreate");
        } else {
            jg->n_graphs = n;
            st = locrec_sg_group_create(g, (int32_t)n, &jg->group);
            if (st != LOCREC_OK) {
                throw_status(env, st);
                free(jg);
                jg = NULL;
            }
        }
    }
    bufs_free(&b);
    return (jlong)(intptr_t)jg;
}

static jni_group *group_of(JNIEnv *env, jlong group, jlongArray vertexIds)
{
    jni_group *jg = (jni_group *)(intptr_t)group;
    if (!jg) {
        iae(env, "null group");
        return NULL;
    }
    if (alen(env, vertexIds) != jg->n_graphs) {
        iae(env, "vertexIds needs one entry per graph of the group (%lld)", (long long)jg->n_graphs);
        return NULL;
    }
    return jg;
}

JNIEXPORT void JNICALL JNI_FN(sgGroupSweeps)(JNIEnv *env, jobject self, jlong group, jlongArray vertexIds, jdouble alpha,
                                             jlong sweeps)
{
    (void)self;
    jni_group *jg = group_of(env, group, vertexIds);
    if (!jg) return;
    bufs b = {{0}, 0};
    int64_t *v = in_longs(env, &b, vertexIds, jg->n_graphs);
    if (v) {
        const int32_t st = locrec_sg_group_sweeps_async(jg->group, v, alpha, (int64_t)sweeps);
        if (st != LOCREC_OK) throw_status(env, st);
    }
    bufs_free(&b);
}

/* sgGroupIterate(group, vertexIds, alpha, epsilon, maxIterations): makeRecommendations' iteration for every graph */
JNIEXPORT void JNICALL JNI_FN(sgGroupIterate)(JNIEnv *env, jobject self, jlong group, jlongArray vertexIds, jdouble alpha,
                                              jdouble epsilon, jlong maxIterations)
{
    (void)self;
    jni_group *jg = group_of(env, group, vertexIds);
    if (!jg) return;
    bufs b = {{0}, 0};
    int64_t *v = in_longs(env, &b, vertexIds, jg->n_graphs);
    if (v) {
        const int32_t st = locrec_sg_group_iterate_async(jg->group, v, alpha, epsilon, (int64_t)maxIterations);
        if (st != LOCREC_OK) throw_status(env, st);
    }
    bufs_free(&b);
}

JNIEXPORT void JNICALL JNI_FN(sgGroupSynchronize)(JNIEnv *env, jobject self, jlong group)
{
    (void)self;
    jni_group *jg = (jni_group *)(intptr_t)group;
    if (!jg) {
        iae(env, "null group");
        return;
    }
    const int32_t st = locrec_sg_group_synchronize(jg->group);
    if (st != LOCREC_OK) throw_status(env, st);
}

JNIEXPORT void JNICALL JNI_FN(sgGroupDestroy)(JNIEnv *env, jobject self, jlong group)
{
    (void)env;
    (void)self;
    jni_group *jg = (jni_group *)(intptr_t)group;
    if (jg) {
        locrec_sg_group_destroy(jg->group);
        free(jg);
    }
}

/* sgFetch(handle, outIds, outProbabilities, outIterationsConverged[2]): Long - the result a group call (or any *_async
 * form) left in one graph: rows, the reference's iteration counter, converged or not (locrec_sg_fetch) */
JNIEXPORT jlong JNICALL JNI_FN(sgFetch)(JNIEnv *env, jobject self, jlong handle, jlongArray outIds,
                                        jdoubleArray outProbabilities, jlongArray outIterationsConverged)
{
    (void)self;
    if (!handle || !outIds || !outProbabilities || !outIterationsConverged || alen(env, outIterationsConverged) < 2)
        return iae(env, "null handle / output array, or outIterationsConverged shorter than 2");
    const int64_t c1 = alen(env, outIds), c2 = alen(env, outProbabilities);
    const int64_t room = c1 < c2 ? c1 : c2;
    int64_t count = room, iterations = 0;
    int32_t converged = 0;
    bufs b = {{0}, 0};
    int32_t st = LOCREC_E_OOM;
    int64_t *ids = (int64_t *)buf_new(env, &b, room, sizeof *ids);
    double *pr = ids ? (double *)buf_new(env, &b, room, sizeof *pr) : NULL;
    if (pr) {
        st = locrec_sg_fetch((locrec_sg_graph *)(intptr_t)handle, ids, pr, &count, &iterations, &converged);
        if (st != LOCREC_OK) {
            throw_status(env, st);
        } else {
            const int64_t wrote = count < room ? count : room;
            out_longs(env, outIds, ids, wrote);
            out_doubles(env, outProbabilities, pr, wrote);
            const jlong ic[2] = {(jlong)iterations, (jlong)converged};
            (*env)->SetLongArrayRegion(env, outIterationsConverged, 0, 2, ic);
        }
    }
    bufs_free(&b);
    return st == LOCREC_OK ? (jlong)count : 0;
}

/* ------------------------------------------------ one graph sharded over several devices (csrc/multi.hip)
 * sgShardedCreate(deviceIds, sourceIds, targetIds, balancedWeights, byTarget): Long; sgShardedRecommend as sgRecommend;
 * sgShardedVertexCount; sgShardedDestroy */
JNIEXPORT jlong JNICALL JNI_FN(sgShardedCreate)(JNIEnv *env, jobject self, jintArray deviceIds, jlongArray sourceIds,
                                                jlongArray targetIds, jdoubleArray weights, jboolean byTarget)
{
    (void)self;
    if (!sourceIds || !targetIds || !weights) return iae(env, "sgShardedCreate: null array");
    const int64_t ne = alen(env, sourceIds), nd = deviceIds ? alen(env, deviceIds) : 0;
    if (alen(env, targetIds) != ne || alen(env, weights) != ne) return iae(env, "sgShardedCreate: edge columns of different lengths");
    bufs b = {{0}, 0};
    locrec_sg_sharded *g = NULL;
    int32_t st = LOCREC_E_OOM;
    int32_t *devs = nd > 0 ? in_ints(env, &b, deviceIds, nd) : NULL;
    int64_t *s = (nd == 0 || devs) ? in_longs(env, &b, sourceIds, ne) : NULL, *t = s ? in_longs(env, &b, targetIds, ne) : NULL;
    double *w = t ? in_doubles(env, &b, weights, ne) : NULL;
    if (w) {
        st = locrec_sg_sharded_create((int32_t)nd, devs, ne, s, t, w, byTarget ? 1 : 0, &g);
        if (st != LOCREC_OK) throw_status(env, st);
    }
    bufs_free(&b);
    return st == LOCREC_OK ? (jlong)(intptr_t)g : 0;
}

JNIEXPORT void JNICALL JNI_FN(sgShardedDestroy)(JNIEnv *env, jobject self, jlong handle)
{
    (void)env;
    (void)self;
    if (handle) locrec_sg_sharded_destroy((locrec_sg_sharded *)(intptr_t)handle);
}

JNIEXPORT jlong JNICALL JNI_FN(sgShardedVertexCount)(JNIEnv *env, jobject self, jlong handle)
{
    (void)self;
    if (!handle) return iae(env, "null handle");
    int64_t v = 0;
    const int32_t st = locrec_sg_sharded_info((const locrec_sg_sharded *)(intptr_t)handle, NULL, NULL, NULL, &v);
    if (st != LOCREC_OK) {
        throw_status(env, st);
        return 0;
    }
    return (jlong)v;
}

JNIEXPORT jlong JNICALL JNI_FN(sgShardedRecommend)(JNIEnv *env, jobject self, jlong handle, jlong vertexId, jdouble alpha,
                                                   jdouble epsilon, jlong maxIterations, jlongArray outIds,
                                                   jdoubleArray outProbabilities, jlongArray outIterationsConverged)
{
    (void)self;
    if (!handle || !outIds || !outProbabilities || !outIterationsConverged || alen(env, outIterationsConverged) < 2)
        return iae(env, "null handle / output array, or outIterationsConverged shorter than 2");
    const int64_t c1 = alen(env, outIds), c2 = alen(env, outProbabilities);
    const int64_t room = c1 < c2 ? c1 : c2;
    int64_t count = room, iterations = 0;
    int32_t converged = 0;
    bufs b = {{0}, 0};
    int32_t st = LOCREC_E_OOM;
    int64_t *ids = (int64_t *)buf_new(env, &b, room, sizeof *ids);
    double *pr = ids ? (double *)buf_new(env, &b, room, sizeof *pr) : NULL;
    if (pr) {
        st = locrec_sg_sharded_recommend((locrec_sg_sharded *)(intptr_t)handle, (int64_t)vertexId, alpha, epsilon,
                                         (int64_t)maxIterations, ids, pr, &count, &iterations, &converged);
        if (st != LOCREC_OK) {
            throw_status(env, st);
        } else {
            const int64_t wrote = count < room ? count : room;
            out_longs(env, outIds, ids, wrote);
            out_doubles(env, outProbabilities, pr, wrote);
            const jlong ic[2] = {(jlong)iterations, (jlong)converged};
            (*env)->SetLongArrayRegion(env, outIterationsConverged, 0, 2, ic);
        }
    }
    bufs_free(&b);
    return st == LOCREC_OK ? (jlong)count : 0;
}

/* ------------------------------------------------ Parquet sets -> device handles by native code (include/locrec_parquet.h)
 * knnCreateFromParquet(placeRatingVectorsPath, categoryRatingVectorsPath, placeRatingsPath): Long
 * sgCreateFromParquet(stochasticGraphPath): Long
 * What a cache miss of the operator classes calls instead of spark.read.parquet(...).collect() when its frames are plain
 * scans of local files (LocrecBackend.localPathOf).  A shim built without LOCREC_WITH_PARQUET throws
 * UnsupportedOperationException, and the classes fall back to collect. */
#ifdef LOCREC_WITH_PARQUET
static void throw_parquet_status(JNIEnv *env, int32_t status)
{
    const char *cls = "java/lang/RuntimeException";
    if (status == LOCREC_E_INVALID_ARG || status == LOCREC_E_NOT_FOUND)
        cls = "java/lang/IllegalArgumentException";
    else if (status == LOCREC_E_OOM)
        cls = "java/lang/OutOfMemoryError";
    throw_new(env, cls, locrec_parquet_last_error());
}
#endif

JNIEXPORT jlong JNICALL JNI_FN(knnCreateFromParquet)(JNIEnv *env, jobject self, jstring placeVectors, jstring categoryVectors,
                                                     jstring placeRatings)
{
    (void)self;
#ifdef LOCREC_WITH_PARQUET
    if (!placeVectors || !categoryVectors || !placeRatings) return iae(env, "knnCreateFromParquet: null path");
    const char *a = (*env)->GetStringUTFChars(env, placeVectors, NULL);
    const char *b = a ? (*env)->GetStringUTFChars(env, categoryVectors, NULL) : NULL;
    const char *c = b ? (*env)->GetStringUTFChars(env, placeRatings, NULL) : NULL;
    locrec_knn_index *h = NULL;
    int32_t st = LOCREC_E_OOM;
    if (c) st = locrec_knn_create_from_parquet(a, b, c, &h);
    if (c) (*env)->ReleaseStringUTFChars(env, placeRatings, c);
    if (b) (*env)->ReleaseStringUTFChars(env, categoryVectors, b);
    if (a) (*env)->ReleaseStringUTFChars(env, placeVectors, a);
    if (!c) return 0; /* OutOfMemoryError pending */
    if (st != LOCREC_OK) {
        throw_parquet_status(env, st);
        return 0;
    }
    return (jlong)(intptr_t)h;
#else
    (void)placeVectors;
    (void)categoryVectors;
    (void)placeRatings;
    throw_new(env, "java/lang/UnsupportedOperationException", "locrec_jni was built without LOCREC_WITH_PARQUET");
    return 0;
#endif
}

JNIEXPORT jlong JNICALL JNI_FN(sgCreateFromParquet)(JNIEnv *env, jobject self, jstring stochasticGraph)
{
    (void)self;
#ifdef LOCREC_WITH_PARQUET
    if (!stochasticGraph) return iae(env, "sgCreateFromParquet: null path");
    const char *a = (*env)->GetStringUTFChars(env, stochasticGraph, NULL);
    if (!a) return 0;
    locrec_sg_graph *g = NULL;
    const int32_t st = locrec_sg_create_from_parquet(a, &g);
    (*env)->ReleaseStringUTFChars(env, stochasticGraph, a);
    if (st != LOCREC_OK) {
        throw_parquet_status(env, st);
        return 0;
    }
    return (jlong)(intptr_t)g;
#else
    (void)stochasticGraph;
    throw_new(env, "java/lang/UnsupportedOperationException", "locrec_jni was built without LOCREC_WITH_PARQUET");
    return 0;
#endif
}
