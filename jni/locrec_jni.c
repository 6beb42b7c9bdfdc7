/*
 * locrec_jni.c -- JNI shim between the reference's two Scala operators and liblocrec.so
 * (include/locrec.h).  One native method per C-ABI entry point the classes in scala/ use; every
 * status other than LOCREC_OK becomes a Java exception of the type the reference throws:
 *   LOCREC_E_INVALID_ARG / LOCREC_E_NOT_FOUND -> IllegalArgumentException (require() failures and
 *       "No such person / vertex": KnnRecommender.scala:17-20,83; StochasticRecommender.scala:33-34,70)
 *   LOCREC_E_OOM                              -> OutOfMemoryError
 *   anything else                             -> RuntimeException
 * Arrays cross the boundary with Get/ReleasePrimitiveArrayCritical: the library copies its inputs to
 * the device inside the call and never retains a JVM pointer; outputs are caller-allocated Java
 * arrays with the "capacity in / rows available out" convention of the header.
 *
 * This file CANNOT be compiled in the build container of this repository (no JDK: no jni.h); it is
 * built by jni/Makefile on a host where JAVA_HOME is set.  Class: com.github.tashoyan.recommender.locrec.LocrecNative
 * (scala/com/github/tashoyan/recommender/locrec/LocrecNative.scala).
 */
#include <jni.h>
#include <stdint.h>
#include <string.h>

#include <stdlib.h>

#include "locrec.h"

#define JNI_FN(name) Java_com_github_tashoyan_recommender_locrec_LocrecNative_00024_##name
/* (LocrecNative is a Scala `object`: its @native methods live in the class LocrecNative$, hence _00024_) */

static void throw_status(JNIEnv *env, int32_t status)
{
    const char *cls = "java/lang/RuntimeException";
    if (status == LOCREC_E_INVALID_ARG || status == LOCREC_E_NOT_FOUND)
        cls = "java/lang/IllegalArgumentException";
    else if (status == LOCREC_E_OOM)
        cls = "java/lang/OutOfMemoryError";
    else if (status == LOCREC_E_ARITHMETIC)   /* RatingVectorsBuilder.checkedCast, RatingVectorsBuilder.scala:36-41 */
        cls = "java/lang/ArithmeticException";
    jclass c = (*env)->FindClass(env, cls);
    if (c) (*env)->ThrowNew(env, c, locrec_last_error());
}

/* a pinned primitive array (NULL array -> NULL pointer) */
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

/* knnCreate(personIds, pRowPtr, pIdx, pVal, pDim, cRowPtr, cIdx, cVal, cDim, rRowPtr, rPlace, rRating): Long */
JNIEXPORT jlong JNICALL JNI_FN(knnCreate)(JNIEnv *env, jobject self, jlongArray personIds, jlongArray pRowPtr,
                                          jintArray pIdx, jdoubleArray pVal, jint pDim, jlongArray cRowPtr,
                                          jintArray cIdx, jdoubleArray cVal, jint cDim, jlongArray rRowPtr,
                                          jlongArray rPlace, jlongArray rRating)
{
    (void)self;
    if (!personIds || !pRowPtr || !cRowPtr) {
        jclass c = (*env)->FindClass(env, "java/lang/IllegalArgumentException");
        if (c) (*env)->ThrowNew(env, c, "knnCreate: null array");
        return 0;
    }
    const jsize n = (*env)->GetArrayLength(env, personIds);
    jarray arrays[11] = {personIds, pRowPtr, pIdx, pVal, cRowPtr, cIdx, cVal, rRowPtr, rPlace, rRating, NULL};
    pinned p[10];
    int ok = 1, got = 0;
    for (; got < 10; ++got) {
        if (!pin(env, arrays[got], &p[got])) {
            ok = 0;
            break;
        }
    }
    locrec_knn_index *h = NULL;
    int32_t st = LOCREC_E_OOM;
    if (ok)
        st = locrec_knn_create((int64_t)n, (const int64_t *)p[0].ptr, (const int64_t *)p[1].ptr, (const int32_t *)p[2].ptr,
                               (const double *)p[3].ptr, (int32_t)pDim, (const int64_t *)p[4].ptr,
                               (const int32_t *)p[5].ptr, (const double *)p[6].ptr, (int32_t)cDim,
                               (const int64_t *)p[7].ptr, (const int64_t *)p[8].ptr, (const int64_t *)p[9].ptr, &h);
    for (int i = got - 1; i >= 0; --i) unpin(env, &p[i], JNI_ABORT); /* inputs: nothing to copy back */
    if (!ok) {
        jclass c = (*env)->FindClass(env, "java/lang/OutOfMemoryError");
        if (c) (*env)->ThrowNew(env, c, "knnCreate: could not pin an input array");
        return 0;
    }
    if (st != LOCREC_OK) {
        throw_status(env, st);
        return 0;
    }
    return (jlong)(intptr_t)h;
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
    if (!handle || !outIds || !outValues) {
        jclass c = (*env)->FindClass(env, "java/lang/IllegalArgumentException");
        if (c) (*env)->ThrowNew(env, c, "null handle or output array");
        return 0;
    }
    const jsize cap_i = (*env)->GetArrayLength(env, outIds), cap_v = (*env)->GetArrayLength(env, outValues);
    int64_t count = cap_i < cap_v ? cap_i : cap_v;
    pinned a, b;
    if (!pin(env, outIds, &a)) return 0;
    if (!pin(env, outValues, &b)) {
        unpin(env, &a, JNI_ABORT);
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

/* knnRecommendBatch(handle, personIds, pw, cw, k, outOffsets[nq + 1], outPlaceIds, outRatings): Long = rows needed */
JNIEXPORT jlong JNICALL JNI_FN(knnRecommendBatch)(JNIEnv *env, jobject self, jlong handle, jlongArray personIds, jdouble pw,
                                                  jdouble cw, jlong k, jlongArray outOffsets, jlongArray outPlaceIds,
                                                  jdoubleArray outRatings)
{
    (void)self;
    if (!handle || !personIds || !outOffsets) {
        jclass c = (*env)->FindClass(env, "java/lang/IllegalArgumentException");
        if (c) (*env)->ThrowNew(env, c, "null handle or array");
        return 0;
    }
    const jsize nq = (*env)->GetArrayLength(env, personIds);
    if ((*env)->GetArrayLength(env, outOffsets) < nq + 1) {
        jclass c = (*env)->FindClass(env, "java/lang/IllegalArgumentException");
        if (c) (*env)->ThrowNew(env, c, "outOffsets needs personIds.length + 1 entries");
        return 0;
    }
    int64_t cap = 0;
    if (outPlaceIds && outRatings) {
        const jsize c1 = (*env)->GetArrayLength(env, outPlaceIds), c2 = (*env)->GetArrayLength(env, outRatings);
        cap = c1 < c2 ? c1 : c2;
    }
    pinned q, o, a, b;
    int ok = pin(env, personIds, &q) && pin(env, outOffsets, &o) && pin(env, cap ? outPlaceIds : NULL, &a) &&
             pin(env, cap ? outRatings : NULL, &b);
    int32_t st = LOCREC_E_OOM;
    if (ok)
        st = locrec_knn_recommend_batch((locrec_knn_index *)(intptr_t)handle, (int64_t)nq, (const int64_t *)q.ptr, pw, cw,
                                        (int64_t)k, (int64_t *)o.ptr, (int64_t *)a.ptr, (double *)b.ptr, &cap);
    unpin(env, &b, 0);
    unpin(env, &a, 0);
    unpin(env, &o, 0);
    unpin(env, &q, JNI_ABORT);
    if (st != LOCREC_OK) {
        throw_status(env, st);
        return 0;
    }
    return (jlong)cap;
}

/* ------------------------------------------------------------------- SG */

/* sgCreate(sourceIds, targetIds, balancedWeights): Long */
JNIEXPORT jlong JNICALL JNI_FN(sgCreate)(JNIEnv *env, jobject self, jlongArray sourceIds, jlongArray targetIds,
                                         jdoubleArray weights)
{
    (void)self;
    if (!sourceIds || !targetIds || !weights) {
        jclass c = (*env)->FindClass(env, "java/lang/IllegalArgumentException");
        if (c) (*env)->ThrowNew(env, c, "sgCreate: null array");
        return 0;
    }
    const jsize ne = (*env)->GetArrayLength(env, sourceIds);
    if ((*env)->GetArrayLength(env, targetIds) != ne || (*env)->GetArrayLength(env, weights) != ne) {
        jclass c = (*env)->FindClass(env, "java/lang/IllegalArgumentException");
        if (c) (*env)->ThrowNew(env, c, "sgCreate: edge columns of different lengths");
        return 0;
    }
    pinned s, t, w;
    int ok = pin(env, sourceIds, &s) && pin(env, targetIds, &t) && pin(env, weights, &w);
    locrec_sg_graph *g = NULL;
    int32_t st = LOCREC_E_OOM;
    if (ok) st = locrec_sg_create((int64_t)ne, (const int64_t *)s.ptr, (const int64_t *)t.ptr, (const double *)w.ptr, &g);
    unpin(env, &w, JNI_ABORT);
    unpin(env, &t, JNI_ABORT);
    unpin(env, &s, JNI_ABORT);
    if (st != LOCREC_OK) {
        throw_status(env, st);
        return 0;
    }
    return (jlong)(intptr_t)g;
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
    if (!handle || !outIds || !outProbabilities || !outIterationsConverged ||
        (*env)->GetArrayLength(env, outIterationsConverged) < 2) {
        jclass c = (*env)->FindClass(env, "java/lang/IllegalArgumentException");
        if (c) (*env)->ThrowNew(env, c, "null handle / output array, or outIterationsConverged shorter than 2");
        return 0;
    }
    const jsize c1 = (*env)->GetArrayLength(env, outIds), c2 = (*env)->GetArrayLength(env, outProbabilities);
    int64_t count = c1 < c2 ? c1 : c2, iterations = 0;
    int32_t converged = 0;
    pinned a, b;
    if (!pin(env, outIds, &a)) return 0;
    if (!pin(env, outProbabilities, &b)) {
        unpin(env, &a, JNI_ABORT);
        return 0;
    }
    const int32_t st = locrec_sg_recommend((locrec_sg_graph *)(intptr_t)handle, (int64_t)vertexId, alpha, epsilon,
                                           (int64_t)maxIterations, (int64_t *)a.ptr, (double *)b.ptr, &count, &iterations,
                                           &converged);
    unpin(env, &b, 0);
    unpin(env, &a, 0);
    if (st != LOCREC_OK) {
        throw_status(env, st);
        return 0;
    }
    const jlong ic[2] = {(jlong)iterations, (jlong)converged};
    (*env)->SetLongArrayRegion(env, outIterationsConverged, 0, 2, ic);
    return (jlong)count;
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

/* ------------------------------------------------ producers, final ranking, graph groups (host arrays) */

/* pins up to 12 primitive arrays; returns 0 (after releasing what it pinned) when one cannot be pinned */
static int pin_all(JNIEnv *env, int n, jarray *arrs, pinned *pins)
{
    for (int i = 0; i < n; ++i)
        if (!pin(env, arrs[i], &pins[i])) {
            for (int j = 0; j < i; ++j) unpin(env, &pins[j], JNI_ABORT);
            return 0;
        }
    return 1;
}

/* calcRatings(personIds, entityIds, topN, outPersonIds, outEntityIds, outRatings): Long = rows written
 * (RatingsBuilder.calcRatings, knn/RatingsBuilder.scala:32-48; the outputs need personIds.length entries) */
JNIEXPORT jlong JNICALL JNI_FN(calcRatings)(JNIEnv *env, jobject self, jlongArray personIds, jlongArray entityIds, jlong topN,
                                            jlongArray outPersonIds, jlongArray outEntityIds, jlongArray outRatings)
{
    (void)self;
    const jsize n = personIds ? (*env)->GetArrayLength(env, personIds) : 0;
    jarray arrs[5] = {personIds, entityIds, outPersonIds, outEntityIds, outRatings};
    pinned p[5];
    if (!pin_all(env, 5, arrs, p)) return 0;
    int64_t count = 0;
    const int32_t st = locrec_calc_ratings((int64_t)n, (const int64_t *)p[0].ptr, (const int64_t *)p[1].ptr, (int64_t)topN,
                                           LOCREC_MEM_HOST, (int64_t *)p[2].ptr, (int64_t *)p[3].ptr, (int64_t *)p[4].ptr, &count);
    for (int i = 4; i >= 2; --i) unpin(env, &p[i], 0);
    unpin(env, &p[1], JNI_ABORT);
    unpin(env, &p[0], JNI_ABORT);
    if (st != LOCREC_OK) {
        throw_status(env, st);
        return 0;
    }
    return (jlong)count;
}

/* calcRatingVectors(personIds, entityIds, ratings, outPersonIds[n], outRowPtr[n + 1], outIdx[n], outVal[n],
 * outCounts[3] = persons, non-zeros, vector size): Unit (RatingVectorsBuilder.calcRatingVectors, :10-25,52-84) */
JNIEXPORT void JNICALL JNI_FN(calcRatingVectors)(JNIEnv *env, jobject self, jlongArray personIds, jlongArray entityIds,
                                                 jlongArray ratings, jlongArray outPersonIds, jlongArray outRowPtr,
                                                 jintArray outIdx, jdoubleArray outVal, jlongArray outCounts)
{
    (void)self;
    const jsize n = personIds ? (*env)->GetArrayLength(env, personIds) : 0;
    jarray arrs[7] = {personIds, entityIds, ratings, outPersonIds, outRowPtr, outIdx, outVal};
    pinned p[7];
    if (!pin_all(env, 7, arrs, p)) return;
    int64_t c[3] = {0, 0, 0};
    const int32_t st = locrec_calc_rating_vectors((int64_t)n, (const int64_t *)p[0].ptr, (const int64_t *)p[1].ptr,
                                                  (const int64_t *)p[2].ptr, LOCREC_MEM_HOST, (int64_t *)p[3].ptr,
                                                  (int64_t *)p[4].ptr, (int32_t *)p[5].ptr, (double *)p[6].ptr, &c[0], &c[1], &c[2]);
    for (int i = 6; i >= 3; --i) unpin(env, &p[i], 0);
    for (int i = 2; i >= 0; --i) unpin(env, &p[i], JNI_ABORT);
    if (st != LOCREC_OK) {
        throw_status(env, st);
        return;
    }
    const jlong jc[3] = {(jlong)c[0], (jlong)c[1], (jlong)c[2]};
    (*env)->SetLongArrayRegion(env, outCounts, 0, 3, jc);
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
    const jsize nv = vPerson ? (*env)->GetArrayLength(env, vPerson) : 0, np = pId ? (*env)->GetArrayLength(env, pId) : 0;
    int64_t count = outPerson ? (*env)->GetArrayLength(env, outPerson) : 0;
    jarray in[10] = {vPerson, vTimestamp, vLat, vLon, vRegion, pId, pLat, pLon, pRegion, pCategory};
    jarray out[5] = {outPerson, outTimestamp, outPlace, outRegion, outCategory};
    pinned pi[10], po[5];
    if (!pin_all(env, 10, in, pi)) return 0;
    if (!pin_all(env, 5, out, po)) {
        for (int i = 9; i >= 0; --i) unpin(env, &pi[i], JNI_ABORT);
        return 0;
    }
    const int32_t st = locrec_calc_place_visits(
        (int64_t)nv, (const int64_t *)pi[0].ptr, (const int64_t *)pi[1].ptr, (const double *)pi[2].ptr, (const double *)pi[3].ptr,
        (const int64_t *)pi[4].ptr, (int64_t)np, (const int64_t *)pi[5].ptr, (const double *)pi[6].ptr, (const double *)pi[7].ptr,
        (const int64_t *)pi[8].ptr, (const int64_t *)pi[9].ptr, (int64_t)visitsFrom, maxMeters, LOCREC_MEM_HOST,
        (int64_t *)po[0].ptr, (int64_t *)po[1].ptr, (int64_t *)po[2].ptr, (int64_t *)po[3].ptr, (int64_t *)po[4].ptr, &count);
    for (int i = 4; i >= 0; --i) unpin(env, &po[i], 0);
    for (int i = 9; i >= 0; --i) unpin(env, &pi[i], JNI_ABORT);
    if (st != LOCREC_OK) {
        throw_status(env, st);
        return 0;
    }
    return (jlong)count;
}

/* rankRecommendations(ids, scores, placeIds, placeRegionIds, targetRegionId, maxRecommendations, outIds, outScores): Long
 * (printRecommendations of both mains, knn/KnnRecommenderMain.scala:90-101) */
JNIEXPORT jlong JNICALL JNI_FN(rankRecommendations)(JNIEnv *env, jobject self, jlongArray ids, jdoubleArray scores,
                                                    jlongArray placeIds, jlongArray placeRegionIds, jlong targetRegionId,
                                                    jlong maxRecommendations, jlongArray outIds, jdoubleArray outScores)
{
    (void)self;
    const jsize n = ids ? (*env)->GetArrayLength(env, ids) : 0, np = placeIds ? (*env)->GetArrayLength(env, placeIds) : 0;
    jarray arrs[6] = {ids, scores, placeIds, placeRegionIds, outIds, outScores};
    pinned p[6];
    if (!pin_all(env, 6, arrs, p)) return 0;
    int64_t count = 0;
    const int32_t st = locrec_rank_recommendations((int64_t)n, (const int64_t *)p[0].ptr, (const double *)p[1].ptr, (int64_t)np,
                                                   (const int64_t *)p[2].ptr, (const int64_t *)p[3].ptr, (int64_t)targetRegionId,
                                                   (int64_t)maxRecommendations, LOCREC_MEM_HOST, (int64_t *)p[4].ptr,
                                                   (double *)p[5].ptr, &count);
    unpin(env, &p[5], 0);
    unpin(env, &p[4], 0);
    for (int i = 3; i >= 0; --i) unpin(env, &p[i], JNI_ABORT);
    if (st != LOCREC_OK) {
        throw_status(env, st);
        return 0;
    }
    return (jlong)count;
}

/* sgGroupCreate(graphHandles): Long; sgGroupSweeps(group, vertexIds, alpha, sweeps); sgGroupSynchronize; sgGroupDestroy -
 * the per-region / per-region-pair graphs of StochasticRecommenderMain iterated together (include/locrec.h) */
JNIEXPORT jlong JNICALL JNI_FN(sgGroupCreate)(JNIEnv *env, jobject self, jlongArray graphHandles)
{
    (void)self;
    const jsize n = graphHandles ? (*env)->GetArrayLength(env, graphHandles) : 0;
    pinned p;
    if (!pin(env, graphHandles, &p)) return 0;
    locrec_sg_group *grp = NULL;
    int32_t st = LOCREC_E_INVALID_ARG;
    if (n > 0 && n <= 65535) {
        locrec_sg_graph *graphs[64];
        locrec_sg_graph **g = n <= 64 ? graphs : (locrec_sg_graph **)malloc((size_t)n * sizeof *g);
        if (g) {
            for (jsize i = 0; i < n; ++i) g[i] = (locrec_sg_graph *)(intptr_t)((const jlong *)p.ptr)[i];
            st = locrec_sg_group_create(g, (int32_t)n, &grp);
            if (g != graphs) free(g);
        } else {
            st = LOCREC_E_OOM;
        }
    }
    unpin(env, &p, JNI_ABORT);
    if (st != LOCREC_OK) {
        throw_status(env, st);
        return 0;
    }
    return (jlong)(intptr_t)grp;
}

JNIEXPORT void JNICALL JNI_FN(sgGroupSweeps)(JNIEnv *env, jobject self, jlong group, jlongArray vertexIds, jdouble alpha,
                                             jlong sweeps)
{
    (void)self;
    pinned p;
    if (!pin(env, vertexIds, &p)) return;
    const int32_t st = locrec_sg_group_sweeps_async((locrec_sg_group *)(intptr_t)group, (const int64_t *)p.ptr, alpha, (int64_t)sweeps);
    unpin(env, &p, JNI_ABORT);
    if (st != LOCREC_OK) throw_status(env, st);
}

/* sgGroupIterate(group, vertexIds, alpha, epsilon, maxIterations): makeRecommendations' iteration for every graph */
JNIEXPORT void JNICALL JNI_FN(sgGroupIterate)(JNIEnv *env, jobject self, jlong group, jlongArray vertexIds, jdouble alpha,
                                              jdouble epsilon, jlong maxIterations)
{
    (void)self;
    pinned p;
    if (!pin(env, vertexIds, &p)) return;
    const int32_t st = locrec_sg_group_iterate_async((locrec_sg_group *)(intptr_t)group, (const int64_t *)p.ptr, alpha, epsilon,
                                                     (int64_t)maxIterations);
    unpin(env, &p, JNI_ABORT);
    if (st != LOCREC_OK) throw_status(env, st);
}

JNIEXPORT void JNICALL JNI_FN(sgGroupSynchronize)(JNIEnv *env, jobject self, jlong group)
{
    (void)self;
    const int32_t st = locrec_sg_group_synchronize((locrec_sg_group *)(intptr_t)group);
    if (st != LOCREC_OK) throw_status(env, st);
}

JNIEXPORT void JNICALL JNI_FN(sgGroupDestroy)(JNIEnv *env, jobject self, jlong group)
{
    (void)env;
    (void)self;
    if (group) locrec_sg_group_destroy((locrec_sg_group *)(intptr_t)group);
}
