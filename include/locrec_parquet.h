/*
 * locrec_parquet.h -- C ABI of liblocrec_parquet.so: the reference's on-disk inputs -> device handles of
 * include/locrec.h without Spark and without Python.
 *
 * Replaces, for a host that wants to skip `spark.read.parquet(...).collect()`:
 *   knn/KnnRecommenderMain.scala:53-57,69-88         the three loads of a request
 *   stochastic/StochasticRecommenderMain.scala:78-84 loadStochasticGraph
 * (file names: DataUtils.scala:34-58; written by knn/RatingVectorsBuilderMain.scala:67-73 and
 *  stochastic/StochasticGraphBuilderMain.scala:68-73).  A path is a Spark output directory (part files; names starting
 *  with '_' or '.' are skipped) or a single Parquet file.
 *
 * The decoder is Apache Arrow's C++ Parquet reader (the libarrow / libparquet shipped inside the pyarrow wheel of the
 * image): a separate, optional library so that liblocrec.so itself depends on nothing but the HIP runtime.  Built by
 * `make -C locations-recommender_amd/csrc parquet` when the pyarrow headers are present (__graft_entry__.build() does).
 *
 * Rating vectors are read in Spark's VectorUDT layout
 *     struct<type: tinyint, size: int, indices: array<int>, values: array<double>>, type 0 = sparse
 * - Spark's layout, not the reference's; no Spark-written file exists in the reference tree, so this is "parity
 * unpinned" against a real file.
 *
 * Conventions: int32 status codes of locrec.h; the message of the last failure ON THIS LIBRARY's side is
 * locrec_parquet_last_error() (a failure inside locrec_knn_create / locrec_sg_create is copied there too).
 */
#ifndef LOCREC_PARQUET_H
#define LOCREC_PARQUET_H

#include <stdint.h>

#include "locrec.h"

#ifdef __cplusplus
extern "C" {
#endif

const char *locrec_parquet_last_error(void);

/* the arrays of locrec_knn_create, malloc'd: persons = union of the three sets' person ids, ascending */
typedef struct locrec_knn_arrays {
    int64_t n;
    int64_t *person_ids;
    int64_t *p_rowptr;
    int32_t *p_idx;
    double *p_val;
    int32_t p_dim;
    int64_t *c_rowptr;
    int32_t *c_idx;
    double *c_val;
    int32_t c_dim;
    int64_t *r_rowptr;
    int64_t *r_place;
    int64_t *r_rating;
} locrec_knn_arrays;

int32_t locrec_parquet_read_knn(const char *place_rating_vectors, const char *category_rating_vectors,
                                const char *place_ratings, locrec_knn_arrays **out_arrays);
void locrec_parquet_free_knn(locrec_knn_arrays *arrays);
/* read + locrec_knn_create (the index is created on the current device, as by locrec_knn_create) */
int32_t locrec_knn_create_from_parquet(const char *place_rating_vectors, const char *category_rating_vectors,
                                       const char *place_ratings, locrec_knn_index **out_index);

typedef struct locrec_sg_edges {
    int64_t n_edges;
    int64_t *source_ids;
    int64_t *target_ids;
    double *balanced_weights;
} locrec_sg_edges;

int32_t locrec_parquet_read_edges(const char *stochastic_graph, locrec_sg_edges **out_edges);
void locrec_parquet_free_edges(locrec_sg_edges *edges);
int32_t locrec_sg_create_from_parquet(const char *stochastic_graph, locrec_sg_graph **out_graph);

#ifdef __cplusplus
}
#endif
#endif /* LOCREC_PARQUET_H */
