// parquet_loader.cpp -- liblocrec_parquet.so: the on-disk inputs of both recommenders straight into device handles,
// without Spark and without Python (include/locrec_parquet.h).
//
// The reference's builders write four Parquet sets per region / region pair (RatingVectorsBuilderMain.scala:67-73,
// StochasticGraphBuilderMain.scala:68-73; named by DataUtils.scala:34-58) and its mains read them back for every
// request (KnnRecommenderMain.scala:69-88, StochasticRecommenderMain.scala:78-84) - spark.read.parquet + a collect
// through the driver.  Here the files are decoded by Apache Arrow's C++ Parquet reader (the libarrow / libparquet that
// ship with the pyarrow wheel of this image: no JVM, no Python interpreter in the process) into the CSR / edge arrays
// of include/locrec.h and handed to locrec_knn_create / locrec_sg_create.  A JVM binds it through JNI and skips
// `collect` on a cache miss; the C ABI is the same plain-pointer style as the main library's.
//
// Layouts.  Edge and rating columns are plain longs / doubles (ids of any integer width are widened,
// StochasticGraphBuilderTest.scala:20-23,56).  A rating vector is Spark's VectorUDT:
//     struct<type: tinyint, size: int, indices: array<int>, values: array<double>>   (type 0 = sparse, 1 = dense)
// That layout is Spark's, not the reference's, and /root/reference holds no Spark-written file: the reader follows
// the published layout and is "parity unpinned" against a real file; the tests write files of
// this layout with pyarrow and compare with the numpy reader of mains.py.
//
// Host-only C++20 (Arrow 25's headers need it), built by `make parquet` when the pyarrow headers are present.
#include <arrow/api.h>
#include <arrow/io/api.h>
#include <parquet/arrow/reader.h>

#include <algorithm>
#include <cstdarg>
#include <cstring>
#include <filesystem>
#include <numeric>
#include <string>
#include <vector>

#include "../../include/locrec.h"
#include "../../include/locrec_parquet.h"

namespace {

thread_local std::string g_err;

int32_t fail(int32_t code, const char *fmt, ...)
{
    char buf[768];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define PQ_TRY(expr)                    \
    do {                                \
        const int32_t s_ = (expr);      \
        if (s_ != LOCREC_OK) return s_; \
    } while (0)

// a Spark output directory (part files; _SUCCESS, .crc and other hidden files skipped) or a single file
int32_t list_files(const std::string &path, std::vector<std::string> &files)
{
    namespace fs = std::filesystem;
    std::error_code ec;
    if (fs::is_directory(path, ec)) {
        for (const auto &e : fs::recursive_directory_iterator(path, ec)) {
            if (!e.is_regular_file()) continue;
            const std::string name = e.path().filename().string();
            if (name.empty() || name[0] == '_' || name[0] == '.') continue;
            files.push_back(e.path().string());
        }
        std::sort(files.begin(), files.end());
    } else if (fs::is_regular_file(path, ec)) {
        files.push_back(path);
    }
    if (files.empty()) return fail(LOCREC_E_INVALID_ARG, "no Parquet file at %s", path.c_str());
    return LOCREC_OK;
}

int32_t read_table(const std::string &path, const std::vector<std::string> &columns, std::shared_ptr<arrow::Table> &out)
{
    std::vector<std::string> files;
    PQ_TRY(list_files(path, files));
    std::vector<std::shared_ptr<arrow::Table>> parts;
    for (const std::string &f : files) {
        auto in = arrow::io::ReadableFile::Open(f);
        if (!in.ok()) return fail(LOCREC_E_INVALID_ARG, "%s: %s", f.c_str(), in.status().ToString().c_str());
        auto rd = parquet::arrow::OpenFile(*in, arrow::default_memory_pool());
        if (!rd.ok()) return fail(LOCREC_E_INVALID_ARG, "%s: %s", f.c_str(), rd.status().ToString().c_str());
        // (ReadTable's column indices count Parquet LEAF columns - a VectorUDT struct has four - so the whole file is
        // read and the wanted top-level columns are selected from the table; the reference's files hold nothing else)
        auto tr = (*rd)->ReadTable();
        if (!tr.ok()) return fail(LOCREC_E_INVALID_ARG, "%s: %s", f.c_str(), tr.status().ToString().c_str());
        std::vector<int> idx;
        for (const std::string &c : columns) {
            const int i = (*tr)->schema()->GetFieldIndex(c);
            if (i < 0) return fail(LOCREC_E_INVALID_ARG, "%s has no column %s", f.c_str(), c.c_str());
            idx.push_back(i);
        }
        tr = (*tr)->SelectColumns(idx);
        if (!tr.ok()) return fail(LOCREC_E_INVALID_ARG, "%s: %s", f.c_str(), tr.status().ToString().c_str());
        parts.push_back(*tr);
    }
    auto cat = arrow::ConcatenateTables(parts);
    if (!cat.ok()) return fail(LOCREC_E_INVALID_ARG, "%s: part files disagree: %s", path.c_str(), cat.status().ToString().c_str());
    out = *cat;
    return LOCREC_OK;
}

// an integer column of any width, widened to int64 (nulls rejected)
int32_t longs_of(const std::shared_ptr<arrow::Table> &t, const std::string &name, std::vector<int64_t> &out)
{
    auto col = t->GetColumnByName(name);
    if (!col) return fail(LOCREC_E_INVALID_ARG, "no column %s", name.c_str());
    out.clear();
    out.reserve((size_t)col->length());
    for (const auto &chunk : col->chunks()) {
        if (chunk->null_count()) return fail(LOCREC_E_INVALID_ARG, "null entry in column %s", name.c_str());
        const int64_t n = chunk->length();
        switch (chunk->type_id()) {
        case arrow::Type::INT64: {
            const auto *p = std::static_pointer_cast<arrow::Int64Array>(chunk)->raw_values();
            out.insert(out.end(), p, p + n);
            break;
        }
        case arrow::Type::INT32: {
            const auto *p = std::static_pointer_cast<arrow::Int32Array>(chunk)->raw_values();
            out.insert(out.end(), p, p + n);
            break;
        }
        case arrow::Type::INT16: {
            const auto *p = std::static_pointer_cast<arrow::Int16Array>(chunk)->raw_values();
            out.insert(out.end(), p, p + n);
            break;
        }
        case arrow::Type::INT8: {
            const auto *p = std::static_pointer_cast<arrow::Int8Array>(chunk)->raw_values();
            out.insert(out.end(), p, p + n);
            break;
        }
        default:
            return fail(LOCREC_E_INVALID_ARG, "column %s is %s, an integer column was expected", name.c_str(),
                        chunk->type()->ToString().c_str());
        }
    }
    return LOCREC_OK;
}

int32_t doubles_of(const std::shared_ptr<arrow::Table> &t, const std::string &name, std::vector<double> &out)
{
    auto col = t->GetColumnByName(name);
    if (!col) return fail(LOCREC_E_INVALID_ARG, "no column %s", name.c_str());
    out.clear();
    out.reserve((size_t)col->length());
    for (const auto &chunk : col->chunks()) {
        if (chunk->null_count()) return fail(LOCREC_E_INVALID_ARG, "null entry in column %s", name.c_str());
        if (chunk->type_id() != arrow::Type::DOUBLE)
            return fail(LOCREC_E_INVALID_ARG, "column %s is %s, double was expected", name.c_str(), chunk->type()->ToString().c_str());
        const auto *p = std::static_pointer_cast<arrow::DoubleArray>(chunk)->raw_values();
        out.insert(out.end(), p, p + chunk->length());
    }
    return LOCREC_OK;
}

struct Vectors {  // one (person_id, rating_vector) set, rows ascending by person id
    std::vector<int64_t> ids, rowptr;
    std::vector<int32_t> idx;
    std::vector<double> val;
    int32_t dim = 0;
};

// (person_id: long, rating_vector: VectorUDT) -> CSR rows sorted by person_id (RatingVectorsBuilder.scala:81-82)
int32_t read_vectors(const std::string &path, Vectors &v)
{
    std::shared_ptr<arrow::Table> t;
    PQ_TRY(read_table(path, {"person_id", "rating_vector"}, t));
    std::vector<int64_t> pid;
    PQ_TRY(longs_of(t, "person_id", pid));
    const int64_t n = (int64_t)pid.size();
    std::vector<int64_t> start((size_t)n), len((size_t)n);  // per file row: its slice of the flat arrays below
    std::vector<int32_t> flat_idx;
    std::vector<double> flat_val;
    int64_t row = 0;
    int64_t size_seen = -1;
    auto col = t->GetColumnByName("rating_vector");
    for (const auto &chunk : col->chunks()) {
        if (chunk->type_id() != arrow::Type::STRUCT) return fail(LOCREC_E_INVALID_ARG, "column rating_vector is not a VectorUDT struct");
        if (chunk->null_count()) return fail(LOCREC_E_INVALID_ARG, "null entry in a vector column");
        const auto st = std::static_pointer_cast<arrow::StructArray>(chunk);
        // (Flatten-style access: the field arrays carry the struct's offset themselves)
        const auto f_type = st->GetFieldByName("type"), f_size = st->GetFieldByName("size");
        const auto f_idx = st->GetFieldByName("indices"), f_val = st->GetFieldByName("values");
        if (!f_type || !f_size || !f_idx || !f_val) return fail(LOCREC_E_INVALID_ARG, "rating_vector lacks a VectorUDT field");
        if (f_type->type_id() != arrow::Type::INT8 || f_size->type_id() != arrow::Type::INT32 || f_idx->type_id() != arrow::Type::LIST ||
            f_val->type_id() != arrow::Type::LIST)
            return fail(LOCREC_E_INVALID_ARG, "rating_vector is not struct<type: tinyint, size: int, indices: array<int>, values: array<double>>");
        const auto a_type = std::static_pointer_cast<arrow::Int8Array>(f_type);
        const auto a_size = std::static_pointer_cast<arrow::Int32Array>(f_size);
        const auto l_idx = std::static_pointer_cast<arrow::ListArray>(f_idx);
        const auto l_val = std::static_pointer_cast<arrow::ListArray>(f_val);
        if (l_idx->values()->type_id() != arrow::Type::INT32 || l_val->values()->type_id() != arrow::Type::DOUBLE)
            return fail(LOCREC_E_INVALID_ARG, "rating_vector's indices / values are not array<int> / array<double>");
        const auto *vi = std::static_pointer_cast<arrow::Int32Array>(l_idx->values())->raw_values();
        const auto *vv = std::static_pointer_cast<arrow::DoubleArray>(l_val->values())->raw_values();
        for (int64_t i = 0; i < st->length(); ++i, ++row) {
            if (a_type->IsNull(i) || a_size->IsNull(i) || l_idx->IsNull(i) || l_val->IsNull(i))
                return fail(LOCREC_E_INVALID_ARG, "null entry in a vector column");
            if (a_type->Value(i) != 0)  // RatingVectorsBuilder.scala:74-77 only ever builds SparseVector
                return fail(LOCREC_E_INVALID_ARG, "dense rating vectors are not produced by the reference's builder");
            const int64_t sz = a_size->Value(i);
            if (size_seen >= 0 && sz != size_seen) return fail(LOCREC_E_INVALID_ARG, "rating vectors of different sizes in one file");
            size_seen = sz;
            const int64_t o = l_idx->value_offset(i), m = l_idx->value_length(i);
            if (l_val->value_length(i) != m) return fail(LOCREC_E_INVALID_ARG, "indices and values of a sparse vector differ in length");
            start[(size_t)row] = (int64_t)flat_idx.size();
            len[(size_t)row] = m;
            flat_idx.insert(flat_idx.end(), vi + o, vi + o + m);
            const int64_t ov = l_val->value_offset(i);
            flat_val.insert(flat_val.end(), vv + ov, vv + ov + m);
        }
    }
    if (row != n) return fail(LOCREC_E_INVALID_ARG, "person_id and rating_vector columns differ in length");
    std::vector<int64_t> order((size_t)n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return pid[(size_t)a] < pid[(size_t)b]; });
    v.dim = n > 0 ? (int32_t)size_seen : 0;
    v.ids.resize((size_t)n);
    v.rowptr.assign((size_t)n + 1, 0);
    v.idx.resize(flat_idx.size());
    v.val.resize(flat_val.size());
    for (int64_t r = 0; r < n; ++r) {
        const int64_t src = order[(size_t)r];
        v.ids[(size_t)r] = pid[(size_t)src];
        v.rowptr[(size_t)r + 1] = v.rowptr[(size_t)r] + len[(size_t)src];
        std::copy(flat_idx.begin() + start[(size_t)src], flat_idx.begin() + start[(size_t)src] + len[(size_t)src],
                  v.idx.begin() + v.rowptr[(size_t)r]);
        std::copy(flat_val.begin() + start[(size_t)src], flat_val.begin() + start[(size_t)src] + len[(size_t)src],
                  v.val.begin() + v.rowptr[(size_t)r]);
    }
    return LOCREC_OK;
}

// one family's rows re-indexed onto the union of person ids (absent persons get empty rows; both lists ascending)
void align(const std::vector<int64_t> &all, const Vectors &v, std::vector<int64_t> &rowptr)
{
    rowptr.assign(all.size() + 1, 0);
    size_t j = 0;
    for (size_t i = 0; i < all.size(); ++i) {
        int64_t l = 0;
        if (j < v.ids.size() && v.ids[j] == all[i]) {
            l = v.rowptr[j + 1] - v.rowptr[j];
            ++j;
        }
        rowptr[i + 1] = rowptr[i] + l;
    }
}

template <class T>
T *dup(const std::vector<T> &v)
{
    T *p = static_cast<T *>(std::malloc(std::max<size_t>(1, v.size()) * sizeof(T)));
    if (p && !v.empty()) std::memcpy(p, v.data(), v.size() * sizeof(T));
    return p;
}

}  // namespace

extern "C" const char *locrec_parquet_last_error(void) { return g_err.c_str(); }

extern "C" void locrec_parquet_free_knn(locrec_knn_arrays *a)
{
    if (!a) return;
    std::free(a->person_ids);
    std::free(a->p_rowptr);
    std::free(a->p_idx);
    std::free(a->p_val);
    std::free(a->c_rowptr);
    std::free(a->c_idx);
    std::free(a->c_val);
    std::free(a->r_rowptr);
    std::free(a->r_place);
    std::free(a->r_rating);
    std::free(a);
}

// KnnRecommenderMain.makeRecommendations' three loads (KnnRecommenderMain.scala:53-57, 69-88) -> the arrays of
// locrec_knn_create: persons = the union of the three sets' person ids, ascending; a person absent from a family gets
// an empty row there; the ratings grouped by person in file order.
extern "C" int32_t locrec_parquet_read_knn(const char *place_rating_vectors, const char *category_rating_vectors,
                                           const char *place_ratings, locrec_knn_arrays **out) try
{
    if (!place_rating_vectors || !category_rating_vectors || !place_ratings || !out) return fail(LOCREC_E_INVALID_ARG, "NULL argument");
    *out = nullptr;
    Vectors vp, vc;
    PQ_TRY(read_vectors(place_rating_vectors, vp));
    PQ_TRY(read_vectors(category_rating_vectors, vc));
    std::shared_ptr<arrow::Table> t;
    PQ_TRY(read_table(place_ratings, {"person_id", "place_id", "rating"}, t));
    std::vector<int64_t> rp, rplace, rrating;
    PQ_TRY(longs_of(t, "person_id", rp));
    PQ_TRY(longs_of(t, "place_id", rplace));
    PQ_TRY(longs_of(t, "rating", rrating));
    std::vector<int64_t> all(vp.ids);
    all.insert(all.end(), vc.ids.begin(), vc.ids.end());
    all.insert(all.end(), rp.begin(), rp.end());
    std::sort(all.begin(), all.end());
    all.erase(std::unique(all.begin(), all.end()), all.end());
    for (const Vectors *v : {&vp, &vc})
        if (std::adjacent_find(v->ids.begin(), v->ids.end()) != v->ids.end())
            return fail(LOCREC_E_INVALID_ARG, "duplicate person_id %lld in a rating-vector set",
                        (long long)*std::adjacent_find(v->ids.begin(), v->ids.end()));
    std::vector<int64_t> prp, crp;
    align(all, vp, prp);
    align(all, vc, crp);
    // ratings: stable counting sort by person row
    std::vector<int64_t> rrp(all.size() + 1, 0), rows(rp.size());
    for (size_t e = 0; e < rp.size(); ++e) {
        rows[e] = std::lower_bound(all.begin(), all.end(), rp[e]) - all.begin();
        ++rrp[(size_t)rows[e] + 1];
    }
    for (size_t i = 0; i < all.size(); ++i) rrp[i + 1] += rrp[i];
    std::vector<int64_t> cur(rrp.begin(), rrp.end() - 1), oplace(rp.size()), orating(rp.size());
    for (size_t e = 0; e < rp.size(); ++e) {
        const int64_t pos = cur[(size_t)rows[e]]++;
        oplace[(size_t)pos] = rplace[e];
        orating[(size_t)pos] = rrating[e];
    }
    auto *a = static_cast<locrec_knn_arrays *>(std::calloc(1, sizeof(locrec_knn_arrays)));
    if (!a) return fail(LOCREC_E_OOM, "host allocation failed");
    a->n = (int64_t)all.size();
    a->p_dim = std::max(1, vp.dim);
    a->c_dim = std::max(1, vc.dim);
    a->person_ids = dup(all);
    a->p_rowptr = dup(prp);
    a->p_idx = dup(vp.idx);
    a->p_val = dup(vp.val);
    a->c_rowptr = dup(crp);
    a->c_idx = dup(vc.idx);
    a->c_val = dup(vc.val);
    a->r_rowptr = dup(rrp);
    a->r_place = dup(oplace);
    a->r_rating = dup(orating);
    if (!a->person_ids || !a->p_rowptr || !a->p_idx || !a->p_val || !a->c_rowptr || !a->c_idx || !a->c_val || !a->r_rowptr ||
        !a->r_place || !a->r_rating) {
        locrec_parquet_free_knn(a);
        return fail(LOCREC_E_OOM, "host allocation failed");
    }
    *out = a;
    return LOCREC_OK;
} catch (const std::exception &e) {
    return fail(LOCREC_E_DEVICE, "internal error: %s", e.what());
} catch (...) {
    return fail(LOCREC_E_DEVICE, "internal error");
}

extern "C" int32_t locrec_knn_create_from_parquet(const char *place_rating_vectors, const char *category_rating_vectors,
                                                  const char *place_ratings, locrec_knn_index **out_index)
{
    if (!out_index) return fail(LOCREC_E_INVALID_ARG, "out_index is NULL");
    *out_index = nullptr;
    locrec_knn_arrays *a = nullptr;
    PQ_TRY(locrec_parquet_read_knn(place_rating_vectors, category_rating_vectors, place_ratings, &a));
    const int32_t st = locrec_knn_create(a->n, a->person_ids, a->p_rowptr, a->p_idx, a->p_val, a->p_dim, a->c_rowptr, a->c_idx, a->c_val,
                                         a->c_dim, a->r_rowptr, a->r_place, a->r_rating, out_index);
    locrec_parquet_free_knn(a);
    if (st != LOCREC_OK) g_err = locrec_last_error();
    return st;
}

extern "C" void locrec_parquet_free_edges(locrec_sg_edges *e)
{
    if (!e) return;
    std::free(e->source_ids);
    std::free(e->target_ids);
    std::free(e->balanced_weights);
    std::free(e);
}

// StochasticRecommenderMain.loadStochasticGraph (:78-84): (source_id, target_id, balanced_weight), file order kept
// (the device layout preserves edge-list order inside a row)
extern "C" int32_t locrec_parquet_read_edges(const char *stochastic_graph, locrec_sg_edges **out) try
{
    if (!stochastic_graph || !out) return fail(LOCREC_E_INVALID_ARG, "NULL argument");
    *out = nullptr;
    std::shared_ptr<arrow::Table> t;
    PQ_TRY(read_table(stochastic_graph, {"source_id", "target_id", "balanced_weight"}, t));
    std::vector<int64_t> s, d;
    std::vector<double> w;
    PQ_TRY(longs_of(t, "source_id", s));
    PQ_TRY(longs_of(t, "target_id", d));
    PQ_TRY(doubles_of(t, "balanced_weight", w));
    auto *e = static_cast<locrec_sg_edges *>(std::calloc(1, sizeof(locrec_sg_edges)));
    if (!e) return fail(LOCREC_E_OOM, "host allocation failed");
    e->n_edges = (int64_t)s.size();
    e->source_ids = dup(s);
    e->target_ids = dup(d);
    e->balanced_weights = dup(w);
    if (!e->source_ids || !e->target_ids || !e->balanced_weights) {
        locrec_parquet_free_edges(e);
        return fail(LOCREC_E_OOM, "host allocation failed");
    }
    *out = e;
    return LOCREC_OK;
} catch (const std::exception &e) {
    return fail(LOCREC_E_DEVICE, "internal error: %s", e.what());
} catch (...) {
    return fail(LOCREC_E_DEVICE, "internal error");
}

extern "C" int32_t locrec_sg_create_from_parquet(const char *stochastic_graph, locrec_sg_graph **out_graph)
{
    if (!out_graph) return fail(LOCREC_E_INVALID_ARG, "out_graph is NULL");
    *out_graph = nullptr;
    locrec_sg_edges *e = nullptr;
    PQ_TRY(locrec_parquet_read_edges(stochastic_graph, &e));
    const int32_t st = locrec_sg_create(e->n_edges, e->source_ids, e->target_ids, e->balanced_weights, out_graph);
    locrec_parquet_free_edges(e);
    if (st != LOCREC_OK) g_err = locrec_last_error();
    return st;
}
