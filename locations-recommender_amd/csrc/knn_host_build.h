// knn_host_build.h -- knn_create_host: the KNN index built on the host (round 1's single-threaded implementation), the A/B
// partner of knn_build.hip's device build behind LOCREC_KNN_HOST_BUILD=1 (tests/test_gpu_build.py).
// A fragment of knn.hip's translation unit: included by knn.hip at file scope, after its helper namespaces.
#pragma once

// The index built on the HOST (the first implementation, single-threaded): kept behind
// LOCREC_KNN_HOST_BUILD=1 as the A/B partner of knn_build.hip's device build (tests/test_gpu_build.py).
static int32_t knn_create_host(
    int64_t n, const int64_t *person_ids,
    const int64_t *p_rowptr, const int32_t *p_idx, const double *p_val, int32_t p_dim,
    const int64_t *c_rowptr, const int32_t *c_idx, const double *c_val, int32_t c_dim,
    const int64_t *r_rowptr, const int64_t *r_place, const int64_t *r_rating,
    locrec_knn_index **out) try
{
    if (!out) return fail(LOCREC_E_INVALID_ARG, "out_index is NULL");
    *out = nullptr;
    if (n < 0 || n >= ((int64_t)1 << 31) - 64) return fail(LOCREC_E_INVALID_ARG, "bad person count");
    if (n > 0 && (!person_ids || !p_rowptr || !c_rowptr)) return fail(LOCREC_E_INVALID_ARG, "NULL input array");
    if (p_dim <= 0 || c_dim <= 0) return fail(LOCREC_E_INVALID_ARG, "vector sizes must be positive");
    LOCREC_TRY(ensure_device());
    std::unique_ptr<locrec_knn_index> ix(new (std::nothrow) locrec_knn_index);
    if (!ix) return fail(LOCREC_E_OOM, "host allocation failed");
    LOCREC_HIP_TRY(hipGetDevice(&ix->device));
    LOCREC_HIP_TRY(hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking));
    ix->own_stream = true;
    ix->n = n;
    ix->nslices = (int32_t)((n + 63) / 64);
    ix->cand_slice0 = 0;
    ix->cand_slice1 = ix->nslices;
    knn_read_env(ix.get());
    const bool force_generic = ix->force_generic;

    const bool dbg_t = debug_env("LOCREC_DEBUG_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!dbg_t) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[locrec knn_create] %-28s %.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
        t_last = now;
    };
    // ---- validation (SparseVector invariants, RatingVectorsBuilder.scala:74-77)
    auto check_family = [&](const char *name, const int64_t *ptr, const int32_t *idx, const double *val,
                            int32_t dim, bool &integral, double &vmax, double &ssmax) -> int32_t {
        if (n == 0) return LOCREC_OK;
        if (ptr[0] != 0) return fail(LOCREC_E_INVALID_ARG, "%s rowptr must start at 0", name);
        for (int64_t r = 0; r < n; ++r) {
            if (ptr[r + 1] < ptr[r]) return fail(LOCREC_E_INVALID_ARG, "%s rowptr not monotone at %lld", name, (long long)r);
            double ss = 0;
            for (int64_t e = ptr[r]; e < ptr[r + 1]; ++e) {
                if (idx[e] < 0 || idx[e] >= dim)
                    return fail(LOCREC_E_INVALID_ARG, "%s index %d out of range [0,%d)", name, idx[e], dim);
                if (e > ptr[r] && idx[e] <= idx[e - 1])
                    return fail(LOCREC_E_INVALID_ARG, "%s indices of person %lld not strictly ascending", name,
                                (long long)person_ids[r]);
                const double v = val[e];
                if (!std::isfinite(v)) return fail(LOCREC_E_INVALID_ARG, "%s value is not finite", name);
                if (!(v >= 1.0) || v != std::floor(v)) integral = false;
                vmax = std::max(vmax, std::fabs(v));
                ss += v * v;
            }
            if (ptr[r + 1] > ptr[r] && !(ss > 0))
                return fail(LOCREC_E_INVALID_ARG, "%s vector of person %lld has zero norm", name,
                            (long long)person_ids[r]);
            ssmax = std::max(ssmax, ss);
        }
        return LOCREC_OK;
    };
    bool integral = true;
    double pvmax = 0, cvmax = 0, pss = 0, css = 0;
    LOCREC_TRY(check_family("place", p_rowptr, p_idx, p_val, p_dim, integral, pvmax, pss));
    LOCREC_TRY(check_family("category", c_rowptr, c_idx, c_val, c_dim, integral, cvmax, css));
    const int p_vbits = std::min(24, 32 - ceil_log2i(p_dim));
    const int c_vbits = std::min(24, 32 - ceil_log2i(c_dim));
    // exact u32 dots need every dot < 2^32; |dot| <= sqrt(ss_a * ss_b) <= max ss
    ix->packed = !force_generic && integral && p_dim < (1 << 20) - 1 && c_dim < (1 << 20) - 1 &&
                 pvmax < (double)(1u << p_vbits) && cvmax < (double)(1u << c_vbits) &&
                 pss < 4294967296.0 && css < 4294967296.0;

    ix->pack16 = ix->packed && pss < 65536.0 && css < 65536.0 && pvmax < 65536.0 && cvmax < 65536.0 &&
                 !ix->no_pack16;

    lap("validation");
    // ---- popularity split of the place family (PACKED formats, hashed panel): place indices are
    // renumbered by descending frequency (a permutation of the dimensions: every dot product is
    // unchanged, and integer sums do not depend on the order of the terms), so that a row's popular
    // indices come first.  new_of_old is empty when the split is not used.
    std::vector<int32_t> new_of_old;
    std::vector<int32_t> npop;  // per input row: number of indices that become < pop_h
    int32_t pop_h = 0;
    // head / tail form (knn_ht.h): PACK16 data whose values fit a byte and whose rows fit 24 bits
    const int ht_qt = 16;
    int32_t ht_h = std::min<int32_t>(p_dim, 512);
    if (ix->env_ht_h > 0) ht_h = std::min<int32_t>(p_dim, ix->env_ht_h);  // tuning
    ht_h = std::min<int32_t>(ht_h, 65536 / (2 * ht_qt));  // the element's low half is the panel row's byte offset
    const bool want_ht = ix->pack16 && !ix->no_ht && !force_generic && n > 0 && n < ((int64_t)1 << 24) && pvmax < 256.0 &&
                         cvmax < 256.0 && c_dim <= kHtCatRows;
    if (ix->packed && !ix->no_pop && n > 0 &&
        (ix->force_hash || (size_t)p_dim * 2 > (size_t)kDirectMaxBytes || want_ht)) {
        std::vector<int64_t> freq((size_t)p_dim, 0);
        for (int64_t e = 0; e < p_rowptr[n]; ++e) ++freq[p_idx[e]];
        std::vector<int32_t> by_freq((size_t)p_dim);
        std::iota(by_freq.begin(), by_freq.end(), 0);
        std::stable_sort(by_freq.begin(), by_freq.end(), [&](int32_t a, int32_t b) { return freq[a] > freq[b]; });
        new_of_old.resize((size_t)p_dim);
        for (int32_t i = 0; i < p_dim; ++i) new_of_old[by_freq[i]] = i;
        pop_h = std::min<int32_t>(p_dim, kPopTable);
        if (ix->env_pop_h > 0) pop_h = std::min<int32_t>(p_dim, ix->env_pop_h);  // tuning
        // third sort key of the rows: their count of popular indices - of HEAD indices when the head / tail
        // form is built, so that the head rows of a slice have (nearly) one length
        const int32_t key_h = want_ht ? ht_h : pop_h;
        npop.assign((size_t)n, 0);
        for (int64_t r = 0; r < n; ++r)
            for (int64_t e = p_rowptr[r]; e < p_rowptr[r + 1]; ++e) npop[r] += new_of_old[p_idx[e]] < key_h ? 1 : 0;
    }

    lap("popularity");
    // ---- row order: ascending (nnz_place, nnz_category[, popular count]), stable
    std::vector<int32_t> order((size_t)n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) {
        const int64_t pa = p_rowptr[a + 1] - p_rowptr[a], pb = p_rowptr[b + 1] - p_rowptr[b];
        if (pa != pb) return pa < pb;
        const int64_t ca = c_rowptr[a + 1] - c_rowptr[a], cb = c_rowptr[b + 1] - c_rowptr[b];
        if (ca != cb) return ca < cb;
        return !npop.empty() && npop[a] < npop[b];
    });
    ix->ids_row.resize((size_t)n);
    ix->row_of_input.resize((size_t)n);
    for (int64_t r = 0; r < n; ++r) {
        ix->ids_row[r] = person_ids[order[r]];
        ix->row_of_input[order[r]] = (int32_t)r;
    }
    {
        ix->row_by_rank.resize((size_t)n);
        std::iota(ix->row_by_rank.begin(), ix->row_by_rank.end(), 0);
        std::sort(ix->row_by_rank.begin(), ix->row_by_rank.end(), [&](int32_t a, int32_t b) { return ix->ids_row[a] < ix->ids_row[b]; });
        ix->ids_sorted.resize((size_t)n);
        for (int64_t k = 0; k < n; ++k) ix->ids_sorted[k] = ix->ids_row[ix->row_by_rank[k]];
        const auto dupit = std::adjacent_find(ix->ids_sorted.begin(), ix->ids_sorted.end());
        if (dupit != ix->ids_sorted.end()) return fail(LOCREC_E_INVALID_ARG, "duplicate person_id %lld", (long long)*dupit);
    }
    lap("row order + id map");
    auto gather = [&](const int64_t *ptr, const int32_t *idx, const double *val, int32_t dim, int vbits,
                      HostFamily &h) {
        h.dim = dim;
        h.vbits = vbits;
        h.ptr.assign((size_t)n + 1, 0);
        for (int64_t r = 0; r < n; ++r) h.ptr[r + 1] = h.ptr[r] + (ptr[order[r] + 1] - ptr[order[r]]);
        h.idx.resize((size_t)h.ptr[n]);
        h.val.resize((size_t)h.ptr[n]);
        for (int64_t r = 0; r < n; ++r) {
            const int64_t b = ptr[order[r]], len = ptr[order[r] + 1] - b;
            std::copy(idx + b, idx + b + len, h.idx.begin() + h.ptr[r]);
            std::copy(val + b, val + b + len, h.val.begin() + h.ptr[r]);
            h.max_nnz = std::max(h.max_nnz, (int32_t)len);
        }
    };
    {
        HostFamily hp, hc;
        gather(p_rowptr, p_idx, p_val, p_dim, p_vbits, hp);
        gather(c_rowptr, c_idx, c_val, c_dim, c_vbits, hc);
        if (!new_of_old.empty()) {
            // the device image of the place family in the renumbered dimensions, rows re-sorted by the
            // new index; hp itself keeps the caller's indices (the default ratings below use them)
            HostFamily hq = hp;
            std::vector<std::pair<int32_t, double>> tmp;
            for (int64_t r = 0; r < n; ++r) {
                tmp.clear();
                for (int64_t e = hp.ptr[r]; e < hp.ptr[r + 1]; ++e) tmp.emplace_back(new_of_old[hp.idx[e]], hp.val[e]);
                std::sort(tmp.begin(), tmp.end());
                for (size_t j = 0; j < tmp.size(); ++j) {
                    hq.idx[hp.ptr[r] + j] = tmp[j].first;
                    hq.val[hp.ptr[r] + j] = tmp[j].second;
                }
            }
            LOCREC_TRY(build_family_device(ix.get(), hq, ix->fp, ix->packed));
            if (want_ht) {
                LOCREC_TRY(build_ht(ix.get(), hq, hc, ht_h, ht_qt));
                lap("head / tail image");
            }
            // leading element groups (dwordx4 = 4 elements) that are popular in EVERY lane of the slice;
            // padding elements are index 0, which is popular
            std::vector<int32_t> split((size_t)ix->nslices, 0);
            for (int32_t sl = 0; sl < ix->nslices; ++sl) {
                int w = 0, g = INT32_MAX;
                for (int64_t r = (int64_t)sl * 64; r < std::min<int64_t>(n, (int64_t)sl * 64 + 64); ++r) {
                    const int len = (int)(hq.ptr[r + 1] - hq.ptr[r]);
                    w = std::max(w, len);
                    int np_r = 0;
                    while (np_r < len && hq.idx[hq.ptr[r] + np_r] < pop_h) ++np_r;
                    if (np_r < len) g = std::min(g, np_r / 4);
                }
                split[sl] = std::min(g, ((w + 3) & ~3) / 4);
            }
            LOCREC_TRY(ix->fp.sell_split.upload(split, ix->stream));
            LOCREC_HIP_TRY(hipStreamSynchronize(ix->stream));
            ix->fp.pop_h = pop_h;
            ix->fp.scan_bytes += (int64_t)ix->nslices * 4;
        } else {
            LOCREC_TRY(build_family_device(ix.get(), hp, ix->fp, ix->packed));
        }
        LOCREC_TRY(build_family_device(ix.get(), hc, ix->fc, ix->packed));
        lap("families (gather, SELL, upload)");
        // ratings CSR in row order
        std::vector<int64_t> rp((size_t)n + 1, 0), rplace;
        std::vector<double> rrating;
        if (r_rowptr) {
            if (n > 0 && (!r_place || !r_rating)) return fail(LOCREC_E_INVALID_ARG, "NULL ratings array");
            for (int64_t r = 0; r < n; ++r) {
                const int64_t len = r_rowptr[order[r] + 1] - r_rowptr[order[r]];
                if (len < 0) return fail(LOCREC_E_INVALID_ARG, "ratings rowptr not monotone");
                rp[r + 1] = rp[r] + len;
            }
            rplace.resize((size_t)rp[n]);
            rrating.resize((size_t)rp[n]);
            for (int64_t r = 0; r < n; ++r) {
                const int64_t b = r_rowptr[order[r]];
                for (int64_t e = 0; e < rp[r + 1] - rp[r]; ++e) {
                    rplace[rp[r] + e] = r_place[b + e];
                    rrating[rp[r] + e] = (double)r_rating[b + e];  // Long * Double promotes (:59)
                }
            }
        } else {
            rp = hp.ptr;
            rplace.resize(hp.idx.size());
            for (size_t e = 0; e < hp.idx.size(); ++e) rplace[e] = hp.idx[e];
            rrating = hp.val;
        }
        for (int64_t r = 0; r < n; ++r) ix->max_r_nnz = std::max(ix->max_r_nnz, rp[r + 1] - rp[r]);
        {
            // place-major transpose of the ratings (rows ascending inside a place: a fixed order)
            std::vector<int64_t> &cpl = ix->cplace_ids;
            std::vector<int32_t> pidx_of(rplace.size());
            int64_t mn = 0, mx = -1;
            if (!rplace.empty()) {
                const auto mm = std::minmax_element(rplace.begin(), rplace.end());
                mn = *mm.first;
                mx = *mm.second;
            }
            if (!rplace.empty() && mx - mn < ((int64_t)1 << 26)) {
                // place ids span a moderate range (they do in the reference: one global id space):
                // distinct ids and their ranks from a presence table, no 25 M-element sort
                std::vector<int32_t> rank((size_t)(mx - mn + 1), 0);
                for (const int64_t pl : rplace) rank[(size_t)(pl - mn)] = 1;
                int32_t acc = 0;
                cpl.clear();
                for (size_t i = 0; i < rank.size(); ++i) {
                    if (rank[i]) {
                        rank[i] = acc++;
                        cpl.push_back(mn + (int64_t)i);
                    } else {
                        rank[i] = -1;
                    }
                }
                for (size_t e = 0; e < rplace.size(); ++e) pidx_of[e] = rank[(size_t)(rplace[e] - mn)];
            } else {
                cpl = rplace;
                std::sort(cpl.begin(), cpl.end());
                cpl.erase(std::unique(cpl.begin(), cpl.end()), cpl.end());
                for (size_t e = 0; e < rplace.size(); ++e)
                    pidx_of[e] = (int32_t)(std::lower_bound(cpl.begin(), cpl.end(), rplace[e]) - cpl.begin());
            }
            const int64_t ncp = (int64_t)cpl.size();
            std::vector<int64_t> cptr((size_t)ncp + 1, 0);
            for (size_t e = 0; e < rplace.size(); ++e) ++cptr[pidx_of[e] + 1];
            for (int64_t i = 0; i < ncp; ++i) cptr[i + 1] += cptr[i];
            std::vector<int64_t> cur(cptr.begin(), cptr.end() - 1);
            std::vector<int32_t> crow(rplace.size());
            std::vector<double> crat(rplace.size());
            for (int64_t r = 0; r < n; ++r)
                for (int64_t e = rp[r]; e < rp[r + 1]; ++e) {
                    const int64_t pos = cur[pidx_of[e]]++;
                    crow[pos] = (int32_t)r;
                    crat[pos] = rrating[e];
                }
            LOCREC_TRY(ix->r_pidx.upload(pidx_of, ix->stream));
            LOCREC_TRY(ix->cplace_dev.upload(cpl, ix->stream));
            LOCREC_TRY(ix->cp_ptr.upload(cptr, ix->stream));
            LOCREC_TRY(ix->cp_row.upload(crow, ix->stream));
            LOCREC_TRY(ix->cp_rating.upload(crat, ix->stream));
            LOCREC_HIP_TRY(hipStreamSynchronize(ix->stream));
        }
        LOCREC_TRY(ix->r_ptr.upload(rp, ix->stream));
        LOCREC_TRY(ix->r_place.upload(rplace, ix->stream));
        LOCREC_TRY(ix->r_rating.upload(rrating, ix->stream));
        LOCREC_HIP_TRY(hipStreamSynchronize(ix->stream));
    }
    lap("ratings (CSR + transpose)");
    // ---- rid: rank of each row's person id (tie-break person_id asc)
    {
        const std::vector<int32_t> &by_id = ix->row_by_rank;
        std::vector<uint32_t> rid((size_t)n);
        std::vector<int64_t> ids_sorted((size_t)n);
        for (int64_t k = 0; k < n; ++k) {
            rid[by_id[k]] = (uint32_t)k;
            ids_sorted[k] = ix->ids_row[by_id[k]];
        }
        LOCREC_TRY(ix->rid.upload(rid, ix->stream));
        if (ix->ht.ready) {  // the same ranks padded to whole slices (knn_scan_ht loads them unconditionally)
            rid.resize((size_t)ix->nslices * 64, 0u);
            LOCREC_TRY(ix->ht.rid.upload(rid, ix->stream));
        }
        LOCREC_TRY(ix->ids_by_rank.upload(ids_sorted, ix->stream));
        LOCREC_TRY(ix->row_of_rid.upload(by_id, ix->stream));
        LOCREC_HIP_TRY(hipStreamSynchronize(ix->stream));
    }
    lap("rid + norms + sync");
    *out = ix.release();
    return LOCREC_OK;
} LOCREC_CATCH_ALL

static_assert(sizeof(HtCold) <= (size_t)cfg::kHtColdBytes, "cfg::kHtColdBytes is the size of the device buffer that holds a launch's HtCold");
static_assert(kHtNP == cfg::kHtNP && kHtCatRows == cfg::kHtCatRows, "knn_ht.h and knn_index.h disagree");
