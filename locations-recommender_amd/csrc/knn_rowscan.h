// knn_rowscan.h -- knn_scan: the row scan of a query tile against all candidates (GENERIC / PACK32 / PACK16 formats; the head / tail form is knn_ht.h)
// A fragment of knn.hip's translation unit: included by knn.hip inside its anonymous namespace, after the parameter
// blocks and the headers it names (it is not a stand-alone header; the split only keeps every file readable).
#pragma once

// MODE 0 = GENERIC (fp64 values), 1 = PACK32, 2 = PACK16; W = waves per block (all share the tile).
// second launch bound = waves per SIMD: an 8-wave block must fit twice per CU (<= 128 VGPRs)
template <int MODE, int QT, int W>
__global__ __launch_bounds__(W * 64, W >= 8 ? (QT >= 32 && W == 8 ? 2 : 4) : 1) void knn_scan(const ScanParams P)
{
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr bool PACKED = MODE != 0;
    double *cand_s = reinterpret_cast<double *>(smem + P.off_cand_s);
    uint32_t *cand_r = reinterpret_cast<uint32_t *>(smem + P.off_cand_rid);
    // misc block: doubles first (alignment)
    double *s_qnp = reinterpret_cast<double *>(smem + P.off_misc);
    double *s_qnc = s_qnp + QT;
    double *tau_s = s_qnc + QT;
    uint32_t *tau_r = reinterpret_cast<uint32_t *>(tau_s + QT);
    int *s_qrow = reinterpret_cast<int *>(tau_r + QT);
    int *cnt = s_qrow + QT;
    float *s_qfp = reinterpret_cast<float *>(cnt + QT);  // pw / |q_place|   (prefilter)
    float *s_qfc = s_qfp + QT;                           // cw / |q_category|
    float *tau32 = s_qfc + QT;
    int *s_nrows = reinterpret_cast<int *>(tau32 + QT);
    int *s_flags = s_nrows + 1;  // [0] overflow seen
    // per-wave survivor queues of the fast path
    double *wq_s = reinterpret_cast<double *>(smem + P.off_queue);
    uint32_t *wq_r = reinterpret_cast<uint32_t *>(wq_s + W * kQueueCap);
    uint32_t *wq_q = wq_r + W * kQueueCap;
    int *wq_cnt = reinterpret_cast<int *>(wq_q + W * kQueueCap);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q0 = blockIdx.y * QT;
    const int nqt = min(QT, P.nq - q0);
    const int K = P.K, S = P.S;
    const double pw = P.pw, cw = P.cw;

    if (P.poison & 1) {
        for (int i = tid; i < P.lds_bytes / 4; i += blockDim.x) reinterpret_cast<uint32_t *>(smem)[i] = 0xA5A5A5A5u;
        __syncthreads();
    }
    if (tid < QT) {
        int row = -1;
        if (tid < nqt) row = P.qrows ? P.qrows[q0 + tid] : P.qrow0 + q0 + tid;
        s_qrow[tid] = row;
        const double np_ = row >= 0 ? P.fp.norm[row] : 0.0;
        const double nc_ = row >= 0 ? P.fc.norm[row] : 0.0;
        s_qnp[tid] = np_;
        s_qnc[tid] = nc_;
        s_qfp[tid] = np_ > 0.0 ? (float)(pw / np_) : 0.0f;
        s_qfc[tid] = nc_ > 0.0 ? (float)(cw / nc_) : 0.0f;
        tau_s[tid] = 0.0;  // every candidate has s > 0, so (0, 0) admits them all
        tau_r[tid] = 0u;
        tau32[tid] = 1.17549435e-38f;  // prefilter threshold / 1.0001, floored at FLT_MIN
        cnt[tid] = 0;
    }
    if (tid < W) wq_cnt[tid] = 0;
    if (tid == 0) {
        s_flags[0] = 0;
        s_flags[1] = 0;
        s_flags[2] = 0;
    }
    __syncthreads();
    // MODE 3 (knn_ht.h): category panel at LDS offset 0, place head panel right behind it - both at
    // compile-time offsets, so that an element's low half IS the ds_read address
    constexpr int kHtCatBytes = kHtCatRows * QT * 2;
    if constexpr (MODE == 3) {
        uint32_t *t32 = reinterpret_cast<uint32_t *>(smem + P.ht.off_tail);
        for (int i = tid; i < W * 64 * QT / 2; i += blockDim.x) t32[i] = 0u;  // (synchronised by the panel builds)
        ht_build_panel<QT>(P.fc, P.ht.c_rows, kHtCatRows, s_qrow, nqt, reinterpret_cast<unsigned short *>(smem));
        ht_build_panel<QT>(P.fp, P.ht.h, cfg::ht_plane_rows(P.ht.h), s_qrow, nqt, reinterpret_cast<unsigned short *>(smem + kHtCatBytes));
    } else if constexpr (MODE == 1) {
        build_panel_packed<QT, uint32_t>(P.fp, s_qrow, nqt, reinterpret_cast<uint32_t *>(smem + P.fp.off_hash),
                                         reinterpret_cast<uint32_t *>(smem + P.fp.off_panel), s_nrows, P.fp.pop_h > 0 && !P.fp.direct ? reinterpret_cast<unsigned short *>(smem + P.fp.off_pop) : nullptr);
        build_panel_packed<QT, uint32_t>(P.fc, s_qrow, nqt, reinterpret_cast<uint32_t *>(smem + P.fc.off_hash),
                                         reinterpret_cast<uint32_t *>(smem + P.fc.off_panel), s_nrows, P.fc.pop_h > 0 && !P.fc.direct ? reinterpret_cast<unsigned short *>(smem + P.fc.off_pop) : nullptr);
    } else if constexpr (MODE == 2) {
        build_panel_packed<QT, uint16_t>(P.fp, s_qrow, nqt, reinterpret_cast<uint32_t *>(smem + P.fp.off_hash),
                                         reinterpret_cast<uint16_t *>(smem + P.fp.off_panel), s_nrows, P.fp.pop_h > 0 && !P.fp.direct ? reinterpret_cast<unsigned short *>(smem + P.fp.off_pop) : nullptr);
        build_panel_packed<QT, uint16_t>(P.fc, s_qrow, nqt, reinterpret_cast<uint32_t *>(smem + P.fc.off_hash),
                                         reinterpret_cast<uint16_t *>(smem + P.fc.off_panel), s_nrows, P.fc.pop_h > 0 && !P.fc.direct ? reinterpret_cast<unsigned short *>(smem + P.fc.off_pop) : nullptr);
    } else {
        build_panel_generic<QT>(P.fp, s_qrow, nqt, reinterpret_cast<uint2 *>(smem + P.fp.off_hash),
                                reinterpret_cast<double *>(smem + P.fp.off_panel), s_nrows);
        build_panel_generic<QT>(P.fc, s_qrow, nqt, reinterpret_cast<uint2 *>(smem + P.fc.off_hash),
                                reinterpret_cast<double *>(smem + P.fc.off_panel), s_nrows);
    }

    const int slice_begin = P.slice0 + blockIdx.x * P.slices_per_chunk;
    const int slice_end = min(slice_begin + P.slices_per_chunk, P.nslices);
    const int iters = (P.slices_per_chunk + W - 1) / W;

    // Insertion mode (block-uniform).  Survivors are inserted synchronously, slice by slice, until
    // every query has a full list AND the block has seen few survivors for kCalmIters iterations
    // in a row; then they go to per-wave LDS queues drained every kFlushEvery slices.  A queue found
    // more than half full at a drain sends the block back to synchronous insertion.
    bool fastmode = false;
    int calm = 0;
    // MODE 3: hits of this tile (knn_ht.h), software-pipelined: while slice s is processed the first 64
    // hits of slice s + W are in flight and the offsets of slice s + 2W are being fetched
    const uint32_t *ht_off_row = nullptr, *ht_hits = nullptr;
    unsigned char *my_tail = nullptr;
    int ht_primed = -1;                          // slice the pipeline registers below are valid for
    uint32_t ht_c0 = 0, ht_c1 = 0, ht_hcur = 0;  // current slice: hit range and its first 64 hits
    uint32_t ht_n0 = 0, ht_n1 = 0;               // next slice (s + W): hit range
    if constexpr (MODE == 3) {
        ht_off_row = P.ht.off + (int64_t)blockIdx.y * P.ht.off_stride - P.slice0;
        ht_hits = P.ht.hits + P.ht.tile_base[blockIdx.y];
        my_tail = smem + P.ht.off_tail + wave * (64 * QT * 2);
    }
    for (int it = 0; it < iters; ++it) {
        const int slice = slice_begin + it * W + wave;  // wave-uniform
        const bool live = slice < slice_end;
        const int row = slice * 64 + lane;
        const bool valid = live && row < P.nrows;
        unsigned pend = 0;
        if constexpr (PACKED) {
            Acc<MODE, QT> accp, accc;
            accp.zero();
            accc.zero();
            float icnp = 0.0f, icnc = 0.0f;
            double cnp = 0.0, cnc = 0.0;
            uint32_t myrid = 0u;
            if constexpr (MODE == 3) {
                if (live) {
                    const u32x4 *bp = reinterpret_cast<const u32x4 *>(P.fp.sell + P.fp.sell_off[slice]) + lane;
                    const u32x4 *bc = reinterpret_cast<const u32x4 *>(P.fc.sell + P.fc.sell_off[slice]) + lane;
                    const int w4p = __builtin_amdgcn_readfirstlane(P.fp.sell_w[slice] >> 2);
                    const int w4c = __builtin_amdgcn_readfirstlane(P.fc.sell_w[slice] >> 2);
                    const Group4 gp = load_group(bp, 0, w4p);
                    const Group4 gc = load_group(bc, 0, w4c);
                    if (valid) {
                        icnp = P.fp.inorm32[row];
                        icnc = P.fc.inorm32[row];
                        cnp = P.fp.norm[row];
                        cnc = P.fc.norm[row];
                        myrid = P.rid[row];
                    }
                    // --- hit pipeline (see the declarations in front of the loop)
                    const int nslice = slice + W;
                    const bool have_next = it + 1 < iters && nslice < slice_end;
                    if (ht_primed != slice) {  // first iteration of the block, or an interval is being replayed
                        ht_c0 = __builtin_amdgcn_readfirstlane(ht_off_row[slice]);
                        ht_c1 = __builtin_amdgcn_readfirstlane(ht_off_row[slice + 1]);
                        ht_hcur = (uint32_t)lane < ht_c1 - ht_c0 ? ht_hits[ht_c0 + lane] : 0u;
                        if (have_next) {
                            ht_n0 = __builtin_amdgcn_readfirstlane(ht_off_row[nslice]);
                            ht_n1 = __builtin_amdgcn_readfirstlane(ht_off_row[nslice + 1]);
                        }
                    }
                    const uint32_t hn = ht_c1 - ht_c0;  // hits of this slice and tile (wave-uniform)
                    const uint32_t hbase = ht_c0;
                    const uint32_t hfirst = ht_hcur;
                    uint32_t hnext = 0u, m0 = 0u, m1 = 0u;
                    if (have_next) {
                        hnext = (uint32_t)lane < ht_n1 - ht_n0 ? ht_hits[ht_n0 + lane] : 0u;
                        if (it + 2 < iters && nslice + W < slice_end) {  // unwaited here: read in the next iteration
                            m0 = ht_off_row[nslice + W];
                            m1 = ht_off_row[nslice + W + 1];
                        }
                    }
                    uint32_t ap[QT / 2], ac[QT / 2];
#pragma unroll
                    for (int i = 0; i < QT / 2; ++i) ap[i] = ac[i] = 0u;
                    ht_family_dots<QT>(smem + kHtCatBytes, cfg::ht_plane_rows(P.ht.h) * 16, bp, w4p, gp, ap);
                    ht_family_dots<QT>(smem, kHtCatRows * 16, bc, w4c, gc, ac);
                    if (hn > 0) {
                        // the tail: this slice's hits go into the wave's private accumulator (a wave's LDS
                        // operations execute in order), each lane folds its own row into its head dots
                        // and the words that were touched are cleared again
                        uint32_t hh = hfirst, waddr = 0u;
                        for (uint32_t done = 0;;) {
                            const uint32_t nb = min(64u, hn - done);
                            if ((uint32_t)lane < nb) {
                                const uint32_t q = (hh >> 16) & 31u;
                                waddr = ht_tail_word<QT>(hh >> 21, q >> 1);
                                atomicAdd(reinterpret_cast<uint32_t *>(my_tail + waddr), (hh & 0xFFFFu) << ((q & 1u) * 16u));
                            }
                            done += nb;
                            if (done >= hn) break;
                            hh = (uint32_t)lane < hn - done ? ht_hits[hbase + done + lane] : 0u;  // > 64 hits: rare
                        }
                        constexpr uint32_t chunks = QT / 8;
                        const uint32_t sw = chunks == 2 ? (((uint32_t)lane >> 3) & 1u) : (((uint32_t)lane >> 2) & 3u);
                        u32x4 trow[chunks];
#pragma unroll
                        for (uint32_t c = 0; c < chunks; ++c)
                            trow[c] = *reinterpret_cast<const u32x4 *>(my_tail + lane * (QT * 2) + (((c ^ sw) & (chunks - 1)) << 4));
                        if (hn <= 64u) {
                            if ((uint32_t)lane < hn) *reinterpret_cast<uint32_t *>(my_tail + waddr) = 0u;
                        } else {
#pragma unroll
                            for (uint32_t c = 0; c < chunks; ++c)
                                *reinterpret_cast<u32x4 *>(my_tail + lane * (QT * 2) + (c << 4)) = u32x4{0u, 0u, 0u, 0u};
                        }
#pragma unroll
                        for (uint32_t c = 0; c < chunks; ++c) {
                            const uint32_t tw[4] = {trow[c].x, trow[c].y, trow[c].z, trow[c].w};
#pragma unroll
                            for (int z = 0; z < 4; ++z)
                                ap[4 * c + z] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(u16x2, ap[4 * c + z]) +
                                                                                 __builtin_bit_cast(u16x2, tw[z]));
                        }
                    }
#pragma unroll
                    for (int i = 0; i < QT / 2; ++i) {
                        accp.a[i] = __builtin_bit_cast(u16x2, ap[i]);
                        accc.a[i] = __builtin_bit_cast(u16x2, ac[i]);
                    }
                    // shift the pipeline
                    ht_c0 = ht_n0;
                    ht_c1 = ht_n1;
                    ht_hcur = hnext;
                    ht_n0 = __builtin_amdgcn_readfirstlane(m0);
                    ht_n1 = __builtin_amdgcn_readfirstlane(m1);
                    ht_primed = have_next ? nslice : -1;
                }
            } else if (live) {
                const HotFam hp = make_hot(P.fp, smem);
                const HotFam hc = make_hot(P.fc, smem);
                const u32x4 *bp = reinterpret_cast<const u32x4 *>(P.fp.sell + P.fp.sell_off[slice]) + lane;
                const u32x4 *bc = reinterpret_cast<const u32x4 *>(P.fc.sell + P.fc.sell_off[slice]) + lane;
                const int w4p = __builtin_amdgcn_readfirstlane(P.fp.sell_w[slice] >> 2);
                const int w4c = __builtin_amdgcn_readfirstlane(P.fc.sell_w[slice] >> 2);
                // both families' first groups and the row scalars go out before any use
                const Group4 gp = load_group(bp, 0, w4p);
                const Group4 gc = load_group(bc, 0, w4c);
                if (valid) {  // all of the row's scalars now: a load issued in the epilogue would stall the wave
                    icnp = P.fp.inorm32[row];
                    icnc = P.fc.inorm32[row];
                    cnp = P.fp.norm[row];
                    cnc = P.fc.norm[row];
                    myrid = P.rid[row];
                }
                const int sp4 = P.fp.sell_split ? __builtin_amdgcn_readfirstlane(P.fp.sell_split[slice]) : 0;
                if constexpr (MODE != 3) {
                    family_dots_packed<MODE, QT>(hp, bp, w4p, gp, accp, sp4);
                    family_dots_packed<MODE, QT>(hc, bc, w4c, gc, accc);
                }
            }
            // f32 upper-bound prefilter: only pairs that can still enter the query's list pay for
            // the fp64 divide.  Relative error of s32 < 1e-6; the 1e-4 margin makes it one-sided.
            unsigned maybe = 0;
            if constexpr (MODE == 3) {
                // the same f32 bound, but every query is tested and - rarely - resolved on the spot: one
                // v_cmp and a scalar branch per pair instead of building a per-lane bit mask
                float fq[QT], gq[QT], tq[QT];
#pragma unroll
                for (int i = 0; i < QT / 4; ++i) {
                    const float4 a = reinterpret_cast<const float4 *>(s_qfp)[i];
                    const float4 b = reinterpret_cast<const float4 *>(s_qfc)[i];
                    const float4 c = reinterpret_cast<const float4 *>(tau32)[i];
                    fq[4 * i] = a.x; fq[4 * i + 1] = a.y; fq[4 * i + 2] = a.z; fq[4 * i + 3] = a.w;
                    gq[4 * i] = b.x; gq[4 * i + 1] = b.y; gq[4 * i + 2] = b.z; gq[4 * i + 3] = b.w;
                    tq[4 * i] = c.x; tq[4 * i + 1] = c.y; tq[4 * i + 2] = c.z; tq[4 * i + 3] = c.w;
                }
#pragma unroll
                for (int q = 0; q < QT; ++q) {
                    const float sp = ((float)accp.get(q) * icnp) * fq[q];
                    const float s32 = __builtin_fmaf((float)accc.get(q) * icnc, gq[q], sp);
                    if (s32 >= tq[q]) {  // tq = threshold / 1.0001, never below FLT_MIN: s32 == 0 fails
                        if (q < nqt && row != s_qrow[q]) {  // person_id =!= personId (:89)
                            double s;
                            if (exact_similarity(accp.get(q), accc.get(q), cnp, cnc, s_qnp[q], s_qnc[q], pw, cw, s) &&
                                better(s, myrid, tau_s[q], tau_r[q]))
                                pend |= 1u << q;
                        }
                    }
                }
            } else if (!(P.poison & 4)) {  // (bit 4: timing experiment without the epilogue; results are wrong)
                // per-query constants come out of LDS in wide reads, all before the arithmetic, and
                // the mask is built without branches (16 dependent LDS round trips otherwise)
                float fq[QT], gq[QT], tq[QT];
                if constexpr (QT >= 4) {
#pragma unroll
                    for (int i = 0; i < QT / 4; ++i) {
                        const float4 a = reinterpret_cast<const float4 *>(s_qfp)[i];
                        const float4 b = reinterpret_cast<const float4 *>(s_qfc)[i];
                        const float4 c = reinterpret_cast<const float4 *>(tau32)[i];
                        fq[4 * i] = a.x; fq[4 * i + 1] = a.y; fq[4 * i + 2] = a.z; fq[4 * i + 3] = a.w;
                        gq[4 * i] = b.x; gq[4 * i + 1] = b.y; gq[4 * i + 2] = b.z; gq[4 * i + 3] = b.w;
                        tq[4 * i] = c.x; tq[4 * i + 1] = c.y; tq[4 * i + 2] = c.z; tq[4 * i + 3] = c.w;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < QT; ++q) {
                        fq[q] = s_qfp[q];
                        gq[q] = s_qfc[q];
                        tq[q] = tau32[q];
                    }
                }
                const bool force = (P.poison & 2) != 0;
#pragma unroll
                for (int q = 0; q < QT; ++q) {
                    const float sp = ((float)accp.get(q) * icnp) * fq[q];
                    const float s32 = __builtin_fmaf((float)accc.get(q) * icnc, gq[q], sp);
                    // tq = threshold / 1.0001, never below FLT_MIN: s32 == 0 (no overlap at all) fails
                    const bool pass = (s32 >= tq[q]) | (force & (s32 > 0.0f));
                    maybe |= pass ? (1u << q) : 0u;
                }
            } else {
#pragma unroll
                for (int q = 0; q < QT; ++q) asm volatile("" ::"v"(accp.get(q)), "v"(accc.get(q)));
            }
            if (maybe) {
#pragma unroll
                for (int q = 0; q < QT; ++q) {
                    if ((maybe & (1u << q)) && q < nqt && row != s_qrow[q]) {  // person_id =!= personId (:89)
                        double s;
                        if (exact_similarity(accp.get(q), accc.get(q), cnp, cnc, s_qnp[q], s_qnc[q], pw, cw, s) &&
                            better(s, myrid, tau_s[q], tau_r[q]))
                            pend |= 1u << q;
                    }
                }
            }
            // Survivors (see "Insertion mode" above).
            if (!fastmode) {
                int np = __syncthreads_count(pend != 0);
                if (P.fast) {
                    bool warm = np <= P.enter_threads * W / 8;
#pragma unroll
                    for (int q = 0; q < QT; ++q) warm = warm && (q >= nqt || tau32[q] > 1.17549435e-38f);
                    calm = warm ? calm + 1 : 0;
                }
                while (np) {
                    if (pend) {
#pragma unroll
                        for (int q = 0; q < QT; ++q) {
                            if (pend & (1u << q)) {
                                double s;
                                exact_similarity(accp.get(q), accc.get(q), cnp, cnc, s_qnp[q], s_qnc[q], pw, cw, s);
                                if (!better(s, myrid, tau_s[q], tau_r[q])) {  // the list tightened meanwhile
                                    pend &= ~(1u << q);
                                    continue;
                                }
                                const int pos = atomicAdd(&cnt[q], 1);
                                if (pos < S) {
                                    cand_s[q * S + pos] = s;
                                    cand_r[q * S + pos] = myrid;
                                    pend &= ~(1u << q);
                                }
                            }
                        }
                    }
                    __syncthreads();
                    for (int q = 0; q < nqt; ++q)
                        if (cnt[q] >= S) compact_query(cand_s, cand_r, cnt, tau_s, tau_r, tau32, q, S, K);
                    np = __syncthreads_count(pend != 0);
                }
                // queues are drained at multiples of kFlushEvery: enter the fast mode on such a boundary
                if (calm >= kCalmIters && ((it + 1) & P.flush_mask) == 0) fastmode = true;
            } else {
                if (pend) {
#pragma unroll
                    for (int q = 0; q < QT; ++q) {
                        if (pend & (1u << q)) {
                            double s;
                            exact_similarity(accp.get(q), accc.get(q), cnp, cnc, s_qnp[q], s_qnc[q], pw, cw, s);
                            const int pos = atomicAdd(&wq_cnt[wave], 1);
                            if (pos < kQueueCap) {
                                wq_s[wave * kQueueCap + pos] = s;
                                wq_r[wave * kQueueCap + pos] = myrid;
                                wq_q[wave * kQueueCap + pos] = (uint32_t)q;
                            } else {
                                s_flags[1] = 1;  // no room: this interval is replayed synchronously (below)
                            }
                        }
                    }
                }
                if (((it + 1) & P.flush_mask) == 0 || it == iters - 1) {
                    __syncthreads();
                    if (s_flags[1]) {
                        // A burst (typically a run of tied candidates) overran a wave's queue.  Nothing
                        // of this interval has reached the lists yet (queues are only drained here) and
                        // the thresholds have not moved, so the interval is simply run again with
                        // synchronous insertion: discard the queues and go back to its first slice.
                        __syncthreads();
                        if (tid < W) wq_cnt[tid] = 0;
                        if (tid == 0) {
                            s_flags[1] = 0;
                            s_flags[2] += 1;  // statistics: replayed intervals of this block
                        }
                        __syncthreads();
                        fastmode = false;
                        calm = 0;
                        it = (it & ~P.flush_mask) - 1;  // ++it -> first iteration of the interval
                        continue;
                    }
                    int rounds = 0, maxfill = 0;
#pragma unroll
                    for (int w = 0; w < W; ++w) {
                        maxfill = max(maxfill, wq_cnt[w]);
                        rounds = max(rounds, (min(wq_cnt[w], kQueueCap) + 63) >> 6);
                    }
                    for (int r = 0; r < rounds; ++r) {
                        const int i = lane + 64 * r;
                        const bool have = i < min(wq_cnt[wave], kQueueCap);
                        const double es = have ? wq_s[wave * kQueueCap + i] : 0.0;
                        const uint32_t er = have ? wq_r[wave * kQueueCap + i] : 0u;
                        const int eq = have ? (int)wq_q[wave * kQueueCap + i] : 0;
                        insert_sync(have, es, er, eq, cand_s, cand_r, cnt, tau_s, tau_r, tau32, nqt, S, K);
                    }
                    __syncthreads();
                    if (tid < W) wq_cnt[tid] = 0;
                    __syncthreads();
                    if (maxfill > kQueueCap / 2) {  // a burst of survivors: back to synchronous insertion
                        fastmode = false;
                        calm = 0;
                    }
                }
            }
        } else {
            double accp[QT], accc[QT];
#pragma unroll
            for (int q = 0; q < QT; ++q) {
                accp[q] = 0.0;
                accc[q] = 0.0;
            }
            if (live) {
                dots_generic<QT>(P.fp, reinterpret_cast<const uint2 *>(smem + P.fp.off_hash),
                                 reinterpret_cast<const double *>(smem + P.fp.off_panel), slice, lane, accp);
                dots_generic<QT>(P.fc, reinterpret_cast<const uint2 *>(smem + P.fc.off_hash),
                                 reinterpret_cast<const double *>(smem + P.fc.off_panel), slice, lane, accc);
            }
            double cnp = 0.0, cnc = 0.0;
            uint32_t myrid = 0u;
            if (valid) {
                cnp = P.fp.norm[row];
                cnc = P.fc.norm[row];
                myrid = P.rid[row];
#pragma unroll
                for (int q = 0; q < QT; ++q) {
                    if (q < nqt && row != s_qrow[q]) {
                        double s;
                        if (exact_similarity(accp[q], accc[q], cnp, cnc, s_qnp[q], s_qnc[q], pw, cw, s) &&
                            better(s, myrid, tau_s[q], tau_r[q]))
                            pend |= 1u << q;
                    }
                }
            }
            while (__syncthreads_or(pend != 0)) {
#pragma unroll
                for (int q = 0; q < QT; ++q) {
                    if (pend & (1u << q)) {
                        double s;
                        exact_similarity(accp[q], accc[q], cnp, cnc, s_qnp[q], s_qnc[q], pw, cw, s);
                        if (!better(s, myrid, tau_s[q], tau_r[q])) {
                            pend &= ~(1u << q);
                            continue;
                        }
                        const int pos = atomicAdd(&cnt[q], 1);
                        if (pos < S) {
                            cand_s[q * S + pos] = s;
                            cand_r[q * S + pos] = myrid;
                            pend &= ~(1u << q);
                        }
                    }
                }
                __syncthreads();
                for (int q = 0; q < nqt; ++q)
                    if (cnt[q] >= S) compact_query(cand_s, cand_r, cnt, tau_s, tau_r, tau32, q, S, K);
            }
        }
    }
    if (tid == 0 && s_flags[0] && P.overflow) atomicAdd(P.overflow, 1);  // (no path sets it any more: kept as a tripwire)
    if (tid == 0 && s_flags[2] && P.overflow) atomicAdd(P.overflow + 1, s_flags[2]);  // locrec_knn_replayed_intervals
    // final compaction and write-out of this chunk's lists
    for (int q = 0; q < nqt; ++q) {
        compact_query(cand_s, cand_r, cnt, tau_s, tau_r, tau32, q, S, K);
        const int m = cnt[q];
        const int64_t base = ((int64_t)(q0 + q) * P.nchunks + blockIdx.x) * K;
        for (int i = tid; i < m; i += blockDim.x) {
            P.part_s[base + i] = cand_s[q * S + i];
            P.part_rid[base + i] = cand_r[q * S + i];
        }
        if (tid == 0) P.part_cnt[(int64_t)(q0 + q) * P.nchunks + blockIdx.x] = m;
        __syncthreads();
    }
}
