// dev_prims.h -- the device-wide primitives of the OFFLINE steps (index build, the producers) and of the one request
// path that sorts (large-K findSimilarPersons): rocPRIM called directly, under the names this code base uses.
// (Rounds 1-2 went through hipCUB, the CUB-compatibility facade over the same rocPRIM kernels)
// Semantics kept from the call sites' point of view: sums and scans accumulate in the INPUT's value type, radix sorts
// are stable, `*_desc` sorts descending, select / unique write the number of kept items to a device counter.
// Every function has the two-phase temporary-storage protocol of rocPRIM: temp == nullptr only sizes.
#pragma once

#include <cstring>  // (rocprim/iterator/texture_cache_iterator.hpp calls memset without including it)
#include <string.h>

#include <rocprim/rocprim.hpp>

#include <iterator>

namespace locrec {
namespace prim {

template <class K, class V>
inline hipError_t sort_pairs(void *temp, size_t &bytes, const K *keys_in, K *keys_out, const V *vals_in, V *vals_out, size_t n,
                             unsigned begin_bit, unsigned end_bit, hipStream_t s)
{
    return rocprim::radix_sort_pairs(temp, bytes, keys_in, keys_out, vals_in, vals_out, n, begin_bit, end_bit, s);
}

template <class K, class V>
inline hipError_t sort_pairs_desc(void *temp, size_t &bytes, const K *keys_in, K *keys_out, const V *vals_in, V *vals_out,
                                  size_t n, unsigned begin_bit, unsigned end_bit, hipStream_t s)
{
    return rocprim::radix_sort_pairs_desc(temp, bytes, keys_in, keys_out, vals_in, vals_out, n, begin_bit, end_bit, s);
}

template <class K>
inline hipError_t sort_keys(void *temp, size_t &bytes, const K *keys_in, K *keys_out, size_t n, unsigned begin_bit,
                            unsigned end_bit, hipStream_t s)
{
    return rocprim::radix_sort_keys(temp, bytes, keys_in, keys_out, n, begin_bit, end_bit, s);
}

template <class In, class Out>
inline hipError_t exclusive_sum(void *temp, size_t &bytes, In in, Out out, size_t n, hipStream_t s)
{
    using T = typename std::iterator_traits<In>::value_type;
    return rocprim::exclusive_scan(temp, bytes, in, out, T(0), n, rocprim::plus<T>(), s);
}

template <class In, class Out>
inline hipError_t inclusive_sum(void *temp, size_t &bytes, In in, Out out, size_t n, hipStream_t s)
{
    using T = typename std::iterator_traits<In>::value_type;
    return rocprim::inclusive_scan(temp, bytes, in, out, n, rocprim::plus<T>(), s);
}

template <class In, class Out>
inline hipError_t inclusive_max(void *temp, size_t &bytes, In in, Out out, size_t n, hipStream_t s)
{
    using T = typename std::iterator_traits<In>::value_type;
    return rocprim::inclusive_scan(temp, bytes, in, out, n, rocprim::maximum<T>(), s);
}

template <class In, class Out, class Op, class T>
inline hipError_t reduce(void *temp, size_t &bytes, In in, Out out, size_t n, Op op, T init, hipStream_t s)
{
    return rocprim::reduce(temp, bytes, in, out, init, n, op, s);
}

// items whose flag is non-zero, in order; *count_out = how many
template <class In, class Flags, class Out, class Count>
inline hipError_t select_flagged(void *temp, size_t &bytes, In in, Flags flags, Out out, Count count_out, size_t n, hipStream_t s)
{
    return rocprim::select(temp, bytes, in, flags, out, count_out, n, s);
}

// the first of every run of equal items
template <class In, class Out, class Count>
inline hipError_t unique(void *temp, size_t &bytes, In in, Out out, Count count_out, size_t n, hipStream_t s)
{
    using T = typename std::iterator_traits<In>::value_type;
    return rocprim::unique(temp, bytes, in, out, count_out, n, rocprim::equal_to<T>(), s);
}

// runs of equal items: the first of every run, its length, and the number of runs
template <class In, class UniqueOut, class CountsOut, class RunsOut>
inline hipError_t run_length_encode(void *temp, size_t &bytes, In in, size_t n, UniqueOut unique_out, CountsOut counts_out,
                                    RunsOut runs_out, hipStream_t s)
{
    return rocprim::run_length_encode(temp, bytes, in, (unsigned int)n, unique_out, counts_out, runs_out, s);
}

template <class T>
using counting_iterator = rocprim::counting_iterator<T>;

}  // namespace prim
}  // namespace locrec
