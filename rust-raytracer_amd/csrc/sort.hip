// Ray-queue ordering: radix sort of (key, index) pairs (rocPRIM device sort).
// Kept in its own translation unit because the sort templates are slow to
// compile.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "kernels.h"

namespace rtc {

// rocPRIM 4.2 (ROCm 7.2) returns a non-permutation from its merge-sort path
// (sizes from a few thousand to about a million pairs) when the sorted bit
// range does not start at bit 0; its onesweep path (large sizes) and full-range
// sorts are correct.  Partial bit ranges are therefore used only well inside
// the onesweep regime, and full 32-bit sorts everywhere else (callers zero the
// key bits they do not care about, so the order is the same).
static void safe_bit_range(unsigned n, unsigned &begin_bit, unsigned &end_bit)
{
    if (n < (1u << 22)) {
        begin_bit = 0;
        end_bit = 32;
    }
}

// Temporary storage the sort of n pairs over key bits [begin_bit, end_bit) needs.
// Never 0: the sort treats a null temporary pointer as "only report the size"
// and would return without sorting.
size_t sort_temp_bytes(unsigned n, unsigned begin_bit, unsigned end_bit)
{
    size_t bytes = 0;
    unsigned *k = nullptr;
    safe_bit_range(n, begin_bit, end_bit);
    hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, (const unsigned *)k, k,
                                             rocprim::counting_iterator<unsigned>(0), k, (size_t)n, begin_bit,
                                             end_bit, (hipStream_t)0);
    if (e != hipSuccess) return 0;
    return bytes < 256 ? 256 : bytes;
}

hipError_t sort_indices_by_key(void *temp, size_t temp_bytes, const unsigned *keys_in, unsigned *keys_out,
                               unsigned *perm_out, unsigned n, unsigned begin_bit, unsigned end_bit,
                               hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    if (temp == nullptr || temp_bytes < sort_temp_bytes(n, begin_bit, end_bit)) return hipErrorInvalidValue;
    safe_bit_range(n, begin_bit, end_bit);
    return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, rocprim::counting_iterator<unsigned>(0),
                                     perm_out, (size_t)n, begin_bit, end_bit, stream);
}

} // namespace rtc
