// Segmented device radix sorts used by the text-locality pre-pass (K7): rocPRIM's primitive, wrapped so that the rest of the library
// does not pay for its templates at compile time.  tmp == nullptr: returns the temporary storage the call needs.
#include <cstring>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_segmented_radix_sort.hpp>

#include "kernels.hpp"

namespace vq {

size_t seg_sort_u32(void* tmp, size_t tmp_bytes, const uint32_t* in, uint32_t* out, uint32_t n, uint32_t nseg, const uint32_t* seg_begin, const uint32_t* seg_end,
                    hipStream_t st) {
    size_t bytes = tmp_bytes;
    const hipError_t e = rocprim::segmented_radix_sort_keys(tmp, bytes, in, out, n, nseg, seg_begin, seg_end, 0u, 32u, st);
    if (e != hipSuccess) return size_t(-1);
    return bytes;
}

size_t seg_sort_u64(void* tmp, size_t tmp_bytes, const unsigned long long* in, unsigned long long* out, uint32_t n, uint32_t nseg, const uint32_t* seg_begin,
                    const uint32_t* seg_end, hipStream_t st) {
    size_t bytes = tmp_bytes;
    const hipError_t e = rocprim::segmented_radix_sort_keys(tmp, bytes, in, out, n, nseg, seg_begin, seg_end, 0u, 64u, st);
    if (e != hipSuccess) return size_t(-1);
    return bytes;
}

}  // namespace vq
