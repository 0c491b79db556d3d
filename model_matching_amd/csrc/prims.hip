// prims.hip -- the library primitives of the path (rocPRIM radix sorts and scans), instantiated ONCE for the whole library.
// Every other file calls these typed wrappers: the same (key, value) combination used to be instantiated in up to four
// translation units (congruent, grid, ingest, lcp), a megabyte of device code each.  Nothing here restates the reference: the
// sorts replace its pointer grids and std::map / std::set containers (reference src/stocs.cpp:806-866, src/rgbd.cpp:123-154),
// the scans its push_backs.  Calling convention as rocPRIM's: tmp == NULL asks for the temporary size.
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "prims.h"

namespace stocs {

hipError_t sort_pairs(void* tmp, size_t& bytes, const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout, size_t n, unsigned b0, unsigned b1, hipStream_t st) {
    return rocprim::radix_sort_pairs(tmp, bytes, (uint32_t*)kin, kout, (uint32_t*)vin, vout, n, b0, b1, st);
}
hipError_t sort_pairs(void* tmp, size_t& bytes, const uint64_t* kin, uint64_t* kout, const uint32_t* vin, uint32_t* vout, size_t n, unsigned b0, unsigned b1, hipStream_t st) {
    return rocprim::radix_sort_pairs(tmp, bytes, (uint64_t*)kin, kout, (uint32_t*)vin, vout, n, b0, b1, st);
}
hipError_t sort_keys(void* tmp, size_t& bytes, const uint64_t* kin, uint64_t* kout, size_t n, unsigned b0, unsigned b1, hipStream_t st) {
    return rocprim::radix_sort_keys(tmp, bytes, (uint64_t*)kin, kout, n, b0, b1, st);
}
hipError_t segmented_sort_keys(void* tmp, size_t& bytes, const uint64_t* kin, uint64_t* kout, unsigned n, unsigned n_seg, const unsigned long long* seg_begin,
                               const unsigned long long* seg_end, unsigned b0, unsigned b1, hipStream_t st) {
    return rocprim::segmented_radix_sort_keys(tmp, bytes, (uint64_t*)kin, kout, n, n_seg, seg_begin, seg_end, b0, b1, st);
}
hipError_t exclusive_scan(void* tmp, size_t& bytes, const uint32_t* in, uint32_t* out, size_t n, hipStream_t st) {
    return rocprim::exclusive_scan(tmp, bytes, (uint32_t*)in, out, 0u, n, rocprim::plus<uint32_t>(), st);
}
hipError_t exclusive_scan(void* tmp, size_t& bytes, const unsigned long long* in, unsigned long long* out, size_t n, hipStream_t st) {
    return rocprim::exclusive_scan(tmp, bytes, (unsigned long long*)in, out, 0ull, n, rocprim::plus<unsigned long long>(), st);
}
hipError_t exclusive_scan(void* tmp, size_t& bytes, const uint32_t* in, unsigned long long* out, size_t n, hipStream_t st) {
    return rocprim::exclusive_scan(tmp, bytes, (uint32_t*)in, out, 0ull, n, rocprim::plus<unsigned long long>(), st);
}

}  // namespace stocs
