// prims.h -- typed entry points of the library primitives (prims.hip): stable radix sorts and exclusive scans, rocPRIM's calling
// convention (tmp == NULL: *bytes receives the temporary size, nothing runs).  Signed 32-bit data goes through the unsigned
// forms (two's complement sums and digit order of non-negative values are the same).
#ifndef STOCS_PRIMS_H
#define STOCS_PRIMS_H
#include <hip/hip_runtime.h>
#include <stdint.h>
namespace stocs {
hipError_t sort_pairs(void* tmp, size_t& bytes, const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout, size_t n, unsigned b0, unsigned b1, hipStream_t st);
hipError_t sort_pairs(void* tmp, size_t& bytes, const uint64_t* kin, uint64_t* kout, const uint32_t* vin, uint32_t* vout, size_t n, unsigned b0, unsigned b1, hipStream_t st);
// the library's own sort of (u32, u32) pairs (sort32.hip): same calling convention
//   n: a host-side bound of the list's length; seg_off (device, n_seg + 1 ascending offsets; NULL: one segment [0, n)): every segment is
//   sorted by key bits [b0, b1) on its own and stays where it is
hipError_t sort_pairs_own(void* tmp, size_t& bytes, const uint32_t* kin, uint32_t* kout, const uint32_t* vin, uint32_t* vout, size_t n, unsigned b0, unsigned b1,
                          const uint32_t* seg_off, uint32_t n_seg, hipStream_t st);
size_t sort_own_err_offset();   // byte offset, in the temporary block of sort_pairs_own, of its error word (non-zero: a look-back wait ran into its bound)
hipError_t sort_keys(void* tmp, size_t& bytes, const uint64_t* kin, uint64_t* kout, size_t n, unsigned b0, unsigned b1, hipStream_t st);
hipError_t segmented_sort_keys(void* tmp, size_t& bytes, const uint64_t* kin, uint64_t* kout, unsigned n, unsigned n_seg, const unsigned long long* seg_begin,
                               const unsigned long long* seg_end, unsigned b0, unsigned b1, hipStream_t st);
hipError_t exclusive_scan(void* tmp, size_t& bytes, const uint32_t* in, uint32_t* out, size_t n, hipStream_t st);                       // init 0
hipError_t exclusive_scan(void* tmp, size_t& bytes, const unsigned long long* in, unsigned long long* out, size_t n, hipStream_t st);
hipError_t exclusive_scan(void* tmp, size_t& bytes, const uint32_t* in, unsigned long long* out, size_t n, hipStream_t st);             // 32-bit counts, 64-bit offsets
}  // namespace stocs
#endif
