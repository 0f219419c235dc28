// What the column calls (orcd_api.hip) and the host reader (orcd_host.cpp) share beyond
// include/orcd.h: the bytes a column handle holds, and the way of a compressed column whose chunks
// are decompressed by the caller.
#pragma once

#include <stddef.h>
#include <stdint.h>

#include "../../include/orcd.h"

extern "C" int orcd_internal_column_bytes(const orcd_column *col, const uint8_t **bytes, size_t *len);
// One compression chunk of a column region read as it is in the file.
struct orcd_internal_chunk {
    uint64_t header;    // offset of the chunk's 3-byte header in its (compressed) stream
    uint64_t at;        // offset of its first byte in the region
    uint32_t len;
    uint32_t stream;    // which of the column's non-ROW_INDEX streams, in file order
    uint32_t original;  // stored as it is
    uint32_t pad;
};
extern "C" int orcd_internal_read_column_raw(orcd_file *f, size_t stripe, const orcd_stripe_footer *ft, uint32_t column,
                                             orcd_column **out);
extern "C" int orcd_internal_column_raw_from(orcd_file *f, size_t stripe, const orcd_stripe_footer *ft, uint32_t column,
                                             const uint8_t *region_bytes, uint64_t region_at, uint64_t region_len,
                                             orcd_column **out);
extern "C" int orcd_internal_column_raw(const orcd_column *col, const uint8_t **bytes, size_t *len,
                                        const orcd_internal_chunk **chunks, size_t *n_chunks, size_t *n_streams);
extern "C" int orcd_internal_column_inflate_host(orcd_file *f, orcd_column *col);
extern "C" int orcd_internal_column_set_plain(orcd_column *col, const uint64_t *plain_off, const uint64_t *plain_len,
                                              const uint64_t *chunk_off);
extern "C" int orcd_internal_column_plain_range(const orcd_column *col, int kind, uint64_t *off, uint64_t *len);
extern "C" uint64_t orcd_internal_file_compression_block(const orcd_file *f);
// Chunks of a compressed region inflated by the host's threads: chunk k is src[0, len), may
// decompress to at most `cap` bytes (and `pitch`), and its bytes go to out + k * pitch; sizes[k]
// receives what it came to, or 0xFFFFFFFF when it does not decompress.  Returns ORCD_OK, or what
// stopped the work as a whole (no memory, a codec that cannot be had).
struct orcd_internal_host_chunk {
    const uint8_t *src;
    uint32_t len, cap;
};
extern "C" int orcd_internal_inflate_list(int codec, const orcd_internal_host_chunk *list, size_t n, uint8_t *out,
                                          size_t pitch, uint32_t *sizes);

// Helper threads -- the readers and copiers of a stripe's bytes, the host's decompression threads --
// are one pool for the whole process, as many as its budget (orcd_helper_threads).
// fn(arg, 0) on the calling thread and fn(arg, slot) -- slot 1 .. max_helpers, each at most once --
// on as many of the process's helper threads as are free to join while the caller is at it; returns
// when all that joined are back.  fn takes pieces of work off a counter of its own.
extern "C" void orcd_internal_parallel_run(int max_helpers, void (*fn)(void *, int), void *arg);
// the CPUs the helper threads keep to (the first call decides)
extern "C" void orcd_internal_pool_bind(const void *cpu_set, size_t cpu_set_bytes);
#ifdef __cplusplus
template <class F>
inline void orcd_parallel(int max_helpers, F &&f) {
    auto *pf = &f;
    orcd_internal_parallel_run(
        max_helpers, [](void *a, int slot) { (*static_cast<decltype(pf)>(a))(slot); }, (void *)pf);
}
#endif
// The CPUs of the NUMA node device memory at PCI address `pci_bus_id` is attached to, as a
// cpu_set_t the helpers of that device's contexts bind themselves to; false when the system does
// not say (one node, no sysfs entry) or ORCD_NUMA_BIND=0.
extern "C" int orcd_internal_numa_cpus(const char *pci_bus_id, void *cpu_set, size_t cpu_set_bytes);
