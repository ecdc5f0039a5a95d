// Measurement aid, not part of the flow path: a memory-only kernel with the CorrBlock lookup's launch shape and access
// pattern (one-wave blocks; per trip five 1 KB loads from random aligned segments of a large buffer and four 1 KB stores
// to the wave's own output stream - the lookup reads 1608 bytes per query for 1296 it writes), timed like the lookup
// (FF_TIME_PROBE).  bench.py runs it next to the lookup so that "fraction of the 8 TB/s peak" has a measured companion:
// what this part delivers to ANY kernel that moves that many bytes in one launch.  tools/proto/hbm_gather.hip is the
// stand-alone version with more patterns.
#include "ff_common.h"

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

__global__ __launch_bounds__(64) void probe_kernel(const char* src, unsigned long long nseg, unsigned seg, int trips, char* dst, unsigned salt,
                                                   unsigned* sink) {
    const unsigned lane = threadIdx.x, wave = blockIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    for (int t = 0; t < trips; ++t) {
        u32x4 v[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const unsigned sidx = lane * 16u / seg;                    // seg <= 1024: 1024 / seg segments per instruction
            const unsigned h = mix(mix(wave * 0x9e3779b9u + (unsigned)t + salt * 0x85ebca6bu) + (unsigned)u * 64u + sidx);
            v[u] = *reinterpret_cast<const u32x4*>(src + (h % nseg) * (unsigned long long)seg + (lane * 16u) % seg);
        }
#pragma unroll
        for (int u = 0; u < 5; ++u) acc ^= v[u];
#pragma unroll
        for (int w = 0; w < 4; ++w)
            *reinterpret_cast<u32x4*>(dst + (((unsigned long long)wave * trips + t) * 4 + w) * 1024ull + lane * 16u) = v[w];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;      // keeps the loads alive
}

}  // namespace

extern "C" int ff_probe_memory_kernel(const void* src, long long src_bytes, void* dst, long long dst_bytes, int seg_bytes, int blocks,
                                      int trips, unsigned int salt, long long* bytes_read, long long* bytes_written, void* stream) {
    FF_REQUIRE(src && dst && ff::aligned16(src) && ff::aligned16(dst), "ff_probe_memory_kernel: null / unaligned buffer");
    FF_REQUIRE(seg_bytes >= 16 && seg_bytes <= 1024 && (seg_bytes & (seg_bytes - 1)) == 0, "ff_probe_memory_kernel: seg_bytes must be a power of two in 16..1024");
    FF_REQUIRE(blocks > 0 && trips > 0 && src_bytes >= seg_bytes, "ff_probe_memory_kernel: bad shape");
    const long long rd = (long long)blocks * trips * 5 * 1024, wr = (long long)blocks * trips * 4 * 1024;
    FF_REQUIRE(dst_bytes >= wr + 4, "ff_probe_memory_kernel: dst holds %lld bytes, the launch writes %lld + 4", dst_bytes, wr);
    hipEvent_t ev0, ev1;
    ff::launch_timing_events(FF_TIME_PROBE, &ev0, &ev1);
    hipExtLaunchKernelGGL(probe_kernel, dim3(blocks), dim3(64), 0, static_cast<hipStream_t>(stream), ev0, ev1, 0, static_cast<const char*>(src),
                          (unsigned long long)(src_bytes / seg_bytes), (unsigned)seg_bytes, trips, static_cast<char*>(dst), salt,
                          reinterpret_cast<unsigned*>(static_cast<char*>(dst) + wr));
    if (bytes_read) *bytes_read = rd;
    if (bytes_written) *bytes_written = wr;
    return ff::check_launch("ff_probe_memory_kernel");
}
