// What HBM delivers for the lookup's access pattern: random aligned segments of 128 .. 4096 bytes out of a buffer larger
// than the last-level cache, against a plain stream over the same buffer.  Context for "fraction of the 8 TB/s peak" of
// the CorrBlock lookup: a query's window is 12 rows x 64-80 bytes in each of four planes of the tiled
// pyramid, i.e. a handful of 128-byte lines per plane, planes 1-12 KB apart.
//   hipcc -O2 --offload-arch=gfx950 -o gpurun_out/hbm_gather tools/proto/hbm_gather.hip && ./gpurun_out/hbm_gather
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned mix(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// One wave per block.  Per trip a wave issues U 1 KB load instructions; the 1 KB of an instruction is 1024 / SEG random
// segments (SEG <= 1024) or a quarter .. of one (SEG > 1024: U instructions walk through consecutive KBs of a segment).
// W > 0: the wave also writes W KB per trip to its own stream of the output buffer (the lookup writes 1296 bytes per query
// for 1608 it reads: U = 5, W = 4).
template <int U, int W>
__global__ __launch_bounds__(64) void gather(const char* buf, unsigned long long bytes, int seg, int trips, int stream, unsigned* out, char* wout, unsigned salt) {
    const unsigned lane = threadIdx.x, wave = blockIdx.x, nw = gridDim.x;
    const unsigned long long nseg = bytes / (unsigned)seg;
    u32x4 acc = {0, 0, 0, 0};
    for (int t = 0; t < trips; ++t) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            unsigned long long off;
            if (stream) {
                off = (((unsigned long long)t * nw + wave) * U + u) * 1024ull + lane * 16u + (unsigned long long)salt * (192ull << 20);
                off %= bytes;
            } else if (seg <= 1024) {
                const unsigned sidx = lane * 16u / (unsigned)seg;
                const unsigned h = mix(mix(wave * 0x9e3779b9u + (unsigned)t + salt * 0x85ebca6bu) + (unsigned)u * 64u + sidx);
                off = (h % nseg) * (unsigned long long)seg + (lane * 16u) % (unsigned)seg;
            } else {
                const unsigned per = (unsigned)seg / 1024u;                      // instructions per segment
                const unsigned h = mix(mix(wave * 0x9e3779b9u + (unsigned)t + salt * 0x85ebca6bu) + (unsigned)(u / per));
                off = (h % nseg) * (unsigned long long)seg + (u % per) * 1024u + lane * 16u;
            }
            v[u] = *reinterpret_cast<const u32x4*>(buf + off);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
#pragma unroll
        for (int w = 0; w < W; ++w)
            *reinterpret_cast<u32x4*>(wout + (((unsigned long long)wave * trips + t) * W + w) * 1024ull + lane * 16u + (salt & 1) * (512ull << 20)) = v[w % U];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[wave] = 1;      // keeps the loads alive
}

int main(int argc, char** argv) {
    const unsigned long long bytes = 4096ull << 20;      // every repetition reads other addresses: nothing is left in the 256 MB last-level cache
    char *buf, *wout; unsigned* out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 1 << 20)); CK(hipMalloc(&wout, 1100ull << 20));      // mixed 1 GB case: 448 MB written
    CK(hipMemset(buf, 1, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int wpc = argc > 1 ? atoi(argv[1]) : 16;
    const int blocks = 256 * wpc;
    printf("buffer %llu MB, %d one-wave blocks; times are the kernel's own start/stop timestamps (hipExtLaunchKernelGGL events)\n", bytes >> 20, blocks);
    const int segs[] = {0, 64, 128, 256, 512, 1024, 4096};
    for (int mixed = 0; mixed < 2; ++mixed)
        for (int total_mb : {72, 144, 1024})
            for (int seg : segs) {
                const int stream = seg == 0;
                const int per_trip_kb = mixed ? 9 : 4;
                const int trips = (int)(((unsigned long long)total_mb << 20) / ((unsigned long long)blocks * per_trip_kb * 1024));
                float best = 1e9f;
                for (int rep = 0; rep < 5; ++rep) {
                    if (mixed) hipExtLaunchKernelGGL((gather<5, 4>), dim3(blocks), dim3(64), 0, 0, e0, e1, 0, buf, bytes, stream ? 1024 : seg, trips, stream, out, wout, (unsigned)(rep + 1));
                    else hipExtLaunchKernelGGL((gather<4, 0>), dim3(blocks), dim3(64), 0, 0, e0, e1, 0, buf, bytes, stream ? 1024 : seg, trips, stream, out, wout, (unsigned)(rep + 1));
                    CK(hipEventSynchronize(e1));
                    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep > 0 && ms < best) best = ms;
                }
                const double b = (double)blocks * trips * per_trip_kb * 1024.0;
                char what[64];
                if (stream) snprintf(what, sizeof what, "stream"); else snprintf(what, sizeof what, "random %4d-byte segments", seg);
                printf("%s %7.1f MB per launch, %-28s: %8.1f us  %6.0f GB/s\n", mixed ? "read 5 : write 4," : "read only,        ", b / 1048576.0, what, best * 1e3, b / best / 1e6);
            }
    return 0;
}
