// Stand-alone lab for the CorrBlock lookup kernels (no torch: starts in a second).  Builds a synthetic tiled pyramid through
// the library's own ff_corr_retile, runs the round-2 kernel (FF_LOOKUP_IMPL=2) and the round-3 LDS-DMA kernel on the same
// queries, compares outputs and taps bit for bit, and times both (back-to-back launches, and launches behind a 512 MB
// write that empties the caches).  Build + run: tools/lookup_lab.sh.
//   usage: lookup_lab B H W half [jitter] [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include <algorithm>
#include <functional>
#include <cmath>
#include "focusflow_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
#define FF(x) do { int r_ = (x); if (r_ != 0) { printf("ff error %d (%s) at %s:%d\n", r_, ff_last_error(), __FILE__, __LINE__); exit(3); } } while (0)

__global__ void fill_random(float* p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned x = (unsigned)i * 2654435761u ^ seed;
        x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; x *= 3266489917u; x ^= x >> 16;
        p[i] = (float)(int)(x >> 8) * (1.0f / 8388608.0f) - 1.0f;
    }
}
__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 1234567) *p = 1; }

static double time_launches(hipStream_t s, int reps, const std::function<void()>& fn) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) fn();
    CK(hipEventRecord(a, s));
    for (int i = 0; i < reps; ++i) fn();
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    return ms * 1e3 / reps;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8, H = argc > 2 ? atoi(argv[2]) : 48, W = argc > 3 ? atoi(argv[3]) : 64;
    const int half = argc > 4 ? atoi(argv[4]) : 0;
    const float jitter = argc > 5 ? (float)atof(argv[5]) : 8.f;
    const int reps = argc > 6 ? atoi(argv[6]) : 50;
    const long long Q = (long long)B * H * W;
    const int esz = half ? 2 : 4, out_ld = 352;
    hipStream_t s; CK(hipStreamCreate(&s));
    // one allocation for the four levels
    size_t off[4], total = 0;
    long long pe[4];
    for (int l = 0; l < 4; ++l) { pe[l] = ff_corr_plane_elems(H, W, l, half); off[l] = total; total += ((size_t)pe[l] * Q * esz + 255) / 256 * 256; }
    char* pyr; CK(hipMalloc(&pyr, total)); CK(hipMemset(pyr, 0, total));
    const void* lv[4];
    for (int l = 0; l < 4; ++l) {
        lv[l] = pyr + off[l];
        const size_t n = (size_t)Q * (H >> l) * (W >> l);
        float* rm; CK(hipMalloc(&rm, n * 4));
        fill_random<<<2048, 256, 0, s>>>(rm, n, 17u + l);
        FF(ff_corr_retile(rm, pyr + off[l], Q, H, W, l, half, 1, s));
        CK(hipStreamSynchronize(s)); CK(hipFree(rm));
    }
    // coordinates: grid + U(-jitter, jitter); a few wild ones and exact integers (the reference's first iteration)
    std::vector<float> hc(Q * 2);
    std::mt19937 g(3); std::uniform_real_distribution<float> u(-jitter, jitter);
    for (long long q = 0; q < Q; ++q) {
        const int x = (int)(q % W), y = (int)((q / W) % H);
        const bool integer = (q % 7) == 3;
        hc[q * 2] = x + (integer ? (float)(int)u(g) : u(g));
        hc[q * 2 + 1] = y + (integer ? (float)(int)u(g) : u(g));
        if (q % 1013 == 5) { hc[q * 2] = -37.25f; hc[q * 2 + 1] = 1e6f; }
        if (q % 1013 == 6) { hc[q * 2] = W + 3.5f; hc[q * 2 + 1] = -2.75f; }
    }
    float* coords; CK(hipMalloc(&coords, Q * 8)); CK(hipMemcpy(coords, hc.data(), Q * 8, hipMemcpyHostToDevice));
    float *o2, *o3; CK(hipMalloc(&o2, Q * out_ld * 4)); CK(hipMalloc(&o3, Q * out_ld * 4));
    int *t2, *t3; CK(hipMalloc(&t2, Q * 72 * 4)); CK(hipMalloc(&t3, Q * 72 * 4));
    CK(hipMemset(o2, 0, Q * out_ld * 4)); CK(hipMemset(o3, 0xff, Q * out_ld * 4));
    CK(hipMemset(t2, 0, Q * 72 * 4)); CK(hipMemset(t3, 0xff, Q * 72 * 4));
    char* junk; const size_t junk_bytes = 512ull << 20; CK(hipMalloc(&junk, junk_bytes));

    // ---- parity: round-2 kernel vs round-3 kernel, outputs and taps ----
    setenv("FF_LOOKUP_IMPL", "2", 1);
    FF(ff_corr_lookup_tiled_fwd(lv, half, coords, Q, H, W, o2, out_ld, t2, s));
    setenv("FF_LOOKUP_IMPL", "3", 1);
    FF(ff_corr_lookup_tiled_fwd(lv, half, coords, Q, H, W, o3, out_ld, t3, s));
    CK(hipStreamSynchronize(s));
    std::vector<float> h2(Q * out_ld), h3(Q * out_ld);
    std::vector<int> ht2(Q * 72), ht3(Q * 72);
    CK(hipMemcpy(h2.data(), o2, Q * out_ld * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h3.data(), o3, Q * out_ld * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ht2.data(), t2, Q * 72 * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ht3.data(), t3, Q * 72 * 4, hipMemcpyDeviceToHost));
    long long bad = 0, badt = 0, first = -1; double sum = 0;
    for (long long q = 0; q < Q; ++q)
        for (int k = 0; k < 324; ++k) {
            const float a = h2[q * out_ld + k], b = h3[q * out_ld + k];
            sum += fabs(a);
            if (memcmp(&a, &b, 4) != 0 && !(a == 0.f && b == 0.f)) { if (!bad) first = q * 1000 + k; ++bad; }
        }
    for (long long i = 0; i < Q * 72; ++i) badt += ht2[i] != ht3[i];
    printf("parity (DBG kernels with taps): %lld differing outputs of %lld (first q*1000+k = %lld), %lld differing taps, mean|out| %.4f\n",
           bad, Q * 324, first, badt, sum / (Q * 324));
    // the production instance (no taps)
    CK(hipMemset(o3, 0xff, Q * out_ld * 4));
    FF(ff_corr_lookup_tiled_fwd(lv, half, coords, Q, H, W, o3, out_ld, nullptr, s));
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h3.data(), o3, Q * out_ld * 4, hipMemcpyDeviceToHost));
    long long bad2 = 0; first = -1;
    for (long long q = 0; q < Q; ++q)
        for (int k = 0; k < 324; ++k) {
            const float a = h2[q * out_ld + k], b = h3[q * out_ld + k];
            if (memcmp(&a, &b, 4) != 0 && !(a == 0.f && b == 0.f)) { if (!bad2) first = q * 1000 + k; ++bad2; }
        }
    printf("parity (production kernel): %lld differing outputs (first %lld)\n", bad2, first);
    if (bad2) {      // where: by iteration of the wave that owns the query (4096 waves) and by level
        long long by_it[8] = {0}, by_lv[4] = {0};
        for (long long q = 0; q < Q; ++q)
            for (int k = 0; k < 324; ++k) {
                const float a = h2[q * out_ld + k], b = h3[q * out_ld + k];
                if (memcmp(&a, &b, 4) != 0 && !(a == 0.f && b == 0.f)) { by_it[std::min<long long>(q / 4096, 7)]++; by_lv[k / 81]++; }
            }
        printf("  by iteration:"); for (int i = 0; i < 8; ++i) printf(" %lld", by_it[i]);
        printf("\n  by level:"); for (int i = 0; i < 4; ++i) printf(" %lld", by_lv[i]);
        printf("\n  first rows:\n");
        int shown = 0;
        for (long long q = 0; q < Q && shown < 6; ++q)
            for (int k = 0; k < 324 && shown < 6; ++k) {
                const float a = h2[q * out_ld + k], b = h3[q * out_ld + k];
                if (memcmp(&a, &b, 4) != 0 && !(a == 0.f && b == 0.f)) { printf("    q %lld k %d: ref %.7g got %.7g (coords %.4f %.4f)\n", q, k, a, b, hc[q * 2], hc[q * 2 + 1]); ++shown; }
            }
    }

    // ---- timing ----
    const double alg = (double)Q * (half ? 2104 : 2904);
    auto report = [&](const char* name, double us) { printf("%-44s %7.2f us  %6.0f GB/s algorithmic = %.3f of 8 TB/s\n", name, us, alg / us / 1e3, alg / us / 1e3 / 8000); };
    report("empty kernel 4096 x 64", time_launches(s, reps, [&] { empty_kernel<<<4096, 64, 0, s>>>(nullptr); }));
    report("empty kernel 1280 x 128", time_launches(s, reps, [&] { empty_kernel<<<1280, 128, 0, s>>>(nullptr); }));
    report("empty kernel 1280 x 128, 31 KB LDS", time_launches(s, reps, [&] { empty_kernel<<<1280, 128, 31488, s>>>(nullptr); }));
    for (int impl = 2; impl <= 3; ++impl) {
        setenv("FF_LOOKUP_IMPL", impl == 2 ? "2" : "3", 1);
        char nm[96];
        snprintf(nm, sizeof nm, "impl %d back-to-back x%d", impl, reps);
        report(nm, time_launches(s, reps, [&] { FF(ff_corr_lookup_tiled_fwd(lv, half, coords, Q, H, W, o3, out_ld, nullptr, s)); }));
        // cold: 512 MB written before every launch, the launch alone between events
        std::vector<double> ts;
        for (int i = 0; i < 12; ++i) {
            CK(hipMemsetAsync(junk, i, junk_bytes, s));
            hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
            CK(hipEventRecord(a, s));
            FF(ff_corr_lookup_tiled_fwd(lv, half, coords, Q, H, W, o3, out_ld, nullptr, s));
            CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); ts.push_back(ms * 1e3);
            CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
        }
        std::sort(ts.begin(), ts.end());
        snprintf(nm, sizeof nm, "impl %d cold, event-bracketed (median of 12)", impl);
        report(nm, ts[ts.size() / 2]);
    }
    return (bad || badt || bad2) ? 1 : 0;
}
