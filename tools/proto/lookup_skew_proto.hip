// Feasibility probe for a query-contiguous ("skewed") correlation pyramid layout: memory pattern only.
// One wave = 64 consecutive queries x one level: 121 coalesced 256-byte loads, 81 outputs per query via an LDS transpose.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int B = 8, H0 = 48, W0 = 64, Q = H0 * W0;

struct Args { const float* lvl[4]; float* out; int jitter; };

__global__ __launch_bounds__(64) void probe(Args a) {
    __shared__ float sm[64 * 82];
    const int lane = threadIdx.x;
    const int task = blockIdx.x;                 // (b, row-wave, level)
    const int l = task & 3, wv = (task >> 2) % (Q / 64), b = task / (4 * (Q / 64));
    const int hl = H0 >> l, wl = W0 >> l;
    const int q = wv * 64 + lane;
    // pseudo-random small displacement per wave (+ optional per-lane jitter of the integer origin)
    unsigned h = (unsigned)task * 2654435761u;
    int ddx = (int)(h % 5) - 2, ddy = (int)((h >> 8) % 5) - 2;
    if (a.jitter) { unsigned g = (unsigned)(q * 40503u + task); ddx += (int)(g % (2 * a.jitter + 1)) - a.jitter; ddy += (int)((g >> 7) % (2 * a.jitter + 1)) - a.jitter; }
    ddx = ((ddx % wl) + wl) % wl; ddy = ((ddy % hl) + hl) % hl;
    const float* base = a.lvl[l] + (long long)b * hl * wl * Q;
    float w[11][11];
#pragma unroll
    for (int r = 0; r < 11; ++r) {
        int dy = ddy + r; if (dy >= hl) dy -= hl;
#pragma unroll
        for (int c = 0; c < 11; ++c) {
            int dx = ddx + c; if (dx >= wl) dx -= wl;
            w[r][c] = base[((long long)dy * wl + dx) * Q + q];
        }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int j = 0; j < 9; ++j)
            sm[lane * 82 + i * 9 + j] = 0.25f * (w[j][i] + w[j][i + 1] + w[j + 1][i] + w[j + 1][i + 1]) + w[j + 2][i + 2] * 1e-3f;
    __syncthreads();
    float* o = a.out + ((long long)b * Q + wv * 64) * 324 + l * 81;
    for (int e = lane; e < 64 * 81; e += 64) { const int row = e / 81, k = e - row * 81; o[(long long)row * 324 + k] = sm[row * 82 + k]; }
}

int main(int argc, char** argv) {
    const int jitter = argc > 1 ? atoi(argv[1]) : 0;
    Args a; a.jitter = jitter;
    size_t tot = 0;
    float* bufs[4];
    for (int l = 0; l < 4; ++l) { size_t n = (size_t)B * (H0 >> l) * (W0 >> l) * Q; CK(hipMalloc(&bufs[l], n * 4)); CK(hipMemset(bufs[l], 0, n * 4)); a.lvl[l] = bufs[l]; tot += n * 4; }
    CK(hipMalloc(&a.out, (size_t)B * Q * 324 * 4));
    float* flush; const size_t fl = 1ull << 30; CK(hipMalloc(&flush, fl));
    const int tasks = B * (Q / 64) * 4;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9, sum = 0; const int reps = 10;
    for (int it = 0; it < reps + 2; ++it) {
        CK(hipMemsetAsync(flush, it, fl, 0));              // evict L2 / MALL between launches (pyramid is cold in the bench)
        CK(hipEventRecord(e0, 0));
        probe<<<tasks, 64>>>(a);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (it >= 2) { sum += ms; best = ms < best ? ms : best; }
    }
    const double alg = (double)B * Q * 2904.0;
    printf("jitter %d: pyramid %.0f MB, tasks %d: avg %.1f us best %.1f us  -> algorithmic %.2f TB/s (%.0f%% of 8 TB/s)\n", jitter, tot / 1e6, tasks,
           sum / reps * 1e3, best * 1e3, alg / (sum / reps * 1e-3) / 1e12, alg / (sum / reps * 1e-3) / 8e12 * 100);
    return 0;
}
