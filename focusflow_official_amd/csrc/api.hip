// libfocusflow_hip: error channel + ABI version.
#include "ff_common.h"
#include <algorithm>
#include <mutex>
#include <utility>
#include <vector>

namespace ff {
char* err_buf() {
    static thread_local char buf[512] = "";
    return buf;
}
}  // namespace ff

extern "C" const char* ff_last_error(void) { return ff::err_buf(); }

// ------------------------------------------------------------------------------------------------------------------
// Launch timing (see ff_common.h: launch_timing_events)
namespace {
struct LaunchTimer {
    bool on = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
};
std::mutex g_timer_mu;
LaunchTimer g_timers[FF_TIME_KINDS];
}  // namespace

void ff::launch_timing_events(int which, hipEvent_t* start, hipEvent_t* stop) {
    *start = *stop = nullptr;
    if (which < 1 || which > FF_TIME_KINDS) return;
    std::lock_guard<std::mutex> lock(g_timer_mu);
    LaunchTimer& t = g_timers[which - 1];
    if (!t.on) return;
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess) return;
    if (hipEventCreate(&b) != hipSuccess) { (void)hipEventDestroy(a); return; }
    t.ev.emplace_back(a, b);
    *start = a;
    *stop = b;
}

extern "C" int ff_launch_timing_begin(int which) {
    FF_REQUIRE(which >= 1 && which <= FF_TIME_KINDS, "ff_launch_timing_begin: unknown launch class %d", which);
    std::lock_guard<std::mutex> lock(g_timer_mu);
    LaunchTimer& t = g_timers[which - 1];
    for (auto& e : t.ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    t.ev.clear();
    t.on = true;
    return FF_OK;
}

extern "C" int ff_launch_timing_end(int which, long long* launches, double* total_us, double* min_us, double* max_us) {
    FF_REQUIRE(which >= 1 && which <= FF_TIME_KINDS, "ff_launch_timing_end: unknown launch class %d", which);
    FF_REQUIRE(launches && total_us, "ff_launch_timing_end: null output");
    std::lock_guard<std::mutex> lock(g_timer_mu);
    LaunchTimer& t = g_timers[which - 1];
    t.on = false;
    double sum = 0, lo = 1e30, hi = 0;
    int rc = FF_OK;
    for (auto& e : t.ev) {
        float ms = 0.f;
        hipError_t err = hipEventSynchronize(e.second);
        if (err == hipSuccess) err = hipEventElapsedTime(&ms, e.first, e.second);
        if (err != hipSuccess && rc == FF_OK) rc = ff::fail(FF_EHIP, "ff_launch_timing_end: %s", hipGetErrorString(err));
        sum += ms * 1e3;
        lo = std::min(lo, ms * 1e3);
        hi = std::max(hi, ms * 1e3);
        (void)hipEventDestroy(e.first);
        (void)hipEventDestroy(e.second);
    }
    *launches = (long long)t.ev.size();
    *total_us = sum;
    if (min_us) *min_us = t.ev.empty() ? 0 : lo;
    if (max_us) *max_us = hi;
    t.ev.clear();
    return rc;
}
// 4: entry points only (ff_pack_weights_table, ff_pack_job_check, ff_unpack_wgrad_group, ff_probe_memory_kernel, FF_TIME_PROBE).  3: FFConvParams + ep_mode ... ep_b_ld (GRU steps in the conv epilogue).  2: FFConvParams grew res2 / res2_ld / res_split / splitk_ws / splitk, ff_norm_bwd gained dx_amax, the row-major corr
// backward entry points went away (round 2).  Callers zero-initialise the WHOLE FFConvParams and check this number.
extern "C" int ff_abi_version(void) { return FF_ABI_VERSION; }

// ------------------------------------------------------------------------------------------------------------------
// Host-side helper of the data loaders (no GPU work): PNG scan-line reconstruction (filter types 0-4, RFC 2083 section 6)
// for the KITTI 16-bit flow maps that frame_utils.read_png16 decodes (core/utils/frame_utils.py:102-120 reads them through
// cv2, absent here).  raw = inflated IDAT stream: h x (1 filter byte + stride bytes); out = h x stride bytes.
extern "C" int ff_png_unfilter(const unsigned char* raw, long long raw_len, int h, int stride, int bpp, unsigned char* out) {
    FF_REQUIRE(raw && out && h > 0 && stride > 0 && bpp > 0 && bpp <= 8, "ff_png_unfilter: bad argument");
    FF_REQUIRE(raw_len >= (long long)h * (stride + 1), "ff_png_unfilter: stream too short (%lld bytes for %d rows of %d)", raw_len, h, stride + 1);
    for (int y = 0; y < h; ++y) {
        const unsigned char* line = raw + (long long)y * (stride + 1);
        const int ft = line[0];
        ++line;
        unsigned char* cur = out + (long long)y * stride;
        const unsigned char* prev = y ? cur - stride : nullptr;
        FF_REQUIRE(ft >= 0 && ft <= 4, "ff_png_unfilter: filter type %d in row %d", ft, y);
        for (int i = 0; i < stride; ++i) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = prev ? prev[i] : 0, c = (prev && i >= bpp) ? prev[i - bpp] : 0;
            int pred = 0;
            if (ft == 1) pred = a;
            else if (ft == 2) pred = b;
            else if (ft == 3) pred = (a + b) >> 1;
            else if (ft == 4) {
                const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            }
            cur[i] = (unsigned char)((line[i] + pred) & 255);
        }
    }
    return FF_OK;
}
