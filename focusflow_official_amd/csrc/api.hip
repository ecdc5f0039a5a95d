// libfocusflow_hip: error channel + ABI version.
#include "ff_common.h"

namespace ff {
char* err_buf() {
    static thread_local char buf[512] = "";
    return buf;
}
}  // namespace ff

extern "C" const char* ff_last_error(void) { return ff::err_buf(); }
// 3: FFConvParams + ep_mode ... ep_b_ld (GRU steps in the conv epilogue).  2: FFConvParams grew res2 / res2_ld / res_split / splitk_ws / splitk, ff_norm_bwd gained dx_amax, the row-major corr
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
