// ssm/png_io.h -- the slice of cv::imread the reference's dataset readers need (src/rgbdframe.cpp:21-22,43-62: TUM and KITTI
// frames are PNG files): a PNG decoder over zlib's inflate.  Non-interlaced, 8 or 16 bits per sample, colour types gray,
// gray+alpha, RGB, RGBA, palette (8 bit).  `flags` follows cv::imread: 1 (default) -> 8-bit BGR, 0 -> 8-bit gray
// (cv::cvtColor's fixed-point weights, the contract of oracle/orb.c), -1 (CV_LOAD_IMAGE_UNCHANGED) -> the file's depth and
// channel count (16-bit samples in host byte order, colour as BGR / BGRA).  Returns an empty Mat when the file is missing or
// not decodable, like cv::imread.  SURVEY.md s.8(f) rank 4.
#pragma once
#include <zlib.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "compat.h"
namespace ssm {
inline cv::Mat imreadPNG(const std::string& path, int flags = 1) {
    cv::Mat empty;
    FILE* fp = fopen(path.c_str(), "rb");
    if (!fp) return empty;
    std::vector<uint8_t> file;
    { uint8_t buf[65536]; size_t n; while ((n = fread(buf, 1, sizeof(buf), fp)) > 0) file.insert(file.end(), buf, buf + n); }
    fclose(fp);
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (file.size() < 8 + 25 || memcmp(file.data(), sig, 8)) return empty;
    auto be32 = [&](size_t o) { return ((uint32_t)file[o] << 24) | ((uint32_t)file[o + 1] << 16) | ((uint32_t)file[o + 2] << 8) | file[o + 3]; };
    uint32_t W = 0, H = 0; int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte;
    for (size_t o = 8; o + 12 <= file.size();) {
        const uint32_t len = be32(o); const char* type = (const char*)&file[o + 4];
        if (o + 12 + (size_t)len > file.size()) return empty;
        const uint8_t* d = &file[o + 8];
        if (!memcmp(type, "IHDR", 4) && len >= 13) { W = be32(o + 8); H = be32(o + 12); depth = d[8]; ctype = d[9]; interlace = d[12]; }
        else if (!memcmp(type, "PLTE", 4)) plte.assign(d, d + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), d, d + len);
        else if (!memcmp(type, "IEND", 4)) break;
        o += 12 + (size_t)len;
    }
    const int nch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!W || !H || W > 32768 || H > 32768 || !nch || interlace || (depth != 8 && depth != 16) || (ctype == 3 && (depth != 8 || plte.size() < 3))) return empty;
    const size_t bpp = (size_t)nch * depth / 8, stride = (size_t)W * bpp;
    std::vector<uint8_t> raw((stride + 1) * H);
    uLongf outlen = (uLongf)raw.size();
    if (uncompress(raw.data(), &outlen, idat.data(), (uLong)idat.size()) != Z_OK || outlen != raw.size()) return empty;
    // undo the per-row filters (PNG spec 9.2) in place; pix[y] = row y without its filter byte
    std::vector<uint8_t> pix(stride * H);
    for (uint32_t y = 0; y < H; y++) {
        const uint8_t ft = raw[(stride + 1) * y]; const uint8_t* src = &raw[(stride + 1) * y + 1];
        uint8_t* cur = &pix[stride * y]; const uint8_t* up = y ? &pix[stride * (y - 1)] : nullptr;
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
            int pred = 0;
            switch (ft) {
                case 0: pred = 0; break; case 1: pred = a; break; case 2: pred = b; break; case 3: pred = (a + b) >> 1; break;
                case 4: { const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); break; }
                default: return empty;
            }
            cur[i] = (uint8_t)(src[i] + pred);
        }
    }
    auto sample = [&](uint32_t y, uint32_t x, int ch) -> int {            // host-order sample value
        const uint8_t* p = &pix[stride * y + (size_t)x * bpp + (size_t)ch * depth / 8];
        return depth == 8 ? p[0] : (p[0] << 8) | p[1];
    };
    auto rgb8 = [&](uint32_t y, uint32_t x, int out[3]) {                  // any colour type -> 8-bit R, G, B
        if (ctype == 3) { const int i = sample(y, x, 0); for (int k = 0; k < 3; k++) out[k] = (size_t)(3 * i + k) < plte.size() ? plte[3 * i + k] : 0; return; }
        for (int k = 0; k < 3; k++) { const int v = sample(y, x, nch >= 3 ? k : 0); out[k] = depth == 8 ? v : v >> 8; }
    };
    cv::Mat m;
    if (flags < 0) {                                                     // unchanged
        const int cvdepth = depth == 8 ? cv::CV_8U : cv::CV_16U;
        if (ctype == 3) { m.create((int)H, (int)W, CV_8UC3); for (uint32_t y = 0; y < H; y++) for (uint32_t x = 0; x < W; x++) { int c[3]; rgb8(y, x, c); uint8_t* o = m.ptr<uint8_t>((int)y) + 3 * x; o[0] = (uint8_t)c[2]; o[1] = (uint8_t)c[1]; o[2] = (uint8_t)c[0]; } return m; }
        m.create((int)H, (int)W, CV_MAKETYPE(cvdepth, nch));
        for (uint32_t y = 0; y < H; y++) for (uint32_t x = 0; x < W; x++) for (int ch = 0; ch < nch; ch++) {
            const int src_ch = (nch >= 3 && ch < 3) ? 2 - ch : ch;          // RGB(A) -> BGR(A)
            const int v = sample(y, x, src_ch);
            if (depth == 8) m.ptr<uint8_t>((int)y)[(size_t)x * nch + ch] = (uint8_t)v; else m.ptr<uint16_t>((int)y)[(size_t)x * nch + ch] = (uint16_t)v;
        }
        return m;
    }
    if (flags > 0) {                                                     // 8-bit BGR
        m.create((int)H, (int)W, CV_8UC3);
        for (uint32_t y = 0; y < H; y++) for (uint32_t x = 0; x < W; x++) { int c[3]; rgb8(y, x, c); uint8_t* o = m.ptr<uint8_t>((int)y) + 3 * x; o[0] = (uint8_t)c[2]; o[1] = (uint8_t)c[1]; o[2] = (uint8_t)c[0]; }
        return m;
    }
    m.create((int)H, (int)W, CV_8UC1);                                   // 8-bit gray
    for (uint32_t y = 0; y < H; y++) for (uint32_t x = 0; x < W; x++) {
        int c[3]; rgb8(y, x, c);
        m.ptr<uint8_t>((int)y)[x] = (nch >= 3 || ctype == 3) ? (uint8_t)((c[2] * 1868 + c[1] * 9617 + c[0] * 4899 + 8192) >> 14) : (uint8_t)c[0];
    }
    return m;
}
}  // namespace ssm
