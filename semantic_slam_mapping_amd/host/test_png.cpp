// test_png -- CPU-only check of ssm::imreadPNG (include/ssm/png_io.h) on the files tests/test_host_cpp.py writes:
// argv[1] = directory holding rgb.png (RGB8), gray.png (L8), depth.png (L16), rgba.png (RGBA8), pal.png (palette), all
// 37 x 23 with pixel(y, x, c) = (7x + 13y + 29c) & 255 and depth(y, x) = (257x + 31y) & 65535.  Prints PASS/FAIL lines.
#include <iostream>
#include "ssm/png_io.h"
static int fails = 0;
#define CHECK(name, cond) do { if (cond) std::cout << "PASS " << name << std::endl; else { std::cout << "FAIL " << name << std::endl; fails++; } } while (0)
int main(int argc, char** argv)
{
    const std::string dir = argc > 1 ? argv[1] : ".";
    const int W = 37, H = 23;
    auto px = [](int y, int x, int c) { return (7 * x + 13 * y + 29 * c) & 255; };
    cv::Mat bgr = ssm::imreadPNG(dir + "/rgb.png", 1);
    bool ok = bgr.rows == H && bgr.cols == W && bgr.type() == CV_8UC3;
    for (int y = 0; ok && y < H; y++) for (int x = 0; x < W; x++) for (int c = 0; c < 3; c++) if (bgr.ptr<uint8_t>(y)[3 * x + c] != px(y, x, 2 - c)) ok = false;     // file is RGB, Mat is BGR
    CHECK("rgb8_as_bgr", ok);
    cv::Mat g = ssm::imreadPNG(dir + "/rgb.png", 0);
    ok = g.rows == H && g.type() == CV_8UC1;
    for (int y = 0; ok && y < H; y++) for (int x = 0; x < W; x++) if (g.ptr<uint8_t>(y)[x] != ((px(y, x, 2) * 1868 + px(y, x, 1) * 9617 + px(y, x, 0) * 4899 + 8192) >> 14)) ok = false;
    CHECK("rgb8_as_gray_uses_cvtColor_weights", ok);
    cv::Mat g8 = ssm::imreadPNG(dir + "/gray.png", 0), g8c = ssm::imreadPNG(dir + "/gray.png", 1);
    ok = g8.type() == CV_8UC1 && g8c.type() == CV_8UC3 && g8.rows == H;
    for (int y = 0; ok && y < H; y++) for (int x = 0; x < W; x++) if (g8.ptr<uint8_t>(y)[x] != px(y, x, 0) || g8c.ptr<uint8_t>(y)[3 * x + 1] != px(y, x, 0)) ok = false;
    CHECK("gray8", ok);
    cv::Mat d = ssm::imreadPNG(dir + "/depth.png", -1);
    ok = d.type() == CV_16UC1 && d.rows == H && d.cols == W;
    for (int y = 0; ok && y < H; y++) for (int x = 0; x < W; x++) if (d.ptr<uint16_t>(y)[x] != ((257 * x + 31 * y) & 65535)) ok = false;
    CHECK("gray16_unchanged_host_byte_order", ok);
    cv::Mat d8 = ssm::imreadPNG(dir + "/depth.png", 0);
    ok = d8.type() == CV_8UC1; for (int y = 0; ok && y < H; y++) for (int x = 0; x < W; x++) if (d8.ptr<uint8_t>(y)[x] != (((257 * x + 31 * y) & 65535) >> 8)) ok = false;
    CHECK("gray16_to_8bit", ok);
    cv::Mat a = ssm::imreadPNG(dir + "/rgba.png", 1), au = ssm::imreadPNG(dir + "/rgba.png", -1);
    ok = a.type() == CV_8UC3 && au.channels() == 4;
    for (int y = 0; ok && y < H; y++) for (int x = 0; x < W; x++) if (a.ptr<uint8_t>(y)[3 * x] != px(y, x, 2) || au.ptr<uint8_t>(y)[4 * x + 3] != px(y, x, 3) || au.ptr<uint8_t>(y)[4 * x] != px(y, x, 2)) ok = false;
    CHECK("rgba8", ok);
    cv::Mat p = ssm::imreadPNG(dir + "/pal.png", 1);
    ok = p.type() == CV_8UC3 && p.rows == H;
    for (int y = 0; ok && y < H; y++) for (int x = 0; x < W; x++) { const int i = (x + y) % 12; if (p.ptr<uint8_t>(y)[3 * x] != ((i * 20 + 2) & 255) || p.ptr<uint8_t>(y)[3 * x + 2] != (i * 20)) ok = false; }
    CHECK("palette", ok);
    CHECK("missing_file_is_empty", ssm::imreadPNG(dir + "/nope.png").empty());
    CHECK("garbage_is_empty", ssm::imreadPNG(dir + "/garbage.png").empty());
    std::cout << (fails ? "FAILED" : "ALL PASSED") << std::endl;
    return fails;
}
