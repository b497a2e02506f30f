// host_sanitize.cpp -- the host side of libslx under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY.md section 5:
// "ASan/UBSan on the CPU restatement" -- here the product's own CPU-side code: file readers of untrusted input, the config
// validation, the launch planner, the gather planner).  Built by g++ from the product's sources (tests/cpp/Makefile, target
// host_sanitize); no GPU, no device code.  Exit code 0 = every check held and no sanitizer report.
//   usage: host_sanitize <scratch directory>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iterator>
#include <limits>
#include <sstream>
#include <random>
#include <string>
#include <vector>

#include "dynaframe.hpp"
#include "sensor.hpp"
#include "slx.h"
#include "slx_kernels.h"

static int g_fail = 0, g_strip_plans = 0, g_generic_plans = 0, g_stream_plans = 0;
#define CHECK(cond)                                                           \
    do {                                                                      \
        if (!(cond)) {                                                        \
            std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            g_fail++;                                                         \
        }                                                                     \
    } while (0)

static void put32(std::vector<unsigned char> &b, size_t at, uint32_t v) { for (int i = 0; i < 4; i++) b[at + i] = (unsigned char)(v >> (8 * i)); }
static void put16(std::vector<unsigned char> &b, size_t at, uint16_t v) { b[at] = (unsigned char)v; b[at + 1] = (unsigned char)(v >> 8); }
static void write_file(const std::string &p, const std::vector<unsigned char> &b) { std::ofstream f(p.c_str(), std::ios::binary); f.write((const char *)b.data(), (std::streamsize)b.size()); }

// a well-formed 8-bit paletted BMP of w x h, pixel (x, y) = (x + 3 y) & 255, bottom-up
static std::vector<unsigned char> good_bmp(int w, int h, uint32_t n_col = 256)
{
    const size_t row = ((size_t)w + 3) & ~(size_t)3, pal = 4 * 256, off = 54 + pal;
    std::vector<unsigned char> b(off + row * (size_t)h, 0);
    b[0] = 'B'; b[1] = 'M';
    put32(b, 2, (uint32_t)b.size()); put32(b, 10, (uint32_t)off); put32(b, 14, 40); put32(b, 18, (uint32_t)w); put32(b, 22, (uint32_t)h);
    put16(b, 26, 1); put16(b, 28, 8); put32(b, 30, 0); put32(b, 46, n_col);
    for (int i = 0; i < 256; i++) b[54 + 4 * i] = b[54 + 4 * i + 1] = b[54 + 4 * i + 2] = (unsigned char)i;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) b[off + row * (size_t)(h - 1 - y) + (size_t)x] = (unsigned char)((x + 3 * y) & 255);
    return b;
}

static void test_bmp(const std::string &dir)
{
    std::vector<uint8_t> px;
    int r = 0, c = 0;
    const std::string p = dir + "/t.bmp";
    write_file(p, good_bmp(13, 7));
    CHECK(slx::ReadBmpGray(p, px, r, c) && r == 7 && c == 13 && px.size() == 91 && px[0] == 0 && px[13 * 2 + 5] == ((5 + 6) & 255));
    // forged headers: every one of them must be refused (or read inside the file), never touch memory outside the buffer
    struct Forge { size_t at; uint32_t v; bool is16; } forges[] = {
        {22, 0x80000000u, false},       // height = INT32_MIN: -h does not exist
        {22, 0x7fffffffu, false},       // height far beyond the file
        {22, 0xfffffff9u, false},       // top-down, 7 rows: legal
        {18, 0x7fffffffu, false},       // width far beyond the file
        {18, 0, false}, {22, 0, false}, // empty image
        {10, 0xfffffff0u, false},       // data offset past the end of the file
        {10, 0, false},                 // data offset inside the header: reads header bytes as pixels, still inside the file
        {46, 100000u, false},           // more palette entries than a palette has
        {46, 2, false},                 // short palette
        {14, 12, false},                // OS/2 header
        {14, 0xffffff00u, false},       // header size past the file: the palette offset must not wrap
        {28, 1, true}, {28, 16, true}, {28, 24, true}, {28, 32, true},   // other bit depths (24 / 32: rows no longer fit)
        {30, 1, false},                 // RLE
    };
    for (const Forge &f : forges) {
        std::vector<unsigned char> b = good_bmp(13, 7);
        if (f.is16) put16(b, f.at, (uint16_t)f.v); else put32(b, f.at, f.v);
        write_file(p, b);
        px.clear();
        const bool ok = slx::ReadBmpGray(p, px, r, c);
        if (ok) CHECK(r > 0 && c > 0 && px.size() == (size_t)r * (size_t)c);
    }
    // truncated at every length
    const std::vector<unsigned char> whole = good_bmp(13, 7);
    for (size_t n = 0; n < whole.size(); n += 7) {
        write_file(p, std::vector<unsigned char>(whole.begin(), whole.begin() + (long)n));
        CHECK(!slx::ReadBmpGray(p, px, r, c));
    }
    // random bytes behind a valid magic
    std::mt19937 rng(7);
    for (int k = 0; k < 300; k++) {
        std::vector<unsigned char> b = good_bmp(9, 5);
        for (int j = 0; j < 6; j++) b[2 + rng() % 52] = (unsigned char)rng();
        write_file(p, b);
        const bool ok = slx::ReadBmpGray(p, px, r, c);
        if (ok) CHECK(px.size() == (size_t)r * (size_t)c);
    }
    // the C entry points: size query, short buffer
    write_file(p, good_bmp(13, 7));
    CHECK(slx_read_bmp_gray(p.c_str(), nullptr, 0, &r, &c) == SLX_OK && r == 7 && c == 13);
    std::vector<uint8_t> small(10);
    CHECK(slx_read_bmp_gray(p.c_str(), small.data(), small.size(), &r, &c) == SLX_ERR_INVALID_ARG);
    CHECK(slx_read_bmp_gray((dir + "/missing.bmp").c_str(), nullptr, 0, &r, &c) == SLX_ERR_UNAVAILABLE);
    // a DIRECTORY where a file is expected (fopen succeeds on one, ftell then answers LONG_MAX): refused, no allocation attempted
    CHECK(slx_read_bmp_gray(dir.c_str(), nullptr, 0, &r, &c) == SLX_ERR_UNAVAILABLE);
    CHECK(slx_read_pgm_gray(dir.c_str(), nullptr, 0, &r, &c) == SLX_ERR_UNAVAILABLE);
    CHECK(slx_read_bmp_gray("/", nullptr, 0, &r, &c) == SLX_ERR_UNAVAILABLE && !slx::ReadBmpGray("/tmp", px, r, c) && !slx::ReadPgmGray("/tmp", px, r, c));
    {
        double a[9], b[9], c9[9], t3[3];
        CHECK(slx_read_calibration_yaml(dir.c_str(), a, b, c9, t3) == SLX_ERR_UNAVAILABLE);
        std::vector<int16_t> lut;
        CHECK(!slx::ReadGrayCodeFile(dir, 6, lut));
    }
}

static void test_pgm(const std::string &dir)
{
    std::vector<uint8_t> px;
    int r = 0, c = 0;
    const std::string p = dir + "/t.pgm";
    auto put = [&](const std::string &text) { std::ofstream f(p.c_str(), std::ios::binary); f << text; };
    put(std::string("P5\n# a comment\n3 2\n255\n") + std::string("\x01\x02\x03\x04\x05\x06", 6));
    CHECK(slx::ReadPgmGray(p, px, r, c) && r == 2 && c == 3 && px[5] == 6);
    for (const char *bad : {"P5\n3 2\n255\n\x01\x02", "P5\n99999999999 2\n255\n", "P5\n3 2\n65535\n", "P5\n-3 2\n255\n", "P5\n3", "P5 1073741824 1073741824 255 ", "P2\n1 1\n255\n0", "P5\n#",
                            "P5\n1073741823 1073741823\n255\nx"}) {
        put(bad);
        CHECK(!slx::ReadPgmGray(p, px, r, c));
    }
}

static void test_yaml_and_gray_table(const std::string &dir)
{
    const std::string p = dir + "/c.yml";
    auto put = [&](const std::string &text) { std::ofstream f(p.c_str()); f << text; };
    auto mat = [](const char *k, int n) {
        std::string s = std::string(k) + ": !!opencv-matrix\n   rows: 3\n   cols: 3\n   dt: d\n   data: [ ";
        for (int i = 0; i < n; i++) s += (i ? ", " : "") + std::to_string(i + 1) + ".5e+000";
        return s + " ]\n";
    };
    slx::Calibration cal;
    put("%YAML:1.0\n" + mat("CamMat", 9) + mat("ProMat", 9) + mat("R", 9) + mat("T", 3));
    CHECK(slx::ReadCalibrationYaml(p, cal) && cal.CamMat[8] == 9.5 && cal.T[2] == 3.5);
    put("%YAML:1.0\n" + mat("CamMat", 9) + mat("ProMat", 9) + mat("R", 10) + mat("T", 3));     // one value too many
    CHECK(!slx::ReadCalibrationYaml(p, cal));
    put("%YAML:1.0\n" + mat("CamMat", 9) + mat("ProMat", 9) + mat("R", 9));                       // a key missing
    CHECK(!slx::ReadCalibrationYaml(p, cal));
    put("CamMat: data: [ 1, 2");                                                                  // no closing bracket
    CHECK(!slx::ReadCalibrationYaml(p, cal));
    put("CamMat:");
    CHECK(!slx::ReadCalibrationYaml(p, cal));
    put("xCamMat: data: [1]\nT: data: [ nan, x, 1 ]\n");
    CHECK(!slx::ReadCalibrationYaml(p, cal));
    double a[9], b[9], c[9], t[3];
    CHECK(slx_read_calibration_yaml((dir + "/none.yml").c_str(), a, b, c, t) == SLX_ERR_UNAVAILABLE);
    // Gray table "<binary> <gray>" rows (R/Patterns/vGrayCode.txt): out-of-range and huge entries must not index outside the table
    const std::string g = dir + "/g.txt";
    { std::ofstream f(g.c_str()); f << "0 0\n1 1\n2 3\n3 2\n7 99\n-5 -1\n99999999999 3\n4 2147483647\n"; }
    std::vector<int16_t> lut;
    CHECK(slx::ReadGrayCodeFile(g, 8, lut) && lut.size() == 8 && lut[1] == 1);
    CHECK(!slx::ReadGrayCodeFile(dir + "/none.txt", 8, lut));
}

// The point-cloud text file against the loop it replaces (R/CCalculation.cpp:351-353: `file << x << ' ' << y << ' ' << z << endl`):
// the same bytes for ordinary coordinates, every exponent range, values that round up to the next power of ten at six digits,
// zeros, infinities and NaNs of both signs, and for point counts on and around the writer's block size.
static void test_point_cloud_text(const std::string &dir)
{
    const std::string p = dir + "/cloud.txt";
    auto reference = [](const std::vector<double> &xyz) {
        std::ostringstream os;
        for (size_t k = 0; k + 2 < xyz.size(); k += 3) {
            os << xyz[k] << ' ';
            os << xyz[k + 1] << ' ';
            os << xyz[k + 2] << std::endl;
        }
        return os.str();
    };
    auto file_bytes = [&]() {
        std::ifstream f(p.c_str(), std::ios::in | std::ios::binary);
        return std::string((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    };
    const double inf = std::numeric_limits<double>::infinity(), nan = std::numeric_limits<double>::quiet_NaN();
    std::vector<double> special = {0.0, -0.0, 1.0, -1.0, 999999.5, 999999.4, 0.0001, 0.00009999995, 0.000099999949, 123456.5, 1234565.0, 1e-5, 1e5, 1e6, 1e-300, -1e300,
                                   5e-324, -5e-324, 1.7976931348623157e308, inf, -inf, nan, -nan, 0.1, 2.5, 3.5, 1234.5678, -987.654321, 100000.5, 99999.95, 0.30000000000000004, 123.4565,
                                   123.4575, 1e21, 1e22, 1e23};
    while (special.size() % 3) special.push_back(42.0);
    CHECK(slx::WritePointCloudText(p, special.data(), special.size() / 3) && file_bytes() == reference(special));
    // values that sit EXACTLY on a rounding boundary at six significant digits (the integer-arithmetic formatter rounds half to
    // even on the exact binary value, like printf): 1000005, 1000015, ... and their binary scalings, both signs
    {
        std::vector<double> ties;
        for (long i = 0; i < 30000; i++) {
            const double base = (double)(1000005 + 10 * i);
            for (double v : {base, -base, base / 16, base * 8, base / 1024, base / 65536.0, base / 1048576.0 / 64, base * 1024 * 1024, base / 1073741824.0 / 128}) ties.push_back(v);
        }
        while (ties.size() % 3) ties.push_back(0.5);
        CHECK(slx::WritePointCloudText(p, ties.data(), ties.size() / 3) && file_bytes() == reference(ties));
    }
    uint64_t x = 88172645463325252ull;
    auto next = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return x; };
    for (size_t n : {(size_t)0, (size_t)1, (size_t)2, (size_t)65535, (size_t)65536, (size_t)65537, (size_t)200001}) {
        std::vector<double> xyz(3 * n);
        for (size_t i = 0; i < xyz.size(); i++) {
            const uint64_t r = next();
            if (i % 7 == 0) {                                             // any bit pattern: every exponent, denormals, NaNs
                std::memcpy(&xyz[i], &r, sizeof r);
            } else {                                                      // the range a depth map lives in
                xyz[i] = ((double)(r >> 11) / 9007199254740992.0 - 0.3) * 1500.0;
            }
        }
        CHECK(slx_write_point_cloud_text(p.c_str(), n ? xyz.data() : nullptr, n) == SLX_OK);
        CHECK(file_bytes() == reference(xyz));
    }
    // SLX_TEXT_MSVC2013, the bytes of the reference as built (MSVC 2013 runtime, text-mode stream): the libstdc++ text with every exponent
    // padded to three digits, that runtime's spellings of the non-finite values, CR LF.  The dialect's own reference, made from
    // libstdc++'s by those three rewrites -- a second statement of the rule, not a third-party oracle (parity unpinned: the reference ships no output).
    {
        auto msvc = [&](const std::vector<double> &xyz) {
            std::string out;
            for (size_t k = 0; k < xyz.size(); k++) {
                const double v = xyz[k];
                std::string t;
                if (std::isnan(v)) t = std::signbit(v) ? "-1.#IND" : "1.#QNAN";
                else if (std::isinf(v)) t = v < 0 ? "-1.#INF" : "1.#INF";
                else {
                    std::ostringstream os;
                    os << v;
                    t = os.str();
                    const size_t e = t.find('e');
                    if (e != std::string::npos && t.size() - (e + 2) == 2) t.insert(e + 2, "0");       // e+05 -> e+005; e-300 stays
                }
                out += t;
                out += k % 3 == 2 ? "\r\n" : " ";
            }
            return out;
        };
        CHECK(slx::WritePointCloudText(p, special.data(), special.size() / 3, SLX_TEXT_MSVC2013) && file_bytes() == msvc(special));
        const std::vector<double> expo = {5e-5, -5e-5, 1.5e-5, 1e6, 1.25e15, -9.99999e-5, 1e-300, 1e300, 1e100, 2.5, 0.0, 100000.0};
        CHECK(slx_write_point_cloud_text_ex(p.c_str(), expo.data(), expo.size() / 3, SLX_TEXT_MSVC2013) == SLX_OK);
        CHECK(file_bytes() == "5e-005 -5e-005 1.5e-005\r\n1e+006 1.25e+015 -9.99999e-005\r\n1e-300 1e+300 1e+100\r\n2.5 0 100000\r\n");
        CHECK(slx_write_point_cloud_text_ex(p.c_str(), expo.data(), expo.size() / 3, SLX_TEXT_LIBSTDCXX) == SLX_OK && file_bytes() == reference(expo));
        std::vector<double> many(3 * 70001);
        for (size_t i = 0; i < many.size(); i++) {
            const uint64_t r = next();
            if (i % 5 == 0) std::memcpy(&many[i], &r, sizeof r);
            else many[i] = ((double)(r >> 11) / 9007199254740992.0 - 0.5) * ((i % 3) ? 1500.0 : 1e-3);
        }
        CHECK(slx_write_point_cloud_text_ex(p.c_str(), many.data(), many.size() / 3, SLX_TEXT_MSVC2013) == SLX_OK && file_bytes() == msvc(many));
        CHECK(slx_write_point_cloud_text_ex(p.c_str(), expo.data(), 1, 7) == SLX_ERR_INVALID_ARG);
    }
    CHECK(slx_write_point_cloud_text(nullptr, nullptr, 0) == SLX_ERR_INVALID_ARG);
    CHECK(slx_write_point_cloud_text(p.c_str(), nullptr, 5) == SLX_ERR_INVALID_ARG);
    CHECK(slx_write_point_cloud_text((dir + "/no/such/directory/cloud.txt").c_str(), special.data(), 1) == SLX_ERR_UNAVAILABLE);
}

static slx_config base_config(int w, int h, int mode, int F, int N, int G)
{
    static int16_t lut[1 << 16];
    slx_config c;
    std::memset(&c, 0, sizeof c);
    c.width = w; c.height = h; c.mode = mode; c.n_freq = F; c.n_steps = N; c.gray_bits = G; c.gray_stripe = 20; c.gray_lut = lut;
    for (int f = 0; f < SLX_MAX_FREQ; f++) c.period[f] = 1920 >> (3 * f);
    c.fov_min = 100; c.fov_max = 1000; c.device = -1;
    c.cam[0] = c.cam[4] = 1200; c.cam[8] = 1; c.pro[0] = c.pro[4] = 2000; c.pro[8] = 1; c.rot[0] = c.rot[4] = c.rot[8] = 1; c.trans[0] = -30;
    return c;
}

static void test_validate_and_create()
{
    char msg[256];
    slx_config c = base_config(64, 48, SLX_MODE_MULTIFREQ, 3, 4, 0);
    CHECK(slx_validate_config(&c, msg, sizeof msg) == SLX_OK);
    CHECK(slx_validate_config(nullptr, msg, sizeof msg) == SLX_ERR_INVALID_ARG);
    for (int w : {0, -1, INT_MAX}) { slx_config d = c; d.width = w; d.height = INT_MAX; CHECK(slx_validate_config(&d, msg, 4) != SLX_OK); }
    for (int m : {-1, 5, INT_MIN}) { slx_config d = c; d.mode = m; CHECK(slx_validate_config(&d, nullptr, 0) != SLX_OK); }
    for (int n : {0, 2, SLX_MAX_STEPS + 1, INT_MAX}) { slx_config d = c; d.n_steps = n; CHECK(slx_validate_config(&d, msg, sizeof msg) != SLX_OK); }
    for (int f : {0, SLX_MAX_FREQ + 1, INT_MIN}) { slx_config d = c; d.n_freq = f; CHECK(slx_validate_config(&d, msg, sizeof msg) != SLX_OK); }
    { slx_config d = c; d.period[1] = 1 << 24; CHECK(slx_validate_config(&d, msg, sizeof msg) != SLX_OK); }
    { slx_config d = base_config(64, 48, SLX_MODE_GRAY_PHASE, 1, 4, 17); CHECK(slx_validate_config(&d, msg, sizeof msg) != SLX_OK); }
    { slx_config d = base_config(64, 48, SLX_MODE_GRAY_PHASE, 1, 4, 6); d.gray_lut = nullptr; CHECK(slx_validate_config(&d, msg, sizeof msg) != SLX_OK); }
    { slx_config d = c; d.aux_outputs = 0xffffffffu; CHECK(slx_validate_config(&d, msg, sizeof msg) != SLX_OK); }
    // No device is visible to this process (tests/test_sanitizers.py hides the GPUs of a GPU box, so that the run is the same
    // everywhere and no runtime allocation needs a leak suppression): never a crash, never a CPU fallback, always SLX_ERR_NO_DEVICE.
    slx_ctx *ctx = nullptr;
    const int rc = slx_create(&c, &ctx);
    CHECK(rc == SLX_ERR_NO_DEVICE);
    CHECK(ctx == nullptr && std::strlen(slx_last_error(nullptr)) > 0);
    slx_destroy(nullptr);
}

// Every plan the launcher can make for a tile: the work items of all tiers must cover every row of a frame-set exactly once
// (rows past the tile are allowed, they are range-checked away), and what the kernel forms in 32 bits must fit.
static void check_plan(int w, int h, int n_sets, int mode, int F, int N, int G, const SlxTuning *tn, bool aux)
{
    SlxKParams kp;
    std::memset(&kp, 0, sizeof kp);
    kp.width = w; kp.height = h; kp.quads_per_row = (unsigned)((w + 3) / 4); kp.n_quads = kp.quads_per_row * (unsigned)h;
    kp.n_freq = F; kp.n_steps = N; kp.gray_bits = G; kp.gray_stripe = 20; kp.aligned = (w % 4) == 0; kp.row_stride = (size_t)w;
    for (int f = 0; f < F; f++) kp.period[f] = 1920 >> (3 * f);
    kp.cx = w / 2.0; kp.cy = h / 2.0; kp.fu = kp.fv = 1200; kp.P00 = 1; kp.P01 = .1; kp.P20 = .01; kp.P21 = .02; kp.K1 = 5; kp.K2 = 7; kp.cA = 3; kp.cB = 2;
    kp.out_set_stride = (size_t)w * (size_t)h;
    const float r = 0.70710677f;
    const float ey[8] = {1.f, r, 0.f, -r, -1.f, -r, 0.f, r}, ex[8] = {0.f, r, 1.f, r, 0.f, -r, -1.f, -r};
    for (int k = 0; k < 8 && N == 8; k++) kp.wy[k] = ey[k], kp.wx[k] = ex[k];
    kp.wscale = 2.0f / (float)N;
    static uint8_t arena[1];                                         // addresses only: the planner never dereferences a plane
    const size_t plane = (size_t)w * (size_t)h;
    for (int i = 0; i < F * N; i++) kp.phase[i] = arena + (size_t)i * plane;
    for (int i = 0; i < 2 * G; i++) kp.gray[i] = arena + (size_t)(F * N + i) * plane;
    kp.phase_set_stride = kp.gray_set_stride = (size_t)(F * N + 2 * G) * plane;
    static double out_arena[2];                                      // addresses only
    if (mode == SLX_MODE_PHASE_ONLY) kp.pix = out_arena;             // the decoder modes write pix / gray, not z
    if (mode == SLX_MODE_GRAY_ONLY) kp.gray_out = out_arena;
    kp.std_gray = 1;
    for (int variant : {SLX_VARIANT_AUTO, SLX_VARIANT_GENERIC, SLX_VARIANT_STRIP}) {
        SlxLaunchPlan plan;
        const int rc = slx_plan_launch(kp, mode, aux, n_sets, variant, tn, &plan);
        if (rc != 0) { CHECK(variant == SLX_VARIANT_STRIP || n_sets <= 0 || n_sets > 65535); continue; }
        CHECK(plan.grid_x >= 1 && plan.grid_y >= 1 && plan.block >= 64 && plan.block <= 256 && plan.lds_bytes <= 160u * 1024u);
        (plan.strip ? g_strip_plans : g_generic_plans)++;
        if (!plan.strip) { CHECK((unsigned long long)plan.grid_x * plan.block >= kp.n_quads && (int)plan.grid_y == n_sets); continue; }
        const SlxKParams &q = plan.kp;
        CHECK(q.interleave >= 1 && q.interleave <= 64 && (q.interleave * q.quads_per_row) % (mode == SLX_MODE_MULTIFREQ_GRAYMASK ? 1u : 64u) == 0);
        CHECK(q.n_tiers >= 1 && q.n_tiers <= SLX_MAX_TIERS && q.tier_row0[0] == 0 && q.tier_first_wg[0] == 0);
        unsigned row = 0;
        unsigned long long wgs = 0;
        for (unsigned t = 0; t < q.n_tiers; t++) {
            CHECK(q.tier_row0[t] == row && q.tier_rows[t] >= 1 && q.tier_rows[t] <= 32 && q.tier_first_wg[t] == wgs);
            CHECK(q.tier_items_per_set[t] % q.chunks_per_group == 0 && q.tier_items[t] == q.tier_items_per_set[t] * (unsigned)n_sets);
            const unsigned groups = q.tier_items_per_set[t] / q.chunks_per_group;
            CHECK(groups >= 1);
            row += groups * q.tier_rows[t] * q.interleave;
            CHECK((unsigned long long)q.tier_wgs[t] * (plan.block / 64) >= q.tier_items[t]);
            wgs += q.tier_wgs[t];
        }
        CHECK(row >= (unsigned)h);                                            // every row belongs to exactly one tier's items
        CHECK(row < (unsigned)h + 64u * 32u + 64u);                           // and the overshoot stays inside what slx_strip_eligible bounded
        CHECK(wgs == plan.grid_x);
        CHECK((unsigned long long)(row + 1) * (unsigned)w * 8ull < (1ull << 32));   // 32-bit byte offsets of the depth stores, rows past the tile included
    }
}

// The stream kernel's plan: every (row group, chunk column) of the launch belongs to exactly one queue position, the scalar
// multiply-high that turns a group number into (frame-set, group) is exact for every group, the groups cover the tile.
static void check_stream_plan(int w, int h, int n_sets, int F, int rows, unsigned n_cus = 0)
{
    SlxKParams kp;
    std::memset(&kp, 0, sizeof kp);
    kp.width = w; kp.height = h; kp.quads_per_row = (unsigned)((w + 3) / 4); kp.n_quads = kp.quads_per_row * (unsigned)h;
    kp.n_freq = F; kp.n_steps = 4; kp.aligned = (w % 4) == 0; kp.row_stride = (size_t)w;
    for (int f = 0; f < F; f++) kp.period[f] = 1920 >> (3 * f);
    kp.cx = w / 2.0; kp.cy = h / 2.0; kp.fu = kp.fv = 1200; kp.P00 = 1; kp.P01 = .1; kp.P20 = .01; kp.P21 = .02; kp.K1 = 5; kp.K2 = 7; kp.cA = 3; kp.cB = 2;
    kp.out_set_stride = (size_t)w * (size_t)h;
    static uint8_t arena[1];
    static unsigned counters[1];
    static double out_arena[2];
    const size_t plane = (size_t)w * (size_t)h;
    for (int i = 0; i < F * 4; i++) kp.phase[i] = arena + (size_t)i * plane;
    kp.phase_set_stride = (size_t)(F * 4) * plane;
    kp.z = out_arena;
    kp.sq_counters = counters;
    kp.n_cus = n_cus;
    SlxTuning tn;
    std::memset(&tn, 0, sizeof tn);
    tn.stream = 2;
    tn.stream_rows = rows;
    SlxLaunchPlan plan;
    if (slx_plan_launch(kp, SLX_MODE_MULTIFREQ, false, n_sets, SLX_VARIANT_AUTO, &tn, &plan) != 0) return;
    if (!plan.stream) return;
    g_stream_plans++;
    const SlxKParams &q = plan.kp;
    CHECK(q.sq_rows >= 2 && q.sq_rows <= 16 && q.sq_m >= 1 && q.sq_queues == q.chunks_per_group * q.sq_m && q.sq_queues <= SLX_STREAM_MAX_QUEUES);
    CHECK(q.chunks_per_group * 64u == q.interleave * q.quads_per_row);
    CHECK((unsigned long long)q.sq_groups_per_set * q.sq_rows * q.interleave >= (unsigned)h);
    CHECK((unsigned long long)(q.sq_groups_per_set - 1) * q.sq_rows * q.interleave < (unsigned)h);          // the last group starts inside the tile
    CHECK(q.sq_groups_total == q.sq_groups_per_set * (unsigned)n_sets);
    CHECK(plan.block == 256 && plan.grid_x >= 1 && plan.grid_x <= 1024 && plan.lds_bytes <= 160u * 1024u);
    {
        // every queue that holds items is polled: a wave takes queue (wave number) % queues, so the launch needs a wave for every
        // queue index below chunks_per_group * min(sq_m, row groups) (a device with few compute units has fewer waves than 255)
        const unsigned long long launched = (unsigned long long)plan.grid_x * (plan.block / 64u);
        const unsigned long long holding = (unsigned long long)q.chunks_per_group * std::min(q.sq_m, q.sq_groups_total);
        CHECK(launched >= holding);
        if (n_cus) CHECK(launched <= (unsigned long long)n_cus * 16ull + 3ull);
    }
    // every group is some queue's k-th item exactly once, and the multiply-high division is exact
    unsigned long long seen = 0;
    for (unsigned j = 0; j < q.sq_m; j++) {
        const unsigned Kq = q.sq_groups_total > j ? (q.sq_groups_total - j + q.sq_m - 1u) / q.sq_m : 0u;
        for (unsigned k = 0; k < Kq; k += (Kq > 4096u ? 97u : 1u)) {
            const unsigned G = k * q.sq_m + j;
            CHECK(G < q.sq_groups_total);
            const unsigned set = q.sq_groups_per_set == 1u ? G : (unsigned)(((unsigned long long)G * q.sq_magic) >> 32);   // as slx_stream_kernel
            CHECK(set == G / q.sq_groups_per_set && set < (unsigned)n_sets);
        }
        seen += Kq;
    }
    CHECK(seen == q.sq_groups_total);
    // the largest offsets the kernel forms in 32 bits
    CHECK((unsigned long long)((unsigned)h + q.sq_rows * q.interleave) * (unsigned)w * 8ull < (1ull << 32));
}

// The same for slx_gstream_kernel (the reference's own mode: 6 Gray bits on the ring + one 4-step frequency): queues, exact division,
// coverage, 32-bit offsets; taken by the planner itself only from 8 items per resident wave on.
static int g_gstream_plans = 0;
static void check_gstream_plan(int w, int h, int n_sets, int rows, int force, unsigned n_cus = 0)
{
    SlxKParams kp;
    std::memset(&kp, 0, sizeof kp);
    kp.width = w; kp.height = h; kp.quads_per_row = (unsigned)((w + 3) / 4); kp.n_quads = kp.quads_per_row * (unsigned)h;
    kp.n_freq = 1; kp.n_steps = 4; kp.aligned = (w % 4) == 0; kp.row_stride = (size_t)w;
    kp.period[0] = 40; kp.gray_bits = 6; kp.gray_stripe = 20; kp.std_gray = 1;
    kp.cx = w / 2.0; kp.cy = h / 2.0; kp.fu = kp.fv = 1200; kp.P00 = 1; kp.P01 = .1; kp.P20 = .01; kp.P21 = .02; kp.K1 = 5; kp.K2 = 7; kp.cA = 3; kp.cB = 2;
    kp.out_set_stride = (size_t)w * (size_t)h;
    static uint8_t arena[1], garena[1];
    static unsigned counters[1];
    static double out_arena[2];
    const size_t plane = (size_t)w * (size_t)h;
    for (int i = 0; i < 4; i++) kp.phase[i] = arena + (size_t)i * plane;
    for (int i = 0; i < 12; i++) kp.gray[i] = garena + (size_t)i * plane;
    kp.phase_set_stride = 4 * plane;
    kp.gray_set_stride = 12 * plane;
    kp.z = out_arena;
    kp.sq_counters = counters;
    kp.n_cus = n_cus;
    SlxTuning tn;
    std::memset(&tn, 0, sizeof tn);
    tn.stream = force ? 2 : 0;
    tn.stream_rows = rows;
    SlxLaunchPlan plan;
    if (slx_plan_launch(kp, SLX_MODE_GRAY_PHASE, false, n_sets, SLX_VARIANT_AUTO, &tn, &plan) != 0) return;
    const SlxKParams &q = plan.kp;
    if (plan.stream != 2) {
        CHECK(plan.stream == 0);                                      // never the Gray-free kernel
        return;
    }
    g_gstream_plans++;
    CHECK(plan.gray_ring_bits == 6 && q.sq_rows == (unsigned)(rows ? rows : 1) && q.sq_m >= 1 && q.sq_queues == q.chunks_per_group * q.sq_m && q.sq_queues <= SLX_STREAM_MAX_QUEUES);
    CHECK(q.chunks_per_group * 64u == q.interleave * q.quads_per_row);
    CHECK((unsigned long long)q.sq_groups_per_set * q.sq_rows * q.interleave >= (unsigned)h);
    CHECK((unsigned long long)(q.sq_groups_per_set - 1) * q.sq_rows * q.interleave < (unsigned)h);
    CHECK(q.sq_groups_total == q.sq_groups_per_set * (unsigned)n_sets);
    CHECK(plan.block == 256 && plan.grid_x >= 1 && plan.grid_x <= 1024 && plan.lds_bytes == 4u * 8192u);
    const unsigned long long launched = (unsigned long long)plan.grid_x * (plan.block / 64u);
    CHECK(launched >= (unsigned long long)q.chunks_per_group * std::min(q.sq_m, q.sq_groups_total));
    if (!force) CHECK((unsigned long long)q.sq_groups_total * q.chunks_per_group >= 8ull * (n_cus ? n_cus : 256u) * 16ull);   // the planner's own threshold
    unsigned long long seen = 0;
    for (unsigned j = 0; j < q.sq_m; j++) {
        const unsigned Kq = q.sq_groups_total > j ? (q.sq_groups_total - j + q.sq_m - 1u) / q.sq_m : 0u;
        for (unsigned k = 0; k < Kq; k += (Kq > 4096u ? 97u : 1u)) {
            const unsigned G = k * q.sq_m + j;
            const unsigned set = q.sq_groups_per_set == 1u ? G : (unsigned)(((unsigned long long)G * q.sq_magic) >> 32);
            CHECK(G < q.sq_groups_total && set == G / q.sq_groups_per_set && set < (unsigned)n_sets);
        }
        seen += Kq;
    }
    CHECK(seen == q.sq_groups_total);
    CHECK((unsigned long long)((unsigned)h + q.sq_rows * q.interleave) * (unsigned)w * 8ull < (1ull << 32));
}

static void test_plans()
{
    for (int w : {1280, 1920, 640, 4096, 516, 64, 252})
        for (int h : {1024, 720, 150, 37, 7, 3000})
            for (int n : {1, 2, 7, 9, 32, 256, 4000})
                for (int rows : {0, 1, 2, 3, 16})
                    for (int force : {0, 1})
                        if ((unsigned long long)w * h * 16ull * (unsigned)n < (1ull << 40)) check_gstream_plan(w, h, n, rows, force);
    for (unsigned cus : {1u, 4u, 15u, 64u, 128u})
        for (int n : {2, 9, 256}) check_gstream_plan(1280, 1024, n, 0, 1, cus);
    CHECK(g_gstream_plans > 300);

    for (int w : {1920, 1280, 640, 4096, 516, 64, 252})
        for (int h : {1200, 720, 150, 37, 7, 3000})
            for (int n : {1, 2, 9, 32, 256, 4000})
                for (int rows : {0, 2, 3, 4, 16})
                    if ((unsigned long long)w * h * 12ull * (unsigned)n < (1ull << 40)) check_stream_plan(w, h, n, 3, rows);
    CHECK(g_stream_plans > 300);
    {
        // small and partitioned devices (the planner reads the device's compute-unit count): fewer resident waves than queues
        const int before = g_stream_plans;
        for (unsigned cus : {1u, 2u, 4u, 8u, 15u, 16u, 32u, 64u, 128u})
            for (int w : {1920, 1280, 64, 516})
                for (int n : {1, 9, 256})
                    check_stream_plan(w, 1200, n, 3, 2, cus);
        CHECK(g_stream_plans > before + 40);
    }
    std::mt19937 rng(11);
    const int shapes[][2] = {{1920, 1200}, {1280, 1024}, {1280, 720}, {640, 480}, {4096, 3000}, {4, 1}, {8, 1200}, {500, 5}, {1920, 150}, {1920, 37}, {4096, 130}, {64, 20}, {252, 3000}, {4092, 17}};
    for (const auto &s : shapes)
        for (int n_sets : {1, 2, 5, 32, 256, 4000})
            for (int cfg = 0; cfg < 6; cfg++) {
                const int mode = cfg == 0 ? SLX_MODE_MULTIFREQ : cfg == 1 ? SLX_MODE_GRAY_PHASE : cfg == 2 ? SLX_MODE_MULTIFREQ_GRAYMASK : cfg == 3 ? SLX_MODE_MULTIFREQ
                                 : cfg == 4 ? SLX_MODE_PHASE_ONLY : SLX_MODE_GRAY_ONLY;           // 4, 5: the decoder objects' strip kernel
                const int F = cfg == 5 ? 0 : (cfg == 1 || cfg == 4) ? 1 : cfg == 3 ? 4 : 3, N = cfg == 3 ? 8 : 4, G = (cfg == 1 || cfg == 2 || cfg == 5) ? 6 : 0;
                if ((unsigned long long)s[0] * s[1] * (unsigned)(F * N + 2 * G) * (unsigned)n_sets >= (1ull << 40)) continue;
                check_plan(s[0], s[1], n_sets, mode, F, N, G, nullptr, false);
                if (cfg < 4) check_plan(s[0], s[1], n_sets, mode, F, N, G, nullptr, true);
                for (int k = 0; k < 6; k++) {
                    SlxTuning tn;
                    std::memset(&tn, 0, sizeof tn);
                    tn.strip_rows = (int)(rng() % 33); tn.tail_pct = (int)(rng() % 101) - 1; tn.tail_rows = (int)(rng() % 33); tn.tiers = (int)(rng() % 5);
                    tn.strip_waves = (int)(rng() % 5); tn.gray_plain = (int)(rng() % 2);
                    check_plan(s[0], s[1], n_sets, mode, F, N, G, &tn, false);
                }
            }
    // random small tiles
    for (int k = 0; k < 3000; k++) check_plan(4 * (1 + (int)(rng() % 600)), 1 + (int)(rng() % 400), 1 + (int)(rng() % 40), SLX_MODE_MULTIFREQ, 3, 4, 0, nullptr, false);
    // the rows-per-item rule itself: always 1..16, monotone sanity
    for (unsigned hh : {1u, 7u, 150u, 720u, 1200u, 3000u, 100000u})
        for (unsigned n : {1u, 3u, 32u, 65535u})
            for (unsigned pref : {0u, 3u, 10u, 16u, 99u}) {
                const unsigned rws = slx_strip_rows_model(hh, 2, 15, n, 16, pref, (n % 3) ? 256u : 0u);
                CHECK(rws >= 1 && rws <= 16);
            }
}

static void test_gather_plans()
{
    // every rank's message list against the others': per ordered pair the k-th send meets the k-th receive with the same length
    for (int world = 1; world <= 8; world++)
        for (int split = 0; split < 2; split++)
            for (int root : {-1, 0, world - 1}) {
                const int H = 37, W = 12, total = 2 * world + 1;
                std::vector<slx_shard> sh((size_t)world);
                for (int r = 0; r < world; r++) {
                    if (split == 0) { const int lo = total * r / world, hi = total * (r + 1) / world; sh[(size_t)r] = {lo, hi - lo, 0, H}; }
                    else { const int lo = H * r / world, hi = H * (r + 1) / world; sh[(size_t)r] = {0, total, lo, hi - lo}; }
                }
                std::vector<std::vector<slx_msg>> plans((size_t)world);
                for (int r = 0; r < world; r++) {
                    int n = 0;
                    CHECK(slx_gather_plan(sh.data(), world, r, H, W, 0, total, 0, root, nullptr, 0, &n) == SLX_OK);
                    plans[(size_t)r].resize((size_t)n);
                    CHECK(slx_gather_plan(sh.data(), world, r, H, W, 0, total, 0, root, plans[(size_t)r].data(), n, &n) == SLX_OK);
                }
                for (int a = 0; a < world; a++)
                    for (int b = 0; b < world; b++) {
                        std::vector<unsigned long long> sends, recvs;
                        for (const slx_msg &m : plans[(size_t)a]) if (m.send && m.peer == b) sends.push_back(m.count);
                        for (const slx_msg &m : plans[(size_t)b]) if (!m.send && m.peer == a) recvs.push_back(m.count);
                        CHECK(sends == recvs);
                        for (const slx_msg &m : plans[(size_t)b]) if (!m.send) CHECK(m.offset + m.count <= (unsigned long long)total * H * W);
                    }
            }
    int n = 0;
    CHECK(slx_gather_plan(nullptr, 2, 0, 4, 4, 0, 1, 0, 0, nullptr, 0, &n) == SLX_ERR_INVALID_ARG);
    // the staged shape (slx_gather_plan_ex) in chunks: sends meet receives, staged receives stay inside the slot the plan asks for,
    // the scatter list reads every staged double once and writes inside the full array
    for (int world = 2; world <= 8; world++)
        for (int root : {0, world - 1})
            for (int chunk : {1, 3, 100}) {
                const int H = 41, W = 10, total = 7;
                std::vector<slx_shard> sh((size_t)world);
                for (int r = 0; r < world; r++) { const int lo = H * r / world, hi = H * (r + 1) / world; sh[(size_t)r] = {0, total, lo, hi - lo}; }
                for (int first = 0; first < total; first += chunk) {
                    std::vector<std::vector<slx_msg>> plans((size_t)world);
                    std::vector<slx_scatter> scat;
                    unsigned long long staging = 0;
                    for (int r = 0; r < world; r++) {
                        int nm = 0, ns = 0;
                        unsigned long long st = 0;
                        const size_t ls = r == root ? (size_t)H * W : 0;            // the root holds its tile in place, the others a dense stack
                        CHECK(slx_gather_plan_ex(sh.data(), world, r, H, W, first, chunk, ls, root, SLX_GATHER_STAGED, nullptr, 0, &nm, nullptr, 0, &ns, &st) == SLX_OK);
                        plans[(size_t)r].resize((size_t)nm);
                        std::vector<slx_scatter> sc((size_t)ns);
                        CHECK(slx_gather_plan_ex(sh.data(), world, r, H, W, first, chunk, ls, root, SLX_GATHER_STAGED, plans[(size_t)r].data(), nm, &nm, sc.data(), ns, &ns, &st) == SLX_OK);
                        if (r == root) { scat = sc; staging = st; }
                        else CHECK(ns == 0 && st == 0);
                    }
                    std::vector<int> hits((size_t)staging, 0);
                    for (int a = 0; a < world; a++) {
                        std::vector<unsigned long long> sends, recvs;
                        for (const slx_msg &m : plans[(size_t)a]) if (m.send == 1 && m.peer == root) sends.push_back(m.count);
                        for (const slx_msg &m : plans[(size_t)root]) if (m.send != 1 && m.peer == a) { recvs.push_back(m.count); CHECK(m.send == 2 && m.offset + m.count <= staging); }
                        if (a != root) CHECK(sends == recvs && sends.size() <= 1);
                    }
                    for (const slx_scatter &q : scat)
                        for (unsigned long long t = 0; t < q.n_runs; t++) {
                            CHECK(q.src + t * q.src_stride + q.run <= staging && q.dst + t * q.dst_stride + q.run <= (unsigned long long)total * H * W);
                            for (unsigned long long i = 0; i < q.run && q.src + t * q.src_stride + i < staging; i++) hits[(size_t)(q.src + t * q.src_stride + i)]++;
                        }
                    for (int h : hits) CHECK(h == 1);
                }
            }
    CHECK(slx_gather_plan_ex(nullptr, 2, 0, 4, 4, 0, 1, 0, 0, SLX_GATHER_STAGED, nullptr, 0, &n, nullptr, 0, nullptr, nullptr) == SLX_ERR_INVALID_ARG);
    {
        slx_shard two[2] = {{0, 2, 0, 2}, {0, 2, 2, 2}};
        CHECK(slx_gather_plan_ex(two, 2, 0, 4, 4, 0, 1, 0, 0, 5, nullptr, 0, &n, nullptr, 0, nullptr, nullptr) == SLX_ERR_INVALID_ARG);   // unknown shape
        CHECK(slx_gather_plan_ex(two, 2, 1, 4, 4, 0, 2, 16, 0, SLX_GATHER_STAGED, nullptr, 0, &n, nullptr, 0, nullptr, nullptr) == SLX_ERR_INVALID_ARG);   // sender not dense
        CHECK(slx_scatter_rows(nullptr, nullptr, 0, nullptr, nullptr, nullptr) == SLX_ERR_INVALID_ARG);
    }
}

// The fused point cloud's plan (slx_cloud_fused_plan): the parts tile the map, the sizes stay inside what the kernel's arrays and a
// launch's LDS allow, a device too small to keep a column group's parts resident is refused.
static void test_cloud_plans()
{
    int planned = 0;
    for (int W : {1, 15, 16, 17, 250, 1280, 1920, 4096, 100000})
        for (int H : {1, 31, 32, 33, 255, 256, 257, 480, 1024, 1200, 3000, 4096, 4097, 7000, 9000, 20000})
            for (unsigned cus : {0u, 1u, 2u, 8u, 256u}) {
                int G = -1, P = -1, R = -1;
                if (!slx_cloud_fused_plan(W, H, cus, &G, &P, &R)) {
                    CHECK(H > 4096 || cus == 1u || cus == 2u || cus == 8u || (unsigned long long)W * H >= (1ull << 31) || W > 16 * (1 << 20));   // the refused ones are the expected ones
                    continue;
                }
                planned++;
                CHECK(G == (W + 15) / 16 && P >= 1 && P <= 16 && R >= 1 && R % 64 == 0);
                CHECK((long long)P * R >= H && (long long)(P - 1) * R < H);                       // the last part starts inside the map
                CHECK(slx_cloud_fused_lds_bytes(R) + 2048u <= 64u * 1024u);
                CHECK(slx_cloud_fused_words(G, P) == (size_t)SLX_CLOUD_COUNTERS * 16u + (size_t)G * P * 17u);
                const unsigned long long resident = (cus ? cus : 256u) * std::min<unsigned long long>(R > 256 ? 2u : 3u, 160u * 1024u / (slx_cloud_fused_lds_bytes(R) + 2048u));
                CHECK(R <= SLX_CLOUD_MAX_ROWS);
                CHECK(resident >= (unsigned long long)P);
            }
    CHECK(planned > 200);
    int G, P, R;
    CHECK(slx_cloud_fused_plan(1920, 1200, 256, &G, &P, &R) && G == 120 && P == 5 && R == 256);
    CHECK(!slx_cloud_fused_plan(0, 10, 256, &G, &P, &R) && !slx_cloud_fused_plan(1920, 1200, 1, &G, &P, &R));   // one CU keeps 3 such workgroups: 5 parts do not fit
}

int main(int argc, char **argv)
{
    // the scratch directory: the caller's, or a fresh one under TMPDIR (never the source tree: a run used to leave cloud.txt etc. in tests/cpp)
    std::string dir;
    char tmpl[] = "/tmp/host_sanitize.XXXXXX";
    bool own_dir = false;
    if (argc > 1) dir = argv[1];
    else if (const char *made = mkdtemp(tmpl)) { dir = made; own_dir = true; }
    else { std::perror("mkdtemp"); return 1; }
    test_bmp(dir);
    test_pgm(dir);
    test_yaml_and_gray_table(dir);
    test_point_cloud_text(dir);
    test_validate_and_create();
    test_plans();
    test_gather_plans();
    test_cloud_plans();
    if (own_dir) {                                                    // leave nothing behind in /tmp
        for (const char *f : {"/t.bmp", "/t.pgm", "/c.yml", "/g.txt", "/cloud.txt"}) std::remove((dir + f).c_str());
        std::remove(dir.c_str());
    }
    if (g_fail) { std::fprintf(stderr, "host_sanitize: %d check(s) failed\n", g_fail); return 1; }
    CHECK(g_strip_plans > 5000 && g_generic_plans > 1000);
    if (g_fail) return 1;
    std::printf("host_sanitize ok: %d strip plans, %d generic plans checked\n", g_strip_plans, g_generic_plans);
    return 0;
}
