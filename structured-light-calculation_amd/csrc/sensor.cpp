// sensor.cpp -- see sensor.hpp.  Host-only file I/O; nothing here touches the GPU.
#include "sensor.hpp"

#include <algorithm>
#include <atomic>
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <new>
#include <sstream>
#include <thread>

#include <sys/stat.h>

namespace slx {

namespace {

uint32_t rd32(const unsigned char *p) { return (uint32_t)p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
uint16_t rd16(const unsigned char *p) { return (uint16_t)(p[0] | p[1] << 8); }

// cv::cvtColor BGR2GRAY of OpenCV 2.4 (8-bit): yuv_shift = 14, B2Y = 1868, G2Y = 9617, R2Y = 4899
inline uint8_t luma(unsigned b, unsigned g, unsigned r) { return (uint8_t)((b * 1868u + g * 9617u + r * 4899u + (1u << 13)) >> 14); }

}  // namespace

// The whole file in one read (a stream iterator takes it byte by byte: 1.5 ms of the 2 ms a 1.3 MB image took to load).
// Only regular files (fopen succeeds on a directory and ftell then answers LONG_MAX), and never more than kMaxFileBytes: an image
// or a calibration file of a data directory is a few megabytes, a size beyond 1 GiB is a mistake or a forgery.
static constexpr size_t kMaxFileBytes = (size_t)1 << 30;
static bool read_file(const std::string &path, std::vector<unsigned char> &buf)
{
    std::FILE *fp = std::fopen(path.c_str(), "rb");
    if (!fp) return false;
    buf.clear();
    struct stat st;
    if (fstat(fileno(fp), &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 0 || (unsigned long long)st.st_size > kMaxFileBytes) {
        std::fclose(fp);
        return false;
    }
    bool ok = true;
    try {
        buf.resize((size_t)st.st_size);
        const size_t got = buf.empty() ? 0 : std::fread(buf.data(), 1, buf.size(), fp);
        buf.resize(got);                                             // a file that shrank meanwhile: what was there
        if (got == (size_t)st.st_size) {                             // ... or grew (or reports size 0, like /proc files): the rest in pieces
            unsigned char piece[4096];
            size_t more;
            while (ok && (more = std::fread(piece, 1, sizeof piece, fp)) > 0) {
                if (buf.size() + more > kMaxFileBytes) ok = false;
                else buf.insert(buf.end(), piece, piece + more);
            }
        }
    } catch (const std::bad_alloc &) {
        ok = false;
    }
    ok = ok && std::ferror(fp) == 0;
    std::fclose(fp);
    if (!ok) buf.clear();
    return ok;
}

bool ReadBmpGray(const std::string &path, std::vector<uint8_t> &pixels, int &rows, int &cols)
{
    std::vector<unsigned char> buf;
    if (!read_file(path, buf)) return false;
    if (buf.size() < 54 || buf[0] != 'B' || buf[1] != 'M') return false;
    const uint32_t data_off = rd32(&buf[10]), hdr = rd32(&buf[14]);
    if (hdr < 40) return false;
    const int32_t w = (int32_t)rd32(&buf[18]);
    int32_t h = (int32_t)rd32(&buf[22]);
    const uint16_t bpp = rd16(&buf[28]);
    const uint32_t comp = rd32(&buf[30]);
    if (w <= 0 || h == 0 || h == INT32_MIN || comp != 0) return false;   // (-INT32_MIN does not exist)
    const bool top_down = h < 0;
    if (top_down) h = -h;
    if (bpp != 8 && bpp != 24 && bpp != 32) return false;
    const size_t row_bytes = (((size_t)w * bpp + 31) / 32) * 4;
    // every row must lie inside the file; by division, so that a forged height cannot wrap the product
    if ((size_t)data_off > buf.size() || (size_t)h > (buf.size() - (size_t)data_off) / row_bytes) return false;
    uint8_t pal[256];
    bool identity = false;                                           // a grey ramp as palette (what imwrite of an 8-bit image stores): rows are copied
    if (bpp == 8) {
        uint32_t n_col = rd32(&buf[46]);
        if (n_col == 0 || n_col > 256) n_col = 256;
        const size_t pal_off = 14 + (size_t)hdr;
        if (pal_off + 4 * (size_t)n_col > buf.size()) return false;
        for (uint32_t i = 0; i < 256; i++) {
            if (i < n_col) {
                const unsigned char *e = &buf[pal_off + 4 * i];
                pal[i] = (e[0] == e[1] && e[1] == e[2]) ? e[0] : luma(e[0], e[1], e[2]);
            } else {
                pal[i] = 0;
            }
        }
        identity = true;
        for (uint32_t i = 0; i < 256; i++) identity = identity && pal[i] == i;
    }
    rows = h;
    cols = w;
    pixels.resize((size_t)w * (size_t)h);
    for (int y = 0; y < h; y++) {
        const unsigned char *src = &buf[data_off + row_bytes * (size_t)(top_down ? y : h - 1 - y)];
        uint8_t *dst = &pixels[(size_t)y * (size_t)w];
        if (bpp == 8 && identity) {
            std::memcpy(dst, src, (size_t)w);
        } else if (bpp == 8) {
            for (int x = 0; x < w; x++) dst[x] = pal[src[x]];
        } else {
            const int step = bpp / 8;
            for (int x = 0; x < w; x++) dst[x] = luma(src[x * step], src[x * step + 1], src[x * step + 2]);
        }
    }
    return true;
}

bool ReadCalibrationYaml(const std::string &path, Calibration &calib)
{
    std::vector<unsigned char> raw;
    if (!read_file(path, raw)) return false;                         // (a regular file of bounded size, like the images)
    const std::string text(raw.begin(), raw.end());
    struct Want { const char *key; double *dst; int n; } wants[] = {
        {"CamMat", calib.CamMat, 9}, {"ProMat", calib.ProMat, 9}, {"R", calib.R, 9}, {"T", calib.T, 3}};
    for (const Want &w : wants) {
        // a top-level key starts a line: "<key>: !!opencv-matrix"
        size_t pos = std::string::npos, from = 0;
        const std::string key = std::string(w.key) + ":";
        while ((from = text.find(key, from)) != std::string::npos) {
            if (from == 0 || text[from - 1] == '\n') { pos = from; break; }
            from += key.size();
        }
        if (pos == std::string::npos) return false;
        const size_t d = text.find("data:", pos);
        const size_t lb = d == std::string::npos ? d : text.find('[', d);
        const size_t rb = lb == std::string::npos ? lb : text.find(']', lb);
        if (rb == std::string::npos) return false;
        std::string list = text.substr(lb + 1, rb - lb - 1);
        for (char &c : list)
            if (c == ',' || c == '\n' || c == '\r') c = ' ';
        std::stringstream ls(list);
        std::string tok;
        int n = 0;
        while (ls >> tok) {
            if (n >= w.n) return false;
            char *end = nullptr;
            const double v = std::strtod(tok.c_str(), &end);      // "0." and "1.2e+003" both parse
            if (end == tok.c_str()) return false;
            w.dst[n++] = v;
        }
        if (n != w.n) return false;
    }
    return true;
}

CSensor::CSensor(const StaticParameters &sp, int dynaFrameMaxNum) : m_sp(sp), m_dynaMax(dynaFrameMaxNum) {}
CSensor::~CSensor() {}

bool CSensor::InitSensor(const std::string &groupDataPath)
{
    m_groupDataPath = groupDataPath;
    if (!m_groupDataPath.empty() && m_groupDataPath[m_groupDataPath.size() - 1] != '/') m_groupDataPath += '/';
    m_iFramePath = "iFrame/";
    m_cFramePath = "cFrame/";
    m_vGrayName = "vGrayCam";
    m_vPhaseName = "vPhaseCam";
    m_dynaName = "dynaCam";
    m_dataFileSuffix = ".bmp";
    return true;
}

bool CSensor::CloseSensor() { return UnloadDatas(); }

bool CSensor::UnloadDatas()
{
    m_dataMats.clear();
    m_rows.clear();
    m_cols.clear();
    m_dataNum = 0;
    m_nowNum = 0;
    return true;
}

bool CSensor::LoadDatas(int groupNum)
{
    if (!m_dataMats.empty()) UnloadDatas();
    std::string filePath, fileName;
    int n = 0;
    if (groupNum == 0) {
        n = m_sp.GRAY_V_NUMDIGIT * 2;
        filePath = m_groupDataPath + m_iFramePath;
        fileName = m_vGrayName;
    } else if (groupNum == 1) {
        n = m_sp.PHASE_NUMDIGIT;
        filePath = m_groupDataPath + m_iFramePath;
        fileName = m_vPhaseName;
    } else if (groupNum == 2) {
        n = m_dynaMax;
        filePath = m_groupDataPath + m_cFramePath;
        fileName = m_dynaName;
    } else {
        return false;
    }
    m_dataMats.resize((size_t)n);
    m_rows.assign((size_t)n, 0);
    m_cols.assign((size_t)n, 0);
    bool ok = true;
    for (int i = 0; i < n; i++) {
        std::ostringstream name;
        name << filePath << fileName << i << m_dataFileSuffix;
        std::ostringstream pgm;                                   // same name as a PGM: data sets converted from the BMPs
        pgm << filePath << fileName << i << ".pgm";
        if (!ReadBmpGray(name.str(), m_dataMats[(size_t)i], m_rows[(size_t)i], m_cols[(size_t)i]) &&
            !ReadPgmGray(pgm.str(), m_dataMats[(size_t)i], m_rows[(size_t)i], m_cols[(size_t)i])) {
            // the reference reports and carries on with an empty Mat (R/CSensorV.cpp:122-129); here the load fails
            m_err = "CSensor::LoadPatterns::<Read>, imread error: " + name.str();
            ok = false;
        }
    }
    m_dataNum = n;
    m_nowNum = 0;
    return ok;
}

bool CSensor::SetProPicture(int nowNum)
{
    if (nowNum < 0 || nowNum >= m_dataNum) return false;
    m_nowNum = nowNum;
    return true;
}

Image8 CSensor::GetCamPicture() const
{
    Image8 im;
    if (m_nowNum < 0 || (size_t)m_nowNum >= m_dataMats.size() || m_dataMats[(size_t)m_nowNum].empty()) return im;
    im.data = m_dataMats[(size_t)m_nowNum].data();
    im.rows = m_rows[(size_t)m_nowNum];
    im.cols = m_cols[(size_t)m_nowNum];
    im.step = (size_t)im.cols;
    im.on_device = false;
    return im;
}

// R/CCalculation.cpp:208-320 with the sensor's group-2 images
int CCalculation::CalculateOther(CSensor &sensor, const std::string &pointCloudPrefix, int recoWindowSize)
{
    if (!m_ctx || !m_done) return 0;
    if (!sensor.LoadDatas(2) && sensor.DataNum() == 0) return 0;
    if (!sensor.SetProPicture(0) || !StripRegression0(sensor.GetCamPicture(), recoWindowSize)) return 0;
    // The reference has every dynaCam image in memory before it walks them (LoadDatas, R/CSensorV.cpp:60-133): the images travel to
    // the device kChunk at a time in ONE transfer (slx_track_stage_frames), and each frame then runs from the device copy -- a
    // frame's point cloud is written before the next frame starts, as in the reference's loop (R/CCalculation.cpp:222-317).
    const int kChunk = 8;                                        // 8 images per transfer measured best (tools/track_bench.py --host --batch k --in-place: 46.5 us per 1920x1200 frame; 16-64: 61)
    int done = 0, frameNum = 1;
    const int total = sensor.DataNum();
    while (frameNum < total) {
        std::vector<Image8> cams;
        for (int f = frameNum; f < total && (int)cams.size() < kChunk; f++) {
            if (!sensor.SetProPicture(f)) break;
            const Image8 cam = sensor.GetCamPicture();
            if (cam.empty() || cam.rows != m_sp.CAMERA_RESROW || cam.cols != m_sp.CAMERA_RESLINE) break;   // fewer images on disk than DYNAFRAME_MAXNUM
            cams.push_back(cam);
        }
        if (cams.empty()) break;
        uint8_t *slab = nullptr;
        size_t stride = 0, istride = 0;
        const uint8_t *dev = nullptr;
        if (slx_track_frames_buffer(m_ctx, (int)cams.size(), &slab, &stride, &istride) != SLX_OK) break;
        for (size_t f = 0; f < cams.size(); f++)
            for (int r = 0; r < cams[f].rows; r++)
                std::memcpy(slab + f * istride + (size_t)r * stride, cams[f].data + (size_t)r * cams[f].step, (size_t)cams[f].cols);
        if (slx_track_stage_frames(m_ctx, slab, stride, istride, (int)cams.size(), &dev) != SLX_OK) break;
        bool ok = true;
        for (size_t f = 0; f < cams.size() && ok; f++) {
            Image8 on_dev;
            on_dev.data = const_cast<uint8_t *>(dev + f * istride);
            on_dev.rows = cams[f].rows;
            on_dev.cols = cams[f].cols;
            on_dev.step = stride;
            on_dev.on_device = true;
            ok = CalculateOtherFrame(frameNum, on_dev);
            if (!ok) break;
            std::ostringstream name;
            name << pointCloudPrefix << frameNum << ".txt";
            ok = Result(name.str(), frameNum);
            if (ok) {
                done++;
                frameNum++;
            }
        }
        if (!ok || (int)cams.size() < kChunk) break;
    }
    return done;
}

// Binary PGM: "P5" <width> <height> <maxval> <one whitespace byte> <rows top-down>; '#' starts a comment in the header.
bool ReadPgmGray(const std::string &path, std::vector<uint8_t> &pixels, int &rows, int &cols)
{
    std::vector<unsigned char> buf;
    if (!read_file(path, buf)) return false;
    if (buf.size() < 7 || buf[0] != 'P' || buf[1] != '5') return false;
    size_t pos = 2;
    long val[3] = {0, 0, 0};
    for (int k = 0; k < 3; k++) {
        for (;;) {                                               // whitespace and comments before the number
            if (pos >= buf.size()) return false;
            const unsigned char ch = buf[pos];
            if (ch == '#') {
                while (pos < buf.size() && buf[pos] != '\n') pos++;
            } else if (ch == ' ' || ch == '\t' || ch == '\r' || ch == '\n') {
                pos++;
            } else {
                break;
            }
        }
        if (buf[pos] < '0' || buf[pos] > '9') return false;
        long v = 0;
        while (pos < buf.size() && buf[pos] >= '0' && buf[pos] <= '9') {
            v = v * 10 + (buf[pos] - '0');
            if (v > (1l << 30)) return false;
            pos++;
        }
        val[k] = v;
    }
    if (pos >= buf.size()) return false;
    pos++;                                                       // the single whitespace byte after maxval
    const long w = val[0], h = val[1], maxval = val[2];
    if (w <= 0 || h <= 0 || maxval <= 0 || maxval > 255) return false;   // 16-bit PGMs are not 8-bit fringe images
    if (pos + (size_t)w * (size_t)h > buf.size()) return false;
    rows = (int)h;
    cols = (int)w;
    pixels.assign(buf.begin() + (long)pos, buf.begin() + (long)pos + w * h);
    return true;
}

namespace {

constexpr size_t kNumberChars = 24;        // "-1.23457e-308" is 13; a non-finite value printed by snprintf stays far below this too
constexpr size_t kLineChars = 3 * kNumberChars + 4;

// One double as printf("%.6g") / `ostream << double` prints it, for the magnitudes a point cloud holds (1e-5 <= |v| < 1e15, and
// zero), by exact integer arithmetic: the six significant digits are round-half-even of the EXACT binary value scaled by a power
// of ten (a 128-bit product shifted, or a 64-bit quotient), which is what a correctly rounding printf computes.  Returns the
// number of characters written, or 0 for a value outside that range (denormals, huge, tiny, non-finite): the caller falls back
// to std::to_chars.  ~4 x faster than to_chars(general, 6) -- formatting IS the cost of CCalculation::Result.
struct Digits3 {
    char t[1000][3];
    constexpr Digits3() : t()
    {
        for (int i = 0; i < 1000; i++) {
            t[i][0] = (char)('0' + i / 100);
            t[i][1] = (char)('0' + i / 10 % 10);
            t[i][2] = (char)('0' + i % 10);
        }
    }
};
static constexpr Digits3 kDigits3{};

static inline int fmt_g6_fast(double v, char *out, bool msvc)
{
    uint64_t bits;
    std::memcpy(&bits, &v, sizeof bits);
    char *o = out;
    if (bits >> 63) *o++ = '-';
    const uint64_t mag = bits & 0x7fffffffffffffffull;
    if (mag == 0) { *o++ = '0'; return (int)(o - out); }
    const int bexp = (int)(mag >> 52);
    // 1e-5 > 2^-17, 1e15 < 2^50: binary exponents outside [-17, 49] never belong to the range
    const int k = bexp - 1023;
    if (k < -17 || k > 49) return 0;
    const double a = v < 0 ? -v : v;
    if (!(a >= 1e-5 && a < 1e15)) return 0;
    const uint64_t m = (mag & 0x000fffffffffffffull) | 0x0010000000000000ull;     // a = m * 2^(k - 52), exactly
    const int s = 52 - k;                                                          // > 0 in this range: a = m / 2^s
    static const uint64_t kPow10[20] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull, 100000000ull, 1000000000ull,
                                        10000000000ull, 100000000000ull, 1000000000000ull, 10000000000000ull, 100000000000000ull,
                                        1000000000000000ull, 10000000000000000ull, 100000000000000000ull, 1000000000000000000ull,
                                        10000000000000000000ull};
    int X = (k * 1233) >> 12;                                                      // floor(k log10 2) for k >= 0, within one of floor(log10 a) always
    if (k < 0) X = -(((-k) * 1233 + 4095) >> 12);
    uint64_t q;
    for (;;) {                                                                     // at most two corrections of the estimate
        const int p = 5 - X;                                                       // digits = a * 10^p, wanted in [10^5, 10^6)
        bool up;                                                                   // round the integer part q up?
        if (p >= 0) {
            const unsigned __int128 T = (unsigned __int128)m * kPow10[p];          // p <= 10: T < 2^87
            q = (uint64_t)(T >> s);
            if (q < 100000ull) { X--; continue; }
            if (q >= 1000000ull) { X++; continue; }
            const unsigned __int128 rem = T & ((((unsigned __int128)1) << s) - 1), half = ((unsigned __int128)1) << (s - 1);
            up = rem > half || (rem == half && (q & 1ull));
        } else {
            const uint64_t D = kPow10[-p] << s;                                    // -p <= 9, s <= 33: < 2^63
            q = m / D;
            if (q < 100000ull) { X--; continue; }
            if (q >= 1000000ull) { X++; continue; }
            const uint64_t rem = m - q * D;
            up = 2 * rem > D || (2 * rem == D && (q & 1ull));
        }
        if (up && ++q == 1000000ull) { q = 100000ull; X++; }
        break;
    }
    char d[6];
    {
        const uint32_t q32 = (uint32_t)q, hi = q32 / 1000u, lo = q32 - hi * 1000u;  // two table look-ups instead of six divisions
        std::memcpy(d, kDigits3.t[hi], 3);
        std::memcpy(d + 3, kDigits3.t[lo], 3);
    }
    int nd = 6;
    while (nd > 1 && d[nd - 1] == '0') nd--;                                       // %g drops trailing zeros
    if (X < -4 || X >= 6) {                                                        // scientific
        *o++ = d[0];
        if (nd > 1) { *o++ = '.'; for (int i = 1; i < nd; i++) *o++ = d[i]; }
        *o++ = 'e';
        int e = X;
        if (e < 0) { *o++ = '-'; e = -e; } else *o++ = '+';
        if (msvc) *o++ = '0';                                                      // that runtime prints three exponent digits: 5e-005
        *o++ = (char)('0' + e / 10);
        *o++ = (char)('0' + e % 10);
    } else if (X >= 0) {
        for (int i = 0; i <= X; i++) *o++ = i < nd ? d[i] : '0';
        if (nd > X + 1) { *o++ = '.'; for (int i = X + 1; i < nd; i++) *o++ = d[i]; }
    } else {
        *o++ = '0'; *o++ = '.';
        for (int i = 0; i < -X - 1; i++) *o++ = '0';
        for (int i = 0; i < nd; i++) *o++ = d[i];
    }
    return (int)(o - out);
}

// one number as `ostream << double` prints it; returns the end.  msvc: the dialect of the reference AS BUILT (MSVC 2013 runtime): at
// least three exponent digits, and that runtime's spellings of the non-finite values at precision 6 ("1.#INF", "-1.#IND", "1.#QNAN")
char *put_number(char *out, double v, bool msvc)
{
    if (const int n = fmt_g6_fast(v, out, msvc)) return out + n;
    if (std::isfinite(v)) {
        char *end = std::to_chars(out, out + kNumberChars, v, std::chars_format::general, 6).ptr;
        if (msvc) {
            char *e = end;
            while (e > out && e[-1] >= '0' && e[-1] <= '9') e--;     // the digits behind "e+" / "e-", when there is an exponent
            if (e - out >= 2 && (e[-1] == '+' || e[-1] == '-') && e[-2] == 'e' && end - e == 2) {
                e[2] = e[1];
                e[1] = e[0];
                e[0] = '0';
                end++;
            }
        }
        return end;
    }
    if (!msvc) return out + std::snprintf(out, kNumberChars, "%g", v);   // nan / inf with their signs: the C library's own spelling
    const char *w = std::isinf(v) ? (v < 0 ? "-1.#INF" : "1.#INF") : (std::signbit(v) ? "-1.#IND" : "1.#QNAN");
    const size_t n = std::strlen(w);
    std::memcpy(out, w, n);
    return out + n;
}

size_t format_points(const double *xyz, size_t n, char *out, bool msvc)
{
    char *o = out;
    for (size_t i = 0; i < n; i++) {
        o = put_number(o, xyz[3 * i + 0], msvc);
        *o++ = ' ';
        o = put_number(o, xyz[3 * i + 1], msvc);
        *o++ = ' ';
        o = put_number(o, xyz[3 * i + 2], msvc);
        if (msvc) *o++ = '\r';                                       // a text-mode stream of that runtime turns endl into CR LF
        *o++ = '\n';
    }
    return (size_t)(o - out);
}

}  // namespace

bool WritePointCloudText(const std::string &path, const double *xyz, size_t n_points, int dialect)
{
    if ((n_points && !xyz) || (dialect != SLX_TEXT_LIBSTDCXX && dialect != SLX_TEXT_MSVC2013)) return false;
    const bool msvc = dialect == SLX_TEXT_MSVC2013;
    std::FILE *f = std::fopen(path.c_str(), "wb");                  // the bytes as formatted: the CR of the MSVC dialect is written out, never added by a stream
    if (!f) return false;
    // rounds of at most `threads` blocks of 64 Ki points: ~6 MB of text per block, so a 12-million-point cloud never holds more
    // than ~100 MB of it; formatting is ~80 ns per number, the write of a round a few milliseconds
    constexpr size_t kBlock = 65536;
    const unsigned hw = std::thread::hardware_concurrency();
    const size_t threads = std::max<size_t>(1, std::min<size_t>({hw ? hw : 1u, 16u, (n_points + kBlock - 1) / kBlock}));
    // two sets of buffers: while one round's text is written out (a thread of its own), the next round is formatted
    std::vector<std::vector<char>> text[2] = {std::vector<std::vector<char>>(threads), std::vector<std::vector<char>>(threads)};
    std::vector<size_t> len[2] = {std::vector<size_t>(threads, 0), std::vector<size_t>(threads, 0)};
    std::atomic<bool> ok{true};
    std::thread writer;
    int cur = 0;
    for (size_t first = 0; first < n_points && ok; first += threads * kBlock, cur ^= 1) {
        auto work = [&, cur](size_t t) {
            const size_t a = std::min(n_points, first + t * kBlock), b = std::min(n_points, a + kBlock);
            text[cur][t].resize((b - a) * kLineChars);
            len[cur][t] = format_points(xyz + 3 * a, b - a, text[cur][t].data(), msvc);
        };
        std::vector<std::thread> pool;
        for (size_t t = 1; t < threads; t++) pool.emplace_back(work, t);
        work(0);
        for (std::thread &th : pool) th.join();
        if (writer.joinable()) writer.join();                     // the round before: its buffers (the other set) are free again after this
        writer = std::thread([&, cur] {
            for (size_t t = 0; t < threads && ok; t++)
                if (len[cur][t] && std::fwrite(text[cur][t].data(), 1, len[cur][t], f) != len[cur][t]) ok = false;
        });
    }
    if (writer.joinable()) writer.join();
    return (std::fclose(f) == 0) && ok;
}

}  // namespace slx

// (No exception crosses the C boundary: a reader that runs out of memory on a forged size answers like a missing file.)
template <class F> static int guarded(F &&f)
{
    try {
        return f();
    } catch (const std::bad_alloc &) {
        return SLX_ERR_OUT_OF_MEMORY;
    } catch (...) {
        return SLX_ERR_UNAVAILABLE;
    }
}

// ---- plain-C access to the file readers (declared in include/slx.h) ----
extern "C" {

int slx_read_bmp_gray(const char *path, uint8_t *pixels, size_t capacity, int *rows, int *cols)
{
    if (!path || !rows || !cols) return SLX_ERR_INVALID_ARG;
    return guarded([&]() -> int {
        std::vector<uint8_t> px;
        int r = 0, c = 0;
        if (!slx::ReadBmpGray(path, px, r, c)) return SLX_ERR_UNAVAILABLE;
        *rows = r;
        *cols = c;
        if (!pixels) return SLX_OK;                               // size query
        if (capacity < px.size()) return SLX_ERR_INVALID_ARG;
        std::memcpy(pixels, px.data(), px.size());
        return SLX_OK;
    });
}

int slx_read_pgm_gray(const char *path, uint8_t *pixels, size_t capacity, int *rows, int *cols)
{
    if (!path || !rows || !cols) return SLX_ERR_INVALID_ARG;
    return guarded([&]() -> int {
        std::vector<uint8_t> px;
        int r = 0, c = 0;
        if (!slx::ReadPgmGray(path, px, r, c)) return SLX_ERR_UNAVAILABLE;
        *rows = r;
        *cols = c;
        if (!pixels) return SLX_OK;                               // size query
        if (capacity < px.size()) return SLX_ERR_INVALID_ARG;
        std::memcpy(pixels, px.data(), px.size());
        return SLX_OK;
    });
}

int slx_write_point_cloud_text_ex(const char *path, const double *xyz, size_t n_points, int dialect)
{
    if (!path || (n_points && !xyz) || (dialect != SLX_TEXT_LIBSTDCXX && dialect != SLX_TEXT_MSVC2013)) return SLX_ERR_INVALID_ARG;
    return guarded([&]() -> int { return slx::WritePointCloudText(path, xyz, n_points, dialect) ? SLX_OK : SLX_ERR_UNAVAILABLE; });
}

int slx_write_point_cloud_text(const char *path, const double *xyz, size_t n_points)
{
    return slx_write_point_cloud_text_ex(path, xyz, n_points, SLX_TEXT_LIBSTDCXX);
}

int slx_read_calibration_yaml(const char *path, double cam[9], double pro[9], double rot[9], double trans[3])
{
    if (!path || !cam || !pro || !rot || !trans) return SLX_ERR_INVALID_ARG;
    return guarded([&]() -> int {
        slx::Calibration c;
        if (!slx::ReadCalibrationYaml(path, c)) return SLX_ERR_UNAVAILABLE;
        std::memcpy(cam, c.CamMat, sizeof c.CamMat);
        std::memcpy(pro, c.ProMat, sizeof c.ProMat);
        std::memcpy(rot, c.R, sizeof c.R);
        std::memcpy(trans, c.T, sizeof c.T);
        return SLX_OK;
    });
}

}  // extern "C"
