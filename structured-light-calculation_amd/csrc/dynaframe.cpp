// dynaframe.cpp -- see dynaframe.hpp.  Thin C++ over the C ABI; no arithmetic lives here.
#include "dynaframe.hpp"

#include <cstdio>
#include <cstring>
#include <fstream>

#include <sys/stat.h>

namespace slx {

namespace {

slx_config base_config(const StaticParameters &sp)
{
    slx_config c;
    std::memset(&c, 0, sizeof c);
    c.width = sp.CAMERA_RESLINE;
    c.height = sp.CAMERA_RESROW;
    c.device = -1;
    return c;
}

bool push_frame(slx_ctx *ctx, int group, int num, const Image8 &pic, int rows, int cols, std::string &err)
{
    if (pic.empty() || pic.rows != rows || pic.cols != cols) {
        err = "image is empty or has the wrong size";
        return false;
    }
    int rc = slx_set_frame(ctx, group, num, pic.data, pic.step, pic.on_device ? SLX_MEM_DEVICE : SLX_MEM_HOST);
    if (rc != SLX_OK) {
        err = slx_last_error(ctx);
        return false;
    }
    return true;
}

std::vector<double> fetch(slx_ctx *ctx, int which, size_t n, std::string &err)
{
    std::vector<double> r(n);
    if (slx_get_output(ctx, which, r.data(), n * sizeof(double), SLX_MEM_HOST) != SLX_OK) {
        err = slx_last_error(ctx);
        r.clear();
    }
    return r;
}

}  // namespace

bool ReadGrayCodeFile(const std::string &path, int grayCodeSize, std::vector<int16_t> &lut)
{
    struct stat st;
    if (stat(path.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) return false;   // (an ifstream "opens" a directory on Linux; the reference's open fails on one)
    std::ifstream codeFile(path.c_str(), std::ios::in);
    if (!codeFile) return false;
    lut.assign((size_t)grayCodeSize, 0);
    for (int i = 0; i < grayCodeSize; i++) {
        int binCode = 0, grayCode = 0;
        codeFile >> binCode >> grayCode;
        if (grayCode >= 0 && grayCode < grayCodeSize) lut[(size_t)grayCode] = (int16_t)binCode;
    }
    return true;
}

// ---------------------------------------------------------------- CDecodePhase
CDecodePhase::CDecodePhase(const StaticParameters &sp) : m_sp(sp) {}
CDecodePhase::~CDecodePhase() { DeleteSpace(); }

bool CDecodePhase::DeleteSpace()
{
    if (m_ctx) slx_destroy(m_ctx);
    m_ctx = nullptr;
    m_decoded = false;
    return true;
}

bool CDecodePhase::SetNumMat(int numMat, int pixperiod)
{
    if (numMat <= 0) return false;
    m_numMat = numMat;
    m_pixPeroid = pixperiod;
    DeleteSpace();
    slx_config c = base_config(m_sp);
    c.mode = SLX_MODE_PHASE_ONLY;
    c.n_freq = 1;
    c.n_steps = numMat;
    c.period[0] = pixperiod;
    if (slx_create(&c, &m_ctx) != SLX_OK) {
        m_err = slx_last_error(nullptr);
        m_ctx = nullptr;
        return false;
    }
    return true;
}

bool CDecodePhase::SetMat(int num, const Image8 &pic)
{
    if (!m_ctx) {
        m_err = "CDecodePhase.SetMat->grePicture Space is not allocated.";
        return false;
    }
    return push_frame(m_ctx, SLX_GROUP_PHASE, num, pic, m_sp.CAMERA_RESROW, m_sp.CAMERA_RESLINE, m_err);
}

bool CDecodePhase::Decode()
{
    // synchronous, like the reference's Decode(): borrowed device images may be rewritten as soon as it returns
    if (!m_ctx || slx_decode(m_ctx, nullptr) != SLX_OK || slx_synchronize(m_ctx) != SLX_OK) {
        m_err = m_ctx ? slx_last_error(m_ctx) : "CDecodePhase.Decode()->CountResult fault";
        return false;
    }
    m_decoded = true;
    return true;
}

std::vector<double> CDecodePhase::GetResult()
{
    if (!m_ctx || !m_decoded) return std::vector<double>();
    return fetch(m_ctx, SLX_OUT_PIX, (size_t)m_sp.CAMERA_RESROW * m_sp.CAMERA_RESLINE, m_err);
}

bool CDecodePhase::GetResult(double *dst, size_t n_elems, bool to_device)
{
    if (!m_ctx || !m_decoded) return false;
    if (slx_get_output(m_ctx, SLX_OUT_PIX, dst, n_elems * sizeof(double), to_device ? SLX_MEM_DEVICE : SLX_MEM_HOST) != SLX_OK) {
        m_err = slx_last_error(m_ctx);
        return false;
    }
    return true;
}

// ----------------------------------------------------------------- CDecodeGray
CDecodeGray::CDecodeGray(const StaticParameters &sp) : m_sp(sp) {}
CDecodeGray::~CDecodeGray() { ReleaseSpace(); }

bool CDecodeGray::ReleaseSpace()
{
    if (m_ctx) slx_destroy(m_ctx);
    m_ctx = nullptr;
    m_decoded = false;
    return true;
}

bool CDecodeGray::SetNumDigit(int numDigit, bool ver)
{
    if ((numDigit <= 0) || (numDigit > 16)) return false;
    m_numDigit = numDigit;
    m_grayCodeSize = 1 << numDigit;
    m_vertical = ver;
    ReleaseSpace();
    slx_config c = base_config(m_sp);
    c.mode = SLX_MODE_GRAY_ONLY;
    c.gray_bits = numDigit;
    // R/CDecodeGray.cpp:182-185, integer division
    c.gray_stripe = (ver ? m_sp.PROJECTOR_RESLINE : m_sp.PROJECTOR_RESROW) / m_grayCodeSize;
    std::vector<int16_t> zeros((size_t)m_grayCodeSize, 0);   // the table itself is read in Decode()
    c.gray_lut = zeros.data();
    if (slx_create(&c, &m_ctx) != SLX_OK) {
        m_err = slx_last_error(nullptr);
        m_ctx = nullptr;
        return false;
    }
    return true;
}

bool CDecodeGray::SetMatFileName(std::string codeFilePath, std::string codeFileName)
{
    m_codeFilePath = codeFilePath;
    m_codeFileName = codeFileName;
    return true;
}

bool CDecodeGray::SetMat(int num, const Image8 &pic)
{
    if (!m_ctx) {
        m_err = "CDecodeGray.SetMat->grePicture Space is not allocated.";
        return false;
    }
    return push_frame(m_ctx, SLX_GROUP_GRAY, num, pic, m_sp.CAMERA_RESROW, m_sp.CAMERA_RESLINE, m_err);
}

bool CDecodeGray::Decode()
{
    if (!m_ctx) return false;
    std::vector<int16_t> lut;
    if (!ReadGrayCodeFile(m_codeFilePath + m_codeFileName, m_grayCodeSize, lut)) {
        m_err = "Gray Decode->Open file error.";
        return false;
    }
    if (slx_set_gray_lut(m_ctx, lut.data(), lut.size()) != SLX_OK || slx_decode(m_ctx, nullptr) != SLX_OK || slx_synchronize(m_ctx) != SLX_OK) {
        m_err = slx_last_error(m_ctx);
        return false;
    }
    m_decoded = true;
    return true;
}

std::vector<double> CDecodeGray::GetResult()
{
    if (!m_ctx || !m_decoded) return std::vector<double>();
    return fetch(m_ctx, SLX_OUT_GRAY, (size_t)m_sp.CAMERA_RESROW * m_sp.CAMERA_RESLINE, m_err);
}

bool CDecodeGray::GetResult(double *dst, size_t n_elems, bool to_device)
{
    if (!m_ctx || !m_decoded) return false;
    if (slx_get_output(m_ctx, SLX_OUT_GRAY, dst, n_elems * sizeof(double), to_device ? SLX_MEM_DEVICE : SLX_MEM_HOST) != SLX_OK) {
        m_err = slx_last_error(m_ctx);
        return false;
    }
    return true;
}

// ---------------------------------------------------------------- CCalculation
CCalculation::CCalculation() {}
CCalculation::~CCalculation() { ReleaseSpace(); }

bool CCalculation::ReleaseSpace()
{
    if (m_ctx) slx_destroy(m_ctx);
    m_ctx = nullptr;
    m_done = false;
    return true;
}

bool CCalculation::Init(const StaticParameters &sp, const Calibration &calib,
                        const std::string &codeFilePath, const std::string &codeFileName)
{
    if (m_ctx != nullptr) return false;             // R/CCalculation.cpp:80-83
    m_sp = sp;
    std::vector<int16_t> lut;
    if (!ReadGrayCodeFile(codeFilePath + codeFileName, 1 << sp.GRAY_V_NUMDIGIT, lut)) {
        m_err = "Gray Decode->Open file error.";
        return false;
    }
    slx_config c = base_config(sp);
    c.mode = SLX_MODE_GRAY_PHASE;
    c.n_freq = 1;
    c.n_steps = sp.PHASE_NUMDIGIT;
    // R/CCalculation.cpp:550: `1 << GRAY_V_NUMDIGIT - 1` parses as 1 << (G-1)
    c.period[0] = sp.PROJECTOR_RESLINE / (1 << (sp.GRAY_V_NUMDIGIT - 1));
    c.gray_bits = sp.GRAY_V_NUMDIGIT;
    c.gray_stripe = sp.PROJECTOR_RESLINE / (1 << sp.GRAY_V_NUMDIGIT);   // :562-563
    c.gray_lut = lut.data();
    c.fov_min = sp.FOV_MIN_DISTANCE;
    c.fov_max = sp.FOV_MAX_DISTANCE;
    std::memcpy(c.cam, calib.CamMat, sizeof c.cam);
    std::memcpy(c.pro, calib.ProMat, sizeof c.pro);
    std::memcpy(c.rot, calib.R, sizeof c.rot);
    std::memcpy(c.trans, calib.T, sizeof c.trans);
    c.aux_outputs = 1u << SLX_OUT_X | 1u << SLX_OUT_Y | 1u << SLX_OUT_U;
    if (slx_create(&c, &m_ctx) != SLX_OK) {
        m_err = slx_last_error(nullptr);
        m_ctx = nullptr;
        return false;
    }
    return true;
}

bool CCalculation::SetSensorFrame(int groupNum, int idx, const Image8 &pic)
{
    if (!m_ctx) return false;
    if (groupNum != 0 && groupNum != 1) return false;   // R/CSensorV.cpp:94-97
    return push_frame(m_ctx, groupNum == 0 ? SLX_GROUP_GRAY : SLX_GROUP_PHASE, idx, pic,
                      m_sp.CAMERA_RESROW, m_sp.CAMERA_RESLINE, m_err);
}

bool CCalculation::CalculateFirst()
{
    if (!m_ctx) return false;                           // R/CCalculation.cpp:176-181
    if (slx_decode(m_ctx, nullptr) != SLX_OK || slx_synchronize(m_ctx) != SLX_OK) {   // synchronous, like the reference's
        m_err = slx_last_error(m_ctx);
        return false;
    }
    m_done = true;
    return true;
}

std::vector<double> CCalculation::Fetch(int which)
{
    if (!m_ctx || !m_done) return std::vector<double>();
    return fetch(m_ctx, which, (size_t)m_sp.CAMERA_RESROW * m_sp.CAMERA_RESLINE, m_err);
}
std::vector<double> CCalculation::GetPointCloud()
{
    std::vector<double> pts;
    if (!m_ctx || !m_done) return pts;
    size_t n = 0;
    int rc = slx_get_point_cloud(m_ctx, nullptr, 0, &n, SLX_MEM_HOST);     // first call: the count
    if (n == 0) {
        if (rc != SLX_OK) m_err = slx_last_error(m_ctx);
        return pts;
    }
    pts.resize(n * 3);
    if (slx_get_point_cloud(m_ctx, pts.data(), n, &n, SLX_MEM_HOST) != SLX_OK) {
        m_err = slx_last_error(m_ctx);
        pts.clear();
    }
    return pts;
}

bool CCalculation::SetTextDialect(int dialect)
{
    if (dialect != SLX_TEXT_LIBSTDCXX && dialect != SLX_TEXT_MSVC2013) return false;
    m_textDialect = dialect;
    return !m_ctx || slx_set_text_dialect(m_ctx, dialect) == SLX_OK;
}

bool CCalculation::Result(std::string fileName, int i)
{
    if (i != m_frame || !m_ctx || !m_done) return false;        // only the current frame's maps exist on the device
    // the reference opens the file first and reports that failure (R/CCalculation.cpp:325-331); so does this
    // The text is formatted on the device and arrives ready to be written (slx_get_point_cloud_text); a frame with a coordinate the
    // device formatter does not take (NaN, infinity, 0 < |v| < 1e-5, |v| >= 1e15) goes through the host formatter below.
    const char *text = nullptr;
    size_t n_bytes = 0;
    (void)slx_set_text_dialect(m_ctx, m_textDialect);
    if (slx_get_point_cloud_text(m_ctx, &text, &n_bytes, nullptr) == SLX_OK) {
        std::FILE *f = std::fopen(fileName.c_str(), "wb");           // the line ends are in the text (CR LF in the MSVC dialect)
        const bool ok = f && (n_bytes == 0 || std::fwrite(text, 1, n_bytes, f) == n_bytes);
        if (!(f && std::fclose(f) == 0 && ok)) {
            m_err = "CCalculation::Result() OpenFile Error:" + fileName;
            return false;
        }
        return true;
    }
    const double *pts = nullptr;                                   // pinned memory of the context: no vector to size and zero first
    size_t n = 0;
    if (slx_get_point_cloud_view(m_ctx, &pts, &n) != SLX_OK) n = 0;      // (as before: a cloud that cannot be had is an empty file)
    if (slx_write_point_cloud_text_ex(fileName.c_str(), pts, n, m_textDialect) != SLX_OK) {
        m_err = "CCalculation::Result() OpenFile Error:" + fileName;
        return false;
    }
    return true;
}

bool CCalculation::StripRegression0(const Image8 &dynaCam0, int recoWindowSize)
{
    if (!m_ctx || !m_done) return false;
    if (dynaCam0.empty() || dynaCam0.rows != m_sp.CAMERA_RESROW || dynaCam0.cols != m_sp.CAMERA_RESLINE) {
        m_err = "image is empty or has the wrong size";
        return false;
    }
    if (slx_track_begin(m_ctx, dynaCam0.data, dynaCam0.step, dynaCam0.on_device ? SLX_MEM_DEVICE : SLX_MEM_HOST, recoWindowSize) != SLX_OK) {
        m_err = slx_last_error(m_ctx);
        return false;
    }
    return true;
}

bool CCalculation::CalculateOtherFrame(int fN, const Image8 &dynaCam)
{
    if (!m_ctx || !m_done || fN != m_frame + 1) return false;   // frames come in order: each builds on the previous one
    if (dynaCam.empty() || dynaCam.rows != m_sp.CAMERA_RESROW || dynaCam.cols != m_sp.CAMERA_RESLINE) {
        m_err = "image is empty or has the wrong size";
        return false;
    }
    if (slx_track_next(m_ctx, dynaCam.data, dynaCam.step, dynaCam.on_device ? SLX_MEM_DEVICE : SLX_MEM_HOST) != SLX_OK) {
        m_err = slx_last_error(m_ctx);
        return false;
    }
    m_frame = fN;
    return true;
}

bool CCalculation::CalculateOtherFrames(int fN0, const Image8 *dynaCams, int n, std::vector<double> *deltaZ)
{
    if (!m_ctx || !m_done || fN0 != m_frame + 1 || !dynaCams || n < 1 || n > SLX_TRACK_MAX_BATCH) return false;
    const size_t hw = (size_t)m_sp.CAMERA_RESROW * (size_t)m_sp.CAMERA_RESLINE;
    for (int f = 0; f < n; f++)
        if (dynaCams[f].empty() || dynaCams[f].rows != m_sp.CAMERA_RESROW || dynaCams[f].cols != m_sp.CAMERA_RESLINE || dynaCams[f].on_device) {
            m_err = "an image is empty, on the device or has the wrong size";
            return false;
        }
    // the images go into the context's pinned slab (the one deep copy the reference's GetCamPicture makes), then one transfer
    uint8_t *slab = nullptr;
    size_t stride = 0, istride = 0;
    if (slx_track_frames_buffer(m_ctx, n, &slab, &stride, &istride) != SLX_OK) {
        m_err = slx_last_error(m_ctx);
        return false;
    }
    for (int f = 0; f < n; f++)
        for (int r = 0; r < dynaCams[f].rows; r++)
            std::memcpy(slab + (size_t)f * istride + (size_t)r * stride, dynaCams[f].data + (size_t)r * dynaCams[f].step, (size_t)dynaCams[f].cols);
    if (deltaZ) deltaZ->resize((size_t)n * hw);
    const bool ok = slx_track_next_batch(m_ctx, slab, stride, istride, n, SLX_MEM_HOST, deltaZ ? deltaZ->data() : nullptr, SLX_MEM_HOST) == SLX_OK &&
                    slx_synchronize(m_ctx) == SLX_OK;
    if (!ok) m_err = slx_last_error(m_ctx);
    if (ok) m_frame = fN0 + n - 1;
    return ok;
}

std::vector<double> CCalculation::GetDeltaZ() { return m_frame > 0 ? Fetch(SLX_OUT_DELTAZ) : std::vector<double>(); }

std::vector<double> CCalculation::GetZ() { return Fetch(SLX_OUT_Z); }
std::vector<double> CCalculation::GetX() { return Fetch(SLX_OUT_X); }
std::vector<double> CCalculation::GetY() { return Fetch(SLX_OUT_Y); }
std::vector<double> CCalculation::GetProjectorU() { return Fetch(SLX_OUT_U); }

}  // namespace slx

// The compiled-in configuration the mirror classes default to -- the reference's R/StaticParameters.cpp -- for bindings in other
// languages and for the test that pins it to the reference's own compiled translation unit.
extern "C" int slx_reference_defaults(int *values, int capacity, int *n)
{
    const slx::StaticParameters sp;
    const int v[] = {sp.PROJECTOR_RESLINE, sp.PROJECTOR_RESROW, sp.CAMERA_RESLINE, sp.CAMERA_RESROW, sp.GRAY_V_NUMDIGIT, sp.PHASE_NUMDIGIT,
                     (int)sp.FOV_MIN_DISTANCE, (int)sp.FOV_MAX_DISTANCE, slx::kRecoWindowSize, slx::kDynaFrameMaxNum};
    const int count = (int)(sizeof v / sizeof v[0]);
    if (n) *n = count;
    if (!values || capacity < count) return SLX_ERR_INVALID_ARG;
    for (int i = 0; i < count; i++) values[i] = v[i];
    return SLX_OK;
}
