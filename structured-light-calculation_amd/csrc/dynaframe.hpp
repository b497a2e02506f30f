// dynaframe.hpp -- host-side mirror of the reference's decoder classes over the C ABI.
//
// Same class names, method names, argument meaning and bool error behaviour as
//   CDecodePhase   R/CDecodePhase.h:12-39
//   CDecodeGray    R/CDecodeGray.h:18-53
//   CCalculation   R/CCalculation.h:10-95   (static path: Init / CalculateFirst / Result)
// with raw 8-bit image views in place of cv::Mat (OpenCV is not a dependency) and run-time
// sizes in place of the compile-time constants of R/StaticParameters.cpp.  A DynaFrame host
// loop (R/CCalculation.cpp:525-559) ports by replacing `Mat` arguments with slx::Image8.
// R/ = DynaFrame/DynaFrame/ of the reference repository.
#ifndef SLX_DYNAFRAME_HPP
#define SLX_DYNAFRAME_HPP

#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "slx.h"

namespace slx {

// An 8-bit single-channel image (CV_8UC1), host or device memory.
struct Image8 {
    const uint8_t *data = nullptr;
    int rows = 0, cols = 0;
    size_t step = 0;           // bytes per row
    bool on_device = false;
    bool empty() const { return data == nullptr || rows <= 0 || cols <= 0; }
};

// R/StaticParameters.cpp:34, :38: the compiled-in defaults of the dynamic-frame loop (pinned, with the struct below, against the
// reference's own compiled translation unit: tests/test_oracle_golden.py::test_reference_static_parameters)
constexpr int kRecoWindowSize = 21;      // RECO_WINDOW_SIZE
constexpr int kDynaFrameMaxNum = 100;    // DYNAFRAME_MAXNUM

// Run-time stand-in for R/StaticParameters.cpp:4-35.
struct StaticParameters {
    int PROJECTOR_RESLINE = 1280, PROJECTOR_RESROW = 800;
    int CAMERA_RESLINE = 1280, CAMERA_RESROW = 1024;
    int GRAY_V_NUMDIGIT = 6, PHASE_NUMDIGIT = 4;
    double FOV_MIN_DISTANCE = 10, FOV_MAX_DISTANCE = 100;
};

// Phase-shift decoder: N grey images in, per-pixel projector offset (f64) out.
class CDecodePhase {
public:
    explicit CDecodePhase(const StaticParameters &sp = StaticParameters());
    ~CDecodePhase();
    CDecodePhase(const CDecodePhase &) = delete;
    CDecodePhase &operator=(const CDecodePhase &) = delete;

    bool SetNumMat(int numMat, int pixperiod);     // R/CDecodePhase.cpp:119
    bool SetMat(int num, const Image8 &pic);       // R/CDecodePhase.cpp:107
    bool Decode();                                 // R/CDecodePhase.cpp:83
    std::vector<double> GetResult();               // R/CDecodePhase.cpp:99 (deep copy, CV_64FC1)
    bool GetResult(double *dst, size_t n_elems, bool to_device = false);
    const std::string &LastError() const { return m_err; }

private:
    bool DeleteSpace();
    StaticParameters m_sp;
    int m_numMat = 0, m_pixPeroid = 16;
    slx_ctx *m_ctx = nullptr;
    bool m_decoded = false;
    std::string m_err;
};

// Gray-code decoder: 2*G grey images (pattern, inverse) in, stripe left edge (f64) out.
class CDecodeGray {
public:
    explicit CDecodeGray(const StaticParameters &sp = StaticParameters());
    ~CDecodeGray();
    CDecodeGray(const CDecodeGray &) = delete;
    CDecodeGray &operator=(const CDecodeGray &) = delete;

    bool SetNumDigit(int numDigit, bool ver);                              // R/CDecodeGray.cpp:36
    bool SetMatFileName(std::string codeFilePath, std::string codeFileName); // R/CDecodeGray.cpp:56
    bool SetMat(int num, const Image8 &pic);                               // R/CDecodeGray.cpp:24
    bool Decode();                                                         // R/CDecodeGray.cpp:108
    std::vector<double> GetResult();                                       // R/CDecodeGray.cpp:142
    bool GetResult(double *dst, size_t n_elems, bool to_device = false);
    const std::string &LastError() const { return m_err; }

private:
    bool ReleaseSpace();
    StaticParameters m_sp;
    int m_numDigit = 0, m_grayCodeSize = 0;
    bool m_vertical = true;
    std::string m_codeFilePath, m_codeFileName;
    slx_ctx *m_ctx = nullptr;
    bool m_decoded = false;
    std::string m_err;
};

// Calibration matrices of the YAML file Init reads (keys CamMat, ProMat, R, T; R/CCalculation.cpp:128-131).
struct Calibration {
    double CamMat[9], ProMat[9], R[9], T[3];
};

class CSensor;

// Static reconstruction of frame 0 (Gray + phase decode, merge, triangulation) and the dynamic frames after it.
class CCalculation {
public:
    CCalculation();
    ~CCalculation();
    CCalculation(const CCalculation &) = delete;
    CCalculation &operator=(const CCalculation &) = delete;

    // R/CCalculation.cpp:77.  The sensor of the reference is replaced by SetSensorFrame.
    bool Init(const StaticParameters &sp, const Calibration &calib,
              const std::string &codeFilePath = "Patterns/", const std::string &codeFileName = "vGrayCode.txt");
    // Role of CSensor::LoadDatas/SetProPicture/GetCamPicture (R/CSensorV.cpp:60-179):
    // groupNum 0 = vGray images (2*G of them), 1 = vPhase images (N of them).
    bool SetSensorFrame(int groupNum, int idx, const Image8 &pic);
    bool CalculateFirst();                          // R/CCalculation.cpp:171
    // R/CCalculation.cpp:323: text point cloud "x y z\n" of the depths inside the FOV, column outer / row inner,
    // default ostream formatting.  Only frame 0 (the static reconstruction) exists here.
    bool Result(std::string fileName, int i = 0);
    // Which bytes `ostream << double` + endl mean (enum slx_text_dialect, include/slx.h): SLX_TEXT_LIBSTDCXX (default: "5e-05", LF -- the
    // reference's loop compiled on Linux) or SLX_TEXT_MSVC2013 (the reference as built: "5e-005", CR LF).  False for another value.
    bool SetTextDialect(int dialect);
    // Dynamic frames, R/CCalculation.cpp:208-320.  StripRegression(0) is what CalculateFirst ends with in the reference
    // (:203); here the camera image comes in explicitly.  CalculateOtherFrame(fN, image) = StripRegression(fN) +
    // FillOtherDeltaProU(fN) + FillCoordinate(fN); afterwards GetZ/GetX/GetY/GetProjectorU/GetDeltaZ/Result refer to frame fN.
    bool StripRegression0(const Image8 &dynaCam0, int recoWindowSize = kRecoWindowSize);
    bool CalculateOtherFrame(int fN, const Image8 &dynaCam);
    // n consecutive frames fN0 .. fN0 + n - 1 in one call (slx_track_next_batch: the host images ride ONE transfer, the frames run
    // back to back); afterwards the getters refer to the last of them.  deltaZ, when given, receives every frame's deltaZ map
    // (n x rows x cols, host memory) -- what the reference keeps in m_deltaZ[fN] (R/CCalculation.cpp:772-775).
    bool CalculateOtherFrames(int fN0, const Image8 *dynaCams, int n, std::vector<double> *deltaZ = nullptr);
    // The whole loop of CalculateOther over the sensor's dynaCam images (group 2): frame fN's point cloud goes to
    // <pointCloudPrefix><fN>.txt like m_pcSucceedName (R/CCalculation.cpp:309-314).  Returns the number of frames done.
    int CalculateOther(CSensor &sensor, const std::string &pointCloudPrefix, int recoWindowSize = kRecoWindowSize);
    std::vector<double> GetDeltaZ();
    std::vector<double> GetPointCloud();            // the same points, packed x y z
    // m_zMat[0], m_xMat[0], m_yMat[0], m_ProjectorU[0] (CV_64FC1, rows x cols)
    std::vector<double> GetZ(), GetX(), GetY(), GetProjectorU();
    const std::string &LastError() const { return m_err; }

private:
    bool ReleaseSpace();
    std::vector<double> Fetch(int which);
    StaticParameters m_sp;
    slx_ctx *m_ctx = nullptr;
    bool m_done = false;
    int m_frame = 0;               // the frame GetZ etc. refer to
    int m_textDialect = SLX_TEXT_LIBSTDCXX;
    std::string m_err;
};

// Reads a Gray-code table file ("bin gray" per line, R/Patterns/vGrayCode.txt) into lut[gray] = bin
// the way R/CDecodeGray.cpp:113-125 does.  Returns false when the file cannot be opened.
bool ReadGrayCodeFile(const std::string &path, int grayCodeSize, std::vector<int16_t> &lut);

}  // namespace slx

#endif
