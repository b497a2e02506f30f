// sensor.hpp -- the reference's offline "virtual sensor" and its file formats, without OpenCV.
//
//   CSensor                      R/CSensorV.h:15-62, R/CSensorV.cpp:29-179: LoadDatas(groupNum) imread()s
//                                <group>/iFrame/vGrayCam<i>.bmp (group 0), vPhaseCam<i>.bmp (group 1),
//                                <group>/cFrame/dynaCam<i>.bmp (group 2) with CV_LOAD_IMAGE_GRAYSCALE
//   cv::FileStorage calibration  R/CCalculation.cpp:124-132 (keys CamMat, ProMat, R, T; format of R/Result.yml)
// A DynaFrame data directory therefore works unchanged (with '/' as the path separator).
// R/ = DynaFrame/DynaFrame/ of the reference repository.
#ifndef SLX_SENSOR_HPP
#define SLX_SENSOR_HPP

#include <string>
#include <vector>

#include "dynaframe.hpp"

namespace slx {

// 8-bit grey image from an uncompressed BMP: 8-bit paletted (grey through the palette, as cv::imread
// does) or 24/32-bit BGR(A) (OpenCV 2.4 fixed-point luma: (B*1868 + G*9617 + R*4899 + 8192) >> 14).
// Rows come out top-down, densely packed.  Returns false on a missing / unsupported file.
bool ReadBmpGray(const std::string &path, std::vector<uint8_t> &pixels, int &rows, int &cols);

// 8-bit binary PGM ("P5", maxval <= 255), the other grey format cv::imread accepts.
bool ReadPgmGray(const std::string &path, std::vector<uint8_t> &pixels, int &rows, int &cols);

// The calibration file Init reads: YAML 1.0 with !!opencv-matrix blocks named CamMat, ProMat, R, T.
bool ReadCalibrationYaml(const std::string &path, Calibration &calib);

// The text file CCalculation::Result writes (R/CCalculation.cpp:323-357): one "x y z" line per point, every number as
// `ostream << double` prints it (precision 6, %g).  Same bytes as that loop, without its flush per line: the numbers are formatted
// by std::to_chars(general, 6) -- specified to give printf("%.6g")'s characters, which is what operator<< gives -- by several
// threads into memory, and written out in order.  xyz: 3 doubles per point.  False when the file cannot be written.
// dialect (enum slx_text_dialect, include/slx.h): SLX_TEXT_LIBSTDCXX = the bytes that loop writes when built with libstdc++ / glibc
// ("5e-05", '\n'); SLX_TEXT_MSVC2013 = the bytes of the reference AS BUILT (MSVC 2013, text-mode stream): "5e-005", CR LF.
bool WritePointCloudText(const std::string &path, const double *xyz, size_t n_points, int dialect = SLX_TEXT_LIBSTDCXX);

class CSensor {
public:
    explicit CSensor(const StaticParameters &sp = StaticParameters(), int dynaFrameMaxNum = kDynaFrameMaxNum);
    ~CSensor();
    // R/CSensorV.cpp:31: fixes the directory layout; groupDataPath replaces DATA_PATH + "20161103\\MoveBoard1103\\"
    bool InitSensor(const std::string &groupDataPath);
    bool CloseSensor();
    bool LoadDatas(int groupNum);          // R/CSensorV.cpp:60
    bool UnloadDatas();                    // R/CSensorV.cpp:136
    bool SetProPicture(int nowNum);        // R/CSensorV.cpp:154
    // R/CSensorV.cpp:171 returns a deep copy; here a view of the sensor's own copy, valid until UnloadDatas
    Image8 GetCamPicture() const;
    int DataNum() const { return m_dataNum; }
    const std::string &LastError() const { return m_err; }

private:
    StaticParameters m_sp;
    int m_dynaMax;
    int m_dataNum = 0, m_nowNum = 0;
    std::string m_groupDataPath, m_iFramePath, m_cFramePath, m_vGrayName, m_vPhaseName, m_dynaName, m_dataFileSuffix;
    std::vector<std::vector<uint8_t>> m_dataMats;
    std::vector<int> m_rows, m_cols;
    std::string m_err;
};

}  // namespace slx

#endif
