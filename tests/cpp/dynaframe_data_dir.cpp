// The reference's main() (R/main.cpp:42-44: Init, CalculateFirst) over a DynaFrame data directory, through the
// C++ mirror classes: calibration from <dir>/parameters.yml (cv::FileStorage YAML), camera images from
// <dir>/<group>/iFrame/vGrayCam<i>.bmp and vPhaseCam<i>.bmp (CSensor), Gray table from <dir>/Patterns/vGrayCode.txt,
// point cloud to <dir>/PointCloud/iFrame.txt (CCalculation::Result).
// Usage: dynaframe_data_dir <dir> <group subdir> <projector width> <fov min> <fov max>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "dynaframe.hpp"
#include "sensor.hpp"

static int die(const char *what, const std::string &why)
{
    std::fprintf(stderr, "%s: %s\n", what, why.c_str());
    return 1;
}

int main(int argc, char **argv)
{
    if (argc != 6) return die("usage", "dir group projW fovMin fovMax");
    const std::string dir = std::string(argv[1]) + "/";
    slx::Calibration cal;
    if (slx::ReadCalibrationYaml(dir + "missing.yml", cal)) return die("ReadCalibrationYaml", "missing file accepted");
    if (!slx::ReadCalibrationYaml(dir + "parameters.yml", cal)) return die("ReadCalibrationYaml", dir + "parameters.yml");

    slx::StaticParameters sp;
    sp.PROJECTOR_RESLINE = std::atoi(argv[3]);
    sp.FOV_MIN_DISTANCE = std::atof(argv[4]);
    sp.FOV_MAX_DISTANCE = std::atof(argv[5]);

    slx::CSensor sensor(sp);
    sensor.InitSensor(dir + argv[2]);
    if (sensor.LoadDatas(7)) return die("CSensor", "unknown group accepted");
    if (!sensor.LoadDatas(0)) return die("LoadDatas(0)", sensor.LastError());
    sensor.SetProPicture(0);
    const slx::Image8 first = sensor.GetCamPicture();
    sp.CAMERA_RESROW = first.rows;                 // the reference compiles the camera size in; here it comes from the data
    sp.CAMERA_RESLINE = first.cols;

    slx::CCalculation calc;
    if (!calc.Init(sp, cal, dir + "Patterns/", "vGrayCode.txt")) return die("Init", calc.LastError());
    for (int i = 0; i < sp.GRAY_V_NUMDIGIT * 2; i++) {          // R/CCalculation.cpp:539-544
        if (!sensor.SetProPicture(i)) return die("SetProPicture", "gray");
        if (!calc.SetSensorFrame(0, i, sensor.GetCamPicture())) return die("SetSensorFrame", calc.LastError());
    }
    if (sensor.SetProPicture(sp.GRAY_V_NUMDIGIT * 2)) return die("SetProPicture", "index past the group accepted");
    if (!sensor.LoadDatas(1)) return die("LoadDatas(1)", sensor.LastError());
    for (int i = 0; i < sp.PHASE_NUMDIGIT; i++) {               // R/CCalculation.cpp:552-557
        sensor.SetProPicture(i);
        if (!calc.SetSensorFrame(1, i, sensor.GetCamPicture())) return die("SetSensorFrame", calc.LastError());
    }
    if (!calc.CalculateFirst()) return die("CalculateFirst", calc.LastError());
    if (!calc.Result(dir + "PointCloud/iFrame.txt", 0)) return die("Result", calc.LastError());
    const std::vector<double> z = calc.GetZ();                  // frame 0, before the dynamic frames overwrite it
    // R/main.cpp:44: CalculateOther over <group>/cFrame/dynaCam<i>.bmp (as many as are on disk)
    const int frames = calc.CalculateOther(sensor, dir + "PointCloud/cFrame");
    FILE *f = std::fopen((dir + "z.bin").c_str(), "wb");
    if (!f) return die("write", "z.bin");
    std::fwrite(z.data(), sizeof(double), z.size(), f);
    std::fclose(f);
    std::printf("ok %d x %d, dynamic frames %d\n", sp.CAMERA_RESLINE, sp.CAMERA_RESROW, frames);
    return 0;
}
