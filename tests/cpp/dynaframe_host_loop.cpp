// A DynaFrame-style host loop over the C++ mirror classes (csrc/dynaframe.hpp), shaped like
// CCalculation::FillFirstProjectorU + CalculateFirst (R/CCalculation.cpp:171-206, :525-592):
// feed 2*G Gray images and N phase images, decode, read the results back.
// Usage: dynaframe_host_loop <in.bin> <out.bin> <W> <H> <projector width> <code file dir/> <code file>
//   in.bin : 12 Gray planes then 4 phase planes, u8, dense W*H each
//   out.bin: gray, pix, z, x, y, U as f64 W*H each
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "dynaframe.hpp"

static int die(const char *what, const std::string &why)
{
    std::fprintf(stderr, "%s: %s\n", what, why.c_str());
    return 1;
}

int main(int argc, char **argv)
{
    if (argc != 8) return die("usage", "in out W H projW codeDir codeFile");
    const int W = std::atoi(argv[3]), H = std::atoi(argv[4]), PW = std::atoi(argv[5]);
    const size_t n = (size_t)W * H;
    slx::StaticParameters sp;
    sp.CAMERA_RESLINE = W;
    sp.CAMERA_RESROW = H;
    sp.PROJECTOR_RESLINE = PW;
    sp.FOV_MIN_DISTANCE = 100;
    sp.FOV_MAX_DISTANCE = 1000;
    const int G = sp.GRAY_V_NUMDIGIT, N = sp.PHASE_NUMDIGIT;

    std::vector<uint8_t> in((size_t)(2 * G + N) * n);
    FILE *f = std::fopen(argv[1], "rb");
    if (!f || std::fread(in.data(), 1, in.size(), f) != in.size()) return die("read", argv[1]);
    std::fclose(f);
    auto image = [&](int i) {
        slx::Image8 im;
        im.data = in.data() + (size_t)i * n;
        im.rows = H;
        im.cols = W;
        im.step = (size_t)W;
        return im;
    };

    // -- the two decoders on their own, as FillFirstProjectorU drives them
    slx::CDecodeGray grayv(sp);
    if (grayv.SetMat(0, image(0))) return die("CDecodeGray", "SetMat before SetNumDigit must fail");
    if (grayv.SetNumDigit(0, true) || grayv.SetNumDigit(17, true)) return die("CDecodeGray", "bad digit count accepted");
    if (!grayv.SetNumDigit(G, true)) return die("SetNumDigit", grayv.LastError());
    grayv.SetMatFileName("/nonexistent/", "nope.txt");
    for (int i = 0; i < 2 * G; i++)
        if (!grayv.SetMat(i, image(i))) return die("CDecodeGray::SetMat", grayv.LastError());
    if (grayv.Decode()) return die("CDecodeGray", "Decode with a missing code file must fail");
    grayv.SetMatFileName(argv[6], argv[7]);
    if (!grayv.Decode()) return die("CDecodeGray::Decode", grayv.LastError());
    std::vector<double> vGrayMat = grayv.GetResult();

    slx::CDecodePhase phasev(sp);
    if (phasev.SetMat(0, image(2 * G))) return die("CDecodePhase", "SetMat before SetNumMat must fail");
    if (phasev.SetNumMat(0, 40)) return die("CDecodePhase", "numMat 0 accepted");
    const int v_pixPeriod = PW / (1 << (G - 1));
    if (!phasev.SetNumMat(N, v_pixPeriod)) return die("SetNumMat", phasev.LastError());
    for (int i = 0; i < N; i++)
        if (!phasev.SetMat(i, image(2 * G + i))) return die("CDecodePhase::SetMat", phasev.LastError());
    if (!phasev.Decode()) return die("CDecodePhase::Decode", phasev.LastError());
    std::vector<double> vPhaseMat = phasev.GetResult();

    // -- the whole static path
    slx::Calibration cal = {
        {1.2138714552009253e+003 * W / 640.0, 0., (W - 1) / 2.0, 0., 1.2159945377703152e+003 * W / 640.0, (H - 1) / 2.0, 0., 0., 1.},
        {2.0288057545415668e+003 * PW / 1280.0, 0., 6.1958898841564314e+002 * PW / 1280.0, 0.,
         2.0319614890033101e+003 * PW / 1280.0, 6.6520739361244557e+002 * PW / 1280.0, 0., 0., 1.},
        {9.9143473372566937e-001, -1.2723342704854930e-002, 1.2998186532253575e-001, 2.5847502916207063e-002,
         9.9467300669012182e-001, -9.9787355687128362e-002, -1.2801982407153850e-001, 1.0229235705783506e-001,
         9.8648223416959957e-001},
        {-3.1747826732013134e+000 * 10.0, -9.2770189525198721e-001 * 10.0, 3.9430125669975382e+000 * 10.0}};
    slx::CCalculation calc;
    if (calc.CalculateFirst()) return die("CCalculation", "CalculateFirst before Init must fail");
    if (!calc.Init(sp, cal, argv[6], argv[7])) return die("Init", calc.LastError());
    if (calc.Init(sp, cal, argv[6], argv[7])) return die("CCalculation", "second Init must fail");
    for (int i = 0; i < 2 * G; i++)
        if (!calc.SetSensorFrame(0, i, image(i))) return die("SetSensorFrame", calc.LastError());
    for (int i = 0; i < N; i++)
        if (!calc.SetSensorFrame(1, i, image(2 * G + i))) return die("SetSensorFrame", calc.LastError());
    if (!calc.CalculateFirst()) return die("CalculateFirst", calc.LastError());
    std::vector<double> z = calc.GetZ(), x = calc.GetX(), y = calc.GetY(), U = calc.GetProjectorU();
    if (vGrayMat.size() != n || vPhaseMat.size() != n || z.size() != n || x.size() != n || y.size() != n || U.size() != n)
        return die("results", "wrong size");

    if (!calc.Result(std::string(argv[2]) + ".txt", 0)) return die("Result", calc.LastError());   // R/CCalculation.cpp:195-200
    if (calc.Result("/nonexistent-dir/cloud.txt", 0)) return die("Result", "unwritable path accepted");

    f = std::fopen(argv[2], "wb");
    if (!f) return die("write", argv[2]);
    for (const std::vector<double> *v : {&vGrayMat, &vPhaseMat, &z, &x, &y, &U}) std::fwrite(v->data(), sizeof(double), n, f);
    std::fclose(f);
    std::printf("ok\n");
    return 0;
}
