// The double-arithmetic shortcut of the device formatter (csrc/slx_text.hip, g6_digits) against the exact integer arithmetic it stands in
// front of (the host formatter's, csrc/sensor.cpp fmt_g6_fast), on the host: the same IEEE operations (a * 10^p, floor, a subtraction;
// no contraction), so what holds here holds on the device.  Random magnitudes over the shortcut's whole range, and the neighbourhoods it
// must hand to the integers or get right: values around (k + 1/2) / 10^p at distances from 2^-20 down to a few ulps, around the powers of
// ten, around the integers.  Usage: text_shortcut [millions of random cases]   (exit code 1 on the first difference)
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

static const uint64_t kPow10[20] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull, 100000000ull, 1000000000ull,
                                    10000000000ull, 100000000000ull, 1000000000000ull, 10000000000000ull, 100000000000000ull,
                                    1000000000000000ull, 10000000000000000ull, 100000000000000000ull, 1000000000000000000ull,
                                    10000000000000000000ull};

// the six digits and the exponent: exact == true takes only the integers, else the shortcut first (as the device does)
static bool digits(double v, bool exact, uint64_t &q_out, int &X_out, bool &used_shortcut)
{
    uint64_t bits;
    std::memcpy(&bits, &v, 8);
    const uint64_t mag = bits & 0x7fffffffffffffffull;
    const int k = (int)(mag >> 52) - 1023;
    const double a = std::fabs(v);
    if (mag == 0 || k < -17 || k > 49 || !(a >= 1e-5 && a < 1e15)) return false;
    const uint64_t m = (mag & 0x000fffffffffffffull) | 0x0010000000000000ull;
    const int s = 52 - k;
    int X = k >= 0 ? (k * 1233) >> 12 : -(((-k) * 1233 + 4095) >> 12);
    uint64_t q = 0;
    used_shortcut = false;
    for (int tries = 0; tries < 4; tries++) {
        const int p = 5 - X;
        if (!exact && p >= 0 && p <= 10) {
            const double P = ((p & 1) ? 10.0 : 1.0) * ((p & 2) ? 100.0 : 1.0) * (((p & 4) ? 1e4 : 1.0) * ((p & 8) ? 1e8 : 1.0));
            const double t = a * P;
            const double f = std::floor(t), fr = t - f;
            if (std::fabs(fr - 0.5) > 0x1p-30 && std::fabs(t - 1e5) > 1e-6 && std::fabs(t - 1e6) > 1e-6) {
                if (f < 1e5) { X--; continue; }
                if (f >= 1e6) { X++; continue; }
                unsigned q32 = (unsigned)f + (fr > 0.5 ? 1u : 0u);
                if (q32 == 1000000u) { q32 = 100000u; X++; }
                q = q32;
                used_shortcut = true;
                break;
            }
        }
        bool up;
        if (p >= 0) {
            const unsigned __int128 T = (unsigned __int128)m * kPow10[p];
            q = (uint64_t)(T >> s);
            if (q < 100000ull) { X--; continue; }
            if (q >= 1000000ull) { X++; continue; }
            const unsigned __int128 rem = T & ((((unsigned __int128)1) << s) - 1), half = ((unsigned __int128)1) << (s - 1);
            up = rem > half || (rem == half && (q & 1ull));
        } else {
            const uint64_t D = kPow10[-p] << s;
            q = m / D;
            if (q < 100000ull) { X--; continue; }
            if (q >= 1000000ull) { X++; continue; }
            const uint64_t rem = m - q * D;
            up = 2 * rem > D || (2 * rem == D && (q & 1ull));
        }
        if (up && ++q == 1000000ull) { q = 100000ull; X++; }
        used_shortcut = false;
        break;
    }
    q_out = q;
    X_out = X;
    return true;
}

static unsigned long long n_cases = 0, n_short = 0;
static bool check(double v)
{
    uint64_t q0, q1;
    int X0, X1;
    bool s0, s1;
    const bool ok0 = digits(v, true, q0, X0, s0), ok1 = digits(v, false, q1, X1, s1);
    n_cases++;
    n_short += ok1 && s1;
    if (ok0 != ok1 || (ok0 && (q0 != q1 || X0 != X1))) {
        std::printf("DIFFERENT: %.17g (%a): integers %llu e%d, shortcut %llu e%d\n", v, v, (unsigned long long)q0, X0, (unsigned long long)q1, X1);
        return false;
    }
    return true;
}

int main(int argc, char **argv)
{
    const unsigned long long millions = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 20;
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> expo(-5.0, 6.2), unit(0.0, 1.0);
    // 1. random magnitudes, log-uniform over [1e-5, 1.6e6)
    for (unsigned long long i = 0; i < millions * 1000000ull; i++) {
        const double v = std::pow(10.0, expo(rng)) * (rng() & 1 ? 1.0 : -1.0);
        if (!check(v)) return 1;
    }
    // 2. the neighbourhoods of the ties: (k + 1/2) / 10^p and what lies 2^-20 ... a few ulps beside them, for every decade
    for (int p = -2; p <= 10; p++) {
        const double scale = std::pow(10.0, -p);
        for (int rep = 0; rep < 200000; rep++) {
            const double kk = 100000.0 + std::floor(unit(rng) * 900000.0) + 0.5;
            double v = kk * scale;
            for (int d = 0; d < 8; d++) {
                if (!check(v) || !check(-v)) return 1;
                v = std::nextafter(v, rep & 1 ? INFINITY : 0.0);
            }
            for (int e = 20; e <= 44; e += 2) {
                if (!check((kk + std::ldexp(1.0, -e)) * scale) || !check((kk - std::ldexp(1.0, -e)) * scale)) return 1;
            }
        }
    }
    // 3. the powers of ten and the integers of the digit range, with their neighbours
    for (int X = -5; X <= 6; X++) {
        double v = std::pow(10.0, X);
        for (int d = 0; d < 2000; d++) v = std::nextafter(v, 0.0);
        for (int d = 0; d < 4000; d++) {
            if (!check(v)) return 1;
            v = std::nextafter(v, INFINITY);
        }
    }
    for (int p = 0; p <= 10; p++) {
        const double scale = std::pow(10.0, -p);
        for (int rep = 0; rep < 200000; rep++) {
            double v = (100000.0 + std::floor(unit(rng) * 900000.0)) * scale;
            for (int d = 0; d < 3; d++) v = std::nextafter(v, 0.0);
            for (int d = 0; d < 7; d++) {
                if (!check(v)) return 1;
                v = std::nextafter(v, INFINITY);
            }
        }
    }
    std::printf("text_shortcut: %llu cases, %llu by the shortcut, 0 differences\n", n_cases, n_short);
    return 0;
}
