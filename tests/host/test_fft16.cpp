// Host check of csrc/fft16.h (the register DFT16 the FIR kernel is built from) against a
// naive double-precision DFT.  Built and run by tests/test_host_logic.py; no GPU needed.
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "../../simplefe_amd/csrc/fft16.h"

template <int DIR>
static double check()
{
    double worst = 0.0;
    for (int trial = 0; trial < 50; trial++) {
        v2f v[16];
        double xr[16], xi[16];
        for (int i = 0; i < 16; i++) {
            xr[i] = (double)rand() / RAND_MAX - 0.5;
            xi[i] = (double)rand() / RAND_MAX - 0.5;
            v[i] = (v2f){(float)xr[i], (float)xi[i]};
            xr[i] = v[i].x;
            xi[i] = v[i].y;
        }
        sfe::dft16<DIR>(v);
        for (int k = 0; k < 16; k++) {
            double sr = 0, si = 0;
            for (int n = 0; n < 16; n++) {
                double a = DIR * 2.0 * M_PI * n * k / 16.0;
                sr += xr[n] * cos(a) - xi[n] * sin(a);
                si += xr[n] * sin(a) + xi[n] * cos(a);
            }
            v2f y = v[sfe::P16(k)];
            worst = fmax(worst, fmax(fabs(y.x - sr), fabs(y.y - si)));
        }
    }
    return worst;
}

// dft16 then dft16_rev of the opposite direction must give 16 * identity, in natural order
static double check_roundtrip()
{
    double worst = 0.0;
    for (int trial = 0; trial < 50; trial++) {
        v2f v[16], x[16];
        for (int i = 0; i < 16; i++) x[i] = v[i] = (v2f){(float)rand() / RAND_MAX - 0.5f, (float)rand() / RAND_MAX - 0.5f};
        sfe::dft16<-1>(v);
        sfe::dft16_rev<+1>(v);
        for (int i = 0; i < 16; i++)
            worst = fmax(worst, fmax(fabs(v[i].x / 16 - x[i].x), fabs(v[i].y / 16 - x[i].y)));
    }
    return worst;
}

int main()
{
    double f = check<-1>(), b = check<+1>();
    double rt = check_roundtrip();
    printf("roundtrip %.3g\n", rt);
    if (rt > 2e-6) return 1;
    // cmul / cmul_conj
    v2f a = {0.3f, -0.7f}, w = {0.6f, 0.8f};
    v2f p = sfe::cmul(a, w), q = sfe::cmul_conj(a, w);
    double e1 = fabs(p.x - (0.3 * 0.6 + 0.7 * 0.8)) + fabs(p.y - (0.3 * 0.8 - 0.7 * 0.6));
    double e2 = fabs(q.x - (0.3 * 0.6 - 0.7 * 0.8)) + fabs(q.y - (-0.3 * 0.8 - 0.7 * 0.6));
    printf("dft16 fwd %.3g inv %.3g cmul %.3g cmul_conj %.3g\n", f, b, e1, e2);
    return (f < 2e-6 && b < 2e-6 && e1 < 1e-6 && e2 < 1e-6) ? 0 : 1;
}
