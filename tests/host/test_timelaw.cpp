// Closed-form runs (time_law_segments) against the literal float32 replay (time_law), which is
// itself checked against the compiled reference's per-call output counts (tests/test_host_logic.py).
// Pure host.  Exhaustive over many rates, upsample factors, chunk sizes and out_len limits.
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#include "../../simplefe_amd/csrc/timelaw.h"

int main()
{
    unsigned seed = 12345;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (seed >> 8) * (1.0 / 16777216.0); };
    long long total = 0, segs_total = 0;
    int cases = 0;
    const float fixed_rates[] = {1.77f, 0.77f, 5.0f / 3.0f, 8.0f, 2.5f, 1.3f, 1.0f, 1.0000001f, 3.1415927f, 0.3333334f,
                                 7.08f / 4.0f, 10.52f, 1.5f, 0.50000006f, 123.456f};
    for (int c = 0; c < 1500; c++) {
        const int U = 1 + (int)(rnd() * 7);
        float rate = c < 15 ? fixed_rates[c] : (float)(1.0 / U + rnd() * rnd() * 12.0);
        if (rate < 1.0f / U) rate = 1.0f / U;
        const int B = c % 7 == 0 ? 4096 : 16 + (int)(rnd() * 5000);
        const bool tight = (c % 11 == 0);           // sometimes hit the out_len limit mid-block
        sfe_rs_timestate a = {0, 0.0f, 0}, b = {0, 0.0f, 0};
        for (int call = 0; call < 12; call++) {
            const int m = (call % 3 == 2) ? 1 + (int)(rnd() * B) : B;
            const int out_len = tight ? (int)floorf(m / rate) : (int)ceilf(m / rate) + 2;
            std::vector<int> pos;
            std::vector<float> mu;
            const int n1 = sfe::time_law(&a, U, m, out_len, rate, [&](int p, float w) { pos.push_back(p); mu.push_back(w); });
            std::vector<sfe::TlSeg> segs;
            const int n2 = sfe::time_law_segments(&b, U, m, out_len, rate, segs);
            if (n1 != n2 || a.pos != b.pos || a.mu != b.mu || a.leftover != b.leftover) {
                printf("FAIL state: case %d U %d rate %.9g B %d call %d: n %d vs %d pos %d/%d mu %.9g/%.9g lo %d/%d\n", c, U, rate, B,
                       call, n1, n2, a.pos, b.pos, a.mu, b.mu, a.leftover, b.leftover);
                return 1;
            }
            int k = 0;
            for (size_t s = 0; s < segs.size(); s++) {
                if (segs[s].k0 != k) { printf("FAIL k0\n"); return 1; }
                for (int i = 0; i < segs[s].count; i++, k++) {
                    const double t = segs[s].t0 + (double)i * (double)segs[s].d;
                    const double fl = floor(t);
                    if ((int)fl != pos[k] || (float)(t - fl) != mu[k]) {
                        printf("FAIL out: case %d U %d rate %.9g B %d call %d k %d: pos %d vs %d mu %.9g vs %.9g\n", c, U, rate, B, call,
                               k, (int)fl, pos[k], (float)(t - fl), mu[k]);
                        return 1;
                    }
                }
            }
            if (k != n1) { printf("FAIL count\n"); return 1; }
            total += n1;
            segs_total += (long long)segs.size();
        }
        cases++;
    }
    printf("timelaw ok: %d cases, %lld outputs in %lld runs (%.1f outputs per run)\n", cases, total, segs_total,
           (double)total / (double)segs_total);
    return 0;
}
