// Probe: time the bulk resample path through the C ABI without torch in the process.
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <vector>
#include "../../include/sfe_dsp.h"
int main(int argc, char **argv)
{
    const size_t n = (size_t)1 << 28;
    std::vector<float> taps(381);
    for (int k = 0; k < 381; k++) { double t = k - 190.0; double s = t == 0 ? 1.0 : sin(0.18 * M_PI * t) / (0.18 * M_PI * t); taps[k] = (float)(0.54 * s * (0.54 - 0.46 * cos(2 * M_PI * k / 380.0))); }
    void *x, *y; sfe_rs_t h;
    if (sfe_dsp_malloc(&x, n * 8) || sfe_dsp_malloc(&y, (n * 3 / 5 + 16) * 8)) return 1;
    sfe_dsp_synth_fill(x, 2 * n, 1, 0, 0, 0);
    if (sfe_dsp_rs_create(taps.data(), 381, 3, 4096, 1, 1, 0, SFE_RS_RESAMPLE, &h)) { puts(sfe_dsp_last_error()); return 1; }
    sfe_timer_t t; sfe_dsp_timer_create(&t);
    size_t no = 0;
    for (int i = 0; i < 5; i++) sfe_dsp_rs_process_stream(h, x, n, n, y, n * 3 / 5 + 16, n * 3 / 5 + 16, 5.0f / 3.0f, &no, 0);
    sfe_dsp_timer_start(t, 0);
    for (int i = 0; i < 20; i++) sfe_dsp_rs_process_stream(h, x, n, n, y, n * 3 / 5 + 16, n * 3 / 5 + 16, 5.0f / 3.0f, &no, 0);
    sfe_dsp_timer_stop(t, 0);
    float ms; sfe_dsp_timer_elapsed_ms(t, &ms);
    printf("n_out %zu  %.4f ms per pass\n", no, ms / 20);
    return 0;
}
