// timelaw.h -- the reference's float32 output-time recurrence, replayed on the host.  HOST ONLY
// (no HIP): tests/host/test_timelaw.cpp checks the closed form against the step-by-step replay
// without a GPU.
//
// The law (libdsp/resample.cxx:89,119-150 == libdsp/decimate.cxx:73,96-127), per process() call
// of n_in samples:   t = (float)m_pos + m_mu;  [leftover output, t += step]
//                    loop { m_pos = floorf(t); m_mu = t - m_pos; stop tests; emit; t += step }
//                    m_pos -= n_in*U
// with step = rate*U in float32 and every `t += step` rounded to float32.  When step is not
// integer-valued the roundings make t drift (~4e-4 per 128-sample block), so outputs must be
// placed by this exact sequence, not by k*step.
//
//   time_law()          emits every output (position, mu): the literal replay.
//   time_law_segments() emits RUNS: inside one binade [2^e, 2^(e+1)) adding a fixed float to a
//       float changes it by a constant (the sum is rounded to the binade's grid u = 2^(e-23);
//       step = q*u + r rounds to q*u or (q+1)*u independent of t, and in the tie case |r| = u/2
//       round-half-even settles after one step), so t advances by a fixed d until it leaves the
//       binade.  A run is (t0, d, count): t_i = t0 + i*d exactly (double holds it exactly).
//       ~3 runs per binade, ~40 per 4096-sample call instead of thousands of outputs; the GPU
//       expands them (polyphase.hip: poly_seg_kernel).
#pragma once
#include <math.h>
#include <stdint.h>

#include <vector>

#include "../../include/sfe_dsp.h"

namespace sfe {

template <class Emit>
static int time_law(sfe_rs_timestate *st, int U, int n_in, int out_len, float rate, Emit emit)
{
    int n_out = 0;
    float t = (float)st->pos + st->mu;
    const float step = rate * (float)U;
    if (st->leftover) {
        emit(-1, st->mu);
        n_out++;
        st->leftover = 0;
        t += step;
    }
    for (;;) {
        st->pos = (int)floorf(t);
        st->mu = t - (float)st->pos;
        const int pos1 = st->pos + 1;
        const int n0 = st->pos / U, n1 = pos1 / U;     // C truncation, as the reference's ints
        if (n0 >= n_in || n_out >= out_len) break;
        if (n1 >= n_in) {
            st->leftover = 1;
            break;
        }
        emit(st->pos, st->mu);
        n_out++;
        t += step;
    }
    st->pos -= n_in * U;
    return n_out;
}

// One run of outputs of one process() call.  Output i of the run (0 <= i < count) sits at
// t = t0 + i*d relative to the call's first sample (upsampled grid): pos = floor(t), mu = t - pos.
struct TlSeg {
    double  t0;
    float   d;        // exact: a multiple of the binade's grid with <= 24 significant bits
    int32_t k0;       // index of the run's first output within the call
    int32_t count;
    int32_t pad;
};

// Appends the runs of one call to `segs`; returns the number of outputs; advances *st exactly as
// time_law() does.
static inline int time_law_segments(sfe_rs_timestate *st, int U, int n_in, int out_len, float rate,
                                    std::vector<TlSeg> &segs)
{
    int n_out = 0;
    float t = (float)st->pos + st->mu;
    const float step = rate * (float)U;
    auto push = [&](double t0, float d, int count) {
        TlSeg s;
        s.t0 = t0;
        s.d = d;
        s.k0 = n_out;
        s.count = count;
        s.pad = 0;
        segs.push_back(s);
        n_out += count;
    };
    if (st->leftover) {
        push(-1.0 + (double)st->mu, 0.0f, 1);      // pos = -1, mu = the stored m_mu, exactly
        st->leftover = 0;
        t += step;
    }
    const double limit = (double)n_in * (double)U - 1.0;    // emit while t < limit  (pos <= n_in*U - 2)
    for (;;) {
        st->pos = (int)floorf(t);
        st->mu = t - (float)st->pos;
        const int pos1 = st->pos + 1;
        const int n0 = st->pos / U, n1 = pos1 / U;
        if (n0 >= n_in || n_out >= out_len) break;
        if (n1 >= n_in) {
            st->leftover = 1;
            break;
        }
        // this output is emitted; how far does a constant-increment run reach?
        int count = 1;
        float d = 0.0f;
        if (t >= 1.0f) {
            const float t1 = t + step, t2 = t1 + step;
            const int e = ilogbf(t);
            const double d1 = (double)t1 - (double)t, d2 = (double)t2 - (double)t1;
            if (ilogbf(t1) == e && ilogbf(t2) == e && d1 == d2 && d1 > 0.0) {
                const double top = ldexp(1.0, e + 1);
                const double bound = top < limit ? top : limit;
                long long j = (long long)ceil((bound - (double)t) / d1) - 1;    // largest j: t + j d1 < bound
                while ((double)t + (double)(j + 1) * d1 < bound) j++;
                while (j > 0 && (double)t + (double)j * d1 >= bound) j--;
                const long long room = (long long)out_len - n_out - 1;
                if (j > room) j = room;
                if (j > 0) {
                    count = (int)j + 1;
                    d = (float)d1;
                }
            }
        }
        push((double)t, d, count);
        const float t_last = (float)((double)t + (double)(count - 1) * (double)d);   // exact
        t = t_last + step;                                                           // the real rounding
    }
    st->pos -= n_in * U;
    return n_out;
}

}  // namespace sfe
