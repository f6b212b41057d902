// fir_fft.hip -- the headline kernel: streaming FIR by overlap-save with a 4096-point complex
// FFT held entirely in one workgroup's registers + LDS (gfx950).
//
// What it computes is the net effect of blkconv::process() over a stream
// (libdsp/blkconv.cxx:77-110): y[n] = sum_k h[k] x[n-k] with carried state.  The reference
// does it by overlap-ADD on real blocks with FFTW; the block scheme does not show in the
// result, so the GPU uses overlap-SAVE (no read-modify-write of the output, no inter-block
// dependency) and transforms I + jQ as ONE complex sequence: complex data and complex taps
// cost the same as real ones.
//
// Shape: N = 4096 = 16 x 16 x 16.  256 threads, thread t owns 16 complex values in VGPRs.
//   index n = n0 + 16 n1 + 256 n2,  bin k = k2 + 16 k1 + 256 k0
//   F1  t = n0+16n1 : DFT16 over n2, times W_4096^(t k2)        -> LDS [k2][n1][n0]
//   F2  t = n0+16k2 : DFT16 over n1, times W_256^(n0 k1)        -> LDS [k2][k1][n0]
//   F3  t = k1+16k2 : DFT16 over n0, times H[k]/N, IDFT16 over k0 -> LDS (same cells)
//   I2  t = n0+16k2 : times conj W_256, IDFT16 over k1          -> LDS [k2][n1][n0]
//   I3  t = n0+16n1 : times conj W_4096, IDFT16 over k2 -> y[t + 256 n2]
// Global traffic is perfectly coalesced in both directions (lane == consecutive sample,
// register == row of 256); the first hl/256 rows of the result are the overlap-save discard.
// LDS rows are padded (272 / 17 complex) so every ds_read_b64/ds_write_b64 is conflict-free.
// Twiddle bases live in registers for the life of the (persistent) workgroup, which walks
// transforms blockIdx.x, +gridDim.x, ...; the taps' spectrum (32 KiB, L2-resident) is
// re-read per transform.  Budget: <= 128 VGPRs and 34 KiB LDS -> 4 workgroups per CU.
#include "common.h"
#include "fft16.h"

namespace sfe {
namespace {

template <bool IN_C>
__device__ __forceinline__ v2f load_sample(const void *p, long long i)
{
    if constexpr (IN_C) return reinterpret_cast<const v2f *>(p)[i];
    else return (v2f){reinterpret_cast<const float *>(p)[i], 0.0f};
}

template <bool IN_C, bool OUT_C>
__global__ __launch_bounds__(256, 3) void fir_fft4096_kernel(FirFftArgs a)
{
    __shared__ v2f lds[FFT_ROWS * LDS_K2_STRIDE];
    const int t = threadIdx.x;
    const int ch = blockIdx.y;
    const int lo = t & 15, hi = t >> 4;
    const int base_a = t;                                   // [k2][t]
    const int base_b = hi * LDS_K2_STRIDE + lo;             // [k2=hi][.][lo]
    const int base_c = hi * LDS_K2_STRIDE + lo * LDS_K1_STRIDE;  // [k2=hi][k1=lo][.]

    const char *in_c = static_cast<const char *>(a.in) + (size_t)ch * a.in_stride * (IN_C ? 8 : 4);
    char *out_c = static_cast<char *>(a.out) + (size_t)ch * a.out_stride * (OUT_C ? 8 : 4);
    const char *hist_c = static_cast<const char *>(a.hist) + (size_t)ch * a.hl * (IN_C ? 8 : 4);

    // Per-thread twiddle bases, resident for the whole launch.  A twiddle with exponent
    // e*(4a+b) is applied as q[a]*p[b], q[a] = W^(4 e a), p[b] = W^(e b): 12 complex registers
    // for the two twiddle stages instead of 30+30.
    v2f p1[4], q1[4], p2[4], q2[4];
#pragma unroll
    for (int k = 1; k < 4; k++) {
        p1[k] = a.tw1[k * 256 + t];          // W_4096^(t k)
        q1[k] = a.tw1[(k + 3) * 256 + t];    // W_4096^(4 t k)
        p2[k] = a.tw2[k * 16 + lo];          // W_256^(lo k)
        q2[k] = a.tw2[(k + 3) * 16 + lo];    // W_256^(4 lo k)
    }
    const v2f *hs_t = a.hs + t;

    const int row0 = a.hl >> 8;   // rows discarded by overlap-save
    for (long long blk = blockIdx.x; blk < a.nblk; blk += gridDim.x) {
        const long long base = blk * a.advance - a.hl;   // stream index of transform element 0
        v2f v[16];
        if (base >= 0 && base + FFT_N <= a.n) {
#pragma unroll
            for (int r = 0; r < 16; r++) v[r] = load_sample<IN_C>(in_c, base + t + 256 * r);
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const long long i = base + t + 256 * r;
                if (i < 0) v[r] = load_sample<IN_C>(hist_c, a.hl + i);
                else if (i < a.n) v[r] = load_sample<IN_C>(in_c, i);
                else v[r] = (v2f){0.0f, 0.0f};
            }
        }

        // ---- F1: over n2, twiddle W_4096^(t k2), scatter to [k2][t]
        dft16<-1>(v);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            v2f x = v[P16(k)];
            if (k >> 2) x = cmul(x, q1[k >> 2]);
            if (k & 3) x = cmul(x, p1[k & 3]);
            lds[base_a + k * LDS_K2_STRIDE] = x;
        }
        __syncthreads();
        // ---- F2: gather n1 for (k2=hi, n0=lo)
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = lds[base_b + 16 * r];
        dft16<-1>(v);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) {
            v2f x = v[P16(k)];
            if (k >> 2) x = cmul(x, q2[k >> 2]);
            if (k & 3) x = cmul(x, p2[k & 3]);
            lds[base_b + k * LDS_K1_STRIDE] = x;
        }
        __syncthreads();
        // ---- F3: gather n0 for (k2=hi, k1=lo); spectrum multiply; first inverse stage
#pragma unroll
        for (int r = 0; r < 16; r++) v[r] = lds[base_c + r];
        v2f hs[16];   // this thread's 16 bins of H/N: streamed from L2 each transform
#pragma unroll
        for (int k = 0; k < 16; k++) hs[k] = hs_t[k * 256];
        dft16<-1>(v);
        {
            v2f y[16];
#pragma unroll
            for (int k = 0; k < 16; k++) y[k] = cmul(v[P16(k)], hs[k]);
            dft16<+1>(y);
            // I1: element n0 goes back to the cell this thread read n0 from (no barrier needed)
#pragma unroll
            for (int k = 0; k < 16; k++) lds[base_c + k] = y[P16(k)];
        }
        __syncthreads();
        // ---- I2: gather k1 for (k2=hi, n0=lo)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            v2f x = lds[base_b + r * LDS_K1_STRIDE];
            if (r >> 2) x = cmul_conj(x, q2[r >> 2]);
            if (r & 3) x = cmul_conj(x, p2[r & 3]);
            v[r] = x;
        }
        dft16<+1>(v);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; k++) lds[base_b + 16 * k] = v[P16(k)];
        __syncthreads();
        // ---- I3: gather k2 for n_lo = t
#pragma unroll
        for (int r = 0; r < 16; r++) {
            v2f x = lds[base_a + r * LDS_K2_STRIDE];
            if (r >> 2) x = cmul_conj(x, q1[r >> 2]);
            if (r & 3) x = cmul_conj(x, p1[r & 3]);
            v[r] = x;
        }
        dft16<+1>(v);
        __syncthreads();   // LDS free for the next transform

        const long long obase = blk * a.advance - a.hl + t;   // + 256*row
        if (blk * a.advance + a.advance <= a.n) {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                if (r >= row0) {
                    const v2f y = v[P16(r)];
                    if constexpr (OUT_C) reinterpret_cast<v2f *>(out_c)[obase + 256 * r] = y;
                    else reinterpret_cast<float *>(out_c)[obase + 256 * r] = y.x;
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const long long o = obase + 256 * r;
                if (r >= row0 && o < a.n) {
                    const v2f y = v[P16(r)];
                    if constexpr (OUT_C) reinterpret_cast<v2f *>(out_c)[o] = y;
                    else reinterpret_cast<float *>(out_c)[o] = y.x;
                }
            }
        }
    }
}

}  // namespace

int launch_fir_fft(const FirFftArgs &a, int in_complex, int out_complex, int n_channels,
                   hipStream_t s)
{
    if (a.nblk <= 0) return SFE_OK;
    if (a.hl <= 0 || (a.hl & 255) || a.hl >= FFT_N || a.advance != FFT_N - a.hl) {
        set_error("fir_fft: bad overlap rows (hl=%d advance=%d)", a.hl, a.advance);
        return SFE_EINVAL;
    }
    // persistent workgroups: enough to fill 256 CUs x 4 resident, shared over channels
    long long gx = a.nblk;
    const long long cap = (256LL * 8 + n_channels - 1) / n_channels;
    if (gx > cap) gx = cap < 1 ? 1 : cap;
    dim3 grid((unsigned)gx, (unsigned)n_channels), block(256);
    if (in_complex && out_complex) hipLaunchKernelGGL((fir_fft4096_kernel<true, true>), grid, block, 0, s, a);
    else if (!in_complex && out_complex) hipLaunchKernelGGL((fir_fft4096_kernel<false, true>), grid, block, 0, s, a);
    else if (!in_complex && !out_complex) hipLaunchKernelGGL((fir_fft4096_kernel<false, false>), grid, block, 0, s, a);
    else {
        set_error("fir_fft: complex input with real output is not a defined combination");
        return SFE_EINVAL;
    }
    SFE_HIP(hipGetLastError());
    return SFE_OK;
}

}  // namespace sfe
