// bfa_stitch.hip -- window stitching, the step immediately in front of the alignment path:
// cosine-weighted overlap-add of per-window network outputs (bournemouth_aligner/cupe2i/windowing.py:103-173,
// called at core.py:422-438 for the phoneme and the group logits).
//
//   combined[b,t,:] = sum_i  x[b,i,t - i*stride,:] * w[t - i*stride]      (windows i in ascending order)
//   out[b,t,:]      = combined / (sum_i w[t - i*stride] + 1e-8)           stride = F / 2
//
// The last window is cut to the frames that still fit (:152-163); a frame no window covers comes out as 0.
// Float32 in the reference's order of operations (product, then accumulate; one IEEE division), so the result is
// bit-identical.  Pure streaming: every input element is read once, every output element written once; the output
// row stride is the caller's (e.g. padded to 68 / 72 floats, the DP's preferred row alignment).
#include <hip/hip_runtime.h>

#include "bfa_types.hpp"

#pragma clang fp contract(off)

namespace bfa {

__global__ __launch_bounds__(256) void k_stitch(const float *__restrict__ win, int B, int NW, int F, int C,
                                                const float *__restrict__ weights, int total_frames,
                                                float *__restrict__ out, int64_t oB, int64_t oT)
{
    // blockIdx.y = utterance; a thread takes elements (t, c) of it in row-major order (32-bit index arithmetic:
    // total_frames * C < 2^31 is checked by the launcher)
    const int stride = F / 2;
    const int b = blockIdx.y;
    const int n = total_frames * C;
    const float *wb = win + (int64_t)b * NW * F * C;
    float *ob = out + (int64_t)b * oB;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n; e += gridDim.x * blockDim.x) {
        const int t = e / C;
        const int c = e - t * C;
        // windows that can cover frame t: i*stride <= t < i*stride + F
        int i_lo, i_hi;
        if (stride > 0) {
            i_hi = min(NW - 1, t / stride);
            const int num = t - F + 1;
            i_lo = (num <= 0) ? 0 : (num + stride - 1) / stride;
        } else { // F == 1: every window starts at frame 0
            i_lo = 0;
            i_hi = (t == 0) ? NW - 1 : -1;
        }
        float comb = 0.0f, wsum = 0.0f;
        for (int i = i_lo; i <= i_hi; ++i) {
            const int start = i * stride;
            const int f = t - start;
            int nf = F;
            if (i == NW - 1 && start + F > total_frames) nf = total_frames - start;
            if (f < 0 || f >= nf) continue;
            const float w = weights[f];
            const float prod = wb[(i * F + f) * C + c] * w;
            comb = comb + prod;
            wsum = wsum + w;
        }
        ob[(int64_t)t * oT + c] = comb / (wsum + 1e-8f);
    }
}

// Even frames-per-window (the model's: 10): stride = F/2, so at most two windows cover a frame -- i1 = t / stride and
// i0 = i1 - 1 -- and both loads can be issued unconditionally (clamped addresses), four elements per thread in flight.
__global__ __launch_bounds__(256) void k_stitch_even(const float *__restrict__ win, int NW, int F, int C,
                                                     const float *__restrict__ weights, int total_frames,
                                                     float *__restrict__ out, int64_t oB, int64_t oT)
{
    const int stride = F / 2;
    const int b = blockIdx.y;
    const int n = total_frames * C;
    const float *wb = win + (int64_t)b * NW * F * C;
    float *ob = out + (int64_t)b * oB;
    const int last_nf = min(F, total_frames - (NW - 1) * stride); // frames the last window contributes
    constexpr int U = 4;
    for (int e0 = (blockIdx.x * U) * blockDim.x + threadIdx.x; e0 < n; e0 += gridDim.x * U * blockDim.x) {
        float x0[U], x1[U], w0[U], w1[U];
        bool h0[U], h1[U];
        int t_[U], c_[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = min(e0 + u * (int)blockDim.x, n - 1);
            const int t = e / C, c = e - t * C;
            t_[u] = t; c_[u] = c;
            const int i1 = t / stride, f1 = t - i1 * stride; // f1 < stride
            const int i0 = i1 - 1, f0 = f1 + stride;         // f0 < F
            h1[u] = i1 <= NW - 1 && (i1 < NW - 1 || f1 < last_nf);
            h0[u] = i0 >= 0 && i0 <= NW - 1 && (i0 < NW - 1 || f0 < last_nf);
            const int j1 = min(i1, NW - 1), j0 = min(max(i0, 0), NW - 1);
            x1[u] = wb[(j1 * F + f1) * C + c];
            x0[u] = wb[(j0 * F + f0) * C + c];
            w1[u] = weights[f1];
            w0[u] = weights[f0];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (e0 + u * (int)blockDim.x >= n) break;
            const float p0 = x0[u] * w0[u], p1 = x1[u] * w1[u];
            float comb = 0.0f, wsum = 0.0f; // ascending window order: i0, then i1
            if (h0[u]) { comb = comb + p0; wsum = wsum + w0[u]; }
            if (h1[u]) { comb = comb + p1; wsum = wsum + w1[u]; }
            ob[(int64_t)t_[u] * oT + c_[u]] = comb / (wsum + 1e-8f);
        }
    }
}

} // namespace bfa

extern "C" int bfa_launch_stitch(const float *win, int B, int NW, int F, int C, const float *weights, int total_frames,
                                 float *out, int64_t oB, int64_t oT, void *stream_)
{
    hipStream_t stream = (hipStream_t)stream_;
    if ((int64_t)total_frames * C >= (1ll << 31) || (int64_t)NW * F * C >= (1ll << 31)) return (int)hipErrorInvalidValue;
    const int n = total_frames * C;
    int bx = (n + 4095) / 4096; // sixteen elements per thread
    if (bx > 1024) bx = 1024;
    if (bx < 1) bx = 1;
    for (int b0 = 0; b0 < B; b0 += 65535) { // gridDim.y limit
        const int nb = (B - b0 < 65535) ? (B - b0) : 65535;
        if ((F & 1) == 0 && NW > 0)
            hipLaunchKernelGGL(bfa::k_stitch_even, dim3((unsigned)bx, (unsigned)nb), dim3(256), 0, stream,
                               win + (int64_t)b0 * NW * F * C, NW, F, C, weights, total_frames, out + (int64_t)b0 * oB, oB, oT);
        else
            hipLaunchKernelGGL(bfa::k_stitch, dim3((unsigned)bx, (unsigned)nb), dim3(256), 0, stream,
                               win + (int64_t)b0 * NW * F * C, nb, NW, F, C, weights, total_frames, out + (int64_t)b0 * oB, oB, oT);
    }
    return (int)hipGetLastError();
}
