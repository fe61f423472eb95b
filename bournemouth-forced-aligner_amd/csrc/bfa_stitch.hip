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
        hipLaunchKernelGGL(bfa::k_stitch, dim3((unsigned)bx, (unsigned)nb), dim3(256), 0, stream,
                           win + (int64_t)b0 * NW * F * C, nb, NW, F, C, weights, total_frames, out + (int64_t)b0 * oB, oB, oT);
    }
    return (int)hipGetLastError();
}
