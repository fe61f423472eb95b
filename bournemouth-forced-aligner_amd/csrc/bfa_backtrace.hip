// bfa_backtrace.hip -- K2: backtrace over the packed backpointers + framewise outputs
// (forced_alignment.py:686-700), plus the non-DP fills (proportional :170-172, silence :382-397).
//
// One wavefront per work item.  The serial dependence state[t-1] = bp[t][state[t]] is only a real
// dependence at frames where the path moves (k != 0): about L of the T frames.  So instead of one
// step per frame, all 64 lanes look at 64 consecutive frames at once for the CURRENT state, a ballot
// finds the latest frame with a move, every frame above it takes the current state, the state is
// updated and the search continues below that frame: ~ (#moves + T/64) wave steps per utterance.
#include <hip/hip_runtime.h>

#include "bfa_math.hpp"
#include "bfa_types.hpp"

namespace bfa {

__device__ __forceinline__ void wave_sync_lds()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(64) void k_backtrace(AlignArgs a)
{
    __shared__ uint32_t sbp[16 * 64]; // one chunk: (16/W) quads x W x (nl <= 64) dwords
    const int lane = threadIdx.x & 63;
    const DevParams &p = a.p;
    const int n_items = a.counters[0];
    for (int i = blockIdx.x; i < n_items; i += gridDim.x) {
        const Item it = a.items[i];
        const int b = it.utt;
        int32_t *oph = a.frame_ph + (int64_t)b * a.Tmax;
        int32_t *oid = a.frame_idx + (int64_t)b * a.Tmax;
        const int32_t *tok = a.tokens + (int64_t)b * a.Smax + it.tok0;
        if (it.kind == ITEM_FILL_BLANK) {
            for (int t = lane; t < it.nout; t += 64) { oph[it.out0 + t] = p.blank; oid[it.out0 + t] = -1; }
            continue;
        }
        if (it.kind == ITEM_FILL_PROP) { // forced_alignment.py:170-172
            for (int t = lane; t < it.nout; t += 64) {
                const int fi = (int)(((int64_t)t * it.nt) / it.nout);
                oph[it.out0 + t] = tok[fi]; oid[it.out0 + t] = it.tok0 + fi;
            }
            continue;
        }
        if (it.kind == ITEM_FILL_SIL) { // forced_alignment.py:382-397
            const double fps = (it.nt > 0) ? (double)it.nout / (double)it.nt : 0.0;
            for (int t = lane; t < it.nout; t += 64) {
                int id = -1;
                if (it.nt > 0) { // the k with int(k*fps) <= t < int((k+1)*fps); the ranges are disjoint
                    const int k = (int)((double)t / fps);
                    for (int kk = max(0, k - 1); kk <= min(it.nt - 1, k + 1); ++kk) {
                        const int f0 = (int)((double)kk * fps), f1 = (int)((double)(kk + 1) * fps);
                        if (t >= f0 && t < f1) id = it.tok0 + kk;
                    }
                }
                oph[it.out0 + t] = p.sil; oid[it.out0 + t] = id;
            }
            continue;
        }
        if (it.kind != ITEM_DP) continue;

        // window items (Item::win = Rw): [nq] window bases, then [quad][64 lanes] dwords; lane l, slot r of quad q
        // holds state base[q] + l*Rw + r (DpCoreW in bfa_dp3.inc)
        const bool win = it.win > 0;
        const int R = win ? it.win : r_class_for_L(it.L);
        const int W = bp_words_for_R(R);
        const int Ts = it.Ts, L = it.L;
        const int nl = win ? 64 : bp_lanes(L, R);
        const uint32_t *bp_base = a.bp + it.bp_off;
        const uint32_t *bp = bp_base + (win ? ((Ts + 3) >> 2) : 0);
        const int invR = 65536 / R + 1; // (d * invR) >> 16 == d / R for the d < 64*R + 8 that occur (R <= 4)
        int s = it.final_state;          // wave-uniform walk state
        int sl = s / R, sr = s - sl * R; // its (lane, register slot)
        // chunk = CQ quads (4*CQ frames, <= 64) = CQ*W*nl <= 1024 dwords, i.e. <= 16 dwords per lane
        const int CQ = 16 / W;
        const int CF = 4 * CQ;
        const int nchunks = (Ts + CF - 1) / CF;
        uint32_t pre[16]; // next chunk, in flight in registers while the current one is walked
        auto fetch = [&](int c) {
            const int q0 = c * CQ;
            const int q1 = min((Ts + 3) >> 2, q0 + CQ);
            const int ndw = (q1 - q0) * W * nl;
            const uint32_t *src = bp + (int64_t)q0 * W * nl;
#pragma unroll
            for (int d = 0; d < 16; ++d) {
                const int idx = d * 64 + lane;
                pre[d] = (idx < ndw) ? src[idx] : 0u;
            }
        };
        fetch(nchunks - 1);
        for (int c = nchunks - 1; c >= 0; --c) {
            const int t0 = c * CF;
            const int t1 = min(Ts, t0 + CF);
            const int q0 = t0 >> 2;
            wave_sync_lds();
#pragma unroll
            for (int d = 0; d < 16; ++d) sbp[d * 64 + lane] = pre[d];
            wave_sync_lds();
            if (c > 0) fetch(c - 1);

            const int t = t0 + lane;    // this lane's frame
            const int qrow = ((t >> 2) - q0) * W;
            const bool mine = t < t1;
            const int wbase = (win && mine) ? (int)bp_base[t >> 2] : 0; // window base of this lane's quad
            int my_state = 0;
            int t_hi = t1 - 1;          // frames (.., t_hi] still to be labelled in this chunk
            while (t_hi >= t0) {
                // backpointer code of this lane's frame for the CURRENT state (state[t-1] = s - k at frame t)
                uint32_t k = 0;
                if (mine && t <= t_hi && t > 0) {
                    // pair (A,B) of (frame t&3, slot sr) in dword sr>>2 (layout: DpCore in bfa_dp3.inc / bfa_dp.inc)
                    int xl = sl, xr = sr;
                    bool inw = true;
                    if (win) { // a frame at which s lies outside the window is below the next move: its code is not used
                        const int d = s - wbase;
                        inw = d >= 0 && d < 64 * R;
                        xl = (d * invR) >> 16; xr = d - xl * R;
                    }
                    if (inw) {
                        const uint32_t wd = sbp[(qrow + (xr >> 2)) * nl + xl];
                        const int Rw = min(4, R - 4 * (xr >> 2));
                        const uint32_t code = (wd >> (2 * (4 * Rw - 1 - ((t & 3) * Rw + (xr & 3))))) & 3u;
                        k = (code >= 2u) ? code - 1u : 0u; // A ? (B ? 2 : 1) : 0
                    }
                }
                const unsigned long long mv = __ballot(k != 0);
                if (mv == 0) { // the path stays in s down to the chunk start
                    if (mine && t <= t_hi) my_state = s;
                    t_hi = t0 - 1;
                    break;
                }
                const int jl = 63 - __builtin_clzll(mv); // latest frame (lane) at which the path moves
                if (mine && t <= t_hi && lane >= jl) my_state = s;
                const int kk = (int)__builtin_amdgcn_readlane((int)k, jl);
                s -= kk; sr -= kk;
                while (sr < 0) { sr += R; sl -= 1; }
                if (s < 0) { s += L; sl = s / R; sr = s - sl * R; } // python negative-index wrap (:692)
                t_hi = t0 + jl - 1;
            }
            if (mine) {
                const int o = t - it.pad_left; // :447-448 trim the boundary padding
                if (o >= 0 && o < it.nout) {
                    int ph = p.blank, id = -1;
                    if (my_state >= 1) {
                        const int q = (my_state - 1) / it.stride;
                        if ((my_state - 1) - q * it.stride == 0 && q < it.nt) { ph = tok[q]; id = it.tok0 + q; }
                    }
                    oph[it.out0 + o] = ph;
                    oid[it.out0 + o] = id;
                }
            }
        }
    }
}

} // namespace bfa

extern "C" void bfa_launch_backtrace(const bfa::AlignArgs *args, int grid, hipStream_t stream)
{
    hipLaunchKernelGGL(bfa::k_backtrace, dim3(grid), dim3(64), 0, stream, *args);
}
