// bfa_backtrace.hip -- K2: backtrace over the packed backpointers + framewise outputs
// (forced_alignment.py:686-700), plus the non-DP fills (proportional :170-172, silence :382-397).
//
// One wavefront per work item.  The serial dependence state[t-1] = bp[t][state[t]] is only a real
// dependence at frames where the path moves (k != 0): about L of the T frames.  So instead of one
// step per frame, all 64 lanes look at 64 consecutive frames at once for the CURRENT state, a ballot
// finds the latest frame with a move, every frame above it takes the current state, the state is
// updated and the search continues below that frame: ~ (#moves + T/64) wave steps per utterance.
//
// Every wave of the launch is resident at once and a walk step is a short dependent chain
// (LDS gather -> extract -> ballot -> readlane), so the kernel time is (steps x step latency) of one
// utterance: the walk is instantiated per backpointer layout (states per lane, window or full) so that the
// per-step address and shift arithmetic folds into a handful of instructions.
#include <hip/hip_runtime.h>


#include "bfa_walk.inc"

namespace bfa {

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_backtrace(AlignArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t sbp[18 * 64]; // one chunk: <= 1024 dwords (dword layouts) / 9 qwords per lane (staged masks)
    __shared__ int32_t stok[1024];    // the item's tokens (nt <= L <= 1024)
    backtrace_body<false>(a, sbp, stok);
}

__global__ __launch_bounds__(64) void k_backtrace_wide(AlignArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t sbp[2 * 33 * 64]; // a chunk's 2R lane masks per frame, R <= 16 (+1 qword: bank spread)
    __shared__ int32_t stok[1024];
    backtrace_body<true>(a, sbp, stok);
}

} // namespace bfa

// `wide` : bit 0 = the narrow kernel may find items, bit 1 = the wide one may (see k2_is_wide)
extern "C" void bfa_launch_backtrace(const bfa::AlignArgs *args, int grid, hipStream_t stream, int wide)
{
    if (wide & 1) hipLaunchKernelGGL(bfa::k_backtrace, dim3(grid), dim3(64), 0, stream, *args);
    if (wide & 2) hipLaunchKernelGGL(bfa::k_backtrace_wide, dim3(grid), dim3(64), 0, stream, *args);
}

// one selection of items (see k2_selected); `fused` = also produce the tuples (no K3a launch follows)
extern "C" void bfa_launch_backtrace_sel(const bfa::AlignArgs *args, int sel, int fused, int grid, hipStream_t stream, int wide)
{
    bfa::AlignArgs a = *args;
    a.k2_sel = sel;
    a.k2_fused_rle = fused;
    bfa_launch_backtrace(&a, grid, stream, wide);
}
