// bfa_pack.hip -- the result records of a call as ONE contiguous record set (CSR), and its index on the receiving side.
//
// The alignment leaves its tuples in a padded [n, seg_cap] array (assort_frames, forced_alignment.py:777-834: one list per
// utterance).  What leaves the GPU -- the copy to the host behind decode_alignments (forced_alignment.py:871,908), the final
// gather of a sharded batch -- needs the valid tuples only: k_pack writes them back to back in the
// order of the utterances, with each utterance's global index, count and offset in front (layout: include/bfa.h,
// bfa_pack_results).  One launch, no host knowledge of the counts: a workgroup takes 64 utterances, finds the offset of its
// first one by summing the counts in front of it (at most n int32 from L2), scans its own 64 counts in LDS and copies its
// tuples one per thread, 16-byte loads and stores, the stores contiguous over the whole workgroup.
#include <hip/hip_runtime.h>

#include <stdint.h>

namespace bfa {

constexpr int PACK_HDR = 8;   // header words: n, total, n_cap, tuple_cap, has_conf, overflow, 0, 0
constexpr int PACK_UTT = 64;  // utterances per workgroup

struct PackLayout {
    int64_t gidx, count, offset, tuples, conf, words;
};

__host__ __device__ inline PackLayout pack_layout(int n_cap, int64_t tuple_cap, int has_conf)
{
    PackLayout l;
    const int64_t n4 = ((int64_t)n_cap + 3) & ~(int64_t)3;
    l.gidx = PACK_HDR;
    l.count = l.gidx + n4;
    l.offset = l.count + n4;
    l.tuples = l.offset + n4;
    l.conf = l.tuples + 4 * tuple_cap;
    l.words = l.conf + (has_conf ? ((tuple_cap + 3) & ~(int64_t)3) : 0);
    return l;
}

__device__ inline int wave_sum(int v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

__global__ __launch_bounds__(256) void k_pack(const int4 *__restrict__ segs, int seg_cap, const int32_t *__restrict__ seg_count,
                                              const float *__restrict__ conf, const int32_t *__restrict__ gidx, int gidx_base,
                                              int n, int n_cap, int tuple_cap, int32_t *__restrict__ out)
{
    __shared__ int s_part[4];
    __shared__ int s_off[PACK_UTT + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j0 = blockIdx.x * PACK_UTT;
    const PackLayout l = pack_layout(n_cap, tuple_cap, conf != nullptr);

    // tuples in front of this workgroup's utterances
    int acc = 0;
    const int lim = min(j0, n);
    for (int i = tid; i < lim; i += 256) acc += min(max(seg_count[i], 0), seg_cap);
    acc = wave_sum(acc);
    if (lane == 0) s_part[wave] = acc;
    // this workgroup's counts: exclusive scan over 64 utterances by the first wave
    const int j = j0 + lane;
    int cnt = 0;
    if (wave == 0) {
        cnt = (j < n) ? min(max(seg_count[j], 0), seg_cap) : 0;
        int inc = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(inc, d, 64);
            if (lane >= d) inc += up;
        }
        s_off[lane + 1] = inc;
        if (lane == 0) s_off[0] = 0;
    }
    __syncthreads();
    const int base = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    const int block_total = s_off[PACK_UTT];
    if (wave == 0 && j < n_cap) {
        out[l.gidx + j] = (j < n) ? (gidx ? gidx[j] : gidx_base + j) : -1;
        out[l.count + j] = cnt;
        out[l.offset + j] = base + s_off[lane];
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) { // the last workgroup knows the total
        const int total = base + block_total;
        out[0] = n;
        out[1] = min(total, tuple_cap);
        out[2] = n_cap;
        out[3] = tuple_cap;
        out[4] = conf != nullptr;
        out[5] = total > tuple_cap; // overflow: the record was cut at tuple_cap
        out[6] = 0;
        out[7] = 0;
    }
    // the workgroup's tuples, one per thread: output position q -> utterance by bisection of the 64 offsets
    int4 *__restrict__ otup = (int4 *)(out + l.tuples);
    float *__restrict__ oconf = (float *)(out + l.conf);
    for (int q = tid; q < block_total; q += 256) {
        int lo = 0;
#pragma unroll
        for (int step = PACK_UTT / 2; step >= 1; step >>= 1)
            if (s_off[lo + step] <= q) lo += step;
        const int k = q - s_off[lo];
        const int64_t src = (int64_t)(j0 + lo) * seg_cap + k;
        const int dst = base + q;
        if (dst < tuple_cap) {
            otup[dst] = segs[src];
            if (conf) oconf[dst] = conf[src];
        }
    }
}


// The same record with 8-byte tuples (phoneme, start, end as uint16, target_idx as int16): for the copy to the host behind
// decode_alignments, where the 2.6 MB of a 4096-utterance batch's tuples ARE the call's host-side cost (PCIe).  The caller
// checks that frame counts and ids fit 16 bits.  Word 6 of the header = 1.  Layout as pack_layout with 2 words per tuple.
__host__ __device__ inline PackLayout pack_layout16(int n_cap, int64_t tuple_cap)
{
    PackLayout l = pack_layout(n_cap, 0, 0);
    l.conf = l.tuples + 2 * tuple_cap;
    l.words = (l.conf + 3) & ~(int64_t)3;
    return l;
}

__global__ __launch_bounds__(256) void k_pack16(const int4 *__restrict__ segs, int seg_cap, const int32_t *__restrict__ seg_count,
                                                int n, int n_cap, int tuple_cap, int32_t *__restrict__ out)
{
    __shared__ int s_part[4];
    __shared__ int s_off[PACK_UTT + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j0 = blockIdx.x * PACK_UTT;
    const PackLayout l = pack_layout16(n_cap, tuple_cap);
    int acc = 0;
    const int lim = min(j0, n);
    for (int i = tid; i < lim; i += 256) acc += min(max(seg_count[i], 0), seg_cap);
    acc = wave_sum(acc);
    if (lane == 0) s_part[wave] = acc;
    const int j = j0 + lane;
    int cnt = 0;
    if (wave == 0) {
        cnt = (j < n) ? min(max(seg_count[j], 0), seg_cap) : 0;
        int inc = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(inc, d, 64);
            if (lane >= d) inc += up;
        }
        s_off[lane + 1] = inc;
        if (lane == 0) s_off[0] = 0;
    }
    __syncthreads();
    const int base = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    const int block_total = s_off[PACK_UTT];
    if (wave == 0 && j < n_cap) {
        out[l.gidx + j] = (j < n) ? j : -1;
        out[l.count + j] = cnt;
        out[l.offset + j] = base + s_off[lane];
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 0) {
        const int total = base + block_total;
        out[0] = n; out[1] = min(total, tuple_cap); out[2] = n_cap; out[3] = tuple_cap; out[4] = 0;
        out[5] = total > tuple_cap; out[6] = 1; out[7] = 0;
    }
    uint2 *__restrict__ otup = (uint2 *)(out + l.tuples);
    for (int q = tid; q < block_total; q += 256) {
        int lo = 0;
#pragma unroll
        for (int step = PACK_UTT / 2; step >= 1; step >>= 1)
            if (s_off[lo + step] <= q) lo += step;
        const int k = q - s_off[lo];
        const int4 v = segs[(int64_t)(j0 + lo) * seg_cap + k];
        const int dst = base + q;
        if (dst < tuple_cap)
            otup[dst] = make_uint2(((uint32_t)v.x & 0xffffu) | ((uint32_t)v.y << 16), ((uint32_t)v.z & 0xffffu) | ((uint32_t)v.w << 16));
    }
}

// Receiving side: records [world][words] as k_pack wrote them (one per rank) -> for every global utterance index g its
// owner (which record), offset (first tuple inside that record's tuple section) and count.  Utterances no record names keep
// what the caller put there (owner -1 / count 0).
__global__ __launch_bounds__(256) void k_index_records(const int32_t *__restrict__ rec, int world, int64_t words, int n_total,
                                                       int32_t *__restrict__ owner, int32_t *__restrict__ offset,
                                                       int32_t *__restrict__ count)
{
    const int r = blockIdx.y;
    const int32_t *__restrict__ p = rec + (int64_t)r * words;
    const int n = p[0], n_cap = p[2];
    const PackLayout l = pack_layout(n_cap, p[3], p[4]);
    for (int j = blockIdx.x * 256 + threadIdx.x; j < n; j += gridDim.x * 256) {
        const int g = p[l.gidx + j];
        if (g < 0 || g >= n_total) continue;
        owner[g] = r;
        offset[g] = p[l.offset + j];
        count[g] = p[l.count + j];
    }
}

// The copy ceiling (bfa_profile_copy): float4 copy, grid-stride, 1024 workgroups (four per CU) -- the fastest of the
// variants of tools/ubench/copy_peak.hip on this part (6.0 TB/s over 1 GiB, 5.6 over 3 GiB; unrolled, chunked and
// non-temporal forms and hipMemcpyAsync reach 4.4-5.9).
__global__ __launch_bounds__(256) void k_copy(float4 *__restrict__ dst, const float4 *__restrict__ src, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

} // namespace bfa

extern "C" int bfa_launch_copy(void *dst, const void *src, size_t bytes, void *stream)
{
    const size_t n = bytes / 16;
    hipLaunchKernelGGL(bfa::k_copy, dim3(1024), dim3(256), 0, (hipStream_t)stream, (float4 *)dst, (const float4 *)src, n);
    return (int)hipGetLastError();
}

extern "C" __attribute__((visibility("default"))) int64_t bfa_pack_words(int n_cap, int64_t tuple_cap, int has_conf)
{
    return bfa::pack_layout(n_cap, tuple_cap, has_conf).words;
}

extern "C" int bfa_launch_pack(const int32_t *segs, int seg_cap, const int32_t *seg_count, const float *conf, const int32_t *gidx,
                               int gidx_base, int n, int n_cap, int tuple_cap, int32_t *out, void *stream)
{
    const int grid = (n_cap + bfa::PACK_UTT - 1) / bfa::PACK_UTT;
    hipLaunchKernelGGL(bfa::k_pack, dim3(grid > 0 ? grid : 1), dim3(256), 0, (hipStream_t)stream, (const int4 *)segs, seg_cap,
                       seg_count, conf, gidx, gidx_base, n, n_cap, tuple_cap, out);
    return (int)hipGetLastError();
}

extern "C" __attribute__((visibility("default"))) int64_t bfa_pack16_words(int n_cap, int64_t tuple_cap) { return bfa::pack_layout16(n_cap, tuple_cap).words; }

extern "C" int bfa_launch_pack16(const int32_t *segs, int seg_cap, const int32_t *seg_count, int n, int n_cap, int tuple_cap,
                                 int32_t *out, void *stream)
{
    const int grid = (n_cap + bfa::PACK_UTT - 1) / bfa::PACK_UTT;
    hipLaunchKernelGGL(bfa::k_pack16, dim3(grid > 0 ? grid : 1), dim3(256), 0, (hipStream_t)stream, (const int4 *)segs, seg_cap,
                       seg_count, n, n_cap, tuple_cap, out);
    return (int)hipGetLastError();
}

extern "C" int bfa_launch_index_records(const int32_t *rec, int world, int64_t words, int n_max, int n_total, int32_t *owner,
                                        int32_t *offset, int32_t *count, void *stream)
{
    int bx = (n_max + 255) / 256;
    if (bx < 1) bx = 1;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(bfa::k_index_records, dim3(bx, world), dim3(256), 0, (hipStream_t)stream, rec, world, words, n_total, owner,
                       offset, count);
    return (int)hipGetLastError();
}
