// tools/exp/hop_union.hip -- EXPERIMENT (end of round 4).  Ran once: output bit-identical to dc_spmm_bf16 on both
// adjacencies of the 100k-point graph; 93 - 98 us per hop against 53 - 55 us for dc_spmm_bf16 on the same box (every
// workgroup runs its phases serially: ~5.8 us of chained latencies); lists 65 / 99 us per side.  DESIGN.md section 8.
//
// The bf16 hop of BASELINE configs[4] (100k-point radius graph, Morton order) gathers E x 512 B = 572 MB of neighbour
// rows per launch from the L2s and runs at ~72 % of the L2 -> CU rate.  Consecutive Morton
// rows share most of their neighbours (tools/exp/neighbour_sharing.py: runs of 32 rows need 3.7 x fewer distinct rows
// than edges, 4.7 x in the dense blob; the largest union of a 32-row run is 313 rows).  This kernel pair stages the
// UNION of a run's neighbour rows in LDS once and lets the run's rows gather from LDS:
//
//   hu_build   one workgroup per run of kRun rows: the run's `other` ids sorted and de-duplicated in LDS ->
//              ulist[run * kUMax ..] (global row ids, ascending), ucnt[run]; lidx[slot] = position of other[slot] in
//              that list (binary search).  A run with more than kSlotMax slots or kUMax distinct rows sets *status.
//   hu_hop     grid (runs, 4 column parts): the part-rows (64 bf16 = 128 B at F = 256) of the run's union go
//              global -> LDS by LDS-DMA with one row address per 8 lanes (8 rows per wave-instruction), then 8 lanes
//              per destination row walk the row's slots in p order: fp32(w[p]) * fp32(row[lidx[p]]) summed in fp32,
//              rounded once to bf16 - the arithmetic of k_spmm_bf16x8 (dc_spmm.hip), so y must be BIT-IDENTICAL to
//              dc_spmm_bf16 (y_is_f32 = 0, no addend).
//
// LDS: kUMax x 128 B = 40 KiB of rows + 12 KiB of list positions / weights per workgroup (three per CU).  Driver:
// tools/exp/hop_union.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

constexpr int kRun = 32;          // destination rows per run
constexpr int kSlotMax = 2048;    // slots of a run sorted in LDS
constexpr int kUMax = 320;        // distinct neighbour rows of a run
constexpr int kPartB = 128;       // bytes of a staged part-row (F = 256 bf16 in kParts column parts)
constexpr int kParts = 4;

__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// ---- per-run neighbour unions -----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
hu_build(const int32_t *ptr, const int32_t *other, int64_t N, int32_t *ulist, int32_t *ucnt, uint16_t *lidx,
         int32_t *status) {
    __shared__ int32_t keys[kSlotMax];
    __shared__ int32_t uniq[kUMax];
    __shared__ int32_t wsum[4], nuniq;
    const int run = blockIdx.x, tid = threadIdx.x;
    const int64_t r0 = (int64_t)run * kRun, r1 = r0 + kRun < N ? r0 + kRun : N;
    const int32_t beg = ptr[r0], m = ptr[r1] - beg;
    if (m > kSlotMax) {
        if (tid == 0) atomicOr(status, 1), ucnt[run] = 0;
        return;
    }
    int P = 2;
    while (P < m) P <<= 1;
    for (int i = tid; i < P; i += 256) keys[i] = i < m ? other[beg + i] : 0x7fffffff;
    __syncthreads();
    // bitonic sort of keys[0..P) (padding = +inf), plain ascending network
    for (int size = 2; size <= P; size <<= 1)
        for (int stride = size >> 1; stride >= 1; stride >>= 1) {
            for (int t = tid; t < (P >> 1); t += 256) {
                const int i = 2 * stride * (t / stride) + (t % stride), j = i + stride;
                const bool up = ((i & size) == 0);
                const int32_t a = keys[i], c = keys[j];
                if ((a > c) == up) keys[i] = c, keys[j] = a;
            }
            __syncthreads();
        }
    // distinct values, in order: flag + block scan (8 elements per thread)
    int flag[8], sum = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = 8 * tid + j;
        flag[j] = (i < m && (i == 0 || keys[i] != keys[i - 1])) ? 1 : 0;
        sum += flag[j];
    }
    int inc = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(inc, d, 64);
        if ((tid & 63) >= d) inc += t;
    }
    if ((tid & 63) == 63) wsum[tid >> 6] = inc;
    __syncthreads();
    int off = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w < (tid >> 6)) off += wsum[w];
        tot += wsum[w];
    }
    if (tid == 0) nuniq = tot;
    if (tot > kUMax) {
        if (tid == 0) atomicOr(status, 2), ucnt[run] = 0;
        return;
    }
    int pos = off + inc - sum;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = 8 * tid + j;
        if (flag[j]) uniq[pos++] = keys[i];
    }
    __syncthreads();
    for (int i = tid; i < tot; i += 256) ulist[(int64_t)run * kUMax + i] = uniq[i];
    if (tid == 0) ucnt[run] = tot;
    // position of every slot's neighbour in the list (lower bound; the value is present)
    for (int i = tid; i < m; i += 256) {
        const int32_t v = other[beg + i];
        int lo = 0, hi = tot - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (uniq[mid] < v) lo = mid + 1;
            else hi = mid;
        }
        lidx[beg + i] = (uint16_t)lo;
    }
}

// ---- the hop ------------------------------------------------------------------------------------------------------------
// grid (runs, kParts column parts of 64 bf16 = 128 B); 256 threads = 8 lanes per destination row, the whole run at once.
// Everything the inner loop touches is in LDS: the union's part-rows (40 KiB), the run's list positions and weights
// (12 KiB), its row offsets - three workgroups per CU.  Order of work: (1) positions / weights / offsets by ordinary
// loads + ds_write and the union's row ids into registers, (2) barrier, (3) all LDS-DMAs of the wave back to back
// (row ids already in registers: nothing between two DMAs waits), (4) vmcnt(0) + barrier, (5) the sums.
__global__ void __launch_bounds__(256)
hu_hop(const int32_t *__restrict__ ptr, const float *__restrict__ w, const uint16_t *__restrict__ lidx,
       const int32_t *__restrict__ ulist, const int32_t *__restrict__ ucnt, const uint16_t *__restrict__ x, int64_t ldx,
       uint16_t *__restrict__ y, int64_t ldy, int64_t N) {
    __shared__ __attribute__((aligned(16))) char rows[kUMax * kPartB];
    __shared__ float s_w[kSlotMax];
    __shared__ uint16_t s_li[kSlotMax];
    __shared__ int32_t s_ptr[kRun + 1];
    const int run = blockIdx.x, part = blockIdx.y, tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int nu = ucnt[run];
    const int32_t *ul = ulist + (int64_t)run * kUMax;
    const int64_t r0 = (int64_t)run * kRun, r1 = r0 + kRun < N ? r0 + kRun : N;
    const int32_t sbeg = ptr[r0], m = ptr[r1] - sbeg;
    for (int i = tid; i < m; i += 256) {
        s_li[i] = lidx[sbeg + i];
        s_w[i] = w ? w[sbeg + i] : 1.0f;
    }
    if (tid <= kRun) s_ptr[tid] = ptr[r0 + tid < r1 ? r0 + tid : r1] - sbeg;
    // wave-instruction g fills part-rows 8 g .. 8 g + 7 (8 lanes x 16 B each); wave wid takes g = wid, wid + 4, ...
    constexpr int kPerWave = kUMax / 8 / 4;
    const int sub = lane & 7;
    int32_t rid[kPerWave];
    const int last = nu > 0 ? nu - 1 : 0;
#pragma unroll
    for (int k = 0; k < kPerWave; ++k) {                           // (ulist holds kUMax entries per run: any index is readable)
        const int r = 8 * (wid + 4 * k) + (lane >> 3);
        rid[k] = ul[r < last ? r : last];                          // the tail of the last group re-reads the last row
    }
    __syncthreads();
    __builtin_amdgcn_s_waitcnt(0x0F70);                            // vmcnt(0): the row ids are in their registers (else hipcc
#pragma unroll                                                     // waits for them - and for the DMAs issued so far - one by one)
    for (int k = 0; k < kPerWave; ++k) {
        const int g = wid + 4 * k;
        if (8 * g < nu)
            __builtin_amdgcn_global_load_lds(
                (const void __attribute__((address_space(1))) *)(x + (int64_t)rid[k] * ldx + part * (kPartB / 2) + 8 * sub),
                (void __attribute__((address_space(3))) *)(rows + g * 1024), 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                            // vmcnt(0): this wave's pieces have landed
    __syncthreads();
    const int rr = tid >> 3;                                       // destination row of the run (8 lanes each)
    const int64_t row = r0 + rr;
    if (row >= N) return;
    const int beg = s_ptr[rr], end = s_ptr[rr + 1];
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.0f;
    for (int p = beg; p < end; p += 4) {                           // four slots' LDS reads in flight, summed in p order
        float wv[4];
        uint4 q[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pj = p + j < end ? p + j : end - 1;
            wv[j] = s_w[pj];
            q[j] = *reinterpret_cast<const uint4 *>(rows + (int)s_li[pj] * kPartB + 16 * (tid & 7));
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (p + j < end) {
                const uint32_t d[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float m0 = wv[j] * __uint_as_float(d[i] << 16);
                    acc[2 * i] = acc[2 * i] + m0;
                    const float m1 = wv[j] * __uint_as_float(d[i] & 0xffff0000u);
                    acc[2 * i + 1] = acc[2 * i + 1] + m1;
                }
            }
    }
    uint4 o;
    o.x = f32_to_bf16_rne(acc[0]) | ((uint32_t)f32_to_bf16_rne(acc[1]) << 16);
    o.y = f32_to_bf16_rne(acc[2]) | ((uint32_t)f32_to_bf16_rne(acc[3]) << 16);
    o.z = f32_to_bf16_rne(acc[4]) | ((uint32_t)f32_to_bf16_rne(acc[5]) << 16);
    o.w = f32_to_bf16_rne(acc[6]) | ((uint32_t)f32_to_bf16_rne(acc[7]) << 16);
    *reinterpret_cast<uint4 *>(y + row * ldy + part * (kPartB / 2) + 8 * (tid & 7)) = o;
}

extern "C" int hu_run_max() { return kUMax; }
extern "C" int hu_run_rows() { return kRun; }

// lists for the adjacency (ptr, other) of N rows: ulist [runs * kUMax], ucnt [runs], lidx [E]; *status: bit 0 a run has
// more than kSlotMax slots, bit 1 more than kUMax distinct neighbours (the caller then keeps the plain hop)
extern "C" int hu_build_lists(const int32_t *ptr, const int32_t *other, int64_t N, int32_t *ulist, int32_t *ucnt,
                              uint16_t *lidx, int32_t *status, void *stream) {
    const unsigned runs = (unsigned)((N + kRun - 1) / kRun);
    (void)hipMemsetAsync(status, 0, 4, (hipStream_t)stream);
    hipLaunchKernelGGL(hu_build, dim3(runs), dim3(256), 0, (hipStream_t)stream, ptr, other, N, ulist, ucnt, lidx, status);
    return (int)hipGetLastError();
}

// y[N, 256] (bf16) = sum over slots of w * x[other]: F = 256 only (kParts column parts of 64)
extern "C" int hu_hop_bf16(const int32_t *ptr, const float *w, const uint16_t *lidx, const int32_t *ulist,
                           const int32_t *ucnt, const uint16_t *x, int64_t ldx, uint16_t *y, int64_t ldy, int64_t N,
                           void *stream) {
    const unsigned runs = (unsigned)((N + kRun - 1) / kRun);
    hipLaunchKernelGGL(hu_hop, dim3(runs, kParts), dim3(256), 0, (hipStream_t)stream, ptr, w, lidx, ulist, ucnt, x, ldx, y, ldy, N);
    return (int)hipGetLastError();
}
