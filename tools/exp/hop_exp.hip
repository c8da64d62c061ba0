// tools/exp/hop_exp.hip -- experimental variants of the F=256 fp32 hop (NOT part of the product
// library): built into tools/exp/libhopexp.so and driven by tools/exp/hop_exp.py to decide what
// goes into csrc/dc_spmm.hip.  All variants compute y[i,:] = sum_p w[p] * x[other[p],:] with the
// product's rounding (separate mul / add, p order), F = 256, 16-byte aligned rows.
#include <hip/hip_runtime.h>
#include <stdint.h>

#pragma clang fp contract(off)

static __device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
    const unsigned q = nblk / 8, r = nblk % 8;
    const unsigned xcd = bid % 8, idx = bid / 8;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

static __device__ __forceinline__ void axpy(float4 &a, float w, const float4 &v) {
    const float mx = w * v.x, my = w * v.y, mz = w * v.z, mw = w * v.w;
    a.x = a.x + mx; a.y = a.y + my; a.z = a.z + mz; a.w = a.w + mw;
}

template <bool NT>
static __device__ __forceinline__ void store4(float *p, const float4 &v) {
    if (NT) {
        __builtin_nontemporal_store(v.x, p);
        __builtin_nontemporal_store(v.y, p + 1);
        __builtin_nontemporal_store(v.z, p + 2);
        __builtin_nontemporal_store(v.w, p + 3);
    } else {
        *reinterpret_cast<float4 *>(p) = v;
    }
}

// ---- copy floor: same geometry as the product kernel, y[row] = x[row] ----
__global__ void __launch_bounds__(256)
k_copy_rows(const float *__restrict__ x, int64_t ldx, float *y, int64_t ldy, int64_t N) {
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t row = lb * 4u + (threadIdx.x >> 6);
    if (row >= N) return;
    const int c = (threadIdx.x & 63) * 4;
    *reinterpret_cast<float4 *>(y + row * ldy + c) = *reinterpret_cast<const float4 *>(x + row * ldx + c);
}

// ---- chunked: each wave walks RW consecutive rows, WPB waves per block; the index loads of the
// next row are issued before the current row's gathers are consumed ----
template <int RW, int WPB, bool NT, int U>
__global__ void __launch_bounds__(WPB * 64)
k_hop_chunk(const int32_t *__restrict__ ptr, const int32_t *__restrict__ other,
            const float *__restrict__ w, const float *__restrict__ x, int64_t ldx, float *y,
            int64_t ldy, int64_t N) {
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t row0 = ((int64_t)lb * WPB + wave) * RW;
    if (row0 >= N) return;
    const int c = (threadIdx.x & 63) * 4;
    const int nrow = (int)((N - row0) < RW ? (N - row0) : RW);
    int beg = ptr[row0], end = ptr[row0 + 1];
    for (int r = 0; r < nrow; ++r) {
        const int64_t row = row0 + r;
        int nbeg = 0, nend = 0;
        if (r + 1 < nrow) { nbeg = end; nend = ptr[row + 2]; }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int p = beg; p < end; p += U) {
            const int n = end - p;
            int s[U]; float ww[U]; float4 v[U];
#pragma unroll
            for (int j = 0; j < U; ++j) if (j < n) { s[j] = other[p + j]; ww[j] = w[p + j]; }
#pragma unroll
            for (int j = 0; j < U; ++j) if (j < n) v[j] = *reinterpret_cast<const float4 *>(x + (int64_t)s[j] * ldx + c);
#pragma unroll
            for (int j = 0; j < U; ++j) if (j < n) axpy(acc, ww[j], v[j]);
        }
        store4<NT>(y + row * ldy + c, acc);
        beg = nbeg; end = nend;
    }
}

// ---- pipelined: the gathers of row r+1 are issued BEFORE row r is reduced and stored (two
// register sets); rows with more than U neighbours fall back to the plain loop for the rest ----
template <int RW, int WPB, bool NT, int U>
__global__ void __launch_bounds__(WPB * 64)
k_hop_pipe(const int32_t *__restrict__ ptr, const int32_t *__restrict__ other,
           const float *__restrict__ w, const float *__restrict__ x, int64_t ldx, float *y,
           int64_t ldy, int64_t N) {
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t row0 = ((int64_t)lb * WPB + wave) * RW;
    if (row0 >= N) return;
    const int c = (threadIdx.x & 63) * 4;
    const int nrow = (int)((N - row0) < RW ? (N - row0) : RW);
    float4 va[U], vb[U];
    float wa[U], wb[U];
    int bega, enda, begb = 0, endb = 0;
    auto issue = [&](int beg, int end, float4 (&v)[U], float (&ww)[U]) {
        const int n = end - beg;
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const bool ok = j < n;
            const int s = ok ? other[beg + j] : 0;
            ww[j] = ok ? w[beg + j] : 0.0f;
            if (ok) v[j] = *reinterpret_cast<const float4 *>(x + (int64_t)s * ldx + c);
        }
    };
    auto finish = [&](int64_t row, int beg, int end, float4 (&v)[U], float (&ww)[U]) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const int n = end - beg;
#pragma unroll
        for (int j = 0; j < U; ++j) if (j < n) axpy(acc, ww[j], v[j]);
        for (int p = beg + U; p < end; ++p) {          // long rows: the tail, one at a time
            const float4 t = *reinterpret_cast<const float4 *>(x + (int64_t)other[p] * ldx + c);
            axpy(acc, w[p], t);
        }
        store4<NT>(y + row * ldy + c, acc);
    };
    bega = ptr[row0]; enda = ptr[row0 + 1];
    issue(bega, enda, va, wa);
    for (int r = 0; r < nrow; r += 2) {
        if (r + 1 < nrow) { begb = enda; endb = ptr[row0 + r + 2]; issue(begb, endb, vb, wb); }
        finish(row0 + r, bega, enda, va, wa);
        if (r + 1 >= nrow) break;
        if (r + 2 < nrow) { bega = endb; enda = ptr[row0 + r + 3]; issue(bega, enda, va, wa); }
        finish(row0 + r + 1, begb, endb, vb, wb);
    }
}

// ---- software-pipelined INDEX stream: a wave walks RW consecutive rows; while the gathers of row r
// are in flight (vmcnt) the segment bounds of row r+2 and the neighbour ids / weights of row r+1 are
// fetched by scalar loads (lgkmcnt - a different counter, so waiting for one does not wait for the
// other).  The dependent chain per row shrinks from bounds -> ids -> gathers -> store to gathers -> store.
template <int RW, bool NT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(80)))
k_hop_swp(const int32_t *__restrict__ ptr, const int32_t *__restrict__ other,
          const float *__restrict__ w, const float *__restrict__ x, int64_t ldx, float *y,
          int64_t ldy, int64_t N) {
    constexpr int U = 8;
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t row0 = ((int64_t)lb * 4 + wave) * RW;
    if (row0 >= N) return;
    const int c = (threadIdx.x & 63) * 4;
    const int nrow = (int)((N - row0) < RW ? (N - row0) : RW);
    int beg = ptr[row0], end = ptr[row0 + 1];
    int s[U], sn[U];
    float ww[U], wn[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const bool ok = beg + j < end;
        s[j] = ok ? other[beg + j] : 0;
        ww[j] = ok ? w[beg + j] : 0.f;
    }
    int nend = nrow > 1 ? ptr[row0 + 2] : end;
    for (int r = 0; r < nrow; ++r) {
        const int64_t row = row0 + r;
        const int n = end - beg;
        float4 v[U];
#pragma unroll
        for (int j = 0; j < U; ++j)
            if (j < n) v[j] = *reinterpret_cast<const float4 *>(x + (int64_t)s[j] * ldx + c);
        // next row's ids / weights and the bounds after it, while those gathers fly
        const int nbeg = end;
        int nnend = nend;
        if (r + 1 < nrow) {
#pragma unroll
            for (int j = 0; j < U; ++j) {
                const bool ok = nbeg + j < nend;
                sn[j] = ok ? other[nbeg + j] : 0;
                wn[j] = ok ? w[nbeg + j] : 0.f;
            }
            if (r + 2 < nrow) nnend = ptr[row + 3];
        }
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int j = 0; j < U; ++j)
            if (j < n) axpy(acc, ww[j], v[j]);
        for (int p = beg + U; p < end; ++p) {          // long rows: the tail, one at a time
            const float4 t = *reinterpret_cast<const float4 *>(x + (int64_t)other[p] * ldx + c);
            axpy(acc, w[p], t);
        }
        store4<NT>(y + row * ldy + c, acc);
        beg = nbeg;
        end = nend;
        nend = nnend;
#pragma unroll
        for (int j = 0; j < U; ++j) { s[j] = sn[j]; ww[j] = wn[j]; }
    }
}

// ---- packed {idx, w} records: ONE scalar load brings the 8 records of a row's first chunk.
// `rec` holds E' + 8 records (8 zero records of padding), so the load needs no bounds logic.
// ELL = true: rec8[row*8 .. row*8+8) is addressed by the row alone (no dependence on ptr): the
// ptr pair and the records are fetched together, one scalar latency instead of two; rows longer
// than 8 continue in the CSR arrays.  CLAMP: all 8 gathers are issued unconditionally (slots past
// the end re-read neighbour 0), their results dropped by a select.
struct Rec { int idx; float w; };

template <bool ELL, bool CLAMP, bool NT>
__global__ void __launch_bounds__(256)
k_hop_rec(const int32_t *__restrict__ ptr, const Rec *__restrict__ rec,
          const int32_t *__restrict__ other, const float *__restrict__ w,
          const float *__restrict__ x, int64_t ldx, float *y, int64_t ldy, int64_t N) {
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t row = __builtin_amdgcn_readfirstlane((int)(lb * 4u + (threadIdx.x >> 6)));
    if (row >= N) return;
    const int c = (threadIdx.x & 63) * 4;
    const Rec *r = ELL ? rec + row * 8 : nullptr;
    Rec q[8];
    if (ELL) {
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = r[j];
    }
    const int beg = ptr[row], end = ptr[row + 1];
    if (!ELL) {
        r = rec + beg;
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = r[j];
    }
    const int n = end - beg;
    float4 v[8];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (CLAMP) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int s = j < n ? q[j].idx : q[0].idx;
            v[j] = *reinterpret_cast<const float4 *>(x + (int64_t)s * ldx + c);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float4 t = acc;
            axpy(t, q[j].w, v[j]);
            if (j < n) acc = t;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < n) v[j] = *reinterpret_cast<const float4 *>(x + (int64_t)q[j].idx * ldx + c);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < n) axpy(acc, q[j].w, v[j]);
    }
    for (int p = beg + 8; p < end; ++p) {            // long rows: the tail from the CSR arrays
        const float4 t = *reinterpret_cast<const float4 *>(x + (int64_t)other[p] * ldx + c);
        axpy(acc, w[p], t);
    }
    store4<NT>(y + row * ldy + c, acc);
}

#define LAUNCH(K, grid, threads) hipLaunchKernelGGL(K, dim3(grid), dim3(threads), 0, (hipStream_t)stream, ptr, other, w, x, ldx, y, ldy, N)

// ---- windowed: a 1024-thread workgroup owns R destination rows x ONE column half (128 floats = 512 B per row)
// and first stages the rows [r0 - H, r0 + R + H) of that half into LDS by LDS-DMA (1 KiB per wave instruction =
// two staged half rows); neighbours inside the window are then read from LDS (ds_read_b64), the others from
// global memory as before.  Meshes whose numbering is local (|src - dst| <= H for most edges) turn E x 1 KiB of
// L2 gathers per launch into (R + 2H) / R staged rows.  Wave-uniform neighbour ids -> the hit test is scalar.
static __device__ __forceinline__ void axpy2(float2 &a, float w, const float2 &v) {
    const float mx = w * v.x, my = w * v.y;
    a.x = a.x + mx; a.y = a.y + my;
}

template <int U>
__global__ void __launch_bounds__(1024)
k_hop_win(const int32_t *__restrict__ ptr, const int32_t *__restrict__ other, const float *__restrict__ w,
          const float *__restrict__ x, int64_t ldx, float *y, int64_t ldy, int64_t N, int R, int H) {
    extern __shared__ __attribute__((aligned(16))) char win[];
    const unsigned lb = xcd_remap(blockIdx.x, gridDim.x);
    const int ch = (int)(lb & 1u);                                   // column half
    const int64_t r0 = (int64_t)(lb >> 1) * R;
    if (r0 >= N) return;
    const int64_t r1 = r0 + R < N ? r0 + R : N;
    const int64_t w0 = r0 - H > 0 ? r0 - H : 0, w1 = r1 + H < N ? r1 + H : N;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const float *xc = x + ch * 128;
    // stage: DMA instruction j moves window rows 2j, 2j+1 (this half) to win + 1024 j
    const int npair = (int)((w1 - w0 + 1) >> 1);
    for (int j = wave; j < npair; j += 16) {
        int64_t row = w0 + 2 * j + (lane >> 5);
        row = row < w1 ? row : w1 - 1;
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1))) *)(xc + row * ldx + (lane & 31) * 4),
                                         (void __attribute__((address_space(3))) *)(win + 1024 * j), 16, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0xF70 | 0);                            // vmcnt(0)
    __syncthreads();
    const int iw0 = (int)w0, iw1 = (int)w1;
    for (int64_t row = r0 + wave; row < r1; row += 16) {
        const int beg = ptr[row], end = ptr[row + 1];
        float2 acc = make_float2(0.f, 0.f);
        for (int p = beg; p < end; p += U) {
            const int n = end - p;
            int s[U]; float ww[U]; float2 v[U];
#pragma unroll
            for (int j = 0; j < U; ++j) if (j < n) { s[j] = other[p + j]; ww[j] = w[p + j]; }
#pragma unroll
            for (int j = 0; j < U; ++j)
                if (j < n) {
                    if (s[j] >= iw0 && s[j] < iw1)
                        v[j] = *reinterpret_cast<const float2 *>(win + (s[j] - iw0) * 512 + lane * 8);
                    else
                        v[j] = *reinterpret_cast<const float2 *>(xc + (int64_t)s[j] * ldx + lane * 2);
                }
#pragma unroll
            for (int j = 0; j < U; ++j) if (j < n) axpy2(acc, ww[j], v[j]);
        }
        *reinterpret_cast<float2 *>(y + row * ldy + ch * 128 + lane * 2) = acc;
    }
}

extern "C" int hopexp_run_win(int R, int H, const int32_t *ptr, const int32_t *other, const float *w,
                              const float *x, int64_t ldx, float *y, int64_t ldy, int64_t N, void *stream) {
    const size_t lds = (size_t)((R + 2 * H + 1) / 2) * 1024;
    static size_t set = 0;
    if (lds > set) {
        if (hipFuncSetAttribute((const void *)k_hop_win<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return 2;
        set = lds;
    }
    const unsigned grid = (unsigned)(2 * ((N + R - 1) / R));
    hipLaunchKernelGGL((k_hop_win<8>), dim3(grid), dim3(1024), lds, (hipStream_t)stream, ptr, other, w, x, ldx, y,
                       ldy, N, R, H);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

extern "C" int hopexp_run_rec(int variant, const int32_t *ptr, const void *rec, const int32_t *other,
                              const float *w, const float *x, int64_t ldx, float *y, int64_t ldy,
                              int64_t N, void *stream) {
    const unsigned grid = (unsigned)((N + 3) / 4);
#define LR(K) hipLaunchKernelGGL(K, dim3(grid), dim3(256), 0, (hipStream_t)stream, ptr, (const Rec *)rec, other, w, x, ldx, y, ldy, N)
    switch (variant) {
    case 20: LR((k_hop_rec<false, false, true>)); break;
    case 21: LR((k_hop_rec<false, true, true>)); break;
    case 22: LR((k_hop_rec<true, false, true>)); break;
    case 23: LR((k_hop_rec<true, true, true>)); break;
    case 24: LR((k_hop_rec<true, false, false>)); break;
    default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int hopexp_run(int variant, const int32_t *ptr, const int32_t *other, const float *w,
                          const float *x, int64_t ldx, float *y, int64_t ldy, int64_t N, void *stream) {
    auto blocks = [&](int rw, int wpb) { return (unsigned)((N + (int64_t)rw * wpb - 1) / ((int64_t)rw * wpb)); };
    switch (variant) {
    case 0: hipLaunchKernelGGL(k_copy_rows, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, ldx, y, ldy, N); break;
    case 1: LAUNCH((k_hop_chunk<1, 4, false, 8>), blocks(1, 4), 256); break;
    case 2: LAUNCH((k_hop_chunk<1, 4, true, 8>), blocks(1, 4), 256); break;
    case 3: LAUNCH((k_hop_chunk<2, 4, false, 8>), blocks(2, 4), 256); break;
    case 4: LAUNCH((k_hop_chunk<4, 4, false, 8>), blocks(4, 4), 256); break;
    case 5: LAUNCH((k_hop_chunk<8, 4, false, 8>), blocks(8, 4), 256); break;
    case 6: LAUNCH((k_hop_chunk<4, 8, false, 8>), blocks(4, 8), 512); break;
    case 7: LAUNCH((k_hop_chunk<1, 16, false, 8>), blocks(1, 16), 1024); break;
    case 8: LAUNCH((k_hop_pipe<4, 4, false, 8>), blocks(4, 4), 256); break;
    case 9: LAUNCH((k_hop_pipe<8, 4, false, 8>), blocks(8, 4), 256); break;
    case 10: LAUNCH((k_hop_pipe<16, 4, false, 8>), blocks(16, 4), 256); break;
    case 11: LAUNCH((k_hop_pipe<8, 4, true, 8>), blocks(8, 4), 256); break;
    case 12: LAUNCH((k_hop_pipe<4, 8, false, 8>), blocks(4, 8), 512); break;
    case 13: LAUNCH((k_hop_pipe<8, 2, false, 8>), blocks(8, 2), 128); break;
    case 14: LAUNCH((k_hop_chunk<1, 4, false, 4>), blocks(1, 4), 256); break;
    case 15: LAUNCH((k_hop_pipe<4, 4, false, 6>), blocks(4, 4), 256); break;
    case 16: LAUNCH((k_hop_swp<2, false>), blocks(2, 4), 256); break;
    case 17: LAUNCH((k_hop_swp<4, false>), blocks(4, 4), 256); break;
    case 18: LAUNCH((k_hop_swp<8, false>), blocks(8, 4), 256); break;
    case 19: LAUNCH((k_hop_swp<16, false>), blocks(16, 4), 256); break;
    default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
