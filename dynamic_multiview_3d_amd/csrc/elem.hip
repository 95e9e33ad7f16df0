// HBM-bound pieces of the train step: warp + bilinear resampler (fwd / grad wrt flow), pixel
// losses, fused TF-Adam, activation fwd/bwd, strided copies.  Built with -ffp-contract=off so the
// fp32 expressions round exactly like the reference's unfused TF kernels (and the numpy oracle).
#include "common.h"
#include "reduce_common.h"
#include <vector>
#include <algorithm>
#include <cstdlib>
#include <atomic>

namespace mv3d {

// ---------------------------------------------------------------- fixed-order loss sums
// A loss kernel ends with one term per workgroup.  They used to meet in a float atomicAdd on the loss word, whose order -- and so
// the last bits of the scalar -- changed from run to run.  Now every workgroup parks its term in its own word of a scratch row
// (an agent-scope atomic exchange: performed at the memory side, visible to every XCD), takes a ticket, and the workgroup that
// draws the last ticket sums the row in index order and adds ONE value to the loss word: launches of a stream add in stream
// order, so the scalar is reproducible bit for bit.  Rows are handed out round-robin per launch (a recorded launch keeps its row).
constexpr int LOSS_ROWS = 32, LOSS_MAXB = 1024;
__device__ unsigned g_loss_part[LOSS_ROWS][LOSS_MAXB];
__device__ unsigned g_loss_ticket[LOSS_ROWS];

// mv3d_loss_overwrite_next(): the next loss call of this thread STORES its sum instead of adding it to the accumulator -- the
// first loss term of a recorded step, which then needs no launch that clears the accumulator (flag = bit 8 of the row).
static thread_local bool g_loss_overwrite = false;
static int next_loss_row() {
    static std::atomic<unsigned> n{0};
    const int flag = g_loss_overwrite ? 256 : 0;
    g_loss_overwrite = false;
    return (int)(n.fetch_add(1) % LOSS_ROWS) | flag;
}

// every thread of a 256-thread workgroup calls this; thread 0 passes the workgroup's term
__device__ __forceinline__ void loss_combine(float term, float* loss, int row_flag) {
    const int row = row_flag & 255;
    __shared__ unsigned s_last;
    __shared__ float s_w[4];
    if (threadIdx.x == 0) {
        (void)__hip_atomic_exchange(&g_loss_part[row][blockIdx.x], __float_as_uint(term), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the exchange has been performed before the ticket is drawn
        const unsigned t = __hip_atomic_fetch_add(&g_loss_ticket[row], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = t == gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    float sum = 0.f;
    for (unsigned i = threadIdx.x; i < gridDim.x; i += 256)
        sum += __uint_as_float(__hip_atomic_fetch_or(&g_loss_part[row][i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float total = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
        if (row_flag & 256) *loss = total; else atomicAdd(loss, total);
        __hip_atomic_store(&g_loss_ticket[row], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---------------------------------------------------------------- warp_pts_layer + resample_layer
// tf_utils.py:35-52 + tf.contrib.resampler.  Output pixel (i,j):
//   x = flow[...,0] + i   (ROW index added to the channel the resampler reads as x / column)
//   y = flow[...,1] + j
// Block = 8 x 32 output pixels; one wave covers 2 x 32 pixels, so its source footprint is the
// transposed 32 x 2 (+flow) patch and stays in L1/L2.
struct ResampleParams {
    const float* src; const float* flow; const float* dgen;
    float* warp; float* gen; float* dflow;
    int N, H, W, Hs, Ws, C, flow_ld, dflow_ld;
};

// One bilinear tap.  The load is unconditional from a clamped (always valid) address and the bounds test only
// selects the result: a load under a branch makes hipcc wait for it at the join, i.e. one memory round trip per tap.
__device__ __forceinline__ float tap(const float* img, int Ws, int Hs, int C, int y, int x, int c) {
    const bool ok = (unsigned)x < (unsigned)Ws && (unsigned)y < (unsigned)Hs;
    const int xc = min(max(x, 0), Ws - 1), yc = min(max(y, 0), Hs - 1);
    const float v = img[((int64_t)yc * Ws + xc) * C + c];
    return ok ? v : 0.f;
}

template <bool BWD>
__global__ __launch_bounds__(256) void resample_kernel(const ResampleParams p) {
    const int tj = threadIdx.x & 31, ti = threadIdx.x >> 5;
    const int j = blockIdx.x * 32 + tj, i = blockIdx.y * 8 + ti, n = blockIdx.z;
    if (i >= p.H || j >= p.W) return;
    const int64_t pix = ((int64_t)n * p.H + i) * p.W + j;
    const float* fl = p.flow + pix * p.flow_ld;
    const float x = fl[0] + (float)i;
    const float y = fl[1] + (float)j;
    if (!BWD && p.warp) { p.warp[pix * 2] = x; p.warp[pix * 2 + 1] = y; }
    const bool valid = x > -1.0f && y > -1.0f && x < (float)p.Ws && y < (float)p.Hs;
    const float fxf = floorf(x), fyf = floorf(y);
    const int fx = (int)fxf, fy = (int)fyf, cx = fx + 1, cy = fy + 1;
    const float dx = (fxf + 1.0f) - x, dy = (fyf + 1.0f) - y;
    const float* img = p.src + (int64_t)n * p.Hs * p.Ws * p.C;
    if (!BWD) {
        float* out = p.gen + pix * p.C;
        for (int c = 0; c < p.C; ++c) {
            float v = 0.f;
            if (valid) {
                const float iff = tap(img, p.Ws, p.Hs, p.C, fy, fx, c), icc = tap(img, p.Ws, p.Hs, p.C, cy, cx, c);
                const float ifc = tap(img, p.Ws, p.Hs, p.C, cy, fx, c), icf = tap(img, p.Ws, p.Hs, p.C, fy, cx, c);
                v = ((dx * dy * iff + (1.0f - dx) * (1.0f - dy) * icc) + dx * (1.0f - dy) * ifc) + (1.0f - dx) * dy * icf;
            }
            out[c] = v;
        }
    } else {
        float gx = 0.f, gy = 0.f;
        if (valid) {
            const float* g = p.dgen + pix * p.C;
            for (int c = 0; c < p.C; ++c) {
                const float iff = tap(img, p.Ws, p.Hs, p.C, fy, fx, c), icc = tap(img, p.Ws, p.Hs, p.C, cy, cx, c);
                const float ifc = tap(img, p.Ws, p.Hs, p.C, cy, fx, c), icf = tap(img, p.Ws, p.Hs, p.C, fy, cx, c);
                gx += g[c] * (dy * (icf - iff) + (1.0f - dy) * (icc - ifc));
                gy += g[c] * (dx * (ifc - iff) + (1.0f - dx) * (icc - icf));
            }
        }
        float* d = p.dflow + pix * p.dflow_ld;
        d[0] = gx; d[1] = gy;
    }
}

// Tiled form of the same arithmetic.  The coords quirk makes the sampler read the TRANSPOSED neighbourhood of an output
// pixel (source row ~ j, source column ~ i), so with one thread per output pixel in row-major order every gather
// instruction touches 32-64 different cache lines and the kernel is bound by the texture-address unit (~25 us for
// 33 MB).  Here a workgroup owns a 32 x 32 output tile and goes through LDS twice:
//   A (j fastest)  coalesced reads of flow (+ target / incoming gradient) -> LDS, warp_pts written
//   B (i fastest)  lanes run along source rows: the four taps of 32 consecutive lanes are (nearly) contiguous;
//                  gen, the loss term and the flow gradient are computed here with the expressions of resample_kernel
//   C (j fastest)  coalesced stores of gen / dflow from LDS
// MODE 0 = forward, 1 = gradient w.r.t. flow from a given dgen, 2 = forward + pixel loss against `tgt` + gradient in one
// pass (the appearance-flow head: gen is only consumed by the loss, so dgen never needs to exist in HBM).
struct ResampleTileParams {
    const float* src; const float* flow; const float* aux;     // aux: dgen (MODE 1) or target (MODE 2), pixel stride aux_ld
    float* warp; float* gen; float* dflow; float* loss; int loss_row;
    int N, H, W, Hs, Ws, flow_ld, aux_ld, dflow_ld, kind;
    float weight;
    int tiles_i, tiles_j, n_tiles;
};

template <int MODE, int C>
__global__ __launch_bounds__(256) void resample_tile_kernel(const ResampleTileParams p) {
    constexpr int P = 33;                                   // LDS pitch: transposed reads hit 32 different banks
    __shared__ float s_xy[2][32 * P];                       // x, y; then gx, gy
    __shared__ float s_c[C][32 * P];                        // target / dgen; then gen
    const int lo = threadIdx.x & 31, hi = threadIdx.x >> 5;
    const float inv_pix = 1.0f / (float)((int64_t)p.N * p.H * p.W);
    const float gscale = (p.kind == 2 ? 2.0f : 1.0f) * p.weight / (float)((int64_t)p.N * p.H * p.W);
    float sum = 0.f;
    for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
        int b = tile;
        const int tj = b % p.tiles_j; b /= p.tiles_j;
        const int ti = b % p.tiles_i;
        const int n = b / p.tiles_i;
        const int i0 = ti * 32, j0 = tj * 32;
        const float* img = p.src + (int64_t)n * p.Hs * p.Ws * C;
        // ---- A: all loads first (clamped indices, no branches), then the LDS writes
        {
            float fx_[4], fy_[4], av[4][C];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = min(i0 + hi + 8 * k, p.H - 1), j = min(j0 + lo, p.W - 1);
                const int64_t pix = ((int64_t)n * p.H + i) * p.W + j;
                const float* fl = p.flow + pix * p.flow_ld;
                fx_[k] = fl[0]; fy_[k] = fl[1];
                if (MODE != 0) {
#pragma unroll
                    for (int c = 0; c < C; ++c) av[k][c] = p.aux[pix * p.aux_ld + c];
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = hi + 8 * k, i = i0 + r, j = j0 + lo;
                const float x = fx_[k] + (float)min(i, p.H - 1), y = fy_[k] + (float)min(j, p.W - 1);
                s_xy[0][r * P + lo] = x; s_xy[1][r * P + lo] = y;
                if (MODE != 0) {
#pragma unroll
                    for (int c = 0; c < C; ++c) s_c[c][r * P + lo] = av[k][c];
                }
                if (MODE != 1 && p.warp && i < p.H && j < p.W) {
                    const int64_t pix = ((int64_t)n * p.H + i) * p.W + j;
                    p.warp[pix * 2] = x; p.warp[pix * 2 + 1] = y;
                }
            }
        }
        __syncthreads();
        // ---- B: the 4 x 4 x C taps of a thread's four pixels are requested before any of them is used
        {
            float tp[4][4][C];          // [pixel][ff, cc, fc, cf][channel]
            float dxs[4], dys[4];
            bool vld[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int q = hi + 8 * k;
                const float x = s_xy[0][lo * P + q], y = s_xy[1][lo * P + q];
                vld[k] = x > -1.0f && y > -1.0f && x < (float)p.Ws && y < (float)p.Hs;
                const float fxf = floorf(x), fyf = floorf(y);
                // the clamp keeps the int conversion defined for far-away / non-finite sample points (they are not valid)
                const int fx = (int)fminf(fmaxf(fxf, -2.0f), (float)p.Ws), fy = (int)fminf(fmaxf(fyf, -2.0f), (float)p.Hs);
                const int cx = fx + 1, cy = fy + 1;
                dxs[k] = (fxf + 1.0f) - x; dys[k] = (fyf + 1.0f) - y;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    tp[k][0][c] = tap(img, p.Ws, p.Hs, C, fy, fx, c); tp[k][1][c] = tap(img, p.Ws, p.Hs, C, cy, cx, c);
                    tp[k][2][c] = tap(img, p.Ws, p.Hs, C, cy, fx, c); tp[k][3][c] = tap(img, p.Ws, p.Hs, C, fy, cx, c);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int r = lo, q = hi + 8 * k;
                const bool inr = i0 + r < p.H && j0 + q < p.W;
                const bool valid = vld[k];
                const float dx = dxs[k], dy = dys[k];
                float gx = 0.f, gy = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float iff = valid ? tp[k][0][c] : 0.f, icc = valid ? tp[k][1][c] : 0.f;
                    const float ifc = valid ? tp[k][2][c] : 0.f, icf = valid ? tp[k][3][c] : 0.f;
                    float g = 0.f;
                    if (MODE != 1) {
                        const float v = valid ? ((dx * dy * iff + (1.0f - dx) * (1.0f - dy) * icc) + dx * (1.0f - dy) * ifc) + (1.0f - dx) * dy * icf : 0.f;
                        if (MODE == 2) {
                            const float d = v - s_c[c][r * P + q];
                            if (p.kind == 2) { sum += inr ? d * d : 0.f; g = d * gscale; }
                            else { sum += inr ? fabsf(d) : 0.f; g = ((d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f)) * gscale; }
                        }
                        s_c[c][r * P + q] = v;
                    } else {
                        g = s_c[c][r * P + q];
                    }
                    if (MODE != 0) {
                        const float tx = g * (dy * (icf - iff) + (1.0f - dy) * (icc - ifc));
                        const float ty = g * (dx * (ifc - iff) + (1.0f - dx) * (icc - icf));
                        gx += valid ? tx : 0.f;
                        gy += valid ? ty : 0.f;
                    }
                }
                if (MODE != 0) { s_xy[0][r * P + q] = gx; s_xy[1][r * P + q] = gy; }
            }
        }
        __syncthreads();
        // ---- C
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = hi + 8 * k, i = i0 + r, j = j0 + lo;
            if (i < p.H && j < p.W) {
                const int64_t pix = ((int64_t)n * p.H + i) * p.W + j;
                if (MODE != 1) {
#pragma unroll
                    for (int c = 0; c < C; ++c) p.gen[pix * C + c] = s_c[c][r * P + lo];
                }
                if (MODE != 0 && p.dflow) { p.dflow[pix * p.dflow_ld] = s_xy[0][r * P + lo]; p.dflow[pix * p.dflow_ld + 1] = s_xy[1][r * P + lo]; }
            }
        }
        __syncthreads();
    }
    if (MODE == 2) {
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
        if ((threadIdx.x & 63) == 0) s_xy[0][threadIdx.x >> 6] = sum;
        __syncthreads();
        loss_combine((s_xy[0][0] + s_xy[0][1] + s_xy[0][2] + s_xy[0][3]) * (p.weight * inv_pix), p.loss, p.loss_row);
    }
}

template <int MODE>
static void launch_resample_tile(const ResampleTileParams& p, int C, int blocks, hipStream_t s) {
    switch (C) {
        case 1: resample_tile_kernel<MODE, 1><<<blocks, 256, 0, s>>>(p); break;
        case 2: resample_tile_kernel<MODE, 2><<<blocks, 256, 0, s>>>(p); break;
        case 3: resample_tile_kernel<MODE, 3><<<blocks, 256, 0, s>>>(p); break;
        default: resample_tile_kernel<MODE, 4><<<blocks, 256, 0, s>>>(p); break;
    }
}

static int resample_tile_blocks(ResampleTileParams& p) {
    p.tiles_i = cdiv(p.H, 32); p.tiles_j = cdiv(p.W, 32);
    p.n_tiles = p.N * p.tiles_i * p.tiles_j;
    return std::min(p.n_tiles, 1024);          // MODE 2 ends in one atomic per workgroup: keep that count small
}

// ---------------------------------------------------------------- pixel losses (tf_utils.py:18-23)
struct PixelLossParams {
    int64_t pixels; int ch;
    const float* a; int a_ld;
    const float* b; int b_ld; float b_scale;
    const float* mask; int mask_ld;
    int kind; float weight;
    float* loss; float* grad; int grad_ld; int loss_row;
};

// a, b, grad: [pixels][ch] views with pixel strides *_ld (channel slices of wider tensors, mv3d/nobg_dm.py:85-92);
// b is read as b * b_scale (mv3d/bg_nodm.py:88: gt_sm * 0.75); mask: one value per pixel (stride mask_ld)
__global__ __launch_bounds__(256) void pixel_loss_kernel(const PixelLossParams p) {
    const int64_t total = p.pixels * p.ch;
    const float gscale = (p.kind == 2 ? 2.0f : 1.0f) * p.weight / (float)p.pixels;
    float sum = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t pix = i / p.ch;
        const int c = (int)(i - pix * p.ch);
        const float m = p.mask ? p.mask[pix * p.mask_ld] : 1.0f;
        const float bv = p.b_scale == 1.0f ? p.b[pix * p.b_ld + c] : p.b[pix * p.b_ld + c] * p.b_scale;
        const float d = (p.a[pix * p.a_ld + c] - bv) * m;
        float g;
        if (p.kind == 2) { sum += d * d; g = d * m * gscale; }
        else { sum += fabsf(d); g = ((d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f)) * m * gscale; }
        if (p.grad) p.grad[pix * p.grad_ld + c] = g;
    }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
    __shared__ float s_part[4];
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = sum;
    __syncthreads();
    loss_combine((s_part[0] + s_part[1] + s_part[2] + s_part[3]) * (p.weight / (float)p.pixels), p.loss, p.loss_row);
}

// dense, unmasked, unscaled operands (the appearance-flow loss): 16-byte loads / stores over the flat index
__global__ __launch_bounds__(256) void pixel_loss_dense_kernel(int64_t total4, int64_t pixels, const float4* __restrict__ a,
                                                              const float4* __restrict__ b, int kind, float weight, float* loss,
                                                              float4* __restrict__ grad, int loss_row) {
    const float gscale = (kind == 2 ? 2.0f : 1.0f) * weight / (float)pixels;
    float sum = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (int64_t)gridDim.x * 256) {
        const float4 x = a[i], y = b[i];
        const float d[4] = {x.x - y.x, x.y - y.y, x.z - y.z, x.w - y.w};
        float g[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (kind == 2) { sum += d[e] * d[e]; g[e] = d[e] * gscale; }
            else { sum += fabsf(d[e]); g[e] = ((d[e] > 0.f) ? 1.f : ((d[e] < 0.f) ? -1.f : 0.f)) * gscale; }
        }
        if (grad) grad[i] = make_float4(g[0], g[1], g[2], g[3]);
    }
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off);
    __shared__ float s_part[4];
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = sum;
    __syncthreads();
    loss_combine((s_part[0] + s_part[1] + s_part[2] + s_part[3]) * (weight / (float)pixels), loss, loss_row);
}

// raw uint8 image bytes -> float32 / 255 (read_tf_records.py:111: tf.cast(image, tf.float32) / 255.0), four pixels' worth per lane
__global__ __launch_bounds__(256) void u8_to_unit_f32_kernel(int64_t count, const unsigned char* __restrict__ src, float* __restrict__ dst) {
    const int64_t nvec = count >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        const uchar4 b = reinterpret_cast<const uchar4*>(src)[i];
        reinterpret_cast<float4*>(dst)[i] = make_float4((float)b.x / 255.0f, (float)b.y / 255.0f, (float)b.z / 255.0f, (float)b.w / 255.0f);
    }
    if (blockIdx.x == 0 && threadIdx.x < (count & 3)) { const int64_t i = (nvec << 2) + threadIdx.x; dst[i] = (float)src[i] / 255.0f; }
}

__global__ __launch_bounds__(256) void fill_kernel(float* dst, int64_t count, float v) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (int64_t)gridDim.x * 256) dst[i] = v;
}

// ---------------------------------------------------------------- Adam (TF ApplyAdam)
// One pass: 16 B per lane loads of p, g, m, v and stores of p, m, v = 28 B/param of HBM traffic.
__global__ __launch_bounds__(256) void adam_kernel(int64_t count, float* __restrict__ p, const float* __restrict__ g,
                                                  float* __restrict__ m, float* __restrict__ v, float alpha,
                                                  float omb1, float omb2, float eps, float gscale) {
    const int64_t nvec = count >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
        float4 pp = reinterpret_cast<float4*>(p)[i], gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        float* pe = &pp.x; float* ge = &gg.x; float* me = &mm.x; float* ve = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = ge[k] * gscale;
            me[k] += (gk - me[k]) * omb1;
            ve[k] += (gk * gk - ve[k]) * omb2;
            pe[k] -= (me[k] * alpha) / (sqrtf(ve[k]) + eps);
        }
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (count & 3)) {
        const int64_t i = (nvec << 2) + threadIdx.x;
        const float gk = g[i] * gscale;
        m[i] += (gk - m[i]) * omb1;
        v[i] += (gk * gk - v[i]) * omb2;
        p[i] -= (m[i] * alpha) / (sqrtf(v[i]) + eps);
    }
}

// Same pass with the scalars taken from the device Adam state (so that a recorded plan replays with the current step's
// bias correction) and up to MV3D_ADAM_MAX_SKIP index ranges left untouched (parameters whose update is fused elsewhere).
struct AdamSkips { int n; int64_t lo[8], hi[8]; };
__global__ __launch_bounds__(256) void adam_dev_kernel(int64_t count, float* __restrict__ p, const float* __restrict__ g,
                                                      float* __restrict__ m, float* __restrict__ v, const float* __restrict__ st, const AdamSkips sk) {
    const float lr = st[0], b1 = st[1], b2 = st[2], eps = st[3], b1p = st[4], b2p = st[5], gscale = st[6];
    const float alpha = lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
    const float omb1 = 1.0f - b1, omb2 = 1.0f - b2;
    // walk the KEPT float4s only: kept index j -> position by adding the lengths of the skipped ranges that start at or before it
    // (ranges are sorted and disjoint: the host checks)
    int64_t kept = count >> 2;
    for (int r = 0; r < sk.n; ++r) kept -= (sk.hi[r] - sk.lo[r]) >> 2;
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < kept; j += (int64_t)gridDim.x * 256) {
        int64_t i = j;
        for (int r = 0; r < sk.n; ++r)
            if (i >= (sk.lo[r] >> 2)) i += (sk.hi[r] - sk.lo[r]) >> 2;
        float4 pp = reinterpret_cast<float4*>(p)[i], gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        float* pe = &pp.x; float* ge = &gg.x; float* me = &mm.x; float* ve = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = ge[k] * gscale;
            me[k] += (gk - me[k]) * omb1;
            ve[k] += (gk * gk - ve[k]) * omb2;
            pe[k] -= (me[k] * alpha) / (sqrtf(ve[k]) + eps);
        }
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
}
__global__ void adam_advance_kernel(float* st) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { st[4] = st[4] * st[1]; st[5] = st[5] * st[2]; }      // beta_power *= beta, in fp32 like TF's update op
}

// ---------------------------------------------------------------- gradient finalisation: ONE launch at the end of the reverse pass
// Every conv / deconv filter gradient leaves its kernel as per-slab partial sums; they used to be summed by one reduce_slabs
// launch per layer (19 per step on AppearanceFlowModel) and the optimiser then re-read the sums.  Here a table of segments
// {partials, slabs, count, position in the flat buffers} drives one launch: a workgroup sums its 64 elements over the slabs in
// the order reduce_slabs_kernel uses (reduce_common.h: same bits) and either stores the gradient or applies TF's ApplyAdam to
// that element right away (the arithmetic of adam_dev_kernel, operation for operation).  Segments with nslab == 0 are gradients
// already final in the flat gradient buffer (biases of fc layers, the angle MLP): optimiser only.
struct FinSeg {
    const float* part; float* out;      // partials [nslab][count]; where the summed gradient goes (gradients-only mode)
    int64_t off;                        // element offset of `out` in the flat gradient / parameter / slot buffers (optimiser mode)
    int64_t count;
    int nslab, vec, blk0, pad;
};

template <bool ADAM>
__global__ __launch_bounds__(256) void grad_finalize_kernel(const FinSeg* __restrict__ segs, const int* __restrict__ seg_of_blk,
                                                           const float* __restrict__ G, float* __restrict__ P, float* __restrict__ M,
                                                           float* __restrict__ V, const float* __restrict__ st) {
    __shared__ float s_sum[4][16][17];
    const FinSeg sg = segs[seg_of_blk[blockIdx.x]];
    const int blk = (int)blockIdx.x - sg.blk0;
    float alpha = 0.f, omb1 = 0.f, omb2 = 0.f, eps = 0.f, gscale = 1.f;
    if (ADAM) {
        const float lr = st[0], b1 = st[1], b2 = st[2], b1p = st[4], b2p = st[5];
        eps = st[3]; gscale = st[6];
        alpha = lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
        omb1 = 1.0f - b1; omb2 = 1.0f - b2;
    }
    auto adam1 = [&](int64_t i, float g) {             // flat index i
        const float gk = g * gscale;
        float m = M[i], v = V[i], pp = P[i];
        m += (gk - m) * omb1;
        v += (gk * gk - v) * omb2;
        pp -= (m * alpha) / (sqrtf(v) + eps);
        P[i] = pp; M[i] = m; V[i] = v;
    };
    if (sg.nslab == 0) {
        // final gradients: 1024 elements per workgroup, 16 bytes per lane (count is padded to a multiple of 4 by the host)
        if (!ADAM) return;
        const int64_t i4 = (int64_t)blk * 256 + threadIdx.x;
        if (i4 * 4 >= sg.count) return;
        const int64_t i = (sg.off >> 2) + i4;
        float4 pp = reinterpret_cast<float4*>(P)[i], gg = reinterpret_cast<const float4*>(G)[i];
        float4 mm = reinterpret_cast<float4*>(M)[i], vv = reinterpret_cast<float4*>(V)[i];
        float* pe = &pp.x; float* ge = &gg.x; float* me = &mm.x; float* ve = &vv.x;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float gk = ge[k] * gscale;
            me[k] += (gk - me[k]) * omb1;
            ve[k] += (gk * gk - ve[k]) * omb2;
            pe[k] -= (me[k] * alpha) / (sqrtf(ve[k]) + eps);
        }
        reinterpret_cast<float4*>(P)[i] = pp;
        reinterpret_cast<float4*>(M)[i] = mm;
        reinterpret_cast<float4*>(V)[i] = vv;
        return;
    }
    auto fin = [&](int64_t i, float t) {
        if (ADAM) adam1(sg.off + i, t);
        else sg.out[i] = t;
    };
    if (sg.vec) reduce_slabs_body<4>(sg.part, sg.nslab, sg.count, blk, s_sum, fin);
    else reduce_slabs_body<1>(sg.part, sg.nslab, sg.count, blk, s_sum, fin);
}

// ---------------------------------------------------------------- activations / copies
__global__ __launch_bounds__(256) void act_fwd_kernel(int64_t rows, int ch, const float* x, int x_ld, float* y, int y_ld,
                                                     int act, float leak) {
    const int64_t total = rows * ch;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / ch; const int c = (int)(i - r * ch);
        y[r * y_ld + c] = act_apply(x[r * x_ld + c], act, leak);
    }
}
__global__ __launch_bounds__(256) void act_bwd_kernel(int64_t rows, int ch, const float* dy, int dy_ld, const float* ref,
                                                     int ref_ld, float* dx, int dx_ld, int act, float leak) {
    const int64_t total = rows * ch;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / ch; const int c = (int)(i - r * ch);
        dx[r * dx_ld + c] = dy[r * dy_ld + c] * act_grad_from_out(ref[r * ref_ld + c], act, leak);
    }
}
__global__ __launch_bounds__(256) void copy2d_kernel(int64_t rows, int ch, const float* src, int64_t src_ld, int64_t src_row_div,
                                                    float* dst, int64_t dst_ld, int accumulate) {
    const int64_t total = rows * ch;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / ch; const int c = (int)(i - r * ch);
        const float v = src[(r / src_row_div) * src_ld + c];
        float* d = dst + r * dst_ld + c;
        *d = accumulate ? (*d + v) : v;
    }
}
__global__ __launch_bounds__(256) void group_sum_kernel(int64_t groups, int group, int ch, const float* src, int64_t src_ld,
                                                       float* dst, int64_t dst_ld) {
    const int64_t total = groups * ch;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t gidx = i / ch; const int c = (int)(i - gidx * ch);
        float s = 0.f;
        for (int r = 0; r < group; ++r) s += src[(gidx * group + r) * src_ld + c];
        dst[gidx * dst_ld + c] = s;
    }
}

static inline int grid_for(int64_t total) { return (int)std::min<int64_t>(cdiv64(total, 256), 8192); }

}  // namespace mv3d

using namespace mv3d;

extern "C" {

int mv3d_warp_resample_fwd(int N, int H, int W, int Hs, int Ws, int C, const void* src, const void* flow, int flow_ld,
                           void* warp_out, void* gen, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0 || Hs <= 0 || Ws <= 0 || C <= 0 || flow_ld < 2) return fail(MV3D_E_INVAL, "mv3d_warp_resample_fwd: bad shape");
    if (!src || !flow || !gen) return fail(MV3D_E_INVAL, "mv3d_warp_resample_fwd: null pointer");
    if (N > 65535) return fail(MV3D_E_UNSUPPORTED, "mv3d_warp_resample_fwd: batch > 65535");
    if (C <= 4 && !(disabled_paths() & 65536)) {
        ResampleTileParams t = {};
        t.src = (const float*)src; t.flow = (const float*)flow; t.warp = (float*)warp_out; t.gen = (float*)gen;
        t.N = N; t.H = H; t.W = W; t.Hs = Hs; t.Ws = Ws; t.flow_ld = flow_ld;
        const int blocks = resample_tile_blocks(t);
        return dispatch(stream, OpInfo{"resample_fwd", 0.0, (double)N * H * W * (8.0 + 8.0 * C)}, [=](hipStream_t s) {
            launch_resample_tile<0>(t, C, blocks, s);
            return launched("resample_tile_kernel<fwd>");
        });
    }
    ResampleParams p = {(const float*)src, (const float*)flow, nullptr, (float*)warp_out, (float*)gen, nullptr, N, H, W, Hs, Ws, C, flow_ld, 0};
    dim3 grid(cdiv(W, 32), cdiv(H, 8), N);
    return dispatch(stream, OpInfo{"resample_fwd", 0.0, (double)N * H * W * (8.0 + 8.0 * C)}, [=](hipStream_t s) {
        resample_kernel<false><<<grid, 256, 0, s>>>(p);
        return launched("resample_kernel<fwd>");
    });
}

int mv3d_warp_resample_bwd(int N, int H, int W, int Hs, int Ws, int C, const void* src, const void* flow, int flow_ld,
                           const void* dgen, void* dflow, int dflow_ld, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0 || Hs <= 0 || Ws <= 0 || C <= 0 || flow_ld < 2 || dflow_ld < 2) return fail(MV3D_E_INVAL, "mv3d_warp_resample_bwd: bad shape");
    if (!src || !flow || !dgen || !dflow) return fail(MV3D_E_INVAL, "mv3d_warp_resample_bwd: null pointer");
    if (N > 65535) return fail(MV3D_E_UNSUPPORTED, "mv3d_warp_resample_bwd: batch > 65535");
    if (C <= 4 && !(disabled_paths() & 65536)) {
        ResampleTileParams t = {};
        t.src = (const float*)src; t.flow = (const float*)flow; t.aux = (const float*)dgen; t.aux_ld = C; t.dflow = (float*)dflow;
        t.N = N; t.H = H; t.W = W; t.Hs = Hs; t.Ws = Ws; t.flow_ld = flow_ld; t.dflow_ld = dflow_ld;
        const int blocks = resample_tile_blocks(t);
        return dispatch(stream, OpInfo{"resample_bwd", 0.0, (double)N * H * W * (16.0 + 8.0 * C)}, [=](hipStream_t s) {
            launch_resample_tile<1>(t, C, blocks, s);
            return launched("resample_tile_kernel<bwd>");
        });
    }
    ResampleParams p = {(const float*)src, (const float*)flow, (const float*)dgen, nullptr, nullptr, (float*)dflow, N, H, W, Hs, Ws, C, flow_ld, dflow_ld};
    dim3 grid(cdiv(W, 32), cdiv(H, 8), N);
    return dispatch(stream, OpInfo{"resample_bwd", 0.0, (double)N * H * W * (16.0 + 8.0 * C)}, [=](hipStream_t s) {
        resample_kernel<true><<<grid, 256, 0, s>>>(p);
        return launched("resample_kernel<bwd>");
    });
}

int mv3d_warp_resample_loss(int N, int H, int W, int Hs, int Ws, int C, const void* src, const void* flow, int flow_ld,
                            const void* target, int target_ld, int kind, float weight, void* warp_out, void* gen,
                            void* dflow, int dflow_ld, void* loss_accum, void* stream) {
    if (N <= 0 || H <= 0 || W <= 0 || Hs <= 0 || Ws <= 0 || C <= 0 || flow_ld < 2 || target_ld < C || (dflow && dflow_ld < 2))
        return fail(MV3D_E_INVAL, "mv3d_warp_resample_loss: bad shape");
    if (kind != 1 && kind != 2) return fail(MV3D_E_INVAL, "mv3d_warp_resample_loss: kind must be 1 (l1) or 2 (euclidean)");
    if (!src || !flow || !target || !gen || !loss_accum) return fail(MV3D_E_INVAL, "mv3d_warp_resample_loss: null pointer");
    if (C > 4) return fail(MV3D_E_UNSUPPORTED, "mv3d_warp_resample_loss: more than 4 channels");
    ResampleTileParams t = {};
    t.src = (const float*)src; t.flow = (const float*)flow; t.aux = (const float*)target; t.aux_ld = target_ld;
    t.warp = (float*)warp_out; t.gen = (float*)gen; t.dflow = (float*)dflow; t.loss = (float*)loss_accum;
    t.N = N; t.H = H; t.W = W; t.Hs = Hs; t.Ws = Ws; t.flow_ld = flow_ld; t.dflow_ld = dflow_ld; t.kind = kind; t.weight = weight;
    t.loss_row = next_loss_row();
    static const int max_blocks = getenv("MV3D_RL_BLOCKS") ? atoi(getenv("MV3D_RL_BLOCKS")) : 512;
    const int blocks = std::min(resample_tile_blocks(t), max_blocks);
    return dispatch(stream, OpInfo{"resample_loss", 0.0, (double)N * H * W * (8.0 + 8.0 + 8.0 + 12.0 * C)}, [=](hipStream_t s) {
        launch_resample_tile<2>(t, C, blocks, s);
        return launched("resample_tile_kernel<fused>");
    });
}

int mv3d_pixel_loss_strided(int64_t pixels, int ch, const void* a, int a_ld, const void* b, int b_ld, float b_scale,
                            const void* mask, int mask_ld, int kind, float weight, void* loss_accum, void* grad, int grad_ld,
                            void* stream) {
    if (pixels <= 0 || ch <= 0 || (kind != 1 && kind != 2)) return fail(MV3D_E_INVAL, "mv3d_pixel_loss: bad arguments");
    if (!a || !b || !loss_accum) return fail(MV3D_E_INVAL, "mv3d_pixel_loss: null pointer");
    if (a_ld < ch || b_ld < ch || (grad && grad_ld < ch) || (mask && mask_ld < 1)) return fail(MV3D_E_INVAL, "mv3d_pixel_loss: pixel stride smaller than the channel count");
    PixelLossParams p = {pixels, ch, (const float*)a, a_ld, (const float*)b, b_ld, b_scale, (const float*)mask, mask_ld, kind, weight,
                         (float*)loss_accum, (float*)grad, grad_ld, next_loss_row()};
    const int64_t total = pixels * ch;
    const bool dense = a_ld == ch && b_ld == ch && (!grad || grad_ld == ch) && !mask && b_scale == 1.0f && (total & 3) == 0 &&
                       ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)grad)) & 15) == 0;
    if (dense) {
        const int64_t total4 = total >> 2;
        // one workgroup per CU
        const int blocks4 = (int)std::min<int64_t>(cdiv64(total4, 256 * 2), 256);
        return dispatch(stream, OpInfo{"pixel_loss", 0.0, (double)total * (grad ? 12.0 : 8.0)}, [=](hipStream_t s) {
            pixel_loss_dense_kernel<<<blocks4, 256, 0, s>>>(total4, pixels, (const float4*)a, (const float4*)b, kind, weight,
                                                           (float*)loss_accum, (float4*)grad, p.loss_row);
            return launched("pixel_loss_dense_kernel");
        });
    }
    const int blocks = (int)std::min<int64_t>(cdiv64(total, 256 * 8), 256);
    return dispatch(stream, OpInfo{"pixel_loss", 0.0, (double)total * (grad ? 12.0 : 8.0)}, [=](hipStream_t s) {
        pixel_loss_kernel<<<blocks, 256, 0, s>>>(p);
        return launched("pixel_loss_kernel");
    });
}

int mv3d_pixel_loss(int64_t pixels, int ch, const void* a, const void* b, const void* mask, int kind, float weight,
                    void* loss_accum, void* grad, void* stream) {
    return mv3d_pixel_loss_strided(pixels, ch, a, ch, b, ch, 1.0f, mask, 1, kind, weight, loss_accum, grad, ch, stream);
}

int mv3d_fill(void* dst, int64_t count, float value, void* stream) {
    if (!dst || count < 0) return fail(MV3D_E_INVAL, "mv3d_fill: bad arguments");
    if (count == 0) return MV3D_OK;
    return dispatch(stream, OpInfo{"fill", 0.0, 4.0 * count}, [=](hipStream_t s) {
        fill_kernel<<<grid_for(count), 256, 0, s>>>((float*)dst, count, value);
        return launched("fill_kernel");
    });
}

int mv3d_u8_to_unit_f32(int64_t count, const void* src, void* dst, void* stream) {
    if (count <= 0 || !src || !dst) return fail(MV3D_E_INVAL, "mv3d_u8_to_unit_f32: bad arguments");
    if (((uintptr_t)src & 3) || ((uintptr_t)dst & 15)) return fail(MV3D_E_INVAL, "mv3d_u8_to_unit_f32: src must be 4-byte and dst 16-byte aligned");
    return dispatch(stream, OpInfo{"u8_to_unit_f32", 0.0, 5.0 * count}, [=](hipStream_t s) {
        u8_to_unit_f32_kernel<<<grid_for(count / 4 + 1), 256, 0, s>>>(count, (const unsigned char*)src, (float*)dst);
        return launched("u8_to_unit_f32_kernel");
    });
}

int mv3d_adam_step(int64_t count, void* p, const void* g, void* m, void* v, float lr, float beta1, float beta2,
                   float eps, float beta1_power, float beta2_power, float grad_scale, void* stream) {
    if (count <= 0 || !p || !g || !m || !v) return fail(MV3D_E_INVAL, "mv3d_adam_step: bad arguments");
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return fail(MV3D_E_INVAL, "mv3d_adam_step: buffers must be 16-byte aligned");
    // alpha in fp32 exactly as TF's functor evaluates it
    const float alpha = lr * sqrtf(1.0f - beta2_power) / (1.0f - beta1_power);
    const float omb1 = 1.0f - beta1, omb2 = 1.0f - beta2;
    const int blocks = (int)std::min<int64_t>(cdiv64(count / 4 + 1, 256), 4096);
    return dispatch(stream, OpInfo{"adam", 0.0, 28.0 * count}, [=](hipStream_t s) {
        adam_kernel<<<blocks, 256, 0, s>>>(count, (float*)p, (const float*)g, (float*)m, (float*)v, alpha, omb1, omb2, eps, grad_scale);
        return launched("adam_kernel");
    });
}

int mv3d_adam_step_dev(int64_t count, void* p, const void* g, void* m, void* v, const void* state, int nskip, const int64_t* skip_lo,
                       const int64_t* skip_hi, void* stream) {
    if (count <= 0 || (count & 3) || !p || !g || !m || !v || !state) return fail(MV3D_E_INVAL, "mv3d_adam_step_dev: bad arguments (count must be a multiple of 4)");
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return fail(MV3D_E_INVAL, "mv3d_adam_step_dev: buffers must be 16-byte aligned");
    if (nskip < 0 || nskip > 8 || (nskip > 0 && (!skip_lo || !skip_hi))) return fail(MV3D_E_INVAL, "mv3d_adam_step_dev: at most 8 skipped ranges");
    AdamSkips sk = {};
    sk.n = nskip;
    int64_t skipped = 0;
    for (int r = 0; r < nskip; ++r) {
        if ((skip_lo[r] & 3) || (skip_hi[r] & 3) || skip_lo[r] < 0 || skip_hi[r] > count || skip_lo[r] > skip_hi[r])
            return fail(MV3D_E_INVAL, "mv3d_adam_step_dev: skipped range %d is not a multiple-of-4 sub-range", r);
        if (r > 0 && skip_lo[r] < skip_hi[r - 1]) return fail(MV3D_E_INVAL, "mv3d_adam_step_dev: skipped ranges must be sorted and disjoint");
        sk.lo[r] = skip_lo[r]; sk.hi[r] = skip_hi[r];
        skipped += skip_hi[r] - skip_lo[r];
    }
    const int blocks = (int)std::min<int64_t>(cdiv64((count - skipped) / 4 + 1, 256), 4096);
    return dispatch(stream, OpInfo{"adam", 0.0, 28.0 * (count - skipped)}, [=](hipStream_t s) {
        adam_dev_kernel<<<blocks, 256, 0, s>>>(count, (float*)p, (const float*)g, (float*)m, (float*)v, (const float*)state, sk);
        return launched("adam_dev_kernel");
    });
}

int mv3d_adam_advance(void* state, void* stream) {
    if (!state) return fail(MV3D_E_INVAL, "mv3d_adam_advance: null state");
    return dispatch(stream, OpInfo{"adam_advance", 0.0, 16.0}, [=](hipStream_t s) {
        adam_advance_kernel<<<1, 64, 0, s>>>((float*)state);
        return launched("adam_advance_kernel");
    });
}

int mv3d_act_fwd(int64_t rows, int ch, const void* x, int x_ld, void* y, int y_ld, int act, float leak, void* stream) {
    if (rows <= 0 || ch <= 0 || !x || !y || x_ld < ch || y_ld < ch) return fail(MV3D_E_INVAL, "mv3d_act_fwd: bad arguments");
    return dispatch(stream, OpInfo{"act_fwd", 0.0, 8.0 * rows * ch}, [=](hipStream_t s) {
        act_fwd_kernel<<<grid_for(rows * ch), 256, 0, s>>>(rows, ch, (const float*)x, x_ld, (float*)y, y_ld, act, leak);
        return launched("act_fwd_kernel");
    });
}

int mv3d_act_bwd(int64_t rows, int ch, const void* dy, int dy_ld, const void* ref, int ref_ld, void* dx, int dx_ld,
                 int act, float leak, void* stream) {
    if (rows <= 0 || ch <= 0 || !dy || !ref || !dx || dy_ld < ch || ref_ld < ch || dx_ld < ch) return fail(MV3D_E_INVAL, "mv3d_act_bwd: bad arguments");
    return dispatch(stream, OpInfo{"act_bwd", 0.0, 12.0 * rows * ch}, [=](hipStream_t s) {
        act_bwd_kernel<<<grid_for(rows * ch), 256, 0, s>>>(rows, ch, (const float*)dy, dy_ld, (const float*)ref, ref_ld, (float*)dx, dx_ld, act, leak);
        return launched("act_bwd_kernel");
    });
}

int mv3d_copy2d(int64_t rows, int ch, const void* src, int64_t src_ld, int64_t src_row_div, void* dst, int64_t dst_ld,
                int accumulate, void* stream) {
    if (rows <= 0 || ch <= 0 || !src || !dst || src_row_div < 1) return fail(MV3D_E_INVAL, "mv3d_copy2d: bad arguments");
    return dispatch(stream, OpInfo{"copy2d", 0.0, 8.0 * rows * ch}, [=](hipStream_t s) {
        copy2d_kernel<<<grid_for(rows * ch), 256, 0, s>>>(rows, ch, (const float*)src, src_ld, src_row_div, (float*)dst, dst_ld, accumulate);
        return launched("copy2d_kernel");
    });
}

int mv3d_group_sum(int64_t groups, int group, int ch, const void* src, int64_t src_ld, void* dst, int64_t dst_ld, void* stream) {
    if (groups <= 0 || group <= 0 || ch <= 0 || !src || !dst) return fail(MV3D_E_INVAL, "mv3d_group_sum: bad arguments");
    return dispatch(stream, OpInfo{"group_sum", 0.0, 4.0 * groups * ch * (group + 1)}, [=](hipStream_t s) {
        group_sum_kernel<<<grid_for(groups * ch), 256, 0, s>>>(groups, group, ch, (const float*)src, src_ld, (float*)dst, dst_ld);
        return launched("group_sum_kernel");
    });
}

int mv3d_loss_overwrite_next(void) { g_loss_overwrite = true; return MV3D_OK; }

// ---- gradient finalisation (see grad_finalize_kernel) -------------------------------------------------------------------------
int mv3d_grad_finalize_begin(void) {
    if (finalize_collecting()) return fail(MV3D_E_INVAL, "mv3d_grad_finalize_begin: a collection is already open on this thread");
    finalize_open();
    return MV3D_OK;
}
int mv3d_grad_finalize_add(void* grad, int64_t count) {
    if (!finalize_collecting()) return fail(MV3D_E_INVAL, "mv3d_grad_finalize_add: no open collection");
    if (!grad || count <= 0 || ((uintptr_t)grad & 15)) return fail(MV3D_E_INVAL, "mv3d_grad_finalize_add: bad range (16-byte aligned, count > 0)");
    finalize_push(nullptr, 0, count, (float*)grad);
    return MV3D_OK;
}
int mv3d_grad_finalize_abort(void) { finalize_take(nullptr); return MV3D_OK; }

static int fin_blocks(const FinSegHost& h) {
    if (h.nslab == 0) return (int)cdiv64(cdiv64(h.count, 4), 256);
    const bool vec = h.count % 4 == 0 && ((uintptr_t)h.part & 15) == 0;
    return (int)cdiv64(h.count, vec ? 64 : 16);
}
size_t mv3d_grad_finalize_table_bytes(void) {
    const std::vector<FinSegHost>* v = finalize_peek();
    if (!v) return 0;
    size_t blocks = 0;
    for (const FinSegHost& h : *v) blocks += fin_blocks(h);
    return ((v->size() * sizeof(FinSeg) + 255) & ~(size_t)255) + blocks * sizeof(int);
}
int mv3d_grad_finalize_commit(void* table, size_t table_bytes, void* grads, void* params, void* adam_m, void* adam_v,
                              const void* adam_state, void* stream) {
    if (!finalize_collecting()) return fail(MV3D_E_INVAL, "mv3d_grad_finalize_commit: no open collection");
    const size_t need = mv3d_grad_finalize_table_bytes();
    std::vector<FinSegHost> segs;
    finalize_take(&segs);
    const bool adam = adam_state != nullptr;
    if (adam && (!grads || !params || !adam_m || !adam_v || (((uintptr_t)grads | (uintptr_t)params | (uintptr_t)adam_m | (uintptr_t)adam_v) & 15)))
        return fail(MV3D_E_INVAL, "mv3d_grad_finalize_commit: the optimiser needs the four flat buffers (16-byte aligned)");
    // one segment per gradient: a range named 'already final' (by the caller, or by a single-slab filter gradient written in place)
    // that a slab segment also produces is the slab segment's, and a range named twice counts once
    std::vector<FinSegHost> keep;
    for (size_t a = 0; a < segs.size(); ++a) {
        const FinSegHost& h = segs[a];
        bool dup = false;
        for (size_t b = 0; b < segs.size(); ++b) {
            if (b == a || segs[b].out != h.out) continue;
            if (h.nslab == 0 && (segs[b].nslab > 0 || b < a)) dup = true;
            if (h.nslab > 0 && segs[b].nslab > 0 && b < a) return fail(MV3D_E_INVAL, "mv3d_grad_finalize_commit: two filter gradients write the same range");
        }
        if (!dup && (adam || h.nslab > 0)) keep.push_back(h);
    }
    if (keep.empty()) return MV3D_OK;
    const bool dry = !table && recording();      // a plan recorded without a device (host-logic tests): never runs, nothing to upload
    if (!dry && (!table || table_bytes < need || ((uintptr_t)table & 15))) return fail(MV3D_E_WORKSPACE, "mv3d_grad_finalize_commit: table %zu < %zu bytes", table_bytes, need);
    std::vector<FinSeg> dev(keep.size());
    std::vector<int> seg_of;
    double bytes = 0.0;
    for (size_t j = 0; j < keep.size(); ++j) {
        const FinSegHost& h = keep[j];
        FinSeg& d = dev[j];
        d.part = h.part; d.out = h.out; d.count = h.count; d.nslab = h.nslab; d.pad = 0;
        d.vec = h.nslab > 0 && h.count % 4 == 0 && ((uintptr_t)h.part & 15) == 0;
        d.off = 0;
        if (adam) {
            const int64_t off = h.out - (float*)grads;
            if (off < 0 || (h.nslab == 0 && (off & 3))) return fail(MV3D_E_INVAL, "mv3d_grad_finalize_commit: segment %zu is not inside the flat gradient buffer", j);
            d.off = off;
        }
        d.blk0 = (int)seg_of.size();
        const int nb = fin_blocks(h);
        seg_of.insert(seg_of.end(), nb, (int)j);
        bytes += 4.0 * h.count * (h.nslab + (adam ? 6 : 1));
    }
    const size_t seg_bytes = (dev.size() * sizeof(FinSeg) + 255) & ~(size_t)255;
    // the table is written NOW (record time): a recorded plan replays the launch below against it
    if (!dry && (hipMemcpy(table, dev.data(), dev.size() * sizeof(FinSeg), hipMemcpyHostToDevice) != hipSuccess ||
                 hipMemcpy((char*)table + seg_bytes, seg_of.data(), seg_of.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess))
        return fail(MV3D_E_HIP, "mv3d_grad_finalize_commit: table upload failed");
    const FinSeg* dsegs = (const FinSeg*)table;
    const int* dmap = (const int*)((char*)table + seg_bytes);
    const int blocks = (int)seg_of.size();
    return dispatch(stream, OpInfo{adam ? "grad_finalize_adam" : "grad_finalize", 0.0, bytes}, [=](hipStream_t s) {
        if (adam) grad_finalize_kernel<true><<<blocks, 256, 0, s>>>(dsegs, dmap, (const float*)grads, (float*)params, (float*)adam_m, (float*)adam_v, (const float*)adam_state);
        else grad_finalize_kernel<false><<<blocks, 256, 0, s>>>(dsegs, dmap, nullptr, nullptr, nullptr, nullptr, nullptr);
        return launched("grad_finalize_kernel");
    });
}

}  // extern "C"
