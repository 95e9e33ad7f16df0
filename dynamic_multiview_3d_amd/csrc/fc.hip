// Linear layers (tf.matmul(x, M) + b, tf_utils.py:67) at batch 32..128: four 4096x4096 fp32 matrices
// = 97 % of the model's parameters.  With M = batch rows the three GEMMs are bound by streaming the
// 67 MB matrix once (fwd, dgrad) or writing it once (wgrad), not by MFMA rate:
//   fwd   y[m][n]  = sum_k x[m][k]  W[k][n]      fc_stream_kernel<false>
//   dgrad dx[m][k] = sum_n dy[m][n] W[k][n]      fc_stream_kernel<true>
//   wgrad dW[k][n] = sum_m x[m][k]  dy[m][n]     fc_wgrad_kernel
// fc_stream: a workgroup owns 32 output columns x all rows and one slice of the reduction; its 4 waves
// split the slice, stream their part of W straight from HBM into MFMA B-fragments (fwd: 128-byte row
// segments per k; dgrad: 16 bytes per lane along n with the permuted k-slot order that hconv uses) and
// read the small activation operand (L2-resident, 1 MB) as 16-byte loads in A-fragment order.  No LDS
// in the main loop; one LDS pass combines the 4 waves; reduction slices are summed (+ bias,
// activation, activation-gradient mask) by igemm_splitk_epilogue in a fixed order.
#include "conv_common.h"
#include <algorithm>

namespace mv3d {

struct FcParams {
    const float* X;      // [M][R] activations (row stride x_ld), R = reduction length
    const float* W;      // fwd: [R][N] ; dgrad: [N][R]   (dense)
    float* Part;         // [nsplit][M][N] partials
    int M, R, N, x_ld;
    int w_ld;            // row stride of W in floats
    int nsplit, chunks_total;
};

// MT = number of 32-row groups (M <= 32*MT).  Workgroup = 128 output columns (one 32-column strip per
// wave) x all rows x one reduction slice.  The activation chunk [M][32] is shared by the 4 waves
// through LDS (double-buffered, one barrier per chunk): read per wave from global it would cost twice
// the L1 bandwidth of the weight stream itself.
template <bool TRANS, int MT>
__global__ __launch_bounds__(256) void fc_stream_kernel(const FcParams p) {
    constexpr int LDA = 33;
    constexpr int ROWS = 32 * MT;
    __shared__ float As[2][ROWS * LDA];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int n0 = blockIdx.x * 128 + wave * 32;
    const int ks = blockIdx.y;
    const int c_begin = (int)((int64_t)p.chunks_total * ks / p.nsplit);
    const int c_end = (int)((int64_t)p.chunks_total * (ks + 1) / p.nsplit);

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    const int col = n0 + li;
    const int colc = col < p.N ? col : p.N - 1;

    // A staging: ROWS x 32 floats per chunk = ROWS*8 float4; thread t -> row t/8 (+32 per pass), float4 t%8
    constexpr int A_PASSES = ROWS / 32;
    float4 ra[A_PASSES];
    auto load_a = [&](int chunk) {
        const int r0 = chunk * 32 + (tid & 7) * 4;
        const bool k_ok = r0 + 4 <= p.R;
#pragma unroll
        for (int ps = 0; ps < A_PASSES; ++ps) {
            const int row = ps * 32 + (tid >> 3);
            const bool ok = k_ok && row < p.M;
            const float4 v = *reinterpret_cast<const float4*>(p.X + (int64_t)(ok ? row : 0) * p.x_ld + (ok ? r0 : 0));
            ra[ps] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_a = [&](int buf) {
#pragma unroll
        for (int ps = 0; ps < A_PASSES; ++ps) {
            float* d = &As[buf][(ps * 32 + (tid >> 3)) * LDA + (tid & 7) * 4];
            d[0] = ra[ps].x; d[1] = ra[ps].y; d[2] = ra[ps].z; d[3] = ra[ps].w;
        }
    };
    // B fragments of one chunk straight from HBM; k-slot lh covers reduction indices r0 + lh*16 .. +15
    float b0[16], b1[16];
    auto load_b = [&](float (&b)[16], int chunk) {
        const int r0 = chunk * 32 + lh * 16;
        const int rc = (r0 + 16 <= p.R) ? r0 : 0;          // past-the-end group: valid dummy (its A rows are zero)
        if constexpr (TRANS) {
            const float4* src = reinterpret_cast<const float4*>(p.W + (int64_t)colc * p.w_ld + rc);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 v = src[j];
                b[4 * j] = v.x; b[4 * j + 1] = v.y; b[4 * j + 2] = v.z; b[4 * j + 3] = v.w;
            }
        } else {
            const float* src = p.W + (int64_t)rc * p.w_ld + colc;
#pragma unroll
            for (int kp = 0; kp < 16; ++kp) { b[kp] = *src; src += p.w_ld; }
        }
    };

    if (c_begin < c_end) {
        load_a(c_begin);
        load_b(b1, c_begin);
        store_a(0);
        __syncthreads();
        int buf = 0;
        for (int c = c_begin; c < c_end; ++c, buf ^= 1) {
            const bool more = c + 1 < c_end;
#pragma unroll
            for (int kp = 0; kp < 16; ++kp) b0[kp] = b1[kp];
            if (more) { load_a(c + 1); load_b(b1, c + 1); }
            const float* ap = &As[buf][li * LDA + lh * 16];
#pragma unroll
            for (int kp = 0; kp < 16; ++kp)
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[m * 32 * LDA + kp], b0[kp], acc[m], 0, 0, 0);
            if (more) store_a(buf ^ 1);
            __syncthreads();
        }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (row < p.M && col < p.N) p.Part[((int64_t)ks * p.M + row) * p.N + col] = acc[m][r];
        }
}


// ---- split-bf16 variants (arithmetic: bconv.hip) --------------------------------------------------------
// With three bf16 MFMA products per fp32 product the 64-row GEMMs need ~2.6 us of matrix-core time per
// 4096 x 4096 layer, so these kernels sit on the HBM stream of the weights (fwd, dgrad) / of dW (wgrad).
typedef __bf16 fbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 fbf16x4 __attribute__((ext_vector_type(4)));
typedef short fs16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void fsplit4(float a, float b, float c, float d, uint2& hi, uint2& lo) {
    fbf16x4 h, l;
    h[0] = (__bf16)a; h[1] = (__bf16)b; h[2] = (__bf16)c; h[3] = (__bf16)d;
    l[0] = (__bf16)(a - (float)h[0]); l[1] = (__bf16)(b - (float)h[1]);
    l[2] = (__bf16)(c - (float)h[2]); l[3] = (__bf16)(d - (float)h[3]);
    hi = __builtin_bit_cast(uint2, h);
    lo = __builtin_bit_cast(uint2, l);
}

// Same decomposition as fc_stream_kernel.  The activation chunk [ROWS][32] is split once into bf16 hi / lo
// planes in LDS (80-byte rows: 16 consecutive rows cover all banks for ds_read_b128); the weights go
// HBM -> registers (two chunks ahead) -> split -> MFMA B operand, each weight is converted exactly once.
template <bool TRANS, int MT>
__global__ __launch_bounds__(256) void fc_stream_b3_kernel(const FcParams p) {
    constexpr int ROWS = 32 * MT;
    constexpr int RSB = 80;                                   // bytes per row per plane
    __shared__ __attribute__((aligned(16))) unsigned char As[2][2][ROWS * RSB];     // [buffer][hi, lo]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int n0 = blockIdx.x * 128 + wave * 32;
    const int ks = blockIdx.y;
    const int c_begin = (int)((int64_t)p.chunks_total * ks / p.nsplit);
    const int c_end = (int)((int64_t)p.chunks_total * (ks + 1) / p.nsplit);

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    const int col = n0 + li;
    const int colc = col < p.N ? col : p.N - 1;

    constexpr int A_PASSES = ROWS / 32;
    constexpr int RING = MT <= 2 ? 6 : 4;
    float4 rq[RING][A_PASSES];
    auto load_a = [&](float4 (&ra)[A_PASSES], int chunk) {
        const int r0 = chunk * 32 + (tid & 7) * 4;
        const bool k_ok = r0 + 4 <= p.R;
#pragma unroll
        for (int ps = 0; ps < A_PASSES; ++ps) {
            const int row = ps * 32 + (tid >> 3);
            const bool ok = k_ok && row < p.M;
            const float4 v = *reinterpret_cast<const float4*>(p.X + (int64_t)(ok ? row : 0) * p.x_ld + (ok ? r0 : 0));
            ra[ps] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_a = [&](const float4 (&ra)[A_PASSES], int buf) {
#pragma unroll
        for (int ps = 0; ps < A_PASSES; ++ps) {
            uint2 hi, lo;
            fsplit4(ra[ps].x, ra[ps].y, ra[ps].z, ra[ps].w, hi, lo);
            const int off = (ps * 32 + (tid >> 3)) * RSB + (tid & 7) * 8;
            *reinterpret_cast<uint2*>(&As[buf][0][off]) = hi;
            *reinterpret_cast<uint2*>(&As[buf][1][off]) = lo;
        }
    };
    // lane half lh holds reduction indices r0 + lh*16 .. +15 of its column (k-step s uses elements 8s..8s+7)
    float bq[RING][16];
    auto load_b = [&](float (&b)[16], int chunk) {
        const int r0 = chunk * 32 + lh * 16;
        const int rc = (r0 + 16 <= p.R) ? r0 : 0;          // past-the-end group: valid dummy (its A columns are zero)
        if constexpr (TRANS) {
            const float4* src = reinterpret_cast<const float4*>(p.W + (int64_t)colc * p.w_ld + rc);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 v = src[j];
                b[4 * j] = v.x; b[4 * j + 1] = v.y; b[4 * j + 2] = v.z; b[4 * j + 3] = v.w;
            }
        } else {
            const float* src = p.W + (int64_t)rc * p.w_ld + colc;
#pragma unroll
            for (int kp = 0; kp < 16; ++kp) { b[kp] = *src; src += p.w_ld; }
        }
    };
    auto mma = [&](const float (&b)[16], int buf) {
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            uint2 h0, l0, h1, l1;
            fsplit4(b[8 * s2], b[8 * s2 + 1], b[8 * s2 + 2], b[8 * s2 + 3], h0, l0);
            fsplit4(b[8 * s2 + 4], b[8 * s2 + 5], b[8 * s2 + 6], b[8 * s2 + 7], h1, l1);
            const fbf16x8 bh = __builtin_bit_cast(fbf16x8, make_uint4(h0.x, h0.y, h1.x, h1.y));
            const fbf16x8 bl = __builtin_bit_cast(fbf16x8, make_uint4(l0.x, l0.y, l1.x, l1.y));
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int off = (m * 32 + li) * RSB + (lh * 16 + 8 * s2) * 2;
                const fbf16x8 ah = __builtin_bit_cast(fbf16x8, *reinterpret_cast<const uint4*>(&As[buf][0][off]));
                const fbf16x8 al = __builtin_bit_cast(fbf16x8, *reinterpret_cast<const uint4*>(&As[buf][1][off]));
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[m], 0, 0, 0);
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[m], 0, 0, 0);
            }
        }
    };

    if (c_begin < c_end) {
        // chunk c in use, c+1 .. c+RING-1 in flight (activations and weights of a chunk are requested together, so
        // waiting for chunk c+1's activations never waits for younger weight loads: vmcnt retires in order);
        // unrolled by RING so every register-set index is a compile-time constant
#pragma unroll
        for (int u = 0; u < RING - 1; ++u) {
            const int cu = c_begin + u < c_end ? c_begin + u : c_end - 1;
            load_a(rq[u], cu);
            load_b(bq[u], cu);
        }
        store_a(rq[0], 0);
        __syncthreads();
        int buf = 0;
        constexpr int NCH = 16;             // 4096-long reduction in 8 slices: straight-line code, exact s_waitcnt counts
        if (c_end - c_begin == NCH) {
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                if (i + RING - 1 < NCH) {
                    load_a(rq[(i + RING - 1) % RING], c_begin + i + RING - 1);
                    load_b(bq[(i + RING - 1) % RING], c_begin + i + RING - 1);
                }
                __builtin_amdgcn_sched_barrier(0);      // keep the look-ahead loads at the top of the step
                mma(bq[i % RING], i & 1);
                if (i + 1 < NCH) store_a(rq[(i + 1) % RING], (i + 1) & 1);
                __syncthreads();
            }
        } else
        for (int c = c_begin; c < c_end; c += RING) {
#pragma unroll
            for (int u = 0; u < RING; ++u) {
                const int cc = c + u;
                if (cc < c_end) {
                    const int cn = cc + RING - 1 < c_end ? cc + RING - 1 : c_end - 1;
                    load_a(rq[(u + RING - 1) % RING], cn);
                    load_b(bq[(u + RING - 1) % RING], cn);
                    mma(bq[u], buf);
                    if (cc + 1 < c_end) store_a(rq[(u + 1) % RING], buf ^ 1);
                    __syncthreads();
                    buf ^= 1;
                }
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (row < p.M && col < p.N) p.Part[((int64_t)ks * p.M + row) * p.N + col] = acc[m][r];
        }
}

// dW[c][k] = sum_b x[b][c] dy[b][k].  A workgroup owns a 128 x 128 tile of dW: the two operand panels
// x[64 b][128 c] and dy[64 b][128 k] are fetched ONCE with 16-byte coalesced loads (512-byte row
// segments) into LDS, each wave then computes a 64 x 64 quadrant (4 accumulators) from conflict-free
// ds_read_b32 fragments.  Operand traffic = output traffic (67 MB per layer); write-bound.
struct FcWgradParams {
    const float* X; const float* DY; float* dW; float* db;
    int B, C, K, x_ld, dy_ld;
    int ktiles;
    // fused optimiser (mv3d_fc_wgrad_adam): dW is the PARAMETER matrix, updated in place from the accumulators
    float* M1; float* V2;            // Adam slots of the matrix
    const float* state;              // device Adam state (include/mv3d_hip.h: MV3D_ADAM_*)
};

// TF ApplyAdam on one element, the arithmetic of adam_kernel (elem.hip) operation for operation (no contraction)
__device__ __forceinline__ void adam_elem(float g, float& pv, float& mv, float& vv, float alpha, float omb1, float omb2, float eps) {
#pragma clang fp contract(off)
    mv += (g - mv) * omb1;
    vv += (g * g - vv) * omb2;
    pv -= (mv * alpha) / (sqrtf(vv) + eps);
}

__global__ __launch_bounds__(256) void fc_wgrad_kernel(const FcWgradParams p) {
    __shared__ __attribute__((aligned(16))) float xs[64 * 128];
    __shared__ __attribute__((aligned(16))) float dys[64 * 128];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int kt = blockIdx.x % p.ktiles, ct = blockIdx.x / p.ktiles;
    const int c0 = ct * 128, k0 = kt * 128;
    const int ws = wave >> 1, wt = wave & 1;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float sb[2] = {0.f, 0.f};
    for (int b0 = 0; b0 < p.B; b0 += 64) {
        if (b0) __syncthreads();
        // panels: 64 rows x 32 float4 each; thread t -> float4 column t%32, rows t/32 + 8*j
        float4 vx[8], vy[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = b0 + (tid >> 5) + 8 * j;
            const int cx = c0 + (tid & 31) * 4, ky = k0 + (tid & 31) * 4;
            const bool okx = row < p.B && cx + 4 <= p.C, oky = row < p.B && ky + 4 <= p.K;
            const float4 tx = *reinterpret_cast<const float4*>(p.X + (okx ? (int64_t)row * p.x_ld + cx : 0));
            const float4 ty = *reinterpret_cast<const float4*>(p.DY + (oky ? (int64_t)row * p.dy_ld + ky : 0));
            vx[j] = okx ? tx : make_float4(0.f, 0.f, 0.f, 0.f);
            vy[j] = oky ? ty : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = (tid >> 5) + 8 * j;
            *reinterpret_cast<float4*>(&xs[r * 128 + (tid & 31) * 4]) = vx[j];
            *reinterpret_cast<float4*>(&dys[r * 128 + (tid & 31) * 4]) = vy[j];
        }
        __syncthreads();
        const float* pa = xs + lh * 128 + ws * 64 + li;
        const float* pb = dys + lh * 128 + wt * 64 + li;
#pragma unroll 4
        for (int bp = 0; bp < 32; ++bp) {
            const float a0 = pa[bp * 256], a1 = pa[bp * 256 + 32];
            const float q0 = pb[bp * 256], q1 = pb[bp * 256 + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, q0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, q1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, q0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, q1, acc[1][1], 0, 0, 0);
            sb[0] += q0; sb[1] += q1;
        }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int k = k0 + wt * 64 + t * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = c0 + ws * 64 + s * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (c < p.C && k < p.K) p.dW[(int64_t)c * p.K + k] = acc[s][t][r];
            }
        }
    if (p.db && ct == 0 && ws == 0) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float v = sb[t] + __shfl_xor(sb[t], 32);
            const int k = k0 + wt * 64 + t * 32 + li;
            if (lh == 0 && k < p.K) p.db[k] = v;
        }
    }
}


__device__ __forceinline__ uint2 ftr_read(const unsigned char* lds_ptr) {
    fs16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) fs16x4*)lds_ptr);
    return __builtin_bit_cast(uint2, v);
}

// dW tile 128 x 128 per workgroup, 64 x 64 quadrant per wave.  The batch is the MFMA k index: both operands
// are needed as "8 consecutive batch rows of one column" per lane while the panels arrive row-major, so they
// are staged as bf16 hi / lo planes [32 b][128 columns] (320-byte rows: 4 consecutive rows x 32 B land on
// distinct banks) and fetched with the transposing ds_read_b64_tr_b16.
template <bool ADAM>
__global__ __launch_bounds__(256) void fc_wgrad_b3_kernel(const FcWgradParams p) {
    constexpr int RS = 320;
    __shared__ __attribute__((aligned(16))) unsigned char xs[2][32 * RS];      // [hi, lo]
    __shared__ __attribute__((aligned(16))) unsigned char dys[2][32 * RS];
    __shared__ float bred[8][128];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int kt = blockIdx.x % p.ktiles, ct = blockIdx.x / p.ktiles;
    const int c0 = ct * 128, k0 = kt * 128;
    const int ws = wave >> 1, wt = wave & 1;
    const int g16 = (lane >> 4) & 1, q = (lane >> 2) & 3, pp = lane & 3;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float4 bs = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b0 = 0; b0 < p.B; b0 += 32) {
        if (b0) __syncthreads();
        // panels: 32 rows x 32 float4 each; thread t -> float4 column t%32, rows t/32 + 8*j
        float4 vx[4], vy[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = b0 + (tid >> 5) + 8 * j;
            const int cx = c0 + (tid & 31) * 4, ky = k0 + (tid & 31) * 4;
            const bool okx = row < p.B && cx + 4 <= p.C, oky = row < p.B && ky + 4 <= p.K;
            const float4 tx = *reinterpret_cast<const float4*>(p.X + (okx ? (int64_t)row * p.x_ld + cx : 0));
            const float4 ty = *reinterpret_cast<const float4*>(p.DY + (oky ? (int64_t)row * p.dy_ld + ky : 0));
            vx[j] = okx ? tx : make_float4(0.f, 0.f, 0.f, 0.f);
            vy[j] = oky ? ty : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int off = ((tid >> 5) + 8 * j) * RS + (tid & 31) * 8;
            uint2 hi, lo;
            fsplit4(vx[j].x, vx[j].y, vx[j].z, vx[j].w, hi, lo);
            *reinterpret_cast<uint2*>(&xs[0][off]) = hi;
            *reinterpret_cast<uint2*>(&xs[1][off]) = lo;
            fsplit4(vy[j].x, vy[j].y, vy[j].z, vy[j].w, hi, lo);
            *reinterpret_cast<uint2*>(&dys[0][off]) = hi;
            *reinterpret_cast<uint2*>(&dys[1][off]) = lo;
            bs.x += vy[j].x; bs.y += vy[j].y; bs.z += vy[j].z; bs.w += vy[j].w;
        }
        __syncthreads();
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            // address supplier role: batch rows 16 s2 + 8 lh + 4 r + q, r = 0, 1
            const int row0 = 16 * s2 + 8 * lh + q;
            const int ra0 = row0 * RS + (16 * g16 + 4 * pp) * 2, ra1 = ra0 + 4 * RS;
            fbf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int ca = (ws * 64 + t * 32) * 2, cb = (wt * 64 + t * 32) * 2;
                uint2 u0 = ftr_read(&xs[0][ra0 + ca]), u1 = ftr_read(&xs[0][ra1 + ca]);
                ah[t] = __builtin_bit_cast(fbf16x8, make_uint4(u0.x, u0.y, u1.x, u1.y));
                u0 = ftr_read(&xs[1][ra0 + ca]); u1 = ftr_read(&xs[1][ra1 + ca]);
                al[t] = __builtin_bit_cast(fbf16x8, make_uint4(u0.x, u0.y, u1.x, u1.y));
                u0 = ftr_read(&dys[0][ra0 + cb]); u1 = ftr_read(&dys[0][ra1 + cb]);
                bh[t] = __builtin_bit_cast(fbf16x8, make_uint4(u0.x, u0.y, u1.x, u1.y));
                u0 = ftr_read(&dys[1][ra0 + cb]); u1 = ftr_read(&dys[1][ra1 + cb]);
                bl[t] = __builtin_bit_cast(fbf16x8, make_uint4(u0.x, u0.y, u1.x, u1.y));
            }
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[a], bh[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bl[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[a], bh[b], acc[a][b], 0, 0, 0);
                }
        }
    }
    float alpha = 0.f, omb1 = 0.f, omb2 = 0.f, eps = 0.f, gscale = 1.f;
    if constexpr (ADAM) {
        const float lr = p.state[0], b1 = p.state[1], b2 = p.state[2], b1p = p.state[4], b2p = p.state[5];
        eps = p.state[3]; gscale = p.state[6];
        {
#pragma clang fp contract(off)
            alpha = lr * sqrtf(1.0f - b2p) / (1.0f - b1p);
            omb1 = 1.0f - b1; omb2 = 1.0f - b2;
        }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int k = k0 + wt * 64 + t * 32 + li;
            if constexpr (ADAM) {
                // the gradient never goes to HBM: parameter and slots are read, updated and written back here.  All sixteen rows of
                // an accumulator tile are requested before the first store (48 loads in flight per lane): vmcnt retires in order and
                // counts stores, so with four rows per round (round 2's first version) every round paid a load round trip behind the
                // previous round's store acknowledgements -- sixteen per workgroup, 95 us per 67 MB matrix; now four, 86 us = 4.8 TB/s.
                // (Requesting the next tile before storing this one needs a second register set: 197 VGPRs, one workgroup fewer per
                // CU, measured slower.)
                {
                    float pv[16], mv[16], vv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int c = c0 + ws * 64 + s * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        const int64_t idx = (c < p.C && k < p.K) ? (int64_t)c * p.K + k : 0;
                        pv[r] = p.dW[idx]; mv[r] = p.M1[idx]; vv[r] = p.V2[idx];
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) adam_elem(acc[s][t][r] * gscale, pv[r], mv[r], vv[r], alpha, omb1, omb2, eps);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int c = c0 + ws * 64 + s * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        if (c < p.C && k < p.K) {
                            const int64_t idx = (int64_t)c * p.K + k;
                            p.dW[idx] = pv[r]; p.M1[idx] = mv[r]; p.V2[idx] = vv[r];
                        }
                    }
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int c = c0 + ws * 64 + s * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (c < p.C && k < p.K) p.dW[(int64_t)c * p.K + k] = acc[s][t][r];
                }
            }
        }
    if (p.db && ct == 0) {
        // bias gradient: per-thread fp32 column sums of dy (4 columns, rows t/32 + 8j), combined in a fixed order
        float* br = &bred[tid >> 5][(tid & 31) * 4];
        br[0] = bs.x; br[1] = bs.y; br[2] = bs.z; br[3] = bs.w;
        __syncthreads();
        if (tid < 128) {
            float t = 0.f;
#pragma unroll
            for (int r = 0; r < 8; ++r) t += bred[r][tid];
            if (k0 + tid < p.K) p.db[k0 + tid] = t;
        }
    }
}

static bool fc_stream_ok(int B, int in, int out, const void* x, int x_ld, const void* W, bool trans) {
    if (B < 1 || B > 128) return false;
    const int R = trans ? out : in;
    if (R < 256 || R % 16 != 0 || x_ld % 4 != 0) return false;
    if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(W) & 15)) return false;
    if (trans && (out % 4 != 0)) return false;
    return !(disabled_paths() & 16);
}

static int fc_target_wgs() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("MV3D_FC_WGS"); v = e ? atoi(e) : 256; }
    return v;
}

size_t fc_stream_ws_bytes(int B, int in, int out, bool trans) {
    const int R = trans ? out : in, N = trans ? in : out;
    const int chunks = cdiv(R, 32);
    int nsplit = std::max(1, std::min(chunks / 4, cdiv(fc_target_wgs(), cdiv(N, 128))));
    return (size_t)nsplit * B * N * sizeof(float);
}

// returns 1 when not applicable (caller falls back to the generic path)
int try_fc_stream(bool trans, int B, int in, int out, const void* x, int x_ld, const void* W, void* y, int y_ld,
                  const mv3d_epilogue* epi, void* ws, size_t wsb, void* stream, const char* who,
                  void (*fill_epi)(IgemmParams&, const mv3d_epilogue*), void (*launch_epi)(const IgemmParams&, int, hipStream_t)) {
    if (!fc_stream_ok(B, in, out, x, x_ld, W, trans)) return 1;
    FcParams p = {};
    p.X = (const float*)x; p.W = (const float*)W; p.Part = (float*)ws;
    p.M = B; p.R = trans ? out : in; p.N = trans ? in : out; p.x_ld = x_ld; p.w_ld = out;
    p.chunks_total = cdiv(p.R, 32);
    p.nsplit = std::max(1, std::min(p.chunks_total / 4, cdiv(fc_target_wgs(), cdiv(p.N, 128))));
    const size_t need = (size_t)p.nsplit * B * p.N * sizeof(float);
    if (!ws || wsb < need) return 1;
    dim3 grid(cdiv(p.N, 128), p.nsplit);
    const int MT = cdiv(B, 32);
    const double flops = 2.0 * B * (double)in * out, bytes = 4.0 * ((double)in * out + (double)B * (in + out));
    const bool b3 = !(disabled_paths() & 4096);
    int rc = dispatch(stream, OpInfo{b3 ? (trans ? "fc_stream_b3<dgrad>" : "fc_stream_b3<fwd>") : (trans ? "fc_stream<dgrad>" : "fc_stream<fwd>"), flops, bytes}, [=](hipStream_t s) {
        if (b3) {
            if (trans) {
                if (MT <= 1) fc_stream_b3_kernel<true, 1><<<grid, 256, 0, s>>>(p);
                else if (MT == 2) fc_stream_b3_kernel<true, 2><<<grid, 256, 0, s>>>(p);
                else fc_stream_b3_kernel<true, 4><<<grid, 256, 0, s>>>(p);
            } else {
                if (MT <= 1) fc_stream_b3_kernel<false, 1><<<grid, 256, 0, s>>>(p);
                else if (MT == 2) fc_stream_b3_kernel<false, 2><<<grid, 256, 0, s>>>(p);
                else fc_stream_b3_kernel<false, 4><<<grid, 256, 0, s>>>(p);
            }
        } else if (trans) {
            if (MT <= 1) fc_stream_kernel<true, 1><<<grid, 256, 0, s>>>(p);
            else if (MT == 2) fc_stream_kernel<true, 2><<<grid, 256, 0, s>>>(p);
            else fc_stream_kernel<true, 4><<<grid, 256, 0, s>>>(p);
        } else {
            if (MT <= 1) fc_stream_kernel<false, 1><<<grid, 256, 0, s>>>(p);
            else if (MT == 2) fc_stream_kernel<false, 2><<<grid, 256, 0, s>>>(p);
            else fc_stream_kernel<false, 4><<<grid, 256, 0, s>>>(p);
        }
        return launched(who);
    });
    if (rc != MV3D_OK) return rc;
    // reduction slices + bias / activation / mask: the split-K epilogue of conv.hip on a [1, B] "image"
    IgemmParams e = {};
    e.N = 1; e.Hc = 1; e.Wc = B; e.Cc = p.N; e.c_ld = y_ld; e.Out = (float*)y; e.Part = (float*)ws; e.ksplit = p.nsplit;
    fill_epi(e, epi);
    const int64_t total = (int64_t)B * p.N;
    const int blocks = (int)std::min<int64_t>(cdiv64(total, 256), 4096);
    return dispatch(stream, OpInfo{"fc_splitk_epilogue", 0.0, (double)total * 4.0 * (p.nsplit + 1)}, [=](hipStream_t s) {
        launch_epi(e, blocks, s);
        return launched("igemm_splitk_epilogue");
    });
}

int try_fc_wgrad(int B, int in, int out, const void* x, int x_ld, const void* dy, int dy_ld, void* dM, void* db, void* stream, const char* who) {
    if (B < 2 || in < 64 || out < 64 || (disabled_paths() & 16)) return 1;
    if (in % 4 || out % 4 || x_ld % 4 || dy_ld % 4 || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(dy) & 15)) return 1;
    FcWgradParams p = {(const float*)x, (const float*)dy, (float*)dM, (float*)db, B, in, out, x_ld, dy_ld, cdiv(out, 128), nullptr, nullptr, nullptr};
    const int items = cdiv(in, 128) * p.ktiles;         // one workgroup per 128 x 128 tile
    const double flops = 2.0 * B * (double)in * out, bytes = 4.0 * ((double)in * out + (double)B * (in + out));
    const bool b3 = !(disabled_paths() & 4096);
    return dispatch(stream, OpInfo{b3 ? "fc_wgrad_b3" : "fc_wgrad", flops, bytes}, [=](hipStream_t s) {
        if (b3) fc_wgrad_b3_kernel<false><<<items, 256, 0, s>>>(p);
        else fc_wgrad_kernel<<<items, 256, 0, s>>>(p);
        return launched(who);
    });
}

// Filter gradient of a large fc layer with tf.train.AdamOptimizer's update of that matrix fused into the epilogue: returns 1 when
// the layer is not one the matrix-core kernel takes (the caller then runs mv3d_fc_wgrad + mv3d_adam_step_dev)
int try_fc_wgrad_adam(int B, int in, int out, const void* x, int x_ld, const void* dy, int dy_ld, void* P, void* M1, void* V2, void* db,
                      const void* state, void* stream, const char* who) {
    if (B < 2 || in < 64 || out < 64 || (disabled_paths() & (16 | 4096))) return 1;
    if (in % 4 || out % 4 || x_ld % 4 || dy_ld % 4 || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(dy) & 15)) return 1;
    FcWgradParams p = {(const float*)x, (const float*)dy, (float*)P, (float*)db, B, in, out, x_ld, dy_ld, cdiv(out, 128), (float*)M1, (float*)V2, (const float*)state};
    const int items = cdiv(in, 128) * p.ktiles;
    // algorithmic bytes: p, m, v read and written once (24 B per parameter), operands once
    const double flops = 2.0 * B * (double)in * out, bytes = 24.0 * (double)in * out + 4.0 * (double)B * (in + out);
    return dispatch(stream, OpInfo{"fc_wgrad_adam_b3", flops, bytes}, [=](hipStream_t s) {
        fc_wgrad_b3_kernel<true><<<items, 256, 0, s>>>(p);
        return launched(who);
    });
}

}  // namespace mv3d
