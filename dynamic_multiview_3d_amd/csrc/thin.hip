// Layers with a 1..3-channel image side at full resolution (appearance_flow_model.py:88,125: e0 = conv2d_msra(image0, 32, 5, 5, 2, 2)
// and the flow head deconv2d_msra(d1_0, [B, 128, 128, 2], 5, 5, 2, 2); main_model.py:60,74: the depth / mask towers and heads).
// They move ~46 MB for ~1 GFLOP: HBM-bound, and the kernels of conv.hip (one 32-pixel item per wave, operands straight from
// global memory, thousands of short workgroups) ran at a quarter of what the bytes allow -- dispatch- and latency-bound.
//
// smallc_band_kernel (image -> feature direction, stride 2: conv forward of e0, data gradient of the flow / depth heads):
//   * a workgroup owns a BAND of output rows of one image x 32 filters.  All input rows of the band are fetched at once
//     (every load of the workgroup in flight together: one memory round trip), split into bf16 hi / lo planes in LDS;
//   * the filter row (kw * C <= 16 contiguous elements of an NHWC row) is the reduction index of one MFMA k-step, so the A
//     fragment of an output pixel is 16 consecutive plane elements starting at column 2 * ow - pad: four dword LDS reads per
//     lane half (plane element 0 is column -pad, so the start 2 C ow + 8 lh is dword aligned);
//   * the filter fragments stay in registers for the whole band; a wave walks 32-pixel tiles: 5 k-steps x 3 products, then
//     bias / activation / gradient mask and sixteen 128-byte row stores.  No global loads inside the tile loop (forward), so
//     the in-order vmcnt queue only ever holds stores.
#include "conv_common.h"
#include <algorithm>
#include <type_traits>

namespace mv3d {

typedef __bf16 tbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 tbf16x2 __attribute__((ext_vector_type(2)));

struct BandParams {
    const float* X; const float* Wt; float* Y;
    int N, H, W, C, Ho, Wo, K, y_ld;
    int kh, kw, sh, sw, pt, pl;
    int RB;                  // output rows per band
    int nrows;               // input rows per band: (RB - 1) * sh + kh
    int xoff;                // plane element of input element 0 = pl * C (so that fragment starts, 2 C ow + 8 lh, are dword aligned)
    int rp;                  // plane row pitch in elements (even)
    int bands_per_img, wtiles;
    unsigned inv_row_f4;     // ceil(2^32 / (W * C / 4)): exact quotients for indices below 2^16
    const float* bias; int act; float leak;
    int gact; float gleak; const float* gref; int g_ld;
    int dbg;                 // MV3D_DBG bit 32: in-kernel stamps
};

__device__ __forceinline__ unsigned tpack2(float a, float b) {
    tbf16x2 v; v[0] = (__bf16)a; v[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, v);
}

// In-kernel stamps (MV3D_DBG bit 32 only): wave w of workgroup b < 512 writes the shader clock of event k to t_stamps[(b * 4 + w) * 16 + k];
// read back with mv3d_debug_band_stamps().  No output value depends on them.
__device__ unsigned long long t_stamps[512 * 4 * 16];
__device__ __forceinline__ void tstamp(bool on, int wave, int lane, int& k) {
    if (on) {
        const unsigned long long t = __builtin_readcyclecounter();
        if (lane == 0 && k < 16 && blockIdx.x < 512 && blockIdx.y == 0) t_stamps[((int)blockIdx.x * 4 + wave) * 16 + k] = t;
        ++k;
    }
}

template <int KH, bool HAS_G>
__global__ __launch_bounds__(256) void smallc_band_kernel(const BandParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    __shared__ __attribute__((aligned(16))) uint4 fsh[KH][2][64];          // [filter row][hi, lo][lane]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int Cf = p.kw * p.C;                                             // <= 16
    const int k0 = blockIdx.y * 32;
    const int band = blockIdx.x % p.bands_per_img, n = blockIdx.x / p.bands_per_img;
    const int oh0 = band * p.RB;
    const int plane_bytes = p.nrows * p.rp * 2;
    const bool st = (p.dbg & 32) != 0;
    int sk = 0;
    tstamp(st, wave, lane, sk);                                            // 0: start
    unsigned char* const hi_pl = lds;
    unsigned char* const lo_pl = lds + plane_bytes;

    // ---- the band's input rows: fp32 global -> bf16 hi | lo planes, and the filter rows -> fragments (split once per workgroup,
    // through LDS).  All loads of the workgroup are issued before the first conversion, rows first: branch-free through a
    // buffer descriptor (rows outside the image / indices past the band read zeros), so that hipcc's vmcnt counts stay exact and the
    // row conversion does not wait for the filter loads behind it.  (The filter used to be loaded AND converted first: two memory
    // round trips in a row, 15 k cycles before the first product.)
    {
        const int row_f4 = (p.W * p.C) >> 2;                               // float4 pieces per input row
        const int total = p.nrows * row_f4;
        const int ih0 = oh0 * p.sh - p.pt;
        constexpr int UB = 8;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.X) + (int64_t)n * p.H * p.W * p.C, 0, p.H * p.W * p.C * 4, 0x00020000);
        auto convert = [&](const float4& t, int lof) {
            const float h0 = (float)(__bf16)t.x, h1 = (float)(__bf16)t.y, h2 = (float)(__bf16)t.z, h3 = (float)(__bf16)t.w;
            const unsigned a = tpack2(t.x, t.y), b = tpack2(t.z, t.w);
            const unsigned c = tpack2(t.x - h0, t.y - h1), d = tpack2(t.z - h2, t.w - h3);
            if (p.xoff & 1) {          // rows start on an odd plane element (pad * C odd): 2-byte aligned pieces
                unsigned short* hp = reinterpret_cast<unsigned short*>(hi_pl + lof);
                unsigned short* lp = reinterpret_cast<unsigned short*>(lo_pl + lof);
                hp[0] = (unsigned short)a; hp[1] = (unsigned short)(a >> 16); hp[2] = (unsigned short)b; hp[3] = (unsigned short)(b >> 16);
                lp[0] = (unsigned short)c; lp[1] = (unsigned short)(c >> 16); lp[2] = (unsigned short)d; lp[3] = (unsigned short)(d >> 16);
            } else {
                *reinterpret_cast<unsigned*>(hi_pl + lof) = a;
                *reinterpret_cast<unsigned*>(hi_pl + lof + 4) = b;
                *reinterpret_cast<unsigned*>(lo_pl + lof) = c;
                *reinterpret_cast<unsigned*>(lo_pl + lof + 4) = d;
            }
        };
        auto row_piece = [&](int idx, float4& v, int& lof) {
            const int r = (int)__umulhi((unsigned)idx, p.inv_row_f4), f = idx - r * row_f4;
            const int ih = ih0 + r;
            const bool ok = idx < total && (unsigned)ih < (unsigned)p.H;
            v = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? (ih * p.W * p.C + 4 * f) * 4 : 0x7ffffff0, 0, 0));
            lof = idx < total ? (r * p.rp + p.xoff + 4 * f) * 2 : -1;
        };
        {
            float4 v[UB];
            int lofs[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) row_piece(u * 256 + tid, v[u], lofs[u]);
            // filter rows of this wave (wave 0: rows 0 and 4 of a 5-row filter): unconditional loads from clamped addresses
            float fv[2][8];
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int pr = min(wave + 4 * it, KH - 1);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int e = min(8 * lh + j, Cf - 1), kk = min(k0 + li, p.K - 1);
                    fv[it][j] = p.Wt[((int64_t)pr * Cf + e) * p.K + kk];
                }
            }
#pragma unroll
            for (int u = 0; u < UB; ++u)
                if (lofs[u] >= 0) convert(v[u], lofs[u]);
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int pr = wave + 4 * it;
                if (pr < KH) {
                    tbf16x8 h, l;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float w = (8 * lh + j < Cf && k0 + li < p.K) ? fv[it][j] : 0.f;
                        h[j] = (__bf16)w; l[j] = (__bf16)(w - (float)h[j]);
                    }
                    fsh[pr][0][lane] = __builtin_bit_cast(uint4, h);
                    fsh[pr][1][lane] = __builtin_bit_cast(uint4, l);
                }
            }
        }
        for (int base = 256 * UB; base < total; base += 256 * UB) {       // wider images: the rest of the band's rows
            float4 v[UB];
            int lofs[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) row_piece(base + u * 256 + tid, v[u], lofs[u]);
#pragma unroll
            for (int u = 0; u < UB; ++u)
                if (lofs[u] >= 0) convert(v[u], lofs[u]);
        }
        tstamp(st, wave, lane, sk);                                        // 1: rows split into the planes
        // zero pads left and right of every row (columns outside the image), element by element (they need not be dword aligned)
        const int lpad = p.xoff, per_row = p.rp - p.W * p.C;              // left pad + right pad elements (host: <= 32)
        const int j = tid & 31;
        if (j < per_row) {
            const int e = j < lpad ? j : p.W * p.C + j;                    // = xoff + W * C + (j - lpad)
            for (int r = tid >> 5; r < p.nrows; r += 8) {
                reinterpret_cast<unsigned short*>(hi_pl)[r * p.rp + e] = 0;
                reinterpret_cast<unsigned short*>(lo_pl)[r * p.rp + e] = 0;
            }
        }
    }
    tstamp(st, wave, lane, sk);                                            // 2: pads zeroed
    tstamp(st, wave, lane, sk);                                            // 3: filter fragments written
    __syncthreads();
    tstamp(st, wave, lane, sk);                                            // 4: barrier passed
    tbf16x8 bh[KH], bl[KH];
#pragma unroll
    for (int pr = 0; pr < KH; ++pr) {
        bh[pr] = __builtin_bit_cast(tbf16x8, fsh[pr][0][lane]);
        bl[pr] = __builtin_bit_cast(tbf16x8, fsh[pr][1][lane]);
    }

    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.Y, 0, (int)((int64_t)p.N * p.Ho * p.Wo * p.y_ld * 4), 0x00020000);
    const float bias = (p.bias && k0 + li < p.K) ? p.bias[k0 + li] : 0.f;
    const int out_lane = k0 + li < p.K ? (4 * lh * p.y_ld + k0 + li) * 4 : (int)0x80000000;     // filters beyond K: stores dropped by the bounds check
    const float c1 = p.act == MV3D_ACT_NONE ? 1.f : (p.act == MV3D_ACT_LRELU ? 0.5f * (1.0f + p.leak) : 0.5f);
    const float c2 = p.act == MV3D_ACT_NONE ? 0.f : (p.act == MV3D_ACT_LRELU ? 0.5f * (1.0f - p.leak) : 0.5f);
    const bool is_relu = p.act == MV3D_ACT_RELU;
    const float g1 = p.gact == MV3D_ACT_LRELU ? 0.5f * (1.0f + p.gleak) : 0.5f;
    const float g2 = p.gact == MV3D_ACT_LRELU ? 0.5f * (1.0f - p.gleak) : 0.5f;
    const bool g_relu = p.gact == MV3D_ACT_RELU;
    const int ntiles = p.RB * p.wtiles;
    // HAS_G (data gradient): the saved outputs behind the activation-gradient mask are the only global loads of the tile loop.
    // vmcnt retires in order and counts stores, so a load issued behind a tile's stores waits for their acknowledgements
    // (~9 k cycles per tile when that happened to every tile: in-kernel stamps); the mask values of tile t + 1 are therefore
    // requested BEFORE the stores of tile t.  Without a mask the loop has no loads at all and never waits for a store.
    float gm[16];
    auto load_mask = [&](int t) {
        if constexpr (HAS_G) {
            const int wt = t % p.wtiles, orow = t / p.wtiles;
            const int pix0 = (n * p.Ho + oh0 + orow) * p.Wo + wt * 32;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int px = pix0 + (q & 3) + 8 * (q >> 2) + 4 * lh;
                gm[q] = (t < ntiles && k0 + li < p.K) ? p.gref[(int64_t)px * p.g_ld + k0 + li] : 0.f;
            }
        }
    };
    load_mask(wave);
    for (int t = wave; t < ntiles; t += 4) {
        const int wt = t % p.wtiles, orow = t / p.wtiles;                  // wave-uniform
        const int oh = oh0 + orow;
        // first plane element of this lane's fragment: column (wt * 32 + li) * sw - pl, element 8 lh of the filter row
        const int e0 = ((wt * 32 + li) * p.sw - p.pl) * p.C + 8 * lh + p.xoff;
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
        for (int pr = 0; pr < KH; ++pr) {
            const int off = ((orow * p.sh + pr) * p.rp + e0) * 2;
            const unsigned* ph = reinterpret_cast<const unsigned*>(hi_pl + off);
            const unsigned* pq = reinterpret_cast<const unsigned*>(lo_pl + off);
            const uint4 ah4 = make_uint4(ph[0], ph[1], ph[2], ph[3]), al4 = make_uint4(pq[0], pq[1], pq[2], pq[3]);
            const tbf16x8 ah = __builtin_bit_cast(tbf16x8, ah4), al = __builtin_bit_cast(tbf16x8, al4);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[pr], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[pr], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[pr], acc, 0, 0, 0);
        }
        // branch-free epilogue (tanh is not taken here: try_smallc_band): y = c1 x + c2 |x| as tf_utils.py:25-33 writes it (two
        // products, one sum; relu keeps -0.0 for x < 0), the mask from the saved output as common.h act_grad_from_out
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float xv = acc[q] + bias;
            float y = __fadd_rn(__fmul_rn(c1, xv), __fmul_rn(c2, fabsf(xv)));
            y = (is_relu && xv < 0.0f) ? -0.0f : y;
            if constexpr (HAS_G) {
                const float go = gm[q];
                const bool neg = g_relu ? (__float_as_uint(go) >> 31) != 0 : go < 0.0f;
                const float sgn = go > 0.0f ? 1.0f : (neg ? -1.0f : 0.0f);
                y *= g1 + g2 * sgn;
            }
            v[q] = y;
        }
        load_mask(t + 4);                                                  // next tile's mask values: in front of this tile's stores
        __builtin_amdgcn_sched_barrier(0);
        const int pix0 = (n * p.Ho + oh) * p.Wo + wt * 32;                 // wave-uniform: first pixel of the tile
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int px = pix0 + (q & 3) + 8 * (q >> 2);                  // + 4 lh in the lane part
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[q]), yr, out_lane, px * p.y_ld * 4, 0);
        }
        tstamp(st, wave, lane, sk);                                        // 5 + i: tile i done (stores issued)
    }
    if (st) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tstamp(st, wave, lane, sk); }      // last: stores acknowledged
}

// thin_head_kernel (feature -> image direction, stride 2, 1..4 image channels, 32 feature channels: the flow / rgb / depth / mask
// heads, appearance_flow_model.py:125, main_model.py:74-79).  The kernel it replaces (conv.hip thin_deconv_s2_tile_kernel)
// multiplies on the vector ALUs: 0.84 GFLOP of packed FMAs per launch, 38 us for 42 MB.  Here the products go to the matrix
// cores as ONE small GEMM per workgroup and the transposed convolution becomes a gather:
//   T[input pixel][(filter row P, filter column Q, image channel c)] = sum_k x[pixel][k] W[P][Q][c][k]      (N = KS^2 CC <= 128 columns)
//   out[2 u + a][2 v + b][c] = bias + sum over the (dh, dw) with P = a + PT - 2 dh, Q = b + PT - 2 dw inside the filter of
//                              T[(u + dh, v + dw)][P][Q][c]
// A workgroup owns 8 x 16 positions of the input grid: the 10 x 18 halo (x 32 channels) is fetched in one round of loads and
// split into bf16 hi / lo records in LDS (144-byte pitch), its 6 blocks of 32 pixels are dealt to the four waves, T goes to LDS as
// fp32 with an odd pitch, and after one barrier a thread adds up the <= 15 T values of each of its output elements in a fixed
// order (thread = (output row parity, position): its two output pixels x CC channels are 8 CC contiguous bytes, a tile row of 16
// threads stores 128 CC contiguous bytes).  T of halo pixels is computed by every workgroup that needs it (1.4 x the products:
// still 2 us of matrix-core time per launch).
template <int KS, int CC>
__global__ __launch_bounds__(256, 2) void thin_head_kernel(const IgemmParams p, int tiles_h, int tiles_w) {
    constexpr int S = 2, PT = (KS - S) / 2;
    constexpr int DMIN = -((KS - 1 - PT) / S), DMAX = (S - 1 + PT) / S;
    constexpr int TH = 8, TW = 16;
    constexpr int HRr = TH + DMAX - DMIN, HCc = TW + DMAX - DMIN, HPIX = HRr * HCc;
    constexpr int MB = (HPIX + 31) / 32;
    constexpr int NCOL = KS * KS * CC, NT = (NCOL + 31) / 32;
    constexpr int TP = NT * 32 + 1;                        // T pitch (floats): odd, so that consecutive pixels fall on consecutive banks
    constexpr int XP = 144;                                // bytes per halo pixel record: 32 x bf16 hi | 32 x bf16 lo | 16 pad
    constexpr int NLOAD = (HPIX * 8 + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* xs = lds;
    float* T = reinterpret_cast<float*>(lds + MB * 32 * XP);
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    int b = blockIdx.x;
    const int tw = b % tiles_w; b /= tiles_w;
    const int th = b % tiles_h;
    const int n = b / tiles_h;
    const int u0 = th * TH, v0 = tw * TW;
    tbf16x8 bh[NT][2], bl[NT][2];
    float4 wv[NT][2][2];
    // ---- halo: every load in flight before the first LDS write; out-of-image pieces get an offset beyond num_records and the
    // hardware returns zeros (a select on a plain load becomes a branch around the load in hipcc's hands, with a wait at every join)
    {
        float4 v[NLOAD];
        const int64_t img_bytes = (int64_t)p.Ha * p.Wa * p.a_ld * 4;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A) + (int64_t)n * p.Ha * p.Wa * p.a_ld, 0, (int)img_bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int idx = tid + 256 * i;
            const int pix = idx >> 3, c4 = idx & 7;
            const int hr = pix / HCc, hc = pix - hr * HCc;
            const int ih = u0 + DMIN + hr, iw = v0 + DMIN + hc;
            const bool ok = pix < HPIX && (unsigned)ih < (unsigned)p.Ha && (unsigned)iw < (unsigned)p.Wa;
            const int off = ok ? ((ih * p.Wa + iw) * p.a_ld + c4 * 4) * 4 : 0x7ffffff0;
            v[i] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0));
        }
        // ---- filter fragments: column nt * 32 + li = (P * KS + Q) * CC + c, channels 16 s + 8 lh + 0..7 (contiguous: w_ks = 1); the
        // loads are unconditional (clamped column) and go out BEHIND the halo's and before its conversion: one round trip for both
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int col = min(nt * 32 + li, NCOL - 1);
            const int tap = col / CC, c = col - tap * CC;
            const float* wsrc = p.Wt + (int64_t)tap * p.w_tap_stride + (int64_t)c * p.w_ns + lh * 8;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                wv[nt][s2][0] = *reinterpret_cast<const float4*>(wsrc + s2 * 16);
                wv[nt][s2][1] = *reinterpret_cast<const float4*>(wsrc + s2 * 16 + 4);
            }
        }
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) {
            const int idx = tid + 256 * i;
            if (idx < HPIX * 8) {
                const float4 t = v[i];
                const float h0 = (float)(__bf16)t.x, h1 = (float)(__bf16)t.y, h2 = (float)(__bf16)t.z, h3 = (float)(__bf16)t.w;
                uint2 hi, lo;
                hi.x = tpack2(t.x, t.y); hi.y = tpack2(t.z, t.w);
                lo.x = tpack2(t.x - h0, t.y - h1); lo.y = tpack2(t.z - h2, t.w - h3);
                unsigned char* d = xs + (idx >> 3) * XP + (idx & 7) * 8;
                *reinterpret_cast<uint2*>(d) = hi;
                *reinterpret_cast<uint2*>(d + 64) = lo;
            }
        }
    }
    {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const bool okc = nt * 32 + li < NCOL;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const float w[8] = {wv[nt][s2][0].x, wv[nt][s2][0].y, wv[nt][s2][0].z, wv[nt][s2][0].w,
                                    wv[nt][s2][1].x, wv[nt][s2][1].y, wv[nt][s2][1].z, wv[nt][s2][1].w};
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float wj = okc ? w[j] : 0.f;
                    const __bf16 h = (__bf16)wj;
                    bh[nt][s2][j] = h;
                    bl[nt][s2][j] = (__bf16)(wj - (float)h);
                }
            }
        }
    }
    __syncthreads();
    // ---- T = X W^T for this wave's pixel blocks
    for (int m = wave; m < MB; m += 4) {
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        const unsigned char* ap = xs + (m * 32 + li) * XP + lh * 16;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const tbf16x8 ah = __builtin_bit_cast(tbf16x8, *reinterpret_cast<const uint4*>(ap + s2 * 32));
            const tbf16x8 al = __builtin_bit_cast(tbf16x8, *reinterpret_cast<const uint4*>(ap + 64 + s2 * 32));
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[nt][s2], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[nt][s2], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[nt][s2], acc[nt], 0, 0, 0);
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) T[(m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh) * TP + nt * 32 + li] = acc[nt][r];
    }
    __syncthreads();
    // ---- gather: thread = (row parity a, tile position (tu, tv)): output row 2 u + a, columns 2 v and 2 v + 1, all channels
    const int tu = (tid >> 4) & 7, tv = tid & 15;
    const int up = u0 + tu, vp = v0 + tv;
    if (up >= p.Ha || vp >= p.Wa) return;
    float bias[CC];
#pragma unroll
    for (int c = 0; c < CC; ++c) bias[c] = p.bias ? p.bias[c] : 0.f;
    const bool is_tanh = p.act == MV3D_ACT_TANH, is_relu = p.act == MV3D_ACT_RELU;
    const float c1 = p.act == MV3D_ACT_NONE ? 1.f : (p.act == MV3D_ACT_LRELU ? 0.5f * (1.0f + p.leak) : 0.5f);
    const float c2 = p.act == MV3D_ACT_NONE ? 0.f : (p.act == MV3D_ACT_LRELU ? 0.5f * (1.0f - p.leak) : 0.5f);
    auto gather = [&](auto pa) {
        constexpr int a = decltype(pa)::value;
        float o[2][CC];
#pragma unroll
        for (int bq = 0; bq < 2; ++bq)
#pragma unroll
            for (int c = 0; c < CC; ++c) o[bq][c] = 0.f;
#pragma unroll
        for (int dh = DMIN; dh <= DMAX; ++dh) {
            constexpr int dummy = 0; (void)dummy;
            const int P = a + PT - S * dh;
            if (P < 0 || P >= KS) continue;
#pragma unroll
            for (int dw = DMIN; dw <= DMAX; ++dw) {
                const float* trow = T + ((tu + dh - DMIN) * HCc + (tv + dw - DMIN)) * TP;
#pragma unroll
                for (int bq = 0; bq < 2; ++bq) {
                    const int Q = bq + PT - S * dw;
                    if (Q < 0 || Q >= KS) continue;
#pragma unroll
                    for (int c = 0; c < CC; ++c) o[bq][c] += trow[(P * KS + Q) * CC + c];
                }
            }
        }
        const int64_t pix0 = (int64_t)(n * p.Hc + up * S + a) * p.Wc + vp * S;
#pragma unroll
        for (int bq = 0; bq < 2; ++bq)
#pragma unroll
            for (int c = 0; c < CC; ++c) {
                const float xv = o[bq][c] + bias[c];
                float y;
                if (is_tanh) y = tanhf(xv);
                else {
                    y = __fadd_rn(__fmul_rn(c1, xv), __fmul_rn(c2, fabsf(xv)));
                    y = (is_relu && xv < 0.0f) ? -0.0f : y;
                }
                o[bq][c] = y;
            }
        float* dst = p.Out + pix0 * p.c_ld;
        bool stored = false;
        if constexpr (CC == 2) {
            if (p.c_ld == 2) { *reinterpret_cast<float4*>(dst) = make_float4(o[0][0], o[0][1], o[1][0], o[1][1]); stored = true; }
        }
        if (!stored) {
#pragma unroll
            for (int bq = 0; bq < 2; ++bq)
#pragma unroll
                for (int c = 0; c < CC; ++c) dst[bq * p.c_ld + c] = o[bq][c];
        }
    };
    if (tid < 128) gather(std::integral_constant<int, 0>{}); else gather(std::integral_constant<int, 1>{});
}

// returns MV3D_OK after dispatching, 1 when the problem is not one of this kernel's (the caller falls back to thin_deconv_s2_tile)
int try_thin_head(const IgemmParams& p, void* stream, const char* who, double flops, double bytes) {
    if (disabled_paths() & (4096 | 536870912)) return 1;
    if (p.gact != MV3D_ACT_NONE) return 1;                      // forward of a head: no gradient mask
    if ((int64_t)p.Ha * p.Wa * p.a_ld * 4 >= 0x7fffffff) return 1;
    const int ntaps_all = p.tap_begin[p.so_h * p.so_w];
    if (p.so_h != 2 || p.so_w != 2 || (ntaps_all != 25 && ntaps_all != 9) || p.Ka != 32 || p.Cc < 1 || p.Cc > 4) return 1;
    if (p.Hc != 2 * p.Ha || p.Wc != 2 * p.Wa || p.a_ld % 4 != 0 || (reinterpret_cast<uintptr_t>(p.A) & 15)) return 1;
    // filter rows contiguous along the 32 feature channels and 16-byte aligned (the reference's [kh, kw, out, in] deconv filters)
    if (p.w_ks != 1 || p.w_ns % 4 != 0 || p.w_tap_stride % 4 != 0 || (reinterpret_cast<uintptr_t>(p.Wt) & 15)) return 1;
    const int KS = ntaps_all == 25 ? 5 : 3;
    const int hpix = KS == 5 ? 10 * 18 : 9 * 17, mb = (hpix + 31) / 32, nt = (KS * KS * p.Cc + 31) / 32;
    const size_t lds = (size_t)mb * 32 * 144 + (size_t)mb * 32 * (nt * 32 + 1) * 4;
    if (lds > 160 * 1024) return 1;
    const int tiles_h = cdiv(p.Ha, 8), tiles_w = cdiv(p.Wa, 16);
    const int blocks = p.N * tiles_h * tiles_w;
    using KernelT = void (*)(const IgemmParams, int, int);
    KernelT kern = nullptr;
    if (KS == 5) { switch (p.Cc) { case 1: kern = &thin_head_kernel<5, 1>; break; case 2: kern = &thin_head_kernel<5, 2>; break; case 3: kern = &thin_head_kernel<5, 3>; break; default: kern = &thin_head_kernel<5, 4>; break; } }
    else { switch (p.Cc) { case 1: kern = &thin_head_kernel<3, 1>; break; case 2: kern = &thin_head_kernel<3, 2>; break; case 3: kern = &thin_head_kernel<3, 3>; break; default: kern = &thin_head_kernel<3, 4>; break; } }
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();            // recording without a device: the attribute call fails and nothing is launched
    IgemmParams q = p;
    q.ksplit = 1;
    static const char* names[4] = {"thin_head<1>", "thin_head<2>", "thin_head<3>", "thin_head<4>"};
    return dispatch(stream, OpInfo{names[p.Cc - 1], flops, bytes}, [=](hipStream_t s) {
        kern<<<blocks, 256, lds, s>>>(q, tiles_h, tiles_w);
        return launched(who);
    });
}

// returns MV3D_OK after dispatching, 1 when the problem is not one of this kernel's (the caller falls back to smallc_b3*)
int try_smallc_band(const mv3d_conv_geom* g, const IgemmParams& ep, int pt, int pl, const void* img, const void* w, void* feat,
                    void* stream, const char* who, double flops, double bytes) {
    if (disabled_paths() & (4096 | 33554432)) return 1;
    if (g->C > 4 || g->img_ld != g->C || g->kw * g->C > 16 || (g->kh != 5 && g->kh != 3) || g->sh != 2 || g->sw != 2) return 1;
    if (g->Wo % 32 != 0 || g->K % 32 != 0 || (g->W * g->C) % 4 != 0 || (reinterpret_cast<uintptr_t>(img) & 15)) return 1;
    if ((int64_t)g->N * g->Ho * g->Wo * g->feat_ld * 4 >= 0x7fffffff) return 1;
    if (ep.gact != MV3D_ACT_NONE && !ep.gref) return 1;
    if (ep.act == MV3D_ACT_TANH || ep.gact == MV3D_ACT_TANH) return 1;
    BandParams p = {};
    p.X = (const float*)img; p.Wt = (const float*)w; p.Y = (float*)feat;
    p.N = g->N; p.H = g->H; p.W = g->W; p.C = g->C; p.Ho = g->Ho; p.Wo = g->Wo; p.K = g->K; p.y_ld = g->feat_ld;
    p.kh = g->kh; p.kw = g->kw; p.sh = g->sh; p.sw = g->sw; p.pt = pt; p.pl = pl;
    p.xoff = pl * g->C;
    // a fragment reads 16 elements from (2 ow - pl) * C + xoff: the last one ends at most 16 elements past the row
    p.rp = (p.xoff + g->W * g->C + 16 + 1) & ~1;
    p.wtiles = g->Wo / 32;
    p.inv_row_f4 = (unsigned)((((uint64_t)1 << 32) + (g->W * g->C / 4) - 1) / (uint64_t)(g->W * g->C / 4));
    // band height: at least 512 workgroups when the layer has them, planes of at most 60 KiB (two workgroups per CU)
    int rb = 16;
    while (rb > 1 && (g->Ho % rb != 0 || (int64_t)g->N * (g->Ho / rb) * (g->K / 32) < 512 || (size_t)((rb - 1) * g->sh + g->kh) * p.rp * 4 > 60 * 1024)) rb >>= 1;
    if (g->Ho % rb != 0) return 1;
    p.RB = rb;
    p.nrows = (rb - 1) * g->sh + g->kh;
    p.bands_per_img = g->Ho / rb;
    const size_t lds = (size_t)p.nrows * p.rp * 4;
    if (lds > 120 * 1024 || p.rp - g->W * g->C > 32 || p.nrows * (g->W * g->C / 4) >= 65536) return 1;
    { static int d = -1; if (d < 0) { const char* e = getenv("MV3D_DBG"); d = e ? atoi(e) : 0; } p.dbg = d; }
    p.bias = ep.bias; p.act = ep.act; p.leak = ep.leak; p.gact = ep.gact; p.gleak = ep.gleak; p.gref = ep.gref; p.g_ld = ep.g_ld;
    const dim3 grid(g->N * p.bands_per_img, g->K / 32, 1);
    const bool k5 = g->kh == 5, hg = ep.gact != MV3D_ACT_NONE;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&smallc_band_kernel<5, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&smallc_band_kernel<5, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&smallc_band_kernel<3, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&smallc_band_kernel<3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        attr_set = true;
    }
    return dispatch(stream, OpInfo{hg ? "smallc_band<gmask>" : "smallc_band", flops, bytes}, [=](hipStream_t s) {
        if (k5) { if (hg) smallc_band_kernel<5, true><<<grid, 256, lds, s>>>(p); else smallc_band_kernel<5, false><<<grid, 256, lds, s>>>(p); }
        else { if (hg) smallc_band_kernel<3, true><<<grid, 256, lds, s>>>(p); else smallc_band_kernel<3, false><<<grid, 256, lds, s>>>(p); }
        return launched(who);
    });
}

}  // namespace mv3d

extern "C" int mv3d_debug_band_stamps(void* dst, size_t bytes) {
    if (!dst || bytes > sizeof(unsigned long long) * 512 * 4 * 16) return mv3d::fail(MV3D_E_INVAL, "mv3d_debug_band_stamps: bad buffer");
    hipError_t e = hipMemcpyFromSymbol(dst, HIP_SYMBOL(mv3d::t_stamps), bytes, 0, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return mv3d::fail(MV3D_E_HIP, "mv3d_debug_band_stamps: %s", hipGetErrorString(e));
    return MV3D_OK;
}

namespace mv3d {

// ---------------------------------------------------------------------------------------------------------------------------------
// Filter gradient of the same layers (Conv2DBackpropFilter of e0; of the flow / depth heads with the roles of x and dy swapped):
//   dW[r][e][k] = sum_{n, oh, ow} X[n][2 oh + r - pt][(2 ow - pl) C + e] * dY[n][oh][ow][k],   e = dw * C + c < kw * C <= 16
// 1 GFLOP over 46 MB.  thin_filtgrad_kernel (conv.hip) does it with rank-1 updates on the vector ALUs (43 us: VALU- and
// latency-bound); here it is three products per fp32 product on the matrix cores with the PIXELS of an output row as the
// reduction index:
//   * a workgroup owns a band of eight output rows of one image; its 19 input rows are fetched once and scattered into
//     S[input row][e][ow] = X[row][(2 ow - pl) C + e] (bf16 hi / lo planes): the sequence a filter element sees along an output
//     row, contiguous in ow -- so an A fragment (8 consecutive pixels of one (r, e)) is ONE 16-byte LDS read.  An M tile is two
//     filter rows x 16 elements; the e-pitch of 72 elements spreads the 16 rows of a lane group over all banks;
//   * dY goes straight from global memory into B fragments: lane (filter k, half h) loads the 8 pixels of its k -- each value
//     is needed by exactly one wave, so LDS would only add a pass.  The bias gradient is the sum of what a lane loaded;
//   * a wave takes every fourth output row: 4 k-steps x 3 M tiles x 3 products per row, accumulators in registers; the four
//     waves' partial filters meet in LDS (fixed order) and leave as this workgroup's slab of the partial-filter buffer.
struct ThinWgParams {
    const float* img; const float* feat; float* out; float* bias_out;
    int N, H, W, C, Ho, Wo, K, feat_ld;
    int kh, kw, pt, pl;
    int RB, nrows, ep;            // band height, input rows per band, e-pitch of S in elements (Wo + 8)
    int rp;                       // pitch of the linear rows in elements: pl * C + W * C + 16, even
    unsigned inv_row_f4;
    int bands_per_img, dbg;
    unsigned inv_per_row, inv_hw;   // ceil(2^32 / (kw C Wo / 2)), ceil(2^32 / (Wo / 2))
};

template <int KH>
__global__ __launch_bounds__(256) void thin_wgrad_kernel(const ThinWgParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int MTL = (KH + 1) / 2;                                      // M tiles: pairs of filter rows
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int Cf = p.kw * p.C;
    const int k0 = blockIdx.y * 32;
    const int band = blockIdx.x % p.bands_per_img, n = blockIdx.x / p.bands_per_img;
    const int oh0 = band * p.RB;
    const int plane_el = (p.nrows + 1) * 16 * p.ep;                        // elements per plane (+ one zero row: the odd filter row of the last M tile)
    unsigned short* const s_hi = reinterpret_cast<unsigned short*>(lds);
    unsigned short* const s_lo = s_hi + plane_el;

    const bool st = (p.dbg & 32) != 0;
    int sk = 0;
    tstamp(st, wave, lane, sk);                                            // 0 start
    constexpr int PF = 4;                                                  // k-steps of dY prefetched (64 pixels)
    const bool k_ok = k0 + li < p.K;
    float pre[PF][8];
    tstamp(st, wave, lane, sk);                                            // 1 dY requested
    // ---- build the planes in two steps.  (1) the band's input rows, coalesced 16-byte loads -> linear bf16 hi / lo rows in LDS
    // (element 0 = column -pl, zero pads left and right, zero rows outside the image); (2) LDS -> LDS: every entry
    // S[ir][e][ow] = row[2 C ow + e], two 2-byte reads and one packed 4-byte store per pair of consecutive ow and plane.
    // (Building S straight from global memory was tried both ways: scattering coalesced row loads costs three positions of
    // index arithmetic and 2-byte stores per element, gathering with 4-byte loads 24 bytes apart keeps the texture path busy
    // for 15 k cycles per band -- in-kernel stamps.)
    unsigned short* const l_hi = s_lo + plane_el;
    unsigned short* const l_lo = l_hi + p.nrows * p.rp;
    {
        const int row_f4 = (p.W * p.C) >> 2;
        const int total = p.nrows * row_f4;
        const int ih0 = oh0 * 2 - p.pt;
        const int xoff = p.pl * p.C;
        constexpr int UB = 5;
        // Branch-free loads through buffer descriptors (rows outside the image and indices past the band get an offset beyond
        // num_records: zeros): with `ok ? load : 0` hipcc branches around every load and then waits with vmcnt(0) -- for the dY
        // prefetch below as well, i.e. for the whole 33 MB dY stream of the layer before the first row is converted.
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.img) + (int64_t)n * p.H * p.W * p.C, 0, p.H * p.W * p.C * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.feat) + (int64_t)n * p.Ho * p.Wo * p.feat_ld, 0, p.Ho * p.Wo * p.feat_ld * 4, 0x00020000);
        auto convert = [&](const float4& t, int lof) {
            const float xs[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const __bf16 h = (__bf16)xs[j];
                const __bf16 l = (__bf16)(xs[j] - (float)h);
                l_hi[lof + j] = __builtin_bit_cast(unsigned short, h);
                l_lo[lof + j] = __builtin_bit_cast(unsigned short, l);
            }
        };
        {
            float4 v[UB];
            int lofs[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int idx = u * 256 + tid;
                const int r = (int)__umulhi((unsigned)idx, p.inv_row_f4), f = idx - r * row_f4;
                const int ih = ih0 + r;
                const bool ok = idx < total && (unsigned)ih < (unsigned)p.H;
                v[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? (ih * p.W * p.C + 4 * f) * 4 : 0x7ffffff0, 0, 0));
                lofs[u] = idx < total ? r * p.rp + xoff + 4 * f : -1;
            }
            // this wave's first output row of dY: requested BEHIND the first round of row loads (vmcnt retires in order: in
            // front of them, the rows waited for the whole dY stream of the layer -- 12 k cycles per band in the stamps)
            // and consumed after the planes are built; the remaining rows of a wave are loaded row by row
            {
                const int frow = (min(oh0 + wave, p.Ho - 1) * p.Wo * p.feat_ld + k0 + li) * 4;
#pragma unroll
                for (int u = 0; u < PF; ++u)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        pre[u][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(fr, (k_ok && u * 16 < p.Wo) ? frow + (u * 16 + 8 * lh + j) * p.feat_ld * 4 : 0x7ffffff0, 0, 0));
            }
#pragma unroll
            for (int u = 0; u < UB; ++u)
                if (lofs[u] >= 0) convert(v[u], lofs[u]);
        }
        for (int base = 256 * UB; base < total; base += 256 * UB) {       // wider images: the rest of the band's rows
            float4 v[UB];
            int lofs[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int idx = base + u * 256 + tid;
                const int r = (int)__umulhi((unsigned)idx, p.inv_row_f4), f = idx - r * row_f4;
                const int ih = ih0 + r;
                const bool ok = idx < total && (unsigned)ih < (unsigned)p.H;
                v[u] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? (ih * p.W * p.C + 4 * f) * 4 : 0x7ffffff0, 0, 0));
                lofs[u] = idx < total ? r * p.rp + xoff + 4 * f : -1;
            }
#pragma unroll
            for (int u = 0; u < UB; ++u)
                if (lofs[u] >= 0) convert(v[u], lofs[u]);
        }
        const int per_row = p.rp - p.W * p.C;                              // left + right pad elements (host: <= 32)
        const int j = tid & 31;
        if (j < per_row) {
            const int e = j < xoff ? j : p.W * p.C + j;
            for (int r = tid >> 5; r < p.nrows; r += 8) { l_hi[r * p.rp + e] = 0; l_lo[r * p.rp + e] = 0; }
        }
    }
    __syncthreads();
    {
        const int hw = p.Wo >> 1;                                          // ow pairs per (row, element)
        const int per_row = Cf * hw;
        const int total = p.nrows * per_row;
        const int c2 = 2 * p.C;
        for (int idx = tid; idx < total; idx += 256) {
            const int ir = (int)__umulhi((unsigned)idx, p.inv_per_row), rem = idx - ir * per_row;
            const int e = (int)__umulhi((unsigned)rem, p.inv_hw), owp = rem - e * hw;
            const int t0 = ir * p.rp + 2 * c2 * owp + e;
            const unsigned h = (unsigned)l_hi[t0] | ((unsigned)l_hi[t0 + c2] << 16);
            const unsigned l = (unsigned)l_lo[t0] | ((unsigned)l_lo[t0 + c2] << 16);
            const int o = (ir * 16 + e) * p.ep + 2 * owp;
            *reinterpret_cast<unsigned*>(s_hi + o) = h;
            *reinterpret_cast<unsigned*>(s_lo + o) = l;
        }
    }
    tstamp(st, wave, lane, sk);                                            // 2 planes written
    __syncthreads();
    tstamp(st, wave, lane, sk);                                            // 3 barrier

    f32x16 acc[MTL];
#pragma unroll
    for (int t = 0; t < MTL; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[t][q] = 0.f;
    float bsum = 0.f;
    const int ksteps = p.Wo >> 4;
    const int a_lane = (((li >> 4) * 16 + (li & 15)) * p.ep + 8 * lh) * 2;     // bytes: filter row parity, element, pixel half
    for (int orow = wave, it = 0; orow < p.RB; orow += 4, ++it) {
        if (oh0 + orow >= p.Ho) break;
        for (int ks = 0; ks < ksteps; ++ks) {
            float b[8];
            if (it == 0 && ks < PF) {
#pragma unroll
                for (int j = 0; j < 8; ++j) b[j] = 0.f;
#pragma unroll
                for (int u = 0; u < PF; ++u)
                    if (u == ks) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) b[j] = pre[u][j];
                    }
            } else {
                const float* frow = p.feat + (int64_t)((n * p.Ho + oh0 + orow) * p.Wo) * p.feat_ld + k0 + li;
#pragma unroll
                for (int j = 0; j < 8; ++j) b[j] = k_ok ? frow[(int64_t)(ks * 16 + 8 * lh + j) * p.feat_ld] : 0.f;
            }
            tbf16x8 bh, bl;
#pragma unroll
            for (int j = 0; j < 8; ++j) { bh[j] = (__bf16)b[j]; bl[j] = (__bf16)(b[j] - (float)bh[j]); bsum += b[j]; }
#pragma unroll
            for (int t = 0; t < MTL; ++t) {
                // M tile t: filter rows 2 t (lanes 0-15) and 2 t + 1 (lanes 16-31) -> input rows 2 orow + 2 t (+ 1)
                const unsigned char* ap = lds + ((orow * 2 + 2 * t) * 16 * p.ep + ks * 16) * 2 + a_lane;
                const tbf16x8 ah = __builtin_bit_cast(tbf16x8, *reinterpret_cast<const uint4*>(ap));
                const tbf16x8 al = __builtin_bit_cast(tbf16x8, *reinterpret_cast<const uint4*>(ap + plane_el * 2));
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[t], 0, 0, 0);
            }
        }
    }
    tstamp(st, wave, lane, sk);                                            // 4 products done
    // ---- the four waves' partial filters: through LDS, added in the order wave 0, 1, 2, 3
    __syncthreads();
    tstamp(st, wave, lane, sk);                                            // 5 barrier
    float* const xch = reinterpret_cast<float*>(lds);                      // [wave][MTL][16][64] + bias [wave][64]
#pragma unroll
    for (int t = 0; t < MTL; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) xch[((wave * MTL + t) * 16 + q) * 64 + lane] = acc[t][q];
    float* const bx = xch + 4 * MTL * 16 * 64;
    bx[wave * 64 + lane] = bsum;
    __syncthreads();
    const int64_t fcount = (int64_t)p.kh * Cf * p.K;
    float* const out = p.out + (int64_t)blockIdx.x * fcount;
    // wave w finishes registers 4 w .. 4 w + 3 of every M tile
#pragma unroll
    for (int t = 0; t < MTL; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int q = wave * 4 + j;
            const float s = ((xch[((0 * MTL + t) * 16 + q) * 64 + lane] + xch[((1 * MTL + t) * 16 + q) * 64 + lane]) +
                             xch[((2 * MTL + t) * 16 + q) * 64 + lane]) + xch[((3 * MTL + t) * 16 + q) * 64 + lane];
            const int m = (q & 3) + 8 * (q >> 2) + 4 * lh;                  // row of the tile: filter row parity * 16 + element
            const int r = 2 * t + (m >> 4), e = m & 15;
            if (r < p.kh && e < Cf && k0 + li < p.K) out[(int64_t)(r * Cf + e) * p.K + k0 + li] = s;
        }
    if (p.bias_out && wave == 0 && lh == 0 && k0 + li < p.K) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) s += bx[w * 64 + li] + bx[w * 64 + 32 + li];
        p.bias_out[(int64_t)blockIdx.x * p.K + k0 + li] = s;
    }
    tstamp(st, wave, lane, sk);                                            // 6 partial filter stored
}

namespace {
// band height of the thin filter gradient: four output rows (eight were measured 2 us slower on e0: one workgroup per CU)
int thin_wg_rb(const mv3d_conv_geom* g) { return g->Ho % 4 == 0 ? 4 : 0; }
unsigned tinv32(int d) { return (unsigned)((((uint64_t)1 << 32) + d - 1) / (uint64_t)d); }
}

// planning: number of slabs (= workgroups along the pixels) of the matrix-core thin filter gradient, 0 when not applicable
int thin_wgrad_slabs(const mv3d_conv_geom* g) {
    if (disabled_paths() & (4096 | 16384 | 67108864)) return 0;
    if (g->C > 4 || g->img_ld != g->C || g->kw * g->C > 16 || (g->kh != 5 && g->kh != 3) || g->sh != 2 || g->sw != 2) return 0;
    if (g->Wo % 16 != 0 || g->K % 32 != 0 || (g->W * g->C) % 4 != 0 || g->W != 2 * g->Wo || g->H != 2 * g->Ho) return 0;
    const int rb = thin_wg_rb(g);
    if (!rb) return 0;
    const int nrows = (rb - 1) * 2 + g->kh;
    const size_t lds = std::max((size_t)(nrows + 1) * 16 * (g->Wo + 8) * 4 + (size_t)nrows * (g->W * g->C + 4 * g->C + 18) * 4, (size_t)(4 * 3 * 16 * 64 + 256) * 4);
    if (lds > 150 * 1024 || nrows * g->kw * g->C * (g->Wo / 2) >= 65536 || nrows * (g->W * g->C / 4) >= 65536) return 0;
    return g->N * (g->Ho / rb);
}

int thin_wgrad_launch(const mv3d_conv_geom* g, const void* img, const void* feat, void* part, void* bias_part, void* stream,
                      const char* who, double flops, double bytes) {
    if (!thin_wgrad_slabs(g)) return 1;
    if ((reinterpret_cast<uintptr_t>(img) & 15)) return 1;
    ThinWgParams p = {};
    int ho, wo;
    same_pad(g->H, g->kh, 2, &ho, &p.pt);
    same_pad(g->W, g->kw, 2, &wo, &p.pl);
    p.img = (const float*)img; p.feat = (const float*)feat; p.out = (float*)part; p.bias_out = (float*)bias_part;
    p.N = g->N; p.H = g->H; p.W = g->W; p.C = g->C; p.Ho = g->Ho; p.Wo = g->Wo; p.K = g->K; p.feat_ld = g->feat_ld;
    p.kh = g->kh; p.kw = g->kw;
    p.RB = thin_wg_rb(g);
    p.nrows = (p.RB - 1) * 2 + g->kh;
    p.ep = g->Wo + 8;
    p.bands_per_img = g->Ho / p.RB;
    { static int d = -1; if (d < 0) { const char* e = getenv("MV3D_DBG"); d = e ? atoi(e) : 0; } p.dbg = d; }
    p.rp = (p.pl * g->C + g->W * g->C + 16 + 1) & ~1;
    p.inv_row_f4 = tinv32(g->W * g->C / 4);
    p.inv_per_row = tinv32(g->kw * g->C * (g->Wo / 2));
    p.inv_hw = tinv32(g->Wo / 2);
    const int mtl = (g->kh + 1) / 2;
    const size_t lds = std::max((size_t)(p.nrows + 1) * 16 * p.ep * 4 + (size_t)p.nrows * p.rp * 4, (size_t)(4 * mtl * 16 * 64 + 256) * 4);
    const dim3 grid(g->N * p.bands_per_img, g->K / 32, 1);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_wgrad_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_wgrad_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const bool k5 = g->kh == 5;
    return dispatch(stream, OpInfo{"thin_wgrad", flops, bytes}, [=](hipStream_t s) {
        if (k5) thin_wgrad_kernel<5><<<grid, 256, lds, s>>>(p);
        else thin_wgrad_kernel<3><<<grid, 256, lds, s>>>(p);
        return launched(who);
    });
}

}  // namespace mv3d
