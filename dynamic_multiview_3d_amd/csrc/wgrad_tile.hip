// Filter gradient (conv / deconv wgrad) for the layers that carry the FLOPs -- tiled, persistent.
//
//   df[tap][c][k] = sum_pixels img[pixel + tap][c] * feat[pixel][k]
//
// A workgroup owns one (32-channel c-tile) x (32*NKT k-tile) block of the filter for ALL taps and
// walks a slab of spatial tiles (128 output pixels each).  Per tile the image halo
// ((TH-1)*s + kh) x ((TW-1)*s + kw) x 32c and the feature tile 128 x 32*NKT are staged in LDS once
// and reused by every tap; the 4 waves split the (tap, k-tile) items and keep their partial filters
// in accumulator registers across the whole slab (up to 13 x 32x32 fp32 tiles = 208 registers per
// lane).  The next tile's global loads are issued before the current tile's MFMAs and written to
// LDS afterwards, so HBM/L2 latency hides under ~30k MFMA cycles per tile.
// MFMA operands: A[i = channel][k-slot = pixel parity], B[k-slot][j = feature channel]; both are
// consecutive-lane ds_read_b32 of 128-byte LDS rows (conflict-free, no padding).
// One partial filter per workgroup is written at the end; reduce_slabs_kernel sums them in a fixed
// order (no float atomics).
#include "conv_common.h"
#include <algorithm>

namespace mv3d {

struct WgTileParams {
    const float* img; const float* feat;
    float* out; float* bias_out;
    int N, H, W, C, img_ld;
    int Ho, Wo, K, feat_ld;
    int sh, sw, pt, pl, kw;
    int TH, TW, tw_shift, tiles_h, tiles_w, HR, HC;
    int TPIX;            // output pixels per spatial tile: 128, or 64 when the (stride-2) halo would not fit twice in LDS
    int ntaps, nitems, ipw;
    int IW, PH;          // 8 waves = IW item groups x PH pixel-pair ranges
    int ctiles;
    int ntiles_total, tiles_per_slab;
    int b3;              // 1: split-bf16 kernel (wgrad_b3_kernel)
    int inv_hc, tpix8_shift;   // ceil(2^20 / HC); log2(TPIX * 8)
    int G, img_shift, HRi, inv_hri;   // small images (4x4 feature maps): a tile is G whole images of 2^img_shift pixels, halos stacked
    int tsplit;          // split-bf16 kernel: the taps are divided over this many workgroups (grid.z); 1 = all taps in one workgroup
};

__device__ float4 g_zero16[4];        // 64 bytes of zeros in the code object: source of out-of-image LDS-DMA lanes

// 16 bytes per lane global -> LDS without a VGPR round trip.  The LDS destination of a wave
// instruction is (wave-uniform base) + lane*16, i.e. 1 KiB contiguous = 8 rows of 128 bytes.
__device__ __forceinline__ void dma16(const float* gsrc, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int TPW, int NKT>
__global__ __launch_bounds__(512) void wgrad_tile_kernel(const WgTileParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // 0..7, two waves per SIMD
    const int iw = wave % p.IW, ph = wave / p.IW;                   // item group, pixel-pair range
    const int li = lane & 31, lh = lane >> 5;
    const int ct = blockIdx.x % p.ctiles, kg = blockIdx.x / p.ctiles;
    const int c0 = ct * 32, k0 = kg * 32 * NKT;
    const int slab = blockIdx.y;
    const int tile_begin = slab * p.tiles_per_slab;
    const int tile_end = min(tile_begin + p.tiles_per_slab, p.ntiles_total);

    const int halo_pix = p.HR * p.HC;
    const int n_img_rows = halo_pix;                    // 128-byte rows: halo pixels, then TPIX*NKT feature rows
    const int n_rows = n_img_rows + p.TPIX * NKT;
    const int n_chunks = (n_rows + 7) >> 3;             // 1 KiB DMA pieces
    const int buf_floats = n_chunks * 256;

    // (tap, k-tile) items of this wave; surplus slots recompute the last real item into an
    // accumulator that is never stored (no branches in the MFMA loop)
    int offA[TPW], offB[TPW], item_tap[TPW], item_kt[TPW];
    bool item_ok[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        int it = iw * p.ipw + i;
        item_ok[i] = (i < p.ipw) && (it < p.nitems);
        if (it >= p.nitems) it = p.nitems - 1;
        const int tap = it % p.ntaps, kt = it / p.ntaps;
        item_tap[i] = tap; item_kt[i] = kt;
        offA[i] = ((tap / p.kw) * p.HC + (tap % p.kw)) * 32 + lh * p.sw * 32 + li;
        offB[i] = (n_img_rows + kt * p.TPIX + lh) * 32 + li;
    }

    f32x16 acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float sb[NKT];
#pragma unroll
    for (int y = 0; y < NKT; ++y) sb[y] = 0.f;
    const bool do_bias = p.bias_out != nullptr && ct == 0;

    // asynchronous staging of one spatial tile into buffer `buf` (LDS-DMA, no registers)
    auto stage = [&](int tile, float* buf) {
        int b = tile;
        const int tw_i = b % p.tiles_w; b /= p.tiles_w;
        const int th_i = b % p.tiles_h;
        const int n = b / p.tiles_h;
        const int oh0 = th_i * p.TH, ow0 = tw_i * p.TW;
        const int ih0 = oh0 * p.sh - p.pt, iw0 = ow0 * p.sw - p.pl;
        const int c4 = lane & 7;
        for (int chunk = wave; chunk < n_chunks; chunk += 8) {
            const int row = chunk * 8 + (lane >> 3);
            const float* src = reinterpret_cast<const float*>(g_zero16);
            if (row < n_img_rows) {
                const int hr = row / p.HC, hc = row - hr * p.HC;
                const int ih = ih0 + hr, iwc = iw0 + hc, ch = c0 + c4 * 4;
                if ((unsigned)ih < (unsigned)p.H && (unsigned)iwc < (unsigned)p.W && ch < p.C)
                    src = p.img + (int64_t)((n * p.H + ih) * p.W + iwc) * p.img_ld + ch;
            } else if (row < n_rows) {
                const int j = row - n_img_rows;
                const int kt = j / p.TPIX, q = j - kt * p.TPIX;
                const int oh = oh0 + (q >> p.tw_shift), ow = ow0 + (q & (p.TW - 1)), kk = k0 + kt * 32 + c4 * 4;
                if (oh < p.Ho && ow < p.Wo && kk < p.K)
                    src = p.feat + (int64_t)((n * p.Ho + oh) * p.Wo + ow) * p.feat_ld + kk;
            }
            dma16(src, buf + chunk * 256);
        }
    };

    const int jshift = p.tw_shift - 1;                       // log2(TW/2)
    const int jmask = (p.TW >> 1) - 1;
    const int strideA = 2 * p.sw * 32, rowA = p.sh * p.HC * 32;
    const int nsteps = (p.TPIX >> 1) / p.PH;                 // pixel pairs of this wave (even)
    const int s_begin = ph * nsteps;

    if (tile_begin < tile_end) stage(tile_begin, smem);
    int cur = 0;
    for (int tile = tile_begin; tile < tile_end; ++tile, cur ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's DMA pieces have landed
        __syncthreads();                                        // everyone's have; previous tile fully consumed
        const float* buf = smem + cur * buf_floats;
        if (tile + 1 < tile_end) stage(tile + 1, smem + (cur ^ 1) * buf_floats);

        // operands of pixel pair s+1 are read while the MFMAs of pair s run; the last iteration's
        // look-ahead wraps to the first pair (a harmless extra read) so the loop needs no tail
        float a0[TPW], b0[TPW], a1[TPW], b1[TPW];
        auto ld = [&](float (&a)[TPW], float (&b)[TPW], int s) {
            const float* pa = buf + (s >> jshift) * rowA + (s & jmask) * strideA;
            const float* pb = buf + s * 64;
#pragma unroll
            for (int i = 0; i < TPW; ++i) { a[i] = pa[offA[i]]; b[i] = pb[offB[i]]; }
        };
        auto mma = [&](const float (&a)[TPW], const float (&b)[TPW]) {
#pragma unroll
            for (int i = 0; i < TPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], acc[i], 0, 0, 0);
        };
        ld(a0, b0, s_begin);
        for (int s = 0; s < nsteps; s += 2) {
            ld(a1, b1, s_begin + s + 1);
            mma(a0, b0);
            ld(a0, b0, s_begin + ((s + 2) & (nsteps - 1)));
            mma(a1, b1);
        }
        if (do_bias && wave == 0) {
            // bias gradient = column sums of the feature tile (lane = (pixel parity, channel))
#pragma unroll
            for (int y = 0; y < NKT; ++y) {
                const float* pb = buf + (n_img_rows + y * p.TPIX + lh) * 32 + li;
                float t = 0.f;
                for (int s = 0; s < (p.TPIX >> 1); ++s) t += pb[s * 64];
                sb[y] += t;
            }
        }
    }

    // combine the PH pixel-range partials of each item group through LDS (fixed order), then store
    __syncthreads();
    for (int r = 1; r < p.PH; ++r) {
        float* xch = smem + (iw * TPW) * 1024 + lane;            // [item][16 regs][64 lanes]
        if (ph == r) {
#pragma unroll
            for (int i = 0; i < TPW; ++i)
#pragma unroll
                for (int q = 0; q < 16; ++q) xch[(i * 16 + q) * 64] = acc[i][q];
        }
        __syncthreads();
        if (ph == 0) {
#pragma unroll
            for (int i = 0; i < TPW; ++i)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][q] += xch[(i * 16 + q) * 64];
        }
        __syncthreads();
    }
    if (ph == 0) {
        // partial filter of this workgroup: D layout col (lane&31) = k, row = c
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            if (!item_ok[i]) continue;
            float* out = p.out + ((int64_t)slab * p.ntaps + item_tap[i]) * p.C * p.K;
            const int k = k0 + item_kt[i] * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = c0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (c < p.C && k < p.K) out[(int64_t)c * p.K + k] = acc[i][r];
            }
        }
    }
    if (do_bias && wave == 0) {
#pragma unroll
        for (int y = 0; y < NKT; ++y) {
            const float s = sb[y] + __shfl_xor(sb[y], 32);
            const int k = k0 + y * 32 + li;
            if (lh == 0 && k < p.K) p.bias_out[(int64_t)slab * p.K + k] = s;
        }
    }
}


// ------------------------------------------------------------------------------------------------
// Split-bf16 variant (see bconv.hip for the arithmetic): same work decomposition, but the reduction
// (pixel) index is the MFMA k index of v_mfma_f32_32x32x16_bf16, 16 pixels per instruction.
//   * staging goes through registers (LDS-DMA cannot convert): every thread prefetches its share of the
//     next tile's fp32 halo + feature tile while the current tile is multiplied, then splits each value
//     into hi/lo bf16 and writes four PLANES per buffer: image hi, image lo ([halo pixel][32 ch], 64-byte
//     rows) and feature hi, lo ([pixel][32 k] per k-tile);
//   * both MFMA operands need "8 consecutive pixels of one channel" per lane while LDS holds
//     [pixel][channel]: ds_read_b64_tr_b16 (gfx950 transposing read) delivers exactly that -- 4 pixels x 16
//     channels per 16-lane group, two reads per operand; four consecutive pixels are 256 contiguous bytes,
//     i.e. conflict-free.  Stride-2 layers store even and odd halo columns apart so that this still holds;
//   * a wave's items all share one k-tile, so the feature operand is read once per 16-pixel step and each
//     tap adds 4 reads + 3 MFMAs.
typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wbf16x4 __attribute__((ext_vector_type(4)));
typedef short ws16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void wsplit4(const float4& v, uint2& hi, uint2& lo) {
    wbf16x4 h, l;
    h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
    l[0] = (__bf16)(v.x - (float)h[0]); l[1] = (__bf16)(v.y - (float)h[1]);
    l[2] = (__bf16)(v.z - (float)h[2]); l[3] = (__bf16)(v.w - (float)h[3]);
    hi = __builtin_bit_cast(uint2, h);
    lo = __builtin_bit_cast(uint2, l);
}

__device__ __forceinline__ uint2 tr_read(const unsigned char* lds_ptr) {
    ws16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ws16x4*)lds_ptr);
    return __builtin_bit_cast(uint2, v);
}

template <int TPW, int NKT, int NI>
__global__ __launch_bounds__(512) void wgrad_b3_kernel(const WgTileParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smb[];
    constexpr int NF = 4;                               // feature float4 slots per thread (TPIX * NKT * 8 / 512 <= 4)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int iw = wave % p.IW;                         // (k-tile, tap group)
    const int ph = wave / p.IW;                         // pixel-step range
    const int tg_per_kt = p.IW / NKT;
    const int kt_w = iw / tg_per_kt, tg = iw - kt_w * tg_per_kt;
    const int li = lane & 31, lh = lane >> 5;
    const int ct = blockIdx.x % p.ctiles, kg = blockIdx.x / p.ctiles;
    const int c0 = ct * 32, k0 = kg * 32 * NKT;
    const int slab = blockIdx.y;
    const int tile_begin = slab * p.tiles_per_slab;
    const int tile_end = min(tile_begin + p.tiles_per_slab, p.ntiles_total);

    const int halo_pix = p.HR * p.HC;
    const int img_plane = halo_pix * 64;                // bytes of one image plane
    const int feat_plane = p.TPIX * 64;
    const int buf_bytes = 2 * img_plane + 2 * NKT * feat_plane;
    const int HCh = (p.HC + 1) >> 1;                    // stride-2: even halo columns first, then the odd ones

    // this wave's taps (same k-tile): surplus slots recompute the last real tap into an accumulator never stored
    int tapoff[TPW], item_tap[TPW];
    bool item_ok[TPW];
    // taps: tg_per_kt groups per k-tile in a workgroup, times the tap split over workgroups (grid.z)
    const int tg_total = tg_per_kt * p.tsplit, tgi = (int)blockIdx.z * tg_per_kt + tg;
    const int tap_lo = (p.ntaps * tgi) / tg_total, tap_n = (p.ntaps * (tgi + 1)) / tg_total - tap_lo;
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        int tap = tap_lo + i;
        item_ok[i] = i < tap_n;
        if (tap >= p.ntaps) tap = p.ntaps - 1;
        item_tap[i] = tap;
        const int dh = tap / p.kw, dw = tap - dh * p.kw;
        const int cs = (p.sw == 2) ? (dw & 1) * HCh + (dw >> 1) : dw;
        tapoff[i] = (dh * p.HC + cs) * 64;
    }

    f32x16 acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    // lane roles of the transposing read: address supplier (row q, column quad pp of 16-channel group g)
    const int g16 = (lane >> 4) & 1, q = (lane >> 2) & 3, pp = lane & 3;
    const int col_bytes = (16 * g16 + 4 * pp) * 2;

    // ---- staging: global -> registers (prefetch) -> split -> LDS planes -------------------------
    float4 vi[NI], vf[NF];
    float4 bsum[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) bsum[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool do_bias = p.bias_out != nullptr && ct == 0 && blockIdx.z == 0;
    const int n_img4 = halo_pix * 8, n_feat4 = p.TPIX * NKT * 8;
    auto fetch = [&](int tile) {
        int b = tile;
        const int tw_i = b % p.tiles_w; b /= p.tiles_w;
        const int th_i = b % p.tiles_h;
        const int n = (b / p.tiles_h) * p.G;                 // first image of the tile (G whole images when G > 1)
        const int oh0 = th_i * p.TH, ow0 = tw_i * p.TW;
        const int ih0 = oh0 * p.sh - p.pt, iw0 = ow0 * p.sw - p.pl;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int idx = j * 512 + tid;
            const int pix = idx >> 3, c4 = idx & 7;
            const int hrt = (int)(((unsigned)pix * (unsigned)p.inv_hc) >> 20), hc = pix - hrt * p.HC;
            const int gi = p.G > 1 ? (int)(((unsigned)hrt * (unsigned)p.inv_hri) >> 20) : 0;
            const int hr = hrt - gi * p.HRi;
            const int ih = ih0 + hr, iwc = iw0 + hc, ch = c0 + c4 * 4;
            const bool ok = idx < n_img4 && n + gi < p.N && (unsigned)ih < (unsigned)p.H && (unsigned)iwc < (unsigned)p.W && ch < p.C;
            const float* src = ok ? p.img + (int64_t)(((n + gi) * p.H + ih) * p.W + iwc) * p.img_ld + ch : p.img;
            const float4 t4 = *reinterpret_cast<const float4*>(src);
            vi[j] = ok ? t4 : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const int idx = j * 512 + tid;
            const int kt = idx >> p.tpix8_shift, rem = idx & ((1 << p.tpix8_shift) - 1);
            const int pq = rem >> 3, c4 = rem & 7;
            const int gi = pq >> p.img_shift, pr = pq & ((1 << p.img_shift) - 1);
            const int oh = oh0 + (pr >> p.tw_shift), ow = ow0 + (pr & (p.TW - 1)), kk = k0 + kt * 32 + c4 * 4;
            const bool ok = idx < n_feat4 && n + gi < p.N && oh < p.Ho && ow < p.Wo && kk < p.K;
            const float* src = ok ? p.feat + (int64_t)(((n + gi) * p.Ho + oh) * p.Wo + ow) * p.feat_ld + kk : p.feat;
            const float4 t4 = *reinterpret_cast<const float4*>(src);
            vf[j] = ok ? t4 : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto commit = [&](unsigned char* buf) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int idx = j * 512 + tid;
            if (idx < n_img4) {
                const int pix = idx >> 3, c4 = idx & 7;
                const int hr = (int)(((unsigned)pix * (unsigned)p.inv_hc) >> 20), hc = pix - hr * p.HC;
                const int cs = (p.sw == 2) ? (hc & 1) * HCh + (hc >> 1) : hc;
                unsigned char* d = buf + (hr * p.HC + cs) * 64 + c4 * 8;
                uint2 hi, lo;
                wsplit4(vi[j], hi, lo);
                *reinterpret_cast<uint2*>(d) = hi;
                *reinterpret_cast<uint2*>(d + img_plane) = lo;
            }
        }
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const int idx = j * 512 + tid;
            if (idx < n_feat4) {
                const int kt = idx >> p.tpix8_shift, rem = idx & ((1 << p.tpix8_shift) - 1);
                unsigned char* d = buf + 2 * img_plane + kt * 2 * feat_plane + (rem >> 3) * 64 + (rem & 7) * 8;
                uint2 hi, lo;
                wsplit4(vf[j], hi, lo);
                *reinterpret_cast<uint2*>(d) = hi;
                *reinterpret_cast<uint2*>(d + feat_plane) = lo;
                bsum[j].x += vf[j].x; bsum[j].y += vf[j].y; bsum[j].z += vf[j].z; bsum[j].w += vf[j].w;
            }
        }
    };

    const int steps_total = p.TPIX >> 4;
    const int nsteps = steps_total / p.PH;
    const int s_begin = ph * nsteps;

    if (tile_begin < tile_end) { fetch(tile_begin); commit(smb); }
    int cur = 0;
    for (int tile = tile_begin; tile < tile_end; ++tile, cur ^= 1) {
        __syncthreads();                                        // buffer `cur` complete; the other one no longer read
        const unsigned char* buf = smb + cur * buf_bytes;
        const bool has_next = tile + 1 < tile_end;
        if (has_next) fetch(tile + 1);                          // in flight under this tile's MFMAs

        const unsigned char* fb = buf + 2 * img_plane + kt_w * 2 * feat_plane + col_bytes;
        for (int s = s_begin; s < s_begin + nsteps; ++s) {
            // pixels of this step handled by this lane as address supplier: 16 s + 8 lh + 4 r + q, r = 0, 1
            const int px0 = 16 * s + 8 * lh + q, px1 = px0 + 4;
            const int im = (1 << p.img_shift) - 1;
            const int r0 = px0 & im, r1 = px1 & im;
            const int a0 = (((px0 >> p.img_shift) * p.HRi + (r0 >> p.tw_shift) * p.sh) * p.HC + (r0 & (p.TW - 1))) * 64 + col_bytes;
            const int a1 = (((px1 >> p.img_shift) * p.HRi + (r1 >> p.tw_shift) * p.sh) * p.HC + (r1 & (p.TW - 1))) * 64 + col_bytes;
            const uint2 bh0 = tr_read(fb + px0 * 64), bh1 = tr_read(fb + px1 * 64);
            const uint2 bl0 = tr_read(fb + feat_plane + px0 * 64), bl1 = tr_read(fb + feat_plane + px1 * 64);
            const wbf16x8 bh = __builtin_bit_cast(wbf16x8, make_uint4(bh0.x, bh0.y, bh1.x, bh1.y));
            const wbf16x8 bl = __builtin_bit_cast(wbf16x8, make_uint4(bl0.x, bl0.y, bl1.x, bl1.y));
            uint2 ar[2][4];
            auto lda = [&](int i, uint2 (&a)[4]) {
                const unsigned char* pa = buf + tapoff[i];
                a[0] = tr_read(pa + a0); a[1] = tr_read(pa + a1);
                a[2] = tr_read(pa + img_plane + a0); a[3] = tr_read(pa + img_plane + a1);
            };
            lda(0, ar[0]);
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                if (i + 1 < TPW) lda(i + 1, ar[(i + 1) & 1]);
                const uint2 (&a)[4] = ar[i & 1];
                const wbf16x8 ah = __builtin_bit_cast(wbf16x8, make_uint4(a[0].x, a[0].y, a[1].x, a[1].y));
                const wbf16x8 al = __builtin_bit_cast(wbf16x8, make_uint4(a[2].x, a[2].y, a[3].x, a[3].y));
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i], 0, 0, 0);
            }
        }
        if (has_next) commit(smb + (cur ^ 1) * buf_bytes);
    }

    // combine the PH pixel-range partials of each item group through LDS (fixed order), then store
    __syncthreads();
    float* smem = reinterpret_cast<float*>(smb);
    for (int r = 1; r < p.PH; ++r) {
        float* xch = smem + (iw * TPW) * 1024 + lane;            // [item][16 regs][64 lanes]
        if (ph == r) {
#pragma unroll
            for (int i = 0; i < TPW; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) xch[(i * 16 + e) * 64] = acc[i][e];
        }
        __syncthreads();
        if (ph == 0) {
#pragma unroll
            for (int i = 0; i < TPW; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][e] += xch[(i * 16 + e) * 64];
        }
        __syncthreads();
    }
    if (ph == 0) {
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            if (!item_ok[i]) continue;
            float* out = p.out + ((int64_t)slab * p.ntaps + item_tap[i]) * p.C * p.K;
            const int k = k0 + kt_w * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = c0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (c < p.C && k < p.K) out[(int64_t)c * p.K + k] = acc[i][r];
            }
        }
    }
    if (do_bias) {
        // bias gradient = column sums of the feature tiles: per-thread fp32 partials (fixed k columns per
        // thread), combined through LDS in a fixed order
        __syncthreads();
        float4* bx = reinterpret_cast<float4*>(smb);
#pragma unroll
        for (int j = 0; j < NF; ++j) bx[j * 512 + tid] = bsum[j];
        __syncthreads();
        if (tid < 32 * NKT) {
            const int kt = tid >> 5, kk = tid & 31;
            // staging slot idx = j*512 + thread = kt*TPIX*8 + pixel*8 + (k>>2) holds that thread's sum over all tiles
            const float* bf = reinterpret_cast<const float*>(smb);
            float t = 0.f;
            for (int pq = 0; pq < p.TPIX; ++pq) t += bf[((kt * p.TPIX + pq) * 8 + (kk >> 2)) * 4 + (kk & 3)];
            const int k = k0 + kt * 32 + kk;
            if (k < p.K) p.bias_out[(int64_t)slab * p.K + k] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Software-pipelined split-bf16 filter gradient (the cconv.hip structure applied to Conv2DBackpropFilter): stride-1 5x5 / 3x3
// layers on feature maps of at least 8 x 16 pixels, one 32 x 32 (channel x filter) tile of the filter per workgroup.
//   * waves 4-7 ("D"): the fp32 halo of x and the tile of dy travel HBM -> registers through buffer descriptors (out-of-image
//     lanes read zeros) two stages ahead, and are split into the bf16 hi / lo planes of stage s+1 while stage s is multiplied.
//     The bias gradient (column sums of dy) is accumulated on the way.
//   * waves 0-3 ("M"): each owns up to 7 taps of the filter tile, accumulators in registers over the whole slab of tiles;
//     both operands come from the planes with the transposing read ds_read_b64_tr_b16 at compile-time offsets from ONE
//     address register per tap (the 16-pixel step loop is unrolled), operands of the next tap in flight under the MFMAs.
//   * one workgroup barrier per stage (= one 8 x 16-pixel tile); the per-slab partial filter leaves once, at the end.
// LDS (5x5): 2 x (x planes 30 KiB + dy planes 16 KiB) = 92 KiB: one workgroup per CU.
struct CwParams {
    const float* img; const float* feat; float* out; float* bias_out;
    int N, H, W, C, img_ld, Ho, Wo, K, feat_ld;
    int pt, pl, tiles_h, tiles_w, ctiles, ntiles_total, tiles_per_slab, ntaps;
    int dbg;             // MV3D_DBG bit 32: in-kernel stamps (diagnostics)
};

// In-kernel stamps (MV3D_DBG bit 32 only; cdna_hip_programming.md section 7): wave w of workgroup (blockIdx.x == 0, slab b < 128)
// writes the shader clock of event k to w_stamps[(b * 8 + w) * 64 + k]; read back with mv3d_debug_cwgrad_stamps().  No output
// value depends on them.
constexpr int CW_NSTAMP = 64;
__device__ unsigned long long w_stamps[128 * 8 * CW_NSTAMP];
__device__ __forceinline__ void wstamp(bool on, int wave, int lane, int& k) {
    if (on) {
        const unsigned long long t = __builtin_readcyclecounter();
        if (lane == 0 && k < CW_NSTAMP && blockIdx.x == 0 && blockIdx.y < 128) w_stamps[((int)blockIdx.y * 8 + wave) * CW_NSTAMP + k] = t;
        ++k;
    }
}

__device__ __forceinline__ void cw_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int KW>
__global__ __launch_bounds__(512) void cwgrad_kernel(const CwParams p) {
    constexpr int NTAPS = KW * KW;
    constexpr int TPW = (NTAPS + 3) / 4;                  // taps per M wave (7 / 3); the first NTAPS % 4 waves take one more than the rest
    constexpr int HR = 7 + KW, HC = 15 + KW;              // halo of an 8 x 16 tile
    constexpr int HPIX = HR * HC;
    constexpr int XPL = HPIX * 64, FPL = 128 * 64;        // bytes of one x / dy plane
    constexpr int BUF = 2 * XPL + 2 * FPL;                // hi + lo of both operands
    constexpr int XPIECES = (HPIX + 7) / 8, NPIECES = XPIECES + 16;
    constexpr int PPW = (NPIECES + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smc[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ct = blockIdx.x % p.ctiles, kt = blockIdx.x / p.ctiles;
    const int c0 = ct * 32, k0 = kt * 32;
    const int slab = blockIdx.y;
    const int tile_begin = slab * p.tiles_per_slab;
    const int tile_end = min(tile_begin + p.tiles_per_slab, p.ntiles_total);
    const int nstages = max(tile_end - tile_begin, 0);
    auto origin = [&](int stage, int& n, int& oh0, int& ow0) {
        int b = tile_begin + stage;
        const int tw_i = b % p.tiles_w; b /= p.tiles_w;
        const int th_i = b % p.tiles_h;
        n = b / p.tiles_h; oh0 = th_i * 8; ow0 = tw_i * 16;
    };

    if (wave >= 4) {
        // ---------------------------------------------------------------------------------------------- D waves
        const int dwv = wave - 4;
        __builtin_amdgcn_s_setprio(2);
        const int c4 = lane & 7, psub = lane >> 3;
        const int64_t xb = (int64_t)p.N * p.H * p.W * p.img_ld * 4, fb = (int64_t)p.N * p.Ho * p.Wo * p.feat_ld * 4;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.img), 0, (int)(xb < 0x7fffffff ? xb : 0x7fffffff), 0x00020000);
        const __amdgpu_buffer_rsrc_t fr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.feat), 0, (int)(fb < 0x7fffffff ? fb : 0x7fffffff), 0x00020000);
        // per-lane constants of this wave's pieces.  x pieces (halo pixels) and dy pieces (tile pixels) are numbered separately,
        // wave w takes x pieces w, w + 4, ... and dy pieces w, w + 4, w + 8, w + 12: which operand a slot holds is a compile-time
        // fact, so the loads below are twelve unconditional instructions (a slot that can be either made hipcc branch between
        // two descriptors and wait for vmcnt(0) at every join)
        constexpr int NXJ = (XPIECES + 3) / 4, NFJ = 4;
        int xvoff[NXJ], xrc[NXJ];                                     // byte offset from the stage origin; (row << 8 | column), 0x7fff00 = no pixel
        int fvoff[NFJ], frc[NFJ];
#pragma unroll
        for (int j = 0; j < NXJ; ++j) {
            const int pix = (dwv + 4 * j) * 8 + psub;
            const int hr = pix / HC, hc = pix - hr * HC;
            xvoff[j] = ((hr * p.W + hc) * p.img_ld + c4 * 4) * 4;
            xrc[j] = pix < HPIX ? (hr << 8 | hc) : 0x7fff00;
        }
#pragma unroll
        for (int j = 0; j < NFJ; ++j) {
            const int pq = (dwv + 4 * j) * 8 + psub;
            const int r = pq >> 4, c = pq & 15;
            fvoff[j] = ((r * p.Wo + c) * p.feat_ld + c4 * 4) * 4;
            frc[j] = r << 8 | c;
        }
        const bool xch_ok = c0 + c4 * 4 < p.C, fch_ok = k0 + c4 * 4 < p.K;
        // fp32 pieces of a stage: HBM -> registers (16 bytes per lane and piece; lanes outside the image / tensor read zeros through
        // the descriptor's bounds check), two stages ahead of the one being converted
        typedef unsigned int wu4 __attribute__((ext_vector_type(4)));
        struct Regs { wu4 x[NXJ]; wu4 f[NFJ]; };
        auto fetch = [&](int stage, Regs& R) {
            int n, oh0, ow0;
            origin(stage, n, oh0, ow0);
            const int ih0 = oh0 - p.pt, iw0 = ow0 - p.pl;
            const int xso = (((n * p.H + ih0) * p.W + iw0) * p.img_ld + c0) * 4;
            const int fso = (((n * p.Ho + oh0) * p.Wo + ow0) * p.feat_ld + k0) * 4;
#pragma unroll
            for (int j = 0; j < NXJ; ++j) {
                const int r = xrc[j] >> 8, c = xrc[j] & 255;
                const bool ok = xch_ok && (unsigned)(ih0 + r) < (unsigned)p.H && (unsigned)(iw0 + c) < (unsigned)p.W;      // the no-pixel marker is row 32767
                const int off = ok ? xvoff[j] + xso : (int)0x80000000;
                R.x[j] = __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < NFJ; ++j) {
                const int r = frc[j] >> 8, c = frc[j] & 255;
                const bool ok = fch_ok && oh0 + r < p.Ho && ow0 + c < p.Wo;
                const int off = ok ? fvoff[j] + fso : (int)0x80000000;
                R.f[j] = __builtin_amdgcn_raw_buffer_load_b128(fr, off, 0, 0);
            }
        };
        // bias-gradient partial sums, one per dy piece slot (compile-time indices: with a run-time index they lived in scratch, and
        // a scratch access waits for vmcnt(0))
        float4 bsum[NFJ];
#pragma unroll
        for (int j = 0; j < NFJ; ++j) bsum[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        auto convert = [&](int stage, const Regs& R) {
            unsigned char* const b = smc + (stage & 1) * BUF;
#pragma unroll
            for (int j = 0; j < NXJ; ++j) {
                const float4 v = make_float4(__uint_as_float(R.x[j][0]), __uint_as_float(R.x[j][1]), __uint_as_float(R.x[j][2]), __uint_as_float(R.x[j][3]));
                uint2 hi, lo;
                wsplit4(v, hi, lo);
                if (xrc[j] != 0x7fff00) {
                    unsigned char* d = b + ((dwv + 4 * j) * 8 + psub) * 64 + c4 * 8;
                    *reinterpret_cast<uint2*>(d) = hi;
                    *reinterpret_cast<uint2*>(d + XPL) = lo;
                }
            }
#pragma unroll
            for (int j = 0; j < NFJ; ++j) {
                const float4 v = make_float4(__uint_as_float(R.f[j][0]), __uint_as_float(R.f[j][1]), __uint_as_float(R.f[j][2]), __uint_as_float(R.f[j][3]));
                uint2 hi, lo;
                wsplit4(v, hi, lo);
                unsigned char* d = b + 2 * XPL + ((dwv + 4 * j) * 8 + psub) * 64 + c4 * 8;
                *reinterpret_cast<uint2*>(d) = hi;
                *reinterpret_cast<uint2*>(d + FPL) = lo;
                // bias gradient: this lane always sees filters k0 + 4 c4 .. + 3 (out-of-range lanes were loaded as zeros)
                bsum[j].x += v.x; bsum[j].y += v.y; bsum[j].z += v.z; bsum[j].w += v.w;
            }
        };
        const bool stp = (p.dbg & 32) != 0;
        int sk = 0;
        wstamp(stp, wave, lane, sk);                                       // 0: start
        // Stage s is multiplied while stage s + 1 is converted from registers fetched two stages earlier and stage s + 3 is
        // requested: a request has two whole stages to land (the LDS-DMA form of round 2's first version had one raw buffer, hence
        // one stage in flight: the data waves waited 8.5 k cycles of a 12 k-cycle stage for it -- stamps, profiles/r02_e_*).
        Regs R0, R1;
        auto clamp = [&](int st) { return st < nstages ? st : nstages - 1; };      // past the end: refetch the last tile (never converted)
        if (nstages > 0) {
            fetch(0, R0);
            fetch(clamp(1), R1);
            convert(0, R0);
            fetch(clamp(2), R0);
        }
        cw_barrier();
        wstamp(stp, wave, lane, sk);                                       // 1: prologue barrier passed
        for (int s = 0; s < nstages; s += 2) {
#ifdef CW_EXP_WAITSTAMP
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NXJ + NFJ) : "memory");
            wstamp(stp, wave, lane, sk);
#endif
            if (s + 1 < nstages) convert(s + 1, R1);
            wstamp(stp, wave, lane, sk);                                   // 2 + 4k: converted
            fetch(clamp(s + 3), R1);
            cw_barrier();
            wstamp(stp, wave, lane, sk);                                   // 3 + 4k: barrier passed
            if (s + 1 >= nstages) break;
            if (s + 2 < nstages) convert(s + 2, R0);
            wstamp(stp, wave, lane, sk);                                   // 4 + 4k: converted
            fetch(clamp(s + 4), R0);
            cw_barrier();
            wstamp(stp, wave, lane, sk);                                   // 5 + 4k: barrier passed
        }
        // bias gradient of this slab: 256 lanes x 4 float4 partial sums -> 32 column sums, fixed order (buffers are free now)
        cw_barrier();                                                      // M waves are past their last reads
        if (p.bias_out != nullptr && ct == 0) {
            float4* bx = reinterpret_cast<float4*>(smc);
            float4 t = bsum[0];
            t.x += bsum[1].x + bsum[2].x + bsum[3].x; t.y += bsum[1].y + bsum[2].y + bsum[3].y;
            t.z += bsum[1].z + bsum[2].z + bsum[3].z; t.w += bsum[1].w + bsum[2].w + bsum[3].w;
            bx[(tid - 256)] = t;
        }
        cw_barrier();
        if (p.bias_out != nullptr && ct == 0 && tid - 256 < 32) {
            const int kk = tid - 256;                                      // column k0 + kk: lanes with c4 == kk >> 2, component kk & 3
            const float* bf = reinterpret_cast<const float*>(smc);
            float t = 0.f;
            for (int l = 0; l < 32; ++l) t += bf[((l * 8 + (kk >> 2)) * 4) + (kk & 3)];
            if (k0 + kk < p.K) p.bias_out[(int64_t)slab * p.K + k0 + kk] = t;
        }
        return;
    }

    // -------------------------------------------------------------------------------------------------- M waves
    const int li = lane & 31, lh = lane >> 5;
    const int g16 = (lane >> 4) & 1, q = (lane >> 2) & 3, pp = lane & 3;
    const int lane_a = (8 * lh + q) * 64 + (16 * g16 + 4 * pp) * 2;     // row of pixel 8 lh + q of a 16-pixel step + column bytes of the transposing read
    // taps of this wave: BASE whole ones, [tap_lo, tap_lo + BASE), and a quarter of the last tap (25 = 4 x 6 + 1, 9 = 4 x 2 + 1):
    // slot BASE of every wave accumulates tap NTAPS - 1 over the wave's own two of the eight pixel steps of a tile; the four
    // partial sums are added through LDS once, at the end.  (One wave taking the odd tap whole made it the stage's critical path:
    // 7 taps against 6.25, and the other three multiplied a duplicate to keep the unrolled sequence uniform.)
    constexpr int REM = NTAPS % 4, BASE = NTAPS / 4;
    static_assert(REM == 1 && TPW == BASE + 1, "tap split written for 4 * BASE + 1 taps");
    const int tap_lo = wave * BASE;
    int tapoff[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int tap = i < BASE ? tap_lo + i : NTAPS - 1;
        tapoff[i] = ((tap / KW) * HC + (tap % KW)) * 64;
    }
    f32x16 acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    cw_barrier();                                                          // stage 0 is in buffer 0
    const bool stp = (p.dbg & 32) != 0;
    int stk = 0;
    wstamp(stp, wave, lane, stk);                                          // 0: first stage starts
    for (int s = 0; s < nstages; ++s) {
        const unsigned char* const buf = smc + (s & 1) * BUF;
        const unsigned char* va[TPW];
#pragma unroll
        for (int i = 0; i < TPW; ++i) va[i] = buf + lane_a + tapoff[i];
        const unsigned char* const vb = buf + 2 * XPL + lane_a;
        // One flat sequence of 8 steps x TPW taps: operands of tap t + PD are issued before the MFMAs of tap t (a single MFMA
        // wave per SIMD has nothing else to cover the LDS latency with), the dy fragments of the next step with its first tap.
        constexpr int PD = 2;
        constexpr int NSEQ = 8 * TPW;
        uint2 ar[PD + 1][4];
        uint2 br[2][4];
        auto lda = [&](int seq, uint2 (&a)[4]) {
            const int st = seq / TPW, i = seq % TPW;
            const unsigned char* pa = va[i] + st * HC * 64;
            a[0] = tr_read(pa); a[1] = tr_read(pa + 256);
            a[2] = tr_read(pa + XPL); a[3] = tr_read(pa + XPL + 256);
        };
        auto ldb = [&](int st, uint2 (&b)[4]) {
            b[0] = tr_read(vb + st * 1024); b[1] = tr_read(vb + st * 1024 + 256);
            b[2] = tr_read(vb + FPL + st * 1024); b[3] = tr_read(vb + FPL + st * 1024 + 256);
        };
        // slot BASE is live only in this wave's two steps (wave-uniform)
        auto live = [&](int seq) { return seq % TPW < BASE || ((seq / TPW) >> 1) == wave; };
        ldb(0, br[0]);
#pragma unroll
        for (int u = 0; u < PD; ++u) if (live(u)) lda(u, ar[u]);
#pragma unroll
        for (int seq = 0; seq < NSEQ; ++seq) {
            const int st = seq / TPW, i = seq % TPW;
            if (seq + PD < NSEQ && live(seq + PD)) lda(seq + PD, ar[(seq + PD) % (PD + 1)]);
            if (i == 0 && st + 1 < 8) ldb(st + 1, br[(st + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            if (!live(seq)) continue;
            const uint2 (&a)[4] = ar[seq % (PD + 1)];
            const uint2 (&b)[4] = br[st & 1];
            const wbf16x8 ah = __builtin_bit_cast(wbf16x8, make_uint4(a[0].x, a[0].y, a[1].x, a[1].y));
            const wbf16x8 al = __builtin_bit_cast(wbf16x8, make_uint4(a[2].x, a[2].y, a[3].x, a[3].y));
            const wbf16x8 bh = __builtin_bit_cast(wbf16x8, make_uint4(b[0].x, b[0].y, b[1].x, b[1].y));
            const wbf16x8 bl = __builtin_bit_cast(wbf16x8, make_uint4(b[2].x, b[2].y, b[3].x, b[3].y));
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[i], 0, 0, 0);
        }
        wstamp(stp, wave, lane, stk);                                      // 1 + 2s: taps done
        cw_barrier();
        wstamp(stp, wave, lane, stk);                                      // 2 + 2s: barrier passed
    }
    // the four partial sums of the shared tap: [wave][register][lane] behind the 4 KiB the data waves use for the bias sums
    // (all staging buffers are free: every multiplying wave is past the last stage's barrier)
    float* const tp = reinterpret_cast<float*>(smc + 8192);
#pragma unroll
    for (int r = 0; r < 16; ++r) tp[(wave * 16 + r) * 64 + lane] = acc[BASE][r];
    cw_barrier();                                                          // pairs with the D waves' two bias barriers
    cw_barrier();
    const int k = k0 + li;
#pragma unroll
    for (int i = 0; i < BASE; ++i) {
        float* out = p.out + ((int64_t)slab * p.ntaps + tap_lo + i) * p.C * p.K;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int c = c0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (c < p.C && k < p.K) out[(int64_t)c * p.K + k] = acc[i][r];
        }
    }
    {   // wave w finishes registers 4 w .. 4 w + 3 of the shared tap (fixed order: partial of wave 0, 1, 2, 3)
        float* out = p.out + ((int64_t)slab * p.ntaps + NTAPS - 1) * p.C * p.K;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = wave * 4 + q;
            const float v = ((tp[(0 * 16 + r) * 64 + lane] + tp[(1 * 16 + r) * 64 + lane]) + tp[(2 * 16 + r) * 64 + lane]) + tp[(3 * 16 + r) * 64 + lane];
            const int c = c0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (c < p.C && k < p.K) out[(int64_t)c * p.K + k] = v;
        }
    }
}

// planning: eligible geometry -> number of slabs (0 = not eligible); every caller (workspace sizing, launch) goes through here
// CUs a filter-gradient launch spreads its partial-filter slabs over: MV3D_WG_CUS (128 = half the chip, because these launches
// share it with the data-gradient chain), or what mv3d_set_wgrad_cus() asked for -- the train step raises it to the whole
// chip for the last filter gradients of the reverse pass, which run when the data-gradient chain has ended.
static int g_wg_cus_override = 0;
static int wgrad_cus(int kh = 5) {
    static int wg_cus = -1, wg_cus3 = -1;
    if (wg_cus < 0) { const char* e = getenv("MV3D_WG_CUS"); wg_cus = e ? atoi(e) : 128; }
    if (wg_cus3 < 0) { const char* e = getenv("MV3D_WG_CUS_3X3"); wg_cus3 = e ? atoi(e) : wg_cus; }      // the 3 x 3 layers (8 x 8 / 4 x 4 maps): latency-bound
    return g_wg_cus_override > 0 ? g_wg_cus_override : (kh == 3 ? wg_cus3 : wg_cus);
}
int set_wgrad_cus(int cus) { const int old = g_wg_cus_override; g_wg_cus_override = cus > 0 ? cus : 0; return old; }

static int cwgrad_plan(const mv3d_conv_geom* g, CwParams* out) {
    if ((disabled_paths() & (2 | 4096 | 2097152)) || g->sh != 1 || g->sw != 1 || g->kh != g->kw || (g->kh != 5 && g->kh != 3)) return 0;
    if (g->Ho < 8 || g->Wo < 16 || g->Ho % 8 || g->Wo % 16 || g->C % 32 || g->K % 32 || g->img_ld % 4 || g->feat_ld % 4) return 0;
    if ((int64_t)g->N * g->H * g->W * g->img_ld * 4 >= 0x7fffffff || (int64_t)g->N * g->Ho * g->Wo * g->feat_ld * 4 >= 0x7fffffff) return 0;
    CwParams p = {};
    int ho, wo;
    same_pad(g->H, g->kh, 1, &ho, &p.pt);
    same_pad(g->W, g->kw, 1, &wo, &p.pl);
    p.N = g->N; p.H = g->H; p.W = g->W; p.C = g->C; p.img_ld = g->img_ld; p.Ho = g->Ho; p.Wo = g->Wo; p.K = g->K; p.feat_ld = g->feat_ld;
    p.tiles_h = g->Ho / 8; p.tiles_w = g->Wo / 16; p.ctiles = g->C / 32; p.ntaps = g->kh * g->kw;
    p.ntiles_total = g->N * p.tiles_h * p.tiles_w;
    { static int d = -1; if (d < 0) { const char* e = getenv("MV3D_DBG"); d = e ? atoi(e) : 0; } p.dbg = d; }
    static int min_tiles = -1;
    if (min_tiles < 0) { const char* e = getenv("MV3D_CW_MINTILES"); min_tiles = e ? atoi(e) : 64; }
    if (p.ntiles_total < min_tiles) return 0;
    const int blocks_xy = p.ctiles * (g->K / 32);
    int nslab = std::max(1, wgrad_cus(g->kh) / blocks_xy);
    if (nslab > p.ntiles_total) nslab = p.ntiles_total;
    p.tiles_per_slab = cdiv(p.ntiles_total, nslab);
    nslab = cdiv(p.ntiles_total, p.tiles_per_slab);
    *out = p;
    return nslab;
}

template <int KW>
static int cwgrad_launch_t(const mv3d_conv_geom* g, CwParams p, int nslab, void* stream, const char* who) {
    const double flops = 2.0 * g->N * g->Ho * g->Wo * g->kh * g->kw * (double)g->C * g->K;
    const double bytes = 4.0 * ((double)g->N * g->H * g->W * g->C + (double)g->N * g->Ho * g->Wo * g->K + (double)g->kh * g->kw * g->C * g->K);
    dim3 grid(p.ctiles * (g->K / 32), nslab);
    constexpr int hpix = (7 + KW) * (15 + KW);
    const size_t lds = 2 * (size_t)(2 * hpix * 64 + 2 * 128 * 64);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cwgrad_kernel<KW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    return dispatch(stream, OpInfo{KW == 5 ? "cwgrad<5x5>" : "cwgrad<3x3>", flops, bytes}, [=](hipStream_t s) {
        cwgrad_kernel<KW><<<grid, 512, lds, s>>>(p);
        return launched(who);
    });
}

static int cwgrad_launch(const mv3d_conv_geom* g, CwParams p, int nslab, void* stream, const char* who) {
    return g->kw == 5 ? cwgrad_launch_t<5>(g, p, nslab, stream, who) : cwgrad_launch_t<3>(g, p, nslab, stream, who);
}

template <int TPW, int NKT>
static int launch_wgt(const WgTileParams& p, dim3 grid, size_t lds, void* stream, const char* name, const char* who,
                      double flops, double bytes) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_tile_kernel<TPW, NKT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    return dispatch(stream, OpInfo{name, flops, bytes}, [=](hipStream_t s) {
        wgrad_tile_kernel<TPW, NKT><<<grid, 512, lds, s>>>(p);
        return launched(who);
    });
}

template <int TPW, int NKT, int NI>
static int launch_wgb(const WgTileParams& p, dim3 grid, size_t lds, void* stream, const char* name, const char* who,
                      double flops, double bytes) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_b3_kernel<TPW, NKT, NI>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    return dispatch(stream, OpInfo{name, flops, bytes}, [=](hipStream_t s) {
        wgrad_b3_kernel<TPW, NKT, NI><<<grid, 512, lds, s>>>(p);
        return launched(who);
    });
}

// Plans the tiled kernel for a geometry.  Returns false when the generic filtgrad kernel must be used.
bool wgrad_tile_plan(const mv3d_conv_geom* g, WgTileParams* out, int* nslab_out, int* cfg_out, size_t* lds_out) {
    if (disabled_paths() & 2) return false;
    if (g->C < 16 || g->C % 4 != 0 || g->K % 4 != 0 || g->img_ld % 4 != 0 || g->feat_ld % 4 != 0) return false;
    const bool b3_on = !(disabled_paths() & 4096);
    // 4 x 4 and 8 x 8 feature maps (split-bf16 kernel only): a 128-pixel tile is 8 / 2 whole images (a 16 x 8 tile on an 8 x 8
    // map is half padding: twice the staging and twice the products)
    const bool small = b3_on && ((g->Ho == 4 && g->Wo == 4 && g->N >= 8) || (g->Ho == 8 && g->Wo == 8 && g->N >= 2));
    if (!small && (g->Ho * g->Wo < 64 || g->Wo < 8)) return false;
    const int ntaps = g->kh * g->kw;
    // 8 waves = IW item groups x PH pixel ranges; TPW = items per wave (template)
    int NKT, TPW, IW, PH, cfg;
    if (ntaps == 25) {
        if (g->K > 32) { NKT = 2; IW = 8; PH = 1; TPW = 7; cfg = 1; }       // 50 items: 7,7,6,6,6,6,6,6
        else { NKT = 1; IW = 4; PH = 2; TPW = 7; cfg = 0; }                 // 25 items: 7,6,6,6
    } else if (ntaps == 9) {
        if (g->K > 32) { NKT = 2; IW = 4; PH = 2; TPW = 5; cfg = 3; }       // 18 items: 5,5,4,4
        else { NKT = 1; IW = 2; PH = 4; TPW = 5; cfg = 2; }                 // 9 items: 5,4
    } else return false;
    const bool b3 = !(disabled_paths() & 4096);          // split-bf16 kernel: waves = (k-tile, tap group) x pixel range
    WgTileParams p = {};
    p.b3 = b3 ? 1 : 0;
    int ho, wo;
    same_pad(g->H, g->kh, g->sh, &ho, &p.pt);
    same_pad(g->W, g->kw, g->sw, &wo, &p.pl);
    // spatial tile: 128 output pixels (64 if two stride-2 halos do not fit in LDS), least padded work
    int64_t best = -1;
    const size_t xchg = (PH > 1) ? (size_t)IW * TPW * 4096 : 0;            // end-of-kernel partial exchange
    p.G = 1; p.img_shift = 7; p.HRi = 0;
    if (small) {
        const int side = g->Ho, ipx = side * side, sshift = side == 4 ? 2 : 3;
        // 128 pixels of whole images per tile, 64 when the stride-2 halos are too big
        for (int G = 128 / ipx; G >= std::max(1, 64 / ipx) && best < 0; G >>= 1) {
            const int tpix = G * ipx;
            if ((tpix >> 4) % PH != 0) continue;
            const int HRi = (side - 1) * g->sh + g->kh, HC = (side - 1) * g->sw + g->kw, HR = G * HRi;
            const size_t lds = std::max(2 * (size_t)(((HR * HC + tpix * NKT + 7) / 8) * 1024), xchg);
            if (lds > 160 * 1024 || HR * HC * 8 > 8 * 512) continue;
            p.G = G; p.img_shift = 2 * sshift; p.TH = side; p.TW = side; p.tw_shift = sshift; p.tiles_h = p.tiles_w = 1; p.TPIX = tpix;
            p.HRi = HRi; p.HC = HC; p.HR = HR;
            best = 1;
        }
    }
    for (int tpix = 128; tpix >= 64 && best < 0; tpix >>= 1) {
        if (b3 ? ((tpix >> 4) % PH != 0) : ((tpix >> 1) / PH < 2)) break;
        for (int sh = 3; sh <= 6; ++sh) {
            const int TW = 1 << sh, TH = tpix / TW;
            if (TH < 1 || TH > g->Ho * 2 || TW > g->Wo * 2) continue;
            const int th = cdiv(g->Ho, TH), tw = cdiv(g->Wo, TW);
            const int HR = (TH - 1) * g->sh + g->kh, HC = (TW - 1) * g->sw + g->kw;
            const size_t lds = std::max(2 * (size_t)(((HR * HC + tpix * NKT + 7) / 8) * 1024), xchg);
            if (lds > 160 * 1024) continue;
            if (b3 && HR * HC * 8 > 8 * 512) continue;                      // staging registers: <= 8 float4 per thread
            const int64_t cost = (int64_t)th * tw * (tpix * 64 + HR * HC);
            if (best < 0 || cost < best) {
                best = cost;
                p.TH = TH; p.TW = TW; p.tw_shift = sh; p.tiles_h = th; p.tiles_w = tw; p.HR = HR; p.HC = HC; p.TPIX = tpix;
            }
        }
    }
    if (best < 0) return false;
    p.N = g->N; p.H = g->H; p.W = g->W; p.C = g->C; p.img_ld = g->img_ld;
    p.Ho = g->Ho; p.Wo = g->Wo; p.K = g->K; p.feat_ld = g->feat_ld;
    p.sh = g->sh; p.sw = g->sw; p.kw = g->kw;
    p.ntaps = ntaps; p.nitems = ntaps * NKT; p.ipw = cdiv(p.nitems, IW); p.IW = IW; p.PH = PH;
    (void)TPW;
    p.ctiles = cdiv(g->C, 32);
    const int kgroups = cdiv(g->K, 32 * NKT);
    if (p.G == 1) { p.HRi = p.HR; p.img_shift = p.TPIX == 128 ? 7 : 6; }
    p.inv_hri = ((1 << 20) + p.HRi - 1) / p.HRi;
    p.ntiles_total = cdiv(g->N, p.G) * p.tiles_h * p.tiles_w;
    // one workgroup per CU in total (operands are prefetched inside the workgroup)
    const int blocks_xy = p.ctiles * kgroups;
    int nslab = std::max(1, wgrad_cus(g->kh) / blocks_xy);      // default 128: half the CUs (measured +3 % on the step over 256: fewer partial filters, room for the main stream)
    if (nslab > p.ntiles_total) nslab = p.ntiles_total;
    p.tiles_per_slab = cdiv(p.ntiles_total, nslab);
    nslab = cdiv(p.ntiles_total, p.tiles_per_slab);
    p.inv_hc = ((1 << 20) + p.HC - 1) / p.HC;
    p.tpix8_shift = p.TPIX == 128 ? 10 : 9;
    // Tap split (split-bf16 kernels): a workgroup's life on these layers is launch + one or a few tiles + the store of its
    // partial filter (102 / 204 KB: the longest piece); with the taps divided over two workgroups each stages the same halo
    // (from L2: the chip is half idle during these launches anyway) and multiplies and stores half -- twice the workgroups,
    // half the chain.
    p.tsplit = 1;
    {
        static int ts = -1;
        if (ts < 0) { const char* e = getenv("MV3D_WG_TAPSPLIT"); ts = e ? atoi(e) : 2; }
        if (b3 && ts == 2) p.tsplit = 2;
    }
    *out = p;
    *nslab_out = nslab;
    *cfg_out = cfg;
    *lds_out = std::max(2 * (size_t)(((p.HR * p.HC + p.TPIX * NKT + 7) / 8) * 1024), xchg);
    if (b3) *lds_out = std::max(*lds_out, (size_t)32 * 1024);              // bias partial exchange
    return true;
}

int wgrad_tile_launch(const mv3d_conv_geom* g, WgTileParams p, int nslab, int cfg, size_t lds, void* stream, const char* who) {
    const double flops = 2.0 * g->N * g->Ho * g->Wo * g->kh * g->kw * (double)g->C * g->K;
    const double bytes = 4.0 * ((double)g->N * g->H * g->W * g->C + (double)g->N * g->Ho * g->Wo * g->K + (double)g->kh * g->kw * g->C * g->K);
    const int NKTv = (cfg & 1) ? 2 : 1;
    dim3 grid(p.ctiles * cdiv(g->K, 32 * NKTv), nslab, p.b3 ? p.tsplit : 1);
    if (p.b3) {
        const bool ni4 = p.HR * p.HC * 8 <= 4 * 512;
        if (p.tsplit == 2) {         // 25 taps over 2 x 4 groups: at most 4 taps per wave
            if (cfg == 0) return ni4 ? launch_wgb<4, 1, 4>(p, grid, lds, stream, "wgrad_b3<5x5,K32,taps/2>", who, flops, bytes) : launch_wgb<4, 1, 8>(p, grid, lds, stream, "wgrad_b3<5x5,K32,taps/2>", who, flops, bytes);
            if (cfg == 1) return ni4 ? launch_wgb<4, 2, 4>(p, grid, lds, stream, "wgrad_b3<5x5,K64,taps/2>", who, flops, bytes) : launch_wgb<4, 2, 8>(p, grid, lds, stream, "wgrad_b3<5x5,K64,taps/2>", who, flops, bytes);
            // 9 taps over 2 x 2 groups: at most 3 per wave
            if (cfg == 2) return ni4 ? launch_wgb<3, 1, 4>(p, grid, lds, stream, "wgrad_b3<3x3,K32,taps/2>", who, flops, bytes) : launch_wgb<3, 1, 8>(p, grid, lds, stream, "wgrad_b3<3x3,K32,taps/2>", who, flops, bytes);
            return ni4 ? launch_wgb<3, 2, 4>(p, grid, lds, stream, "wgrad_b3<3x3,K64,taps/2>", who, flops, bytes) : launch_wgb<3, 2, 8>(p, grid, lds, stream, "wgrad_b3<3x3,K64,taps/2>", who, flops, bytes);
        }
        switch (cfg) {
            case 0: return ni4 ? launch_wgb<7, 1, 4>(p, grid, lds, stream, "wgrad_b3<5x5,K32>", who, flops, bytes) : launch_wgb<7, 1, 8>(p, grid, lds, stream, "wgrad_b3<5x5,K32>", who, flops, bytes);
            case 1: return ni4 ? launch_wgb<7, 2, 4>(p, grid, lds, stream, "wgrad_b3<5x5,K64>", who, flops, bytes) : launch_wgb<7, 2, 8>(p, grid, lds, stream, "wgrad_b3<5x5,K64>", who, flops, bytes);
            case 2: return ni4 ? launch_wgb<5, 1, 4>(p, grid, lds, stream, "wgrad_b3<3x3,K32>", who, flops, bytes) : launch_wgb<5, 1, 8>(p, grid, lds, stream, "wgrad_b3<3x3,K32>", who, flops, bytes);
            default: return ni4 ? launch_wgb<5, 2, 4>(p, grid, lds, stream, "wgrad_b3<3x3,K64>", who, flops, bytes) : launch_wgb<5, 2, 8>(p, grid, lds, stream, "wgrad_b3<3x3,K64>", who, flops, bytes);
        }
    }
    switch (cfg) {
        case 0: return launch_wgt<7, 1>(p, grid, lds, stream, "wgrad_tile<5x5,K32>", who, flops, bytes);
        case 1: return launch_wgt<7, 2>(p, grid, lds, stream, "wgrad_tile<5x5,K64>", who, flops, bytes);
        case 2: return launch_wgt<5, 1>(p, grid, lds, stream, "wgrad_tile<3x3,K32>", who, flops, bytes);
        default: return launch_wgt<5, 2>(p, grid, lds, stream, "wgrad_tile<3x3,K64>", who, flops, bytes);
    }
}

int wgrad_tile_nslab(const mv3d_conv_geom* g) {
    { CwParams cp; const int ns = cwgrad_plan(g, &cp); if (ns > 0) return ns; }
    WgTileParams p; int nslab, cfg; size_t lds;
    return wgrad_tile_plan(g, &p, &nslab, &cfg, &lds) ? nslab : 0;
}

int wgrad_tile_launch_erased(const mv3d_conv_geom* g, const void* img, const void* feat, void* out, void* bias_out, void* stream,
                             const char* who, int* nslab_out) {
    {
        CwParams cp;
        const int ns = cwgrad_plan(g, &cp);
        if (ns > 0) {
            cp.img = (const float*)img; cp.feat = (const float*)feat; cp.out = (float*)out; cp.bias_out = (float*)bias_out;
            *nslab_out = ns;
            return cwgrad_launch(g, cp, ns, stream, who);
        }
    }
    WgTileParams p; int nslab, cfg; size_t lds;
    if (!wgrad_tile_plan(g, &p, &nslab, &cfg, &lds)) return fail(MV3D_E_INVAL, "%s: tiled wgrad not applicable", who);
    p.img = (const float*)img; p.feat = (const float*)feat; p.out = (float*)out; p.bias_out = (float*)bias_out;
    *nslab_out = nslab;
    return wgrad_tile_launch(g, p, nslab, cfg, lds, stream, who);
}

}  // namespace mv3d

extern "C" int mv3d_debug_cwgrad_stamps(void* dst, size_t bytes) {
    if (!dst || bytes > sizeof(unsigned long long) * 128 * 8 * mv3d::CW_NSTAMP) return mv3d::fail(MV3D_E_INVAL, "mv3d_debug_cwgrad_stamps: bad buffer");
    hipError_t e = hipMemcpyFromSymbol(dst, HIP_SYMBOL(mv3d::w_stamps), bytes, 0, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return mv3d::fail(MV3D_E_HIP, "mv3d_debug_cwgrad_stamps: %s", hipGetErrorString(e));
    return MV3D_OK;
}
