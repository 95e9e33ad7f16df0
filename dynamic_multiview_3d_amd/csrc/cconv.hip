// Software-pipelined split-bf16 convolution for the layers that carry the FLOPs: single-phase, stride-1, 5x5 / 3x3,
// 16 x 16-pixel output tiles x 32 filters (conv2d_msra forward and the data gradient of stride-1 layers:
// e0_0 / e1_0 / e2_0 / d1_0 / d2_0 / d3_0 of appearance_flow_model.py:89-122).
//
// bconvu (bconv.hip) runs a workgroup as a serial chain -- fetch the halo (an HBM round trip through registers), split,
// multiply, store -- and only two such workgroups fit a CU, so the matrix cores are busy a third of the time.  Here ONE
// persistent workgroup of 8 waves per CU splits the roles and every transfer is asynchronous:
//   * waves 4-7 ("D") move data.  The fp32 halo of stage s+2 travels HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, no
//     registers, in flight for a whole stage); the halo of stage s+1, which landed long ago, is split into bf16 hi / lo
//     planes LDS -> LDS while stage s is multiplied.  A wave converts exactly the 1 KiB pieces its own DMA wrote, so the
//     only ordering needed is its own vmcnt.
//   * waves 0-3 ("M") multiply: the tap loop is fully unrolled, A fragments come from the hi / lo planes with
//     ds_read_b128 at compile-time offsets (one address register per filter column and k-step), B fragments from the
//     prepared filter through a register ring (the only global loads these waves wait for).  The outputs of the PREVIOUS
//     tile leave from a second accumulator set, a few stores per tap inside the tap loop, so neither the epilogue nor the
//     store acknowledgements ever stall the matrix pipe.
//   * one workgroup barrier per stage (a stage = one tile x one 32-channel chunk).
// LDS (5x5): 2 buffers x (hi + lo) x 400 halo pixels x 64 B = 100 KiB + 50 KiB of raw fp32 = 150 KiB: one workgroup per CU.
//
// LDS image of a plane: pixel (hr, hc) of the 20-column halo owns 64 bytes (32 channels of bf16) = four 16-byte slots; slot q
// is stored at q ^ ((hc >> 2) & 3).  A ds_read_b128 is served in groups of 16 lanes that hold 16 different columns mod 16
// (lanes {0-3, 12-15} of one tile row and {4-11} of the next, MI355X_MICROARCH.md LDS table); with the 20-pixel pitch
// (a multiple of 4) their bank slots 4 * (hc & 3) + (q ^ ((hc >> 2) & 3)) are 16 different ones: conflict-free for every tap.
#include "conv_common.h"
#include <algorithm>
#include <type_traits>

namespace mv3d {

typedef __bf16 cbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 cbf16x4 __attribute__((ext_vector_type(4)));

// In-kernel stamps (diagnostics, MV3D_DBG bit 32 only; cdna_hip_programming.md section 7): wave `w` of workgroup `b` writes the
// shader clock of event k to c_stamps[(b * 8 + w) * CC_NSTAMP + k]; tools/cconv_stamps.py reads them back through
// mv3d_debug_cconv_stamps().  No output value depends on them.
constexpr int CC_NSTAMP = 64;
__device__ unsigned long long c_stamps[256 * 8 * CC_NSTAMP];
__device__ __forceinline__ void cstamp(bool on, int wave, int lane, int& k) {
    if (on) {
        const unsigned long long t = __builtin_readcyclecounter();
        if (lane == 0 && k < CC_NSTAMP && blockIdx.x < 256 && blockIdx.y == 0) c_stamps[((int)blockIdx.x * 8 + wave) * CC_NSTAMP + k] = t;
        ++k;
    }
}

__device__ __forceinline__ void csplit4(const float4& v, uint2& hi, uint2& lo) {
    cbf16x4 h, l;
    h[0] = (__bf16)v.x; h[1] = (__bf16)v.y; h[2] = (__bf16)v.z; h[3] = (__bf16)v.w;
    l[0] = (__bf16)(v.x - (float)h[0]); l[1] = (__bf16)(v.y - (float)h[1]);
    l[2] = (__bf16)(v.z - (float)h[2]); l[3] = (__bf16)(v.w - (float)h[3]);
    hi = __builtin_bit_cast(uint2, h);
    lo = __builtin_bit_cast(uint2, l);
}

__device__ __forceinline__ void cdma16(const float* gsrc, unsigned char* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// workgroup barrier that neither drains the vector-memory queue (LDS-DMA and output stores stay in flight across it) nor
// lets the compiler move LDS accesses over it
__device__ __forceinline__ void cbarrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifndef CC_LA5
#define CC_LA5 3
#endif
// smallest ring depth R of the saved-output registers such that the loads of group g + R (top of tap ts(g + R) - D) come
// after the stores of group g (middle of tap ts(g)), with ts(g) = D + g * (NTAPS - D) / NG
constexpr int cc_gr_depth(int ntaps, int d, int ng) {
    for (int r = 1; r < ng; ++r) {
        bool ok = true;
        for (int g = 0; g + r < ng; ++g)
            if (!((d + ((g + r) * (ntaps - d)) / ng) - d > d + (g * (ntaps - d)) / ng)) ok = false;
        if (ok) return r;
    }
    return ng;
}
constexpr int CC_HC = 20;             // halo pitch in pixels (15 + kw rounded up to a multiple of 4)

template <int KW, bool REV, bool HAS_G, int U, int D, int MT>
__global__ __launch_bounds__(512) void cconv_kernel(const IgemmParams p, const HconvExtra x, const uint4* __restrict__ Wf, int ntiles) {
    constexpr int NTAPS = KW * KW;
    static_assert(NTAPS % U == 0, "ring slots must line up across stages");
    static_assert(MT == 1 || MT == 2, "tile height 8 or 16: one or two 32-pixel groups (two tile rows each) per multiplying wave");
    constexpr int TH = 8 * MT;
    static_assert(D >= 1 && D < U, "look-ahead distance of the filter ring, in taps");
    constexpr int HR = TH - 1 + KW;                       // halo rows
    constexpr int HPIX = HR * CC_HC;
    constexpr int PL = HPIX * 64;                         // bytes of one bf16 plane
    constexpr int NPIECES = (HPIX + 7) / 8;               // 1 KiB pieces of the raw fp32 halo (8 pixels x 128 B)
    constexpr int PPW = (NPIECES + 3) / 4;                // pieces per D wave
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    unsigned char* const raw = lds + 4 * PL;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.y * 32;
    const int chunks = x.chunks;
    const int my_tiles = (x.n_tiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int nstages = my_tiles * chunks;
    auto origin = [&](int stage, int& n, int& oh0, int& ow0) {
        int bq = (int)blockIdx.x + (stage / chunks) * (int)gridDim.x;
        const int tw_i = bq % x.tiles_w; bq /= x.tiles_w;
        const int th_i = bq % x.tiles_h;
        n = bq / x.tiles_h; oh0 = th_i * TH; ow0 = tw_i * 16;
    };

    if (wave >= 4) {
        // ---------------------------------------------------------------------------------------------- D waves
        const int dwv = wave - 4;
        __builtin_amdgcn_s_setprio(2);          // few instructions, all on the critical path of the next stage: win the issue arbitration
        const int c4 = lane & 7, psub = lane >> 3;
        // buffer descriptor over the activation tensor: lanes outside the image get an offset beyond num_records and the
        // LDS-DMA writes zeros for them (hardware bounds check), so the staging loop has no pointer selects
        const int64_t a_bytes = (int64_t)p.N * p.Ha * p.Wa * p.a_ld * 4;
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.A), 0, (int)(a_bytes < 0x7fffffff ? a_bytes : 0x7fffffff), 0x00020000);
        // per-lane constants of this wave's pieces: byte offset relative to the halo origin, (halo row << 8 | halo column)
        int voff[PPW], hrc[PPW];
#pragma unroll
        for (int j = 0; j < PPW; ++j) {
            const int pc = dwv + 4 * j;
            const int pix = pc * 8 + psub;
            const int hr = pix / CC_HC, hc = pix - hr * CC_HC;
            const bool in = pc < NPIECES && pix < HPIX;
            voff[j] = ((hr * p.Wa + hc) * p.a_ld + c4 * 4) * 4;
            hrc[j] = in ? (hr << 8 | hc) : 0x7fff00;                   // row 32767 + ih0 is never inside an image
        }
        auto issue = [&](int stage) {
            int n, oh0, ow0;
            origin(stage, n, oh0, ow0);
            const int cc = stage % chunks;
            const int ih0 = oh0 + x.dh_min, iw0 = ow0 + x.dw_min;
            const int soff = (((n * p.Ha + ih0) * p.Wa + iw0) * p.a_ld + cc * 32) * 4;      // wave-uniform, may wrap below zero at the top / left border
            const bool ch_ok = cc * 32 + c4 * 4 < p.Ka;
#pragma unroll
            for (int j = 0; j < PPW; ++j) {
                const int pc = dwv + 4 * j;
                if (pc < NPIECES) {
                    const int ih = ih0 + (hrc[j] >> 8), iw = iw0 + (hrc[j] & 255);
                    const bool ok = ch_ok && (unsigned)ih < (unsigned)p.Ha && (unsigned)iw < (unsigned)p.Wa;
                    const int off = ok ? voff[j] + soff : (int)0x80000000;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(raw + pc * 1024), 16, off, 0, 0, 0);
                }
            }
        };
        auto convert = [&](int stage) {
            unsigned char* const hi_pl = lds + (stage & 1) * 2 * PL;
#pragma unroll
            for (int j = 0; j < PPW; ++j) {
                const int pc = dwv + 4 * j;
                if (pc < NPIECES) {
                    const int hc = hrc[j] & 255;
                    const float4 v = *reinterpret_cast<const float4*>(raw + pc * 1024 + lane * 16);
                    uint2 hi, lo;
                    csplit4(v, hi, lo);
                    if (hrc[j] != 0x7fff00) {
                        const int off = (pc * 8 + psub) * 64 + ((((c4 >> 1) ^ ((hc >> 2) & 3))) << 4) + (c4 & 1) * 8;
                        *reinterpret_cast<uint2*>(hi_pl + off) = hi;
                        *reinterpret_cast<uint2*>(hi_pl + PL + off) = lo;
                    }
                }
            }
        };
        const bool st = (x.dbg & 32) != 0;
        int sk = 0;
        cstamp(st, wave, lane, sk);                                    // 0: start
        if (nstages > 0) {
            issue(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            cstamp(st, wave, lane, sk);                                // 1: first halo landed
            convert(0);
            if (nstages > 1) issue(1);
        }
        cbarrier();                                                    // stage 0 is in buffer 0
        cstamp(st, wave, lane, sk);                                    // 2: prologue barrier passed
        for (int s = 0; s < nstages; ++s) {
            if (s + 1 < nstages) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's pieces of stage s+1 have landed (issued a stage ago)
                cstamp(st, wave, lane, sk);                            // 3 + 4s: landed
                convert(s + 1);                                        // into the buffer stage s-1 used: free since the last barrier
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                cstamp(st, wave, lane, sk);                            // 4 + 4s: converted
                if (s + 2 < nstages) issue(s + 2);                     // this wave has read its pieces: their raw slots are free
                cstamp(st, wave, lane, sk);                            // 5 + 4s: issued
            }
            cbarrier();
            cstamp(st, wave, lane, sk);                                // 6 + 4s: barrier passed
        }
        return;
    }

    // -------------------------------------------------------------------------------------------------- M waves
    const int li = lane & 31, lh = lane >> 5;
    const int tc = li & 15, tr0 = wave * 2 * MT + (li >> 4);         // pixel group m covers tile rows tr0 + 2m
    // A-operand address registers: one per (filter column, k-step); everything else is a compile-time offset
    int vq[KW][2];
#pragma unroll
    for (int q = 0; q < KW; ++q) {
        const int dwq = REV ? KW - 1 - q : q;
        const int key = ((tc + dwq) >> 2) & 3;
#pragma unroll
        for (int s = 0; s < 2; ++s) vq[q][s] = (tr0 * CC_HC + tc) * 64 + (((s * 2 + lh) ^ key) << 4);
    }
    auto tap_imm = [](int t) {                                          // (dh', dw') of tap t -> byte offset inside a plane
        const int r = t / KW, q = t % KW;
        const int dh = REV ? KW - 1 - r : r, dw = REV ? KW - 1 - q : q;
        return (dh * CC_HC + dw) * 64;
    };

    f32x16 acc[MT], prev[MT];
    struct BSet { uint4 b[2][2]; };
    BSet ring[U];
    uint4 a[MT][2][2];
    // filter fragments through a buffer descriptor: ONE per-lane offset register for every load, the (tap, chunk) part is scalar
    typedef unsigned int cu4 __attribute__((ext_vector_type(4)));
    const int wf_bytes = NTAPS * chunks * ntiles * 4096;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(Wf), 0, wf_bytes, 0x00020000);
    const int wlane = lane * 16;
    auto load_b = [&](BSet& f, int t, int cc) {
        const int so = ((t * chunks + cc) * ntiles + (int)blockIdx.y) * 4096;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f.b[s][0] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane + (s * 2) * 1024, so, 0));
            f.b[s][1] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane + (s * 2 + 1) * 1024, so, 0));
        }
    };
    // Epilogue of the previous tile, one group of four consecutive pixels at a time (accumulator rows q = (wave*2+m)*32 +
    // 8*grp + 4*lh + j).  Branch-free: tiles are whole (the planner only sends images whose sides are multiples of 16 and
    // filter counts that are multiples of 32 here), the activation is y = c1*x + c2*|x| for none / lrelu / relu alike
    // (tf_utils.py:25-33 written as the reference writes it; relu keeps -0.0 for x < 0, common.h act_apply), its derivative
    // from the saved output g1 + g2 * sign (common.h act_grad_from_out).  Addresses: a wave-uniform 64-bit base per store
    // (scalar arithmetic) + ONE per-lane 32-bit offset.
    const int col = n0 + li;
    const float bias = p.bias ? p.bias[col] : 0.f;
    const float c1 = p.act == MV3D_ACT_NONE ? 1.f : (p.act == MV3D_ACT_LRELU ? 0.5f * (1.0f + p.leak) : 0.5f);
    const float c2 = p.act == MV3D_ACT_NONE ? 0.f : (p.act == MV3D_ACT_LRELU ? 0.5f * (1.0f - p.leak) : 0.5f);
    const bool is_relu = p.act == MV3D_ACT_RELU;
    const float g1 = p.gact == MV3D_ACT_LRELU ? 0.5f * (1.0f + p.gleak) : 0.5f;
    const float g2 = p.gact == MV3D_ACT_LRELU ? 0.5f * (1.0f - p.gleak) : 0.5f;
    const bool g_relu = p.gact == MV3D_ACT_RELU;
    const int out_lane = ((wave * 2 * MT * p.Wc + 4 * lh) * p.c_ld + col) * 4;             // bytes
    const int ref_lane = HAS_G ? ((wave * 2 * MT * p.Wc + 4 * lh) * p.g_ld + col) * 4 : 0;
    const int64_t o_bytes = (int64_t)p.N * p.Hc * p.Wc * p.c_ld * 4, r_bytes = (int64_t)p.N * p.Hc * p.Wc * p.g_ld * 4;
    const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(p.Out, 0, (int)(o_bytes < 0x7fffffff ? o_bytes : 0x7fffffff), 0x00020000);
    const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(HAS_G ? p.gref : p.Out), 0,
                                                                             (int)(r_bytes < 0x7fffffff ? r_bytes : 0x7fffffff), 0x00020000);
    int ptile = 0;                                                     // first pixel of the previous tile (wave-uniform)
    // store schedule of the tap loop: group g leaves in the middle of tap ts(g); its saved-output loads are issued at the top of
    // tap ts(g) - D, in front of that tap's filter look-ahead loads, so the in-order vmcnt wait that covers the filter ring also
    // covers them.  The saved-output registers are a ring of GRD groups: slot g % GRD must not be refilled before group g has
    // left, i.e. ts(g + GRD) - D > ts(g) for every g (3x3 with 16 x 16 tiles needs four slots, every other instance two).
    constexpr int NG = 4 * MT;                                         // store groups per tile and wave
    auto ts_of = [](int g8) { return D + (g8 * (NTAPS - D)) / NG; };
    constexpr int GRD = cc_gr_depth(NTAPS, D, NG);
    static_assert(GRD >= 1 && GRD <= NG, "saved-output ring");
    float gr[GRD][4];
    auto group_pix = [&](int g8, int j) {                              // pixel offset of element j of group g8 inside the tile, minus the lane part
        const int m = g8 >> 2, grp = g8 & 3;
        return (2 * m + (grp >> 1)) * p.Wc + 8 * (grp & 1) + j;
    };
    auto pre_group = [&](int g8) {                                     // issue the saved-output loads of a group (data gradient only)
        if constexpr (HAS_G) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                gr[g8 % GRD][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrsrc, ref_lane, (ptile + group_pix(g8, j)) * p.g_ld * 4, 0));
            }
        }
    };
    auto fin_group = [&](int g8) {
        const int m = g8 >> 2, grp = g8 & 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float xv = prev[m][4 * grp + j] + bias;
            float v = __fadd_rn(__fmul_rn(c1, xv), __fmul_rn(c2, fabsf(xv)));
            v = (is_relu && xv < 0.0f) ? -0.0f : v;
            if constexpr (HAS_G) {
                const float y = gr[g8 % GRD][j];
                const bool neg = g_relu ? (__float_as_uint(y) >> 31) != 0 : y < 0.0f;
                const float sgn = y > 0.0f ? 1.0f : (neg ? -1.0f : 0.0f);
                v *= g1 + g2 * sgn;
            }
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), orsrc, out_lane, (ptile + group_pix(g8, j)) * p.c_ld * 4, 0);
        }
    };
#pragma unroll
    for (int u = 0; u < D; ++u) load_b(ring[u], u, 0);
    cbarrier();                                                        // stage 0 is in buffer 0
    const bool st = (x.dbg & 32) != 0;
    int stk = 0;

    // one stage = NTAPS taps on one 32-channel chunk of one tile; PEND: the previous tile's outputs leave inside the tap loop
    auto stage_body = [&](int s, auto pend_tag) {
        constexpr bool PEND = decltype(pend_tag)::value;
        const int cc = s % chunks;
        const int hb = (s & 1) * 2 * PL;
        const unsigned char* const hbase = lds + hb;
        if (cc == 0) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        }
        auto read_a = [&](int t, int m, int sk) {
            const unsigned char* ap = hbase + vq[t % KW][sk] + (tap_imm(t) + m * (2 * CC_HC * 64));
            a[m][sk][0] = *reinterpret_cast<const uint4*>(ap);
            a[m][sk][1] = *reinterpret_cast<const uint4*>(ap + PL);
        };
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int sk = 0; sk < 2; ++sk) read_a(0, m, sk);
#pragma unroll
        for (int t = 0; t < NTAPS; ++t) {
#ifdef CC_TAP_STAMPS
            if (t % 6 == 0) cstamp(st, wave, lane, stk);
#endif
            if constexpr (PEND) {
#pragma unroll
                for (int g8 = 0; g8 < NG; ++g8)
                    if (t == ts_of(g8) - D) pre_group(g8);
            }
            {   // look-ahead tap: this chunk, or the first taps of the next stage's chunk
                const int tn = t + D;
                const int ccn = cc + 1 < chunks ? cc + 1 : 0;
                load_b(ring[tn % U], tn < NTAPS ? tn : tn - NTAPS, tn < NTAPS ? cc : ccn);
            }
            __builtin_amdgcn_sched_barrier(0);                         // keep the look-ahead load here (hipcc sinks it to its first use)
            const BSet& f = ring[t % U];
#pragma unroll
            for (int sk = 0; sk < 2; ++sk)
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const cbf16x8 ah = __builtin_bit_cast(cbf16x8, a[m][sk][0]), al = __builtin_bit_cast(cbf16x8, a[m][sk][1]);
                    const cbf16x8 bh = __builtin_bit_cast(cbf16x8, f.b[sk][0]), bl = __builtin_bit_cast(cbf16x8, f.b[sk][1]);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[m], 0, 0, 0);
                    // refill this operand pair for the next tap right here: nine MFMAs (~290 cycles) lie between this read and its
                    // first use; pinned, because hipcc otherwise sinks the reads down to their uses
                    if (t + 1 < NTAPS) read_a(t + 1 < NTAPS ? t + 1 : t, m, sk);
                    if constexpr (PEND) {
                        if (sk == 0 && m == MT - 1) {
#pragma unroll
                            for (int g8 = 0; g8 < NG; ++g8)
                                if (t == ts_of(g8)) fin_group(g8);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
    };

    bool pending = false;
    cstamp(st, wave, lane, stk);                                        // 0: first stage starts
    for (int s = 0; s < nstages; ++s) {
        const int cc = s % chunks;
        if (pending && cc == 0) { stage_body(s, std::true_type{}); pending = false; }
        else stage_body(s, std::false_type{});
        cstamp(st, wave, lane, stk);                                    // 1 + 2s: taps done
        if (cc == chunks - 1) {
#pragma unroll
            for (int m = 0; m < MT; ++m) prev[m] = acc[m];
            int pn, poh0, pow0;
            origin(s, pn, poh0, pow0);
            ptile = (pn * p.Hc + poh0) * p.Wc + pow0;
            pending = true;
        }
        cbarrier();
        cstamp(st, wave, lane, stk);                                    // 2 + 2s: barrier passed
    }
    if (pending) {
#pragma unroll
        for (int g8 = 0; g8 < NG; ++g8) { pre_group(g8); fin_group(g8); }
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------------
bool cconv_eligible(const IgemmParams& p, int* kw_out, bool* rev_out) {
    if (p.so_h != 1 || p.so_w != 1 || p.sa_h != 1 || p.sa_w != 1 || p.fold) return false;
    const int ntaps = p.tap_begin[1];
    const int kw = ntaps == 25 ? 5 : (ntaps == 9 ? 3 : 0);
    if (!kw) return false;
    if (p.Hp[0] < 16 || p.Wp[0] < 16 || p.Hp[0] % 16 || p.Wp[0] % 16 || p.Cc % 32) return false;      // whole tiles only (branch-free epilogue); 16 | Hp covers the 8-row tiles too
    if (p.act == MV3D_ACT_TANH || p.gact == MV3D_ACT_TANH) return false;
    // buffer descriptors address the tensors with 32-bit byte offsets
    if ((int64_t)p.N * p.Ha * p.Wa * p.a_ld * 4 >= 0x7fffffff || (int64_t)p.N * p.Hc * p.Wc * std::max(p.c_ld, p.g_ld) * 4 >= 0x7fffffff || p.Ha > 16384) return false;
    if (p.Ka % 4 != 0 || p.a_ld % 4 != 0 || (reinterpret_cast<uintptr_t>(p.A) & 15)) return false;
    int dh_min = 127, dw_min = 127;
    for (int t = 0; t < ntaps; ++t) { dh_min = std::min<int>(dh_min, p.taps[t].dh); dw_min = std::min<int>(dw_min, p.taps[t].dw); }
    bool fwd = true, rev = true;
    for (int t = 0; t < ntaps; ++t) {
        const int r = t / kw, q = t % kw;
        const int dh = p.taps[t].dh - dh_min, dw = p.taps[t].dw - dw_min;
        if (dh != r || dw != q) fwd = false;
        if (dh != kw - 1 - r || dw != kw - 1 - q) rev = false;
    }
    if (!fwd && !rev) return false;
    *kw_out = kw; *rev_out = rev && !fwd;
    return true;
}

template <int KW, bool REV, bool HAS_G, int MT>
static int launch_cconv_t(const IgemmParams& p, const HconvExtra& x, dim3 grid, const uint4* wf, int ntiles, void* stream,
                          const char* who, double flops, double bytes) {
    constexpr int U = KW == 5 ? 5 : 3;
    constexpr int LA = KW == 5 ? CC_LA5 : 2;             // taps of look-ahead of the filter ring (16 registers each)
    constexpr int HPIX = (8 * MT - 1 + KW) * CC_HC;
    const size_t lds = (size_t)4 * HPIX * 64 + (size_t)((HPIX + 7) / 8) * 1024;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cconv_kernel<KW, REV, HAS_G, U, LA, MT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const char* name = intern_label("cconv<%s,%dpx,N32%s>", KW == 5 ? "5x5" : "3x3", 128 * MT, HAS_G ? ",gmask" : "");
    return dispatch(stream, OpInfo{name, flops, bytes}, [=](hipStream_t s) {
        cconv_kernel<KW, REV, HAS_G, U, LA, MT><<<grid, 512, lds, s>>>(p, x, wf, ntiles);
        return launched(who);
    });
}

// p must satisfy cconv_eligible; x carries tiles_h / tiles_w / dh_min / dw_min / chunks / dbg; wf = prepared filter
int launch_cconv(const IgemmParams& p, const HconvExtra& x, const void* wf, int ntiles, void* stream, const char* who, double flops, double bytes) {
    int kw = 0; bool rev = false;
    if (!cconv_eligible(p, &kw, &rev)) return fail(MV3D_E_UNSUPPORTED, "%s: not a pipelined-conv problem", who);
    HconvExtra xp = x;
    xp.n_tiles = p.N * x.tiles_h * x.tiles_w;
    const int ny = cdiv(p.Cc, 32);
    dim3 grid(std::min(xp.n_tiles, std::max(1, 256 / ny)), ny, 1);
    const uint4* w4 = reinterpret_cast<const uint4*>(wf);
    // x.TH = 16: 16 x 16 tiles (two pixel groups per multiplying wave); 8: 8 x 16 tiles, for layers with fewer than two large
    // tiles per CU (twice the workgroups, half the halo to stage before the first MFMA)
    if (x.TH != 16 && x.TH != 8) return fail(MV3D_E_INVAL, "%s: pipelined conv tile height %d", who, x.TH);
#define MV3D_CC(KW_, REV_, G_) (x.TH == 16 ? launch_cconv_t<KW_, REV_, G_, 2>(p, xp, grid, w4, ntiles, stream, who, flops, bytes) \
                                           : launch_cconv_t<KW_, REV_, G_, 1>(p, xp, grid, w4, ntiles, stream, who, flops, bytes))
    const bool g = p.gact != MV3D_ACT_NONE;
    if (kw == 5) return rev ? (g ? MV3D_CC(5, true, true) : MV3D_CC(5, true, false)) : (g ? MV3D_CC(5, false, true) : MV3D_CC(5, false, false));
    return rev ? (g ? MV3D_CC(3, true, true) : MV3D_CC(3, true, false)) : (g ? MV3D_CC(3, false, true) : MV3D_CC(3, false, false));
#undef MV3D_CC
}

}  // namespace mv3d

extern "C" int mv3d_debug_cconv_stamps(void* dst, size_t bytes) {
    if (!dst || bytes > sizeof(unsigned long long) * 256 * 8 * mv3d::CC_NSTAMP) return mv3d::fail(MV3D_E_INVAL, "mv3d_debug_cconv_stamps: bad buffer");
    hipError_t e = hipMemcpyFromSymbol(dst, HIP_SYMBOL(mv3d::c_stamps), bytes, 0, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return mv3d::fail(MV3D_E_HIP, "mv3d_debug_cconv_stamps: %s", hipGetErrorString(e));
    return MV3D_OK;
}
