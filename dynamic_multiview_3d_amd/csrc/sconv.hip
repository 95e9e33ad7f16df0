// Small-image split-bf16 convolution: the 8 x 8 and 4 x 4 layers around the bottleneck (e3 .. e4_0, d4 .. d3 of
// appearance_flow_model.py:93-97,111-114; conv2d_msra / deconv2d_msra forward and data gradient in every direction and stride).
//
// At batch 64 these layers are GEMMs with 1 k .. 4 k rows, 64 .. 256 columns and a reduction of 576 .. 2304: too few rows to fill
// the chip with (pixel tile x filter tile) workgroups that each walk the whole reduction chunk by chunk, which is why the generic
// kernel (bconv.hip) splits the channel chunks over workgroups and a second launch (igemm_splitk_epilogue) sums the partial
// outputs -- 28 launches per step whose cost is the launch chain, not the arithmetic.  Here the reduction is split INSIDE the
// workgroup instead:
//   * a workgroup owns 32 pixels of the output's phase grid (two 4 x 4 images, or four rows of an 8 x 8 image) x 32 filters;
//   * the input halo of those pixels is staged ONCE, for ALL channels (<= 150 KiB: 8-channel units of bf16 hi | lo, 32 bytes
//     each, in place of the 32 bytes of fp32 they came from; pixel records padded by 32 bytes so that the 32 pixels of an
//     A-fragment read fall on different banks), one barrier;
//   * the (tap, 16-channel) steps of the reduction are dealt round-robin to the four waves; a wave streams its filter
//     fragments from the prepared filter (bconv.hip: one coalesced 1 KiB line group per load) through an 8-deep register ring,
//     reads its A fragments from the planes and accumulates a partial 32 x 32 tile per output phase;
//   * the four partial tiles meet in LDS and are added in a fixed order (wave 0 + 1 + 2 + 3: reproducible), then bias,
//     activation and the activation-gradient mask are applied and the tile is stored -- no partial sums in HBM, no second launch.
// Stride-2 transposed convolutions (and the data gradient of stride-2 convolutions) are four output phases with their own tap
// sets: grid.z is the phase, a workgroup stages the (small) halo and walks its phase's taps only, so the accumulator index is a
// constant everywhere (one kernel with the phase selected per step by a branch kept the tile in flight through copies and made
// hipcc wait for the filter ring at every join).
//
// The step loop is straight-line code (SC_MAXSTEPS steps, left early): hipcc's vmcnt bookkeeping is exact inside a basic-block
// chain but conservative across a loop's back edge, where it drained the filter ring every R steps (one L2 round trip each
// time).  Inline-asm loads with hand-counted waits were tried and dropped: hipcc copied ring registers whose loads were still
// in flight at the branches around the waits (cdna_hip_programming.md 5.7, item 1).
#include "conv_common.h"
#include <algorithm>
#include <type_traits>
#include <utility>

namespace mv3d {

typedef __bf16 sbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 sbf16x4 __attribute__((ext_vector_type(4)));
constexpr int SC_MAXSTEPS = 48;      // reduction steps per wave the straight-line loop covers (9 taps x 320 channels / 16 / 4 = 45)

struct SconvParams {
    int TH;                 // phase-grid rows of one image slot covered by a tile
    int G;                  // image slots per tile (2: 4 x 4 phase grids, 1 otherwise)
    int tiles_h, tiles_w;   // row / column tiles per image (tiles_w: sconv4 only)
    int tw_shift;           // log2(phase-grid width)
    int slot_shift;         // log2(pixels per image slot) = log2(TH * Wp)
    int HRi, HC, PS;        // halo rows per slot, halo columns, bytes per halo pixel record (4 * Ka + 32)
    int dh_min, dw_min;
    int nk16;               // Ka / 16: reduction steps per tap
    unsigned inv_nk16;      // ceil(2^32 / nk16)
    int nsteps;             // taps * nk16
    int c8;                 // Ka / 8: 32-byte units per pixel
    unsigned inv_c8, inv_hc, inv_hri;
    int units;              // G * HRi * HC * c8
    int ntiles;             // 32-filter tiles of the prepared filter
    int HCp, HCe;           // s2conv: padded halo columns per LDS row, even columns of the halo (the odd ones follow them)
};

__device__ __forceinline__ void ssplit8(const float4& a, const float4& b, uint4& hi, uint4& lo) {
    sbf16x8 h, l;
    h[0] = (__bf16)a.x; h[1] = (__bf16)a.y; h[2] = (__bf16)a.z; h[3] = (__bf16)a.w;
    h[4] = (__bf16)b.x; h[5] = (__bf16)b.y; h[6] = (__bf16)b.z; h[7] = (__bf16)b.w;
    l[0] = (__bf16)(a.x - (float)h[0]); l[1] = (__bf16)(a.y - (float)h[1]); l[2] = (__bf16)(a.z - (float)h[2]); l[3] = (__bf16)(a.w - (float)h[3]);
    l[4] = (__bf16)(b.x - (float)h[4]); l[5] = (__bf16)(b.y - (float)h[5]); l[6] = (__bf16)(b.z - (float)h[6]); l[7] = (__bf16)(b.w - (float)h[7]);
    hi = __builtin_bit_cast(uint4, h);
    lo = __builtin_bit_cast(uint4, l);
}

template <class F, int... Is>
__device__ __forceinline__ void sc_chain(F& f, std::integer_sequence<int, Is...>) { (void)(f(std::integral_constant<int, Is>{}) && ...); }

__global__ __launch_bounds__(256, 2) void sconv_kernel(const IgemmParams p, const SconvParams x, const uint4* __restrict__ Wf) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int R = 8;                                  // filter ring: R - 1 steps of look-ahead
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    int b = blockIdx.x;
    const int th_i = b % x.tiles_h;
    const int n = (b / x.tiles_h) * x.G;                  // first image of the tile
    const int oh0 = th_i * x.TH;
    const int n0 = blockIdx.y * 32;
    const int ph = blockIdx.z;                            // output phase (0 for single-phase problems)
    const int tap_lo = p.tap_begin[ph], tap_hi = p.tap_begin[ph + 1];

    // ---- this wave's steps of the reduction: ks = ks0 + wave, + 4, ...  (ks = tap * nk16 + 16-channel group)
    constexpr int KS = 4;                                 // a wave takes every fourth step
    const int ks0 = tap_lo * x.nk16 + wave;
    const int nsteps = (tap_hi - tap_lo) * x.nk16;
    const int n_w = nsteps > wave ? (nsteps - wave + 3) >> 2 : 0;
    uint4 rhi[R], rlo[R];
    const int wf_bytes = (x.nsteps >> 1) * x.ntiles * 4096;       // nk16 is even: two steps per 32-channel chunk
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(Wf), 0, wf_bytes, 0x00020000);
    const int wlane = lane * 16;
    auto load_b = [&](uint4& hi, uint4& lo, int i) {
        i = i < n_w ? i : n_w - 1;                                   // look-ahead past the end re-reads the last step
        const int ks = ks0 + KS * i;
        const int so = ((ks >> 1) * x.ntiles + (int)blockIdx.y) * 4096 + (ks & 1) * 2048;
        hi = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, so, 0));
        lo = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane + 1024, so, 0));
    };
    if (n_w > 0) {
#pragma unroll
        for (int u = 0; u < R - 1; ++u) load_b(rhi[u], rlo[u], u);  // in flight under the halo staging
    }

    // ---- halo of the tile, all channels: fp32 global -> bf16 hi | lo units in LDS
    {
        const int ih0 = oh0 * p.sa_h + x.dh_min, iw0 = x.dw_min;
        constexpr int UB = 6;                                       // units (2 x 16 bytes) per thread and round
        // first round straight-line (most halos fit it): the loads go out without the wait for the filter ring that hipcc puts
        // in front of a loop header; further rounds for larger halos
        auto stage_round = [&](int base) {
            float4 v[UB][2];
            int lofs[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int idx = base + u * 256 + tid;
                const int pix = (int)__umulhi((unsigned)idx, x.inv_c8), cu = idx - pix * x.c8;
                const int hrv = (int)__umulhi((unsigned)pix, x.inv_hc), hc = pix - hrv * x.HC;
                const int g = x.G > 1 ? (int)__umulhi((unsigned)hrv, x.inv_hri) : 0;
                const int hr = hrv - g * x.HRi;
                const int ih = ih0 + hr, iw = iw0 + hc;
                const bool ok = idx < x.units && n + g < p.N && (unsigned)ih < (unsigned)p.Ha && (unsigned)iw < (unsigned)p.Wa;
                const float* src = ok ? p.A + (int64_t)(((n + g) * p.Ha + ih) * p.Wa + iw) * p.a_ld + cu * 8 : p.A;
                const float4 t0 = reinterpret_cast<const float4*>(src)[0], t1 = reinterpret_cast<const float4*>(src)[1];
                v[u][0] = ok ? t0 : make_float4(0.f, 0.f, 0.f, 0.f);
                v[u][1] = ok ? t1 : make_float4(0.f, 0.f, 0.f, 0.f);
                lofs[u] = idx < x.units ? pix * x.PS + cu * 32 : -1;
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                if (lofs[u] >= 0) {
                    uint4 hi, lo;
                    ssplit8(v[u][0], v[u][1], hi, lo);
                    *reinterpret_cast<uint4*>(lds + lofs[u]) = hi;
                    *reinterpret_cast<uint4*>(lds + lofs[u] + 16) = lo;
                }
            }
        };
        stage_round(0);
        for (int base = 256 * UB; base < x.units; base += 256 * UB) stage_round(base);
    }
    __syncthreads();

    // ---- A-operand addressing: pixel li of the tile, channel unit lh of a step; the tap part comes from lane `tap` of lane_off
    int a_base;
    {
        const int g = li >> x.slot_shift, pr = li & ((1 << x.slot_shift) - 1);
        const int tr = pr >> x.tw_shift, tc = pr & ((1 << x.tw_shift) - 1);
        a_base = ((g * x.HRi + tr * p.sa_h) * x.HC + tc * p.sa_w) * x.PS + lh * 32;
    }
    int lane_off;
    {
        const IgemmTap tap = p.taps[lane < 36 ? lane : 0];
        lane_off = ((tap.dh - x.dh_min) * x.HC + (tap.dw - x.dw_min)) * x.PS;
    }
    auto step_off = [&](int i) {                                  // LDS byte offset (tap + channel group) of step i of this wave
        const int ks = ks0 + KS * i;
        const int t = (int)__umulhi((unsigned)ks, x.inv_nk16);
        return __builtin_amdgcn_readlane(lane_off, t) + (ks - t * x.nk16) * 64;
    };

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    uint4 ab[2][2];                                                 // [buffer][hi, lo]
    auto read_a = [&](int buf, int aoff) {
        const unsigned char* ap = lds + a_base + aoff;
        ab[buf][0] = *reinterpret_cast<const uint4*>(ap);
        ab[buf][1] = *reinterpret_cast<const uint4*>(ap + 16);
    };
    if (n_w > 0) read_a(0, step_off(0));
    // step I of this wave; false = past the end (the && fold below leaves the chain there)
    auto step = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if (i >= n_w) return false;
        load_b(rhi[(i + R - 1) % R], rlo[(i + R - 1) % R], i + R - 1);
        if (i + 1 < n_w) read_a((i + 1) & 1, step_off(i + 1));
        __builtin_amdgcn_sched_barrier(0);                        // keep the look-ahead loads in front of this step's MFMAs
        const sbf16x8 ah = __builtin_bit_cast(sbf16x8, ab[i & 1][0]), al = __builtin_bit_cast(sbf16x8, ab[i & 1][1]);
        const sbf16x8 bh = __builtin_bit_cast(sbf16x8, rhi[i % R]), bl = __builtin_bit_cast(sbf16x8, rlo[i % R]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
        return true;
    };
    sc_chain(step, std::make_integer_sequence<int, SC_MAXSTEPS>{});

    const int col = n0 + li;
    const float bias = (p.bias && col < p.Cc) ? p.bias[col] : 0.f;
    const int phh = ph / p.so_w, phw = ph % p.so_w;
    // ---- the four waves' partial tiles: exchanged through LDS, added in the order wave 0, 1, 2, 3
    __syncthreads();                                                // every wave is past its last halo read
    float* const xch = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int r = 0; r < 16; ++r) xch[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    // wave w finishes accumulator registers 4 w .. 4 w + 3 (pixels 8 w + 4 lh + j of the tile)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = wave * 4 + j;
        const float v = ((xch[(0 * 16 + r) * 64 + lane] + xch[(1 * 16 + r) * 64 + lane]) + xch[(2 * 16 + r) * 64 + lane]) + xch[(3 * 16 + r) * 64 + lane];
        const int q = j + 8 * wave + 4 * lh;
        const int g = q >> x.slot_shift, pr = q & ((1 << x.slot_shift) - 1);
        const int tr = pr >> x.tw_shift, tc = pr & ((1 << x.tw_shift) - 1);
        if (n + g < p.N && col < p.Cc) {
            const int64_t pix = (int64_t)((n + g) * p.Hc + (oh0 + tr) * p.so_h + phh) * p.Wc + tc * p.so_w + phw;
            float o = act_apply(v + bias, p.act, p.leak);
            if (p.gact != MV3D_ACT_NONE) o *= act_grad_from_out(p.gref[pix * p.g_ld + col], p.gact, p.gleak);
            p.Out[pix * p.c_ld + col] = o;
        }
    }
}

// Stride-2 transposed convolution (and the data gradient of a stride-2 convolution), ALL FOUR output phases from one halo:
// a workgroup owns an 8 x 16-pixel tile of the output's phase grid x 32 filters, each wave 32 of its pixels (two tile rows) for the
// WHOLE reduction -- no exchange; the halo of the tile (10 x 18 pixels x all channels, <= 52 KiB: three workgroups per CU, whose
// staging and multiplying phases overlap) is staged once, the filter fragments stream from L2.  (A first version gave every
// phase its own workgroup and so staged the halo four times: the staging weighed as much as the products, 110 TFLOP/s against
// 177 for the fused form.)  Here the taps of the four phases are one straight-line chain of KSZ^2 x NK16
// steps (taps are listed phase by phase: 4 + 6 + 6 + 9 for 5 x 5 with SAME padding 1, 4 + 2 + 2 + 1 for 3 x 3; NK16 = channels / 16 -- both
// template parameters, so every phase boundary is a compile-time position in the chain): at a boundary the wave stores the
// finished 32 x 32 tile of that phase and clears its ONE accumulator.  The filter ring runs across the boundaries; the saved
// outputs behind a gradient mask are requested at the START of their phase, so that they are older than the ring's look-ahead
// loads and have landed long before the phase's epilogue (vmcnt retires in order).
template <int KSZ> struct PhaseTaps;
template <> struct PhaseTaps<5> { static constexpr int b[5] = {0, 4, 10, 16, 25}; };
template <> struct PhaseTaps<3> { static constexpr int b[5] = {0, 4, 6, 8, 9}; };

// TR = 8: as described.  TR = 4 (layers with fewer than 256 8 x 16 tiles): 4 x 16 tiles, waves 0 / 1 own the two 32-pixel halves
// for phases 0 and 3 (4 + 9 of the 25 taps), waves 2 / 3 the same pixels for phases 1 and 2 (6 + 6): twice the workgroups, half the chain.
template <int KSZ, int NK16, bool HAS_G, int TR>
__global__ __launch_bounds__(256, 2) void sconv4_kernel(const IgemmParams p, const SconvParams x, const uint4* __restrict__ Wf) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int R = 8;
    constexpr int NSTEP = KSZ * KSZ * NK16;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    int b = blockIdx.x;
    const int tw_i = b % x.tiles_w; b /= x.tiles_w;
    const int th_i = b % x.tiles_h;
    const int n = b / x.tiles_h;
    const int oh0 = th_i * TR, ow0 = tw_i * 16;
    const int pgw = TR == 8 ? wave : (wave & 1);                      // pixel group (two tile rows) of this wave
    const int half = TR == 8 ? 0 : (wave >> 1);                        // TR = 4: which pair of phases
    const int n0 = blockIdx.y * 32;

    uint4 rhi[R], rlo[R];
    const int wf_bytes = (NSTEP >> 1) * x.ntiles * 4096;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(Wf), 0, wf_bytes, 0x00020000);
    const int wlane = lane * 16;
    auto load_b = [&](uint4& hi, uint4& lo, int ks) {
        ks = ks < NSTEP ? ks : NSTEP - 1;
        const int so = ((ks >> 1) * x.ntiles + (int)blockIdx.y) * 4096 + (ks & 1) * 2048;
        hi = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, so, 0));
        lo = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane + 1024, so, 0));
    };
    // global step (tap * NK16 + channel group) of local step j of this wave's chain
    constexpr int B1 = PhaseTaps<KSZ>::b[1] * NK16, B3 = PhaseTaps<KSZ>::b[3] * NK16;
    auto gstep = [&](int j) { return TR == 8 ? j : (half == 0 ? (j < B1 ? j : j + (B3 - B1)) : j + B1); };
#pragma unroll
    for (int u = 0; u < R - 1; ++u) load_b(rhi[u], rlo[u], gstep(u));

    // ---- halo of the tile, all channels: fp32 global -> bf16 hi | lo units in LDS
    {
        const int ih0 = oh0 + x.dh_min, iw0 = ow0 + x.dw_min;
        constexpr int UB = 6;
        // first round straight-line (most halos fit it): the loads go out without the wait for the filter ring that hipcc puts
        // in front of a loop header; further rounds for larger halos
        auto stage_round = [&](int base) {
            float4 v[UB][2];
            int lofs[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int idx = base + u * 256 + tid;
                const int pix = (int)__umulhi((unsigned)idx, x.inv_c8), cu = idx - pix * x.c8;
                const int hr = (int)__umulhi((unsigned)pix, x.inv_hc), hc = pix - hr * x.HC;
                const int ih = ih0 + hr, iw = iw0 + hc;
                const bool ok = idx < x.units && (unsigned)ih < (unsigned)p.Ha && (unsigned)iw < (unsigned)p.Wa;
                const float* src = ok ? p.A + (int64_t)((n * p.Ha + ih) * p.Wa + iw) * p.a_ld + cu * 8 : p.A;
                const float4 t0 = reinterpret_cast<const float4*>(src)[0], t1 = reinterpret_cast<const float4*>(src)[1];
                v[u][0] = ok ? t0 : make_float4(0.f, 0.f, 0.f, 0.f);
                v[u][1] = ok ? t1 : make_float4(0.f, 0.f, 0.f, 0.f);
                lofs[u] = idx < x.units ? pix * x.PS + cu * 32 : -1;
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                if (lofs[u] >= 0) {
                    uint4 hi, lo;
                    ssplit8(v[u][0], v[u][1], hi, lo);
                    *reinterpret_cast<uint4*>(lds + lofs[u]) = hi;
                    *reinterpret_cast<uint4*>(lds + lofs[u] + 16) = lo;
                }
            }
        };
        stage_round(0);
        for (int base = 256 * UB; base < x.units; base += 256 * UB) stage_round(base);
    }
    __syncthreads();

    const int tr0 = 2 * pgw + (li >> 4), tc0 = li & 15;                   // this lane's pixel of the tile (A operand)
    const int a_base = (tr0 * x.HC + tc0) * x.PS + lh * 32;
    int lane_off;
    {
        const IgemmTap tap = p.taps[lane < KSZ * KSZ ? lane : 0];
        lane_off = ((tap.dh - x.dh_min) * x.HC + (tap.dw - x.dw_min)) * x.PS;
    }
    f32x16 acc;
    uint4 ab[2][2];
    auto read_a = [&](int buf, int ks) {
        const unsigned char* ap = lds + a_base + __builtin_amdgcn_readlane(lane_off, ks / NK16) + (ks % NK16) * 64;
        ab[buf][0] = *reinterpret_cast<const uint4*>(ap);
        ab[buf][1] = *reinterpret_cast<const uint4*>(ap + 16);
    };
    const int col = n0 + li;
    const float bias = (p.bias && col < p.Cc) ? p.bias[col] : 0.f;
    const float c1 = p.act == MV3D_ACT_NONE ? 1.f : (p.act == MV3D_ACT_LRELU ? 0.5f * (1.0f + p.leak) : 0.5f);
    const float c2 = p.act == MV3D_ACT_NONE ? 0.f : (p.act == MV3D_ACT_LRELU ? 0.5f * (1.0f - p.leak) : 0.5f);
    const bool is_relu = p.act == MV3D_ACT_RELU;
    const float g1 = p.gact == MV3D_ACT_LRELU ? 0.5f * (1.0f + p.gleak) : 0.5f;
    const float g2 = p.gact == MV3D_ACT_LRELU ? 0.5f * (1.0f - p.gleak) : 0.5f;
    const bool g_relu = p.gact == MV3D_ACT_RELU;
    // output pixel of accumulator register r of phase (phh, phw), minus the lane-independent part
    auto out_pix = [&](int r, int phh, int phw) {
        const int q = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int tr = 2 * pgw + (q >> 4), tc = q & 15;
        return ((n * p.Hc + (oh0 + tr) * 2 + phh) * p.Wc + (ow0 + tc) * 2 + phw);
    };
    float gm[16];
    // HALF = -1: all four phases (TR = 8); 0: phases 0 and 3; 1: phases 1 and 2.  J = local step of the chain, I = global step.
    auto run = [&](auto hsel) {
        constexpr int HALF = decltype(hsel)::value;
        constexpr int NLOC = HALF < 0 ? NSTEP : (HALF == 0 ? B1 + (NSTEP - B3) : B3 - B1);
        constexpr int OFF1 = HALF < 0 ? 0 : (HALF == 0 ? 0 : B1);
        read_a(0, OFF1);
        auto step = [&](auto ic) {
            constexpr int j = decltype(ic)::value;
            constexpr int i = HALF == 0 ? (j < B1 ? j : j + (B3 - B1)) : j + OFF1;
            constexpr int inext = HALF == 0 ? (j + 1 < B1 ? j + 1 : j + 1 + (B3 - B1)) : j + 1 + OFF1;
            constexpr int tap = i / NK16;
            constexpr int ph = tap < PhaseTaps<KSZ>::b[1] ? 0 : (tap < PhaseTaps<KSZ>::b[2] ? 1 : (tap < PhaseTaps<KSZ>::b[3] ? 2 : 3));
            constexpr bool first = i == PhaseTaps<KSZ>::b[ph] * NK16, last = i + 1 == PhaseTaps<KSZ>::b[ph + 1] * NK16;
            if constexpr (first) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                if constexpr (HAS_G) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) gm[r] = col < p.Cc ? p.gref[(int64_t)out_pix(r, ph >> 1, ph & 1) * p.g_ld + col] : 0.f;
                }
            }
            {
                constexpr int jl = j + R - 1;                          // look-ahead: local step jl (past the end: the last step again)
                constexpr int jc = jl < NLOC ? jl : NLOC - 1;
                constexpr int il = HALF == 0 ? (jc < B1 ? jc : jc + (B3 - B1)) : jc + OFF1;
                load_b(rhi[(j + R - 1) % R], rlo[(j + R - 1) % R], il);
            }
            if constexpr (j + 1 < NLOC) read_a((j + 1) & 1, inext);
            __builtin_amdgcn_sched_barrier(0);
            const sbf16x8 ah = __builtin_bit_cast(sbf16x8, ab[j & 1][0]), al = __builtin_bit_cast(sbf16x8, ab[j & 1][1]);
            const sbf16x8 bh = __builtin_bit_cast(sbf16x8, rhi[j % R]), bl = __builtin_bit_cast(sbf16x8, rlo[j % R]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
            if constexpr (last) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float xv = acc[r] + bias;
                    float y = __fadd_rn(__fmul_rn(c1, xv), __fmul_rn(c2, fabsf(xv)));
                    y = (is_relu && xv < 0.0f) ? -0.0f : y;
                    if constexpr (HAS_G) {
                        const float go = gm[r];
                        const bool neg = g_relu ? (__float_as_uint(go) >> 31) != 0 : go < 0.0f;
                        y *= g1 + g2 * (go > 0.0f ? 1.0f : (neg ? -1.0f : 0.0f));
                    }
                    if (col < p.Cc) p.Out[(int64_t)out_pix(r, ph >> 1, ph & 1) * p.c_ld + col] = y;
                }
            }
            return true;
        };
        sc_chain(step, std::make_integer_sequence<int, NLOC>{});
    };
    if constexpr (TR == 8) run(std::integral_constant<int, -1>{});
    else { if (half == 0) run(std::integral_constant<int, 0>{}); else run(std::integral_constant<int, 1>{}); }
}

// Stride-2 convolution forward (and the data gradient of a stride-2 transposed convolution) on images of whole 4 x 16 output
// tiles: e1 / e2 forward, d1 / d2 data gradient (appearance_flow_model.py:89-91,121-123 -- 32 input channels each).
// A workgroup owns a 4 x 16 tile of the OUTPUT; the input halo of the tile (11 x 35 pixels for 5 x 5) is staged once for all
// channels in ONE round of loads, with its even and odd COLUMNS apart, so that the 16 output pixels of a tile row read 16
// consecutive pixel records for every tap (column 2 c + kx is record c + (kx >> 1) of parity kx & 1) and the A-fragment
// reads stay conflict-free at the record pitch of sconv4; the row pitch is padded until two halo rows (one tile row) are 64
// bytes apart modulo the 256-byte bank row, as one halo row is there.  64 KiB of LDS: two workgroups per CU, one staging or
// storing while the other multiplies.  Waves 0 / 1 own the two 32-pixel halves of the tile, and so do waves 2 / 3, for
//   KSPLIT = false: the second 32-filter tile (64 filters per workgroup, whole reduction per wave);
//   KSPLIT = true:  the second 16-channel half of every tap (32 filters per workgroup; the pair's partial tiles meet in LDS,
//                   channels 0-15 + channels 16-31, and each wave finishes eight of the sixteen accumulator registers).
// The taps are one straight-line chain with the filter ring running through it.  (The generic kernel, bconvu, stages a
// 64-pixel tile per 128-thread workgroup in three rounds of loads and multiplies behind a barrier: 84 - 96 TFLOP/s.)
// S = 1 (round 3, later): the same kernel without the column split takes the stride-1 layers on 32 x 32 and 16 x 16 maps (e1_0,
// e2_0, d2_0, d3_0, forward and data gradient) from the pipelined kernel, whose persistent one-workgroup-per-CU pipeline has an
// 11 k-cycle prologue that layers of one or two stages per CU live in.  NK16 = channels / 16 (2 or 4).
// MT = pixel groups per wave: 1 = 4 x 16 tile; 2 = 8 x 16 tile (each filter fragment feeds two products: the 64 x 64 layers, stride 1)
template <int NTAPS, int S, int NK16, bool KSPLIT, bool HAS_G, int MT>
__global__ __launch_bounds__(256, 2) void s2conv_kernel(const IgemmParams p, const SconvParams x, const uint4* __restrict__ Wf) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    constexpr int R = 8;
    constexpr int KG = KSPLIT ? NK16 / 2 : NK16;                    // 16-channel groups of a tap this wave multiplies
    constexpr int NSTEP = NTAPS * KG;                               // this wave's steps
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int pg = wave & 1, half = wave >> 1;                      // pixel group; filter tile (or channel half) of this wave
    int b = blockIdx.x;
    const int tw_i = b % x.tiles_w; b /= x.tiles_w;
    const int th_i = b % x.tiles_h;
    const int n = b / x.tiles_h;
    const int oh0 = th_i * 4 * MT, ow0 = tw_i * 16;
    const int ytile = KSPLIT ? (int)blockIdx.y : (int)blockIdx.y * 2 + half;
    const int n0 = ytile * 32;

    uint4 rhi[R], rlo[R];
    const int wf_bytes = NTAPS * (NK16 / 2) * x.ntiles * 4096;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(Wf), 0, wf_bytes, 0x00020000);
    const int wlane = lane * 16;
    const int kofs = KSPLIT ? half * KG : 0;                        // first 16-channel group of this wave
    auto load_b = [&](uint4& hi, uint4& lo, int i) {               // step i of this wave: tap i / KG, group kofs + i % KG
        i = i < NSTEP ? i : NSTEP - 1;
        const int ks = (i / KG) * NK16 + (i % KG) + kofs;
        const int so = ((ks >> 1) * x.ntiles + ytile) * 4096 + (ks & 1) * 2048;
        hi = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane, so, 0));
        lo = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wlane + 1024, so, 0));
    };
#pragma unroll
    for (int u = 0; u < R - 1; ++u) load_b(rhi[u], rlo[u], u);

    // ---- halo of the tile, all channels: fp32 global -> bf16 hi | lo units in LDS, even columns first
    {
        const int ih0 = oh0 * S + x.dh_min, iw0 = ow0 * S + x.dw_min;
        constexpr int UB = 7;
        // first round straight-line (most halos fit it): the loads go out without the wait for the filter ring that hipcc puts
        // in front of a loop header; further rounds for larger halos
        auto stage_round = [&](int base) {
            float4 v[UB][2];
            int lofs[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int idx = base + u * 256 + tid;
                const int pix = (int)__umulhi((unsigned)idx, x.inv_c8), cu = idx - pix * x.c8;
                const int hr = (int)__umulhi((unsigned)pix, x.inv_hc), hc = pix - hr * x.HC;
                const int ih = ih0 + hr, iw = iw0 + hc;
                const bool ok = idx < x.units && (unsigned)ih < (unsigned)p.Ha && (unsigned)iw < (unsigned)p.Wa;
                const float* src = ok ? p.A + (int64_t)((n * p.Ha + ih) * p.Wa + iw) * p.a_ld + cu * 8 : p.A;
                const float4 t0 = reinterpret_cast<const float4*>(src)[0], t1 = reinterpret_cast<const float4*>(src)[1];
                v[u][0] = ok ? t0 : make_float4(0.f, 0.f, 0.f, 0.f);
                v[u][1] = ok ? t1 : make_float4(0.f, 0.f, 0.f, 0.f);
                const int cs = S == 2 ? (hc & 1) * x.HCe + (hc >> 1) : hc;
                lofs[u] = idx < x.units ? (hr * x.HCp + cs) * x.PS + cu * 32 : -1;
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                if (lofs[u] >= 0) {
                    uint4 hi, lo;
                    ssplit8(v[u][0], v[u][1], hi, lo);
                    *reinterpret_cast<uint4*>(lds + lofs[u]) = hi;
                    *reinterpret_cast<uint4*>(lds + lofs[u] + 16) = lo;
                }
            }
        };
        stage_round(0);
        for (int base = 256 * UB; base < x.units; base += 256 * UB) stage_round(base);
    }
    __syncthreads();

    const int tr0 = 2 * pg + (li >> 4), tc0 = li & 15;                    // this lane's pixel of the 4 x 16 tile (A operand)
    const int a_base = (S * tr0 * x.HCp + tc0) * x.PS + lh * 32 + kofs * 64;
    const int a_mt = 4 * S * x.HCp * x.PS;                          // second pixel group: four tile rows further down
    int lane_off;
    {
        const IgemmTap tap = p.taps[lane < NTAPS ? lane : 0];
        const int dc = tap.dw - x.dw_min;
        lane_off = ((tap.dh - x.dh_min) * x.HCp + (S == 2 ? (dc & 1) * x.HCe + (dc >> 1) : dc)) * x.PS;
    }
    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
    uint4 ab[2][MT][2];
    auto read_a = [&](int buf, int i) {
        const unsigned char* ap = lds + a_base + __builtin_amdgcn_readlane(lane_off, i / KG) + (i % KG) * 64;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            ab[buf][mt][0] = *reinterpret_cast<const uint4*>(ap + mt * a_mt);
            ab[buf][mt][1] = *reinterpret_cast<const uint4*>(ap + mt * a_mt + 16);
        }
    };
    const int col = n0 + li;
    const float bias = (p.bias && col < p.Cc) ? p.bias[col] : 0.f;
    const float c1 = p.act == MV3D_ACT_NONE ? 1.f : (p.act == MV3D_ACT_LRELU ? 0.5f * (1.0f + p.leak) : 0.5f);
    const float c2 = p.act == MV3D_ACT_NONE ? 0.f : (p.act == MV3D_ACT_LRELU ? 0.5f * (1.0f - p.leak) : 0.5f);
    const bool is_relu = p.act == MV3D_ACT_RELU;
    const float g1 = p.gact == MV3D_ACT_LRELU ? 0.5f * (1.0f + p.gleak) : 0.5f;
    const float g2 = p.gact == MV3D_ACT_LRELU ? 0.5f * (1.0f - p.gleak) : 0.5f;
    const bool g_relu = p.gact == MV3D_ACT_RELU;
    auto out_pix = [&](int mt, int r) {
        const int q = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int tr = 4 * mt + 2 * pg + (q >> 4), tc = q & 15;
        return (n * p.Hc + oh0 + tr) * p.Wc + ow0 + tc;
    };
    // the accumulator registers this wave finishes: all sixteen, or (KSPLIT) eight -- 0-7 by the pair's first wave, 8-15 by its second
    constexpr int NFIN = KSPLIT ? 8 : 16;
    const int rbase = KSPLIT ? half * 8 : 0;
    float gm[MT][NFIN];
    if constexpr (HAS_G) {          // requested before the ring's look-ahead loads of the chain: landed long before the epilogue
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < NFIN; ++j) gm[mt][j] = col < p.Cc ? p.gref[(int64_t)out_pix(mt, rbase + j) * p.g_ld + col] : 0.f;
    }
    read_a(0, 0);
    auto step = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        load_b(rhi[(i + R - 1) % R], rlo[(i + R - 1) % R], i + R - 1);
        if constexpr (i + 1 < NSTEP) read_a((i + 1) & 1, i + 1);
        __builtin_amdgcn_sched_barrier(0);
        const sbf16x8 bh = __builtin_bit_cast(sbf16x8, rhi[i % R]), bl = __builtin_bit_cast(sbf16x8, rlo[i % R]);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const sbf16x8 ah = __builtin_bit_cast(sbf16x8, ab[i & 1][mt][0]), al = __builtin_bit_cast(sbf16x8, ab[i & 1][mt][1]);
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[mt], 0, 0, 0);
        }
        return true;
    };
    sc_chain(step, std::make_integer_sequence<int, NSTEP>{});
    float fin[MT][NFIN];
    if constexpr (KSPLIT) {
        // the pair's partial tiles: each wave hands over the eight registers the other one finishes; lower channel half + upper channel half
        __syncthreads();                                            // every wave is past its last halo read
        float* const xch = reinterpret_cast<float*>(lds);           // [pixel group][wave][8][64]
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (half == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) xch[((mt * 4 + wave) * 8 + j) * 64 + lane] = acc[mt][8 + j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) xch[((mt * 4 + wave) * 8 + j) * 64 + lane] = acc[mt][j];
            }
        }
        __syncthreads();
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            if (half == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) fin[mt][j] = acc[mt][j] + xch[((mt * 4 + wave + 2) * 8 + j) * 64 + lane];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) fin[mt][j] = xch[((mt * 4 + wave - 2) * 8 + j) * 64 + lane] + acc[mt][8 + j];
            }
        }
    } else {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int j = 0; j < 16; ++j) fin[mt][j] = acc[mt][j];
    }
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int j = 0; j < NFIN; ++j) {
            const float xv = fin[mt][j] + bias;
            float o = __fadd_rn(__fmul_rn(c1, xv), __fmul_rn(c2, fabsf(xv)));
            o = (is_relu && xv < 0.0f) ? -0.0f : o;
            if constexpr (HAS_G) {
                const float go = gm[mt][j];
                const bool neg = g_relu ? (__float_as_uint(go) >> 31) != 0 : go < 0.0f;
                o *= g1 + g2 * (go > 0.0f ? 1.0f : (neg ? -1.0f : 0.0f));
            }
            if (col < p.Cc) p.Out[(int64_t)out_pix(mt, rbase + j) * p.c_ld + col] = o;
        }
}

// ---- host side -------------------------------------------------------------------------------------------------------------
static unsigned inv32(int d) { return (unsigned)((((uint64_t)1 << 32) + d - 1) / (uint64_t)d); }

// returns MV3D_OK after dispatching, 1 when the problem is not one of this kernel's
int try_sconv(const IgemmParams& p, void* ws, size_t ws_bytes, void* stream, const char* who, double flops, double bytes) {
    const int nph = p.so_h * p.so_w;
    if (nph != 1 && nph != 4) return 1;
    if (p.fold || p.Ka % 16 != 0 || p.Ka < 32 || p.a_ld % 4 != 0 || (reinterpret_cast<uintptr_t>(p.A) & 15) || p.Cc < 16) return 1;
    const int Hp = p.Hp[0], Wp = p.Wp[0];
    if (nph == 4 && (p.so_h != 2 || p.so_w != 2 || p.Hp[1] != Hp || p.Wp[1] != Wp)) return 1;
    const int ipx = Hp * Wp;
    // sconv4_kernel: 4-phase problems on phase grids of whole 8 x 16 tiles (the small-image form below takes grids up to 64 pixels)
    // (4 x 16 tiles where the layer has fewer than 256 workgroups of 8 x 16)
    const bool tile4 = nph == 4 && Hp % 8 == 0 && Wp % 16 == 0 && p.N * (Hp / 8) * (Wp / 16) * cdiv(p.Cc, 32) < 256 && !getenv("MV3D_S4_NO_TR4");
    const bool tile = nph == 4 && ipx > 64 && Hp % 8 == 0 && Wp % 16 == 0 && p.sa_h == 1 && p.sa_w == 1 && !(disabled_paths() & 134217728);
    if (!tile && ((ipx & (ipx - 1)) || (Wp & (Wp - 1)) || Wp < 4 || Wp > 32 || ipx < 16 || ipx > 64)) return 1;
    const int ntaps = p.tap_begin[nph];
    if (ntaps < 2 || ntaps > 36) return 1;
    for (int ph = 0; ph < nph; ++ph) if (p.tap_begin[ph + 1] == p.tap_begin[ph]) return 1;
    int dh_min = 127, dh_max = -127, dw_min = 127, dw_max = -127;
    for (int t = 0; t < ntaps; ++t) {
        dh_min = std::min<int>(dh_min, p.taps[t].dh); dh_max = std::max<int>(dh_max, p.taps[t].dh);
        dw_min = std::min<int>(dw_min, p.taps[t].dw); dw_max = std::max<int>(dw_max, p.taps[t].dw);
    }
    SconvParams x = {};
    x.tiles_w = 1;
    if (tile) { x.G = 1; x.TH = tile4 ? 4 : 8; x.tiles_h = Hp / x.TH; x.tiles_w = Wp / 16; }
    else if (ipx == 16) { x.G = 2; x.TH = Hp; x.tiles_h = 1; }
    else { x.G = 1; x.TH = 32 / Wp; x.tiles_h = Hp / x.TH; }
    if (x.TH < 1 || x.TH * x.tiles_h != Hp) return 1;
    x.tw_shift = 0; while ((1 << x.tw_shift) < Wp) ++x.tw_shift;
    x.slot_shift = 0; while ((1 << x.slot_shift) < x.TH * Wp) ++x.slot_shift;
    if (!tile && (x.G << x.slot_shift) != 32) return 1;
    x.HRi = (x.TH - 1) * p.sa_h + (dh_max - dh_min + 1);
    x.HC = ((tile ? 16 : Wp) - 1) * p.sa_w + (dw_max - dw_min + 1);
    x.PS = p.Ka * 4 + 32;
    x.dh_min = dh_min; x.dw_min = dw_min;
    x.nk16 = p.Ka / 16;
    if (x.nk16 & 1) return 1;                                       // the prepared filter is laid out in 32-channel chunks
    x.inv_nk16 = inv32(x.nk16);
    x.nsteps = ntaps * x.nk16;
    x.c8 = p.Ka / 8; x.inv_c8 = inv32(x.c8); x.inv_hc = inv32(x.HC); x.inv_hri = inv32(x.HRi);
    x.units = x.G * x.HRi * x.HC * x.c8;
    if (x.units >= 65536) return 1;                                 // the multiply-high divisions are exact below 2^16
    for (int ph = 0; ph < nph; ++ph) if (cdiv((p.tap_begin[ph + 1] - p.tap_begin[ph]) * x.nk16, tile ? 1 : 4) > SC_MAXSTEPS) return 1;
    const size_t halo = (size_t)x.G * x.HRi * x.HC * x.PS;
    const size_t lds = tile ? halo : std::max(halo, (size_t)16 * 1024);
    if (lds > 160 * 1024 || (tile && lds > 52 * 1024)) return 1;      // TILE: three workgroups per CU or the generic kernel
    int rc = MV3D_OK;
    const uint4* wf = bconv_get_filter(p, ws, ws_bytes, stream, &x.ntiles, &rc);
    if (!wf) return rc;
    const dim3 grid(cdiv(p.N, x.G) * x.tiles_h * x.tiles_w, cdiv(p.Cc, 32), nph);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sconv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (getenv("MV3D_TRACE"))
        fprintf(stderr, "[mv3d] %-22s sconv<%dph> N=%d in %dx%dx%d (stride %d) out %dx%dx%d taps=%d tile %dx%dx%d halo %dx%d lds=%zu grid=%dx%dx%d %.2f GFLOP\n",
                who, nph, p.N, p.Ha, p.Wa, p.Ka, p.sa_h, p.Hc, p.Wc, p.Cc, ntaps, x.G, x.TH, Wp, x.G * x.HRi, x.HC, lds, grid.x, grid.y, grid.z, flops * 1e-9);
    const IgemmParams pc = p;
    if (tile && (ntaps == 25 || ntaps == 9) && (x.nk16 == 2 || x.nk16 == 4) &&
        p.act != MV3D_ACT_TANH && p.gact != MV3D_ACT_TANH && (p.gact == MV3D_ACT_NONE || p.gref)) {
        // all four phases from one halo (sconv4_kernel): the tap list must be the phase-by-phase list the kernel is compiled for
        const int* tb = ntaps == 25 ? PhaseTaps<5>::b : PhaseTaps<3>::b;
        bool same = true;
        for (int ph = 0; ph <= 4; ++ph) same = same && p.tap_begin[ph] == tb[ph];
        if (same) {
            const dim3 g4(grid.x, grid.y, 1);
            const bool hg = p.gact != MV3D_ACT_NONE;
            const int nk = x.nk16;
            const bool k5 = ntaps == 25;
            static bool attr4 = false;
            if (!attr4) {
#define MV3D_S4_ATTR(K_, N_, G_) do { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sconv4_kernel<K_, N_, G_, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
                                      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&sconv4_kernel<K_, N_, G_, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); } while (0)
                MV3D_S4_ATTR(5, 2, false); MV3D_S4_ATTR(5, 2, true); MV3D_S4_ATTR(5, 4, false); MV3D_S4_ATTR(5, 4, true);
                MV3D_S4_ATTR(3, 2, false); MV3D_S4_ATTR(3, 2, true); MV3D_S4_ATTR(3, 4, false); MV3D_S4_ATTR(3, 4, true);
#undef MV3D_S4_ATTR
                attr4 = true;
            }
            const bool t4 = x.TH == 4;
            const char* name = intern_label("sconv4<%s,C%d%s%s>", k5 ? "5x5" : "3x3", nk * 16, t4 ? ",64px" : "", hg ? ",gmask" : "");
            return dispatch(stream, OpInfo{name, flops, bytes}, [=](hipStream_t s) {
#define MV3D_S4(K_, N_) do { if (t4) { if (hg) sconv4_kernel<K_, N_, true, 4><<<g4, 256, lds, s>>>(pc, x, wf); else sconv4_kernel<K_, N_, false, 4><<<g4, 256, lds, s>>>(pc, x, wf); } \
                             else { if (hg) sconv4_kernel<K_, N_, true, 8><<<g4, 256, lds, s>>>(pc, x, wf); else sconv4_kernel<K_, N_, false, 8><<<g4, 256, lds, s>>>(pc, x, wf); } } while (0)
                if (k5) { if (nk == 4) MV3D_S4(5, 4); else MV3D_S4(5, 2); }
                else { if (nk == 4) MV3D_S4(3, 4); else MV3D_S4(3, 2); }
#undef MV3D_S4
                return launched(who);
            });
        }
    }
    if (tile) return 1;                                             // a tap layout or channel count sconv4 is not compiled for: the generic kernel
    return dispatch(stream, OpInfo{nph == 4 ? "sconv<4ph,32px,N32>" : "sconv<1ph,32px,N32>", flops, bytes}, [=](hipStream_t s) {
        sconv_kernel<<<grid, 256, lds, s>>>(pc, x, wf);
        return launched(who);
    });
}

// Single-phase problems on output grids of whole 4 x 16 tiles (s2conv_kernel): stride 2 with 32 channels, stride 1 with 32 or 64
// channels where the pipelined kernel would run its 8 x 16-tile instance; returns 1 when not one of them.
int try_s2conv(const IgemmParams& p, void* ws, size_t ws_bytes, void* stream, const char* who, double flops, double bytes) {
    if (disabled_paths() & 268435456) return 1;
    if (p.so_h != 1 || p.so_w != 1 || p.sa_h != p.sa_w || (p.sa_h != 1 && p.sa_h != 2)) return 1;
    const int S = p.sa_h;
    if (p.fold || (p.Ka != 32 && !(S == 1 && p.Ka == 64)) || p.a_ld % 4 != 0 || (reinterpret_cast<uintptr_t>(p.A) & 15) || p.Cc < 16) return 1;
    const int Hp = p.Hp[0], Wp = p.Wp[0];
    if (Hp % 4 != 0 || Wp % 16 != 0 || Hp * Wp <= 64) return 1;
    const int ntaps = p.tap_begin[1];
    if (ntaps != 25 && ntaps != 9) return 1;
    if (p.act == MV3D_ACT_TANH || p.gact == MV3D_ACT_TANH || (p.gact != MV3D_ACT_NONE && !p.gref)) return 1;
    if (S == 1) {
        // MV3D_TC_S1_BELOW: only stride-1 layers with fewer 16 x 16 x 32-filter tasks (default: all of them -- with 8 x 16 tiles the
        // 64 x 64 layers run at 260 - 268 TFLOP/s against the pipelined kernel's 270 - 285, and the step is 1.3 % faster without its
        // persistent 150 KiB workgroups on every CU)
        static int s1_below = -1;
        if (s1_below < 0) { const char* e = getenv("MV3D_TC_S1_BELOW"); s1_below = e ? atoi(e) : (1 << 30); }
        if (p.N * cdiv(Hp, 16) * cdiv(Wp, 16) * cdiv(p.Cc, 32) >= s1_below) return 1;
    }
    int dh_min = 127, dh_max = -127, dw_min = 127, dw_max = -127;
    for (int t = 0; t < ntaps; ++t) {
        dh_min = std::min<int>(dh_min, p.taps[t].dh); dh_max = std::max<int>(dh_max, p.taps[t].dh);
        dw_min = std::min<int>(dw_min, p.taps[t].dw); dw_max = std::max<int>(dw_max, p.taps[t].dw);
    }
    if (ntaps == 25 ? (dh_max - dh_min != 4 || dw_max - dw_min != 4) : (dh_max - dh_min != 2 || dw_max - dw_min != 2)) return 1;
    // 8 x 16 tiles (two pixel groups per wave) for stride-1 32-channel layers with at least MV3D_TC_MT2_MIN 4 x 16 tiles
    static int mt2_min = -1;
    if (mt2_min < 0) { const char* e = getenv("MV3D_TC_MT2_MIN"); mt2_min = e ? atoi(e) : 1000; }
    const int MTv = (S == 1 && p.Ka == 32 && Hp % 8 == 0 && p.N * (Hp / 4) * (Wp / 16) >= mt2_min) ? 2 : 1;
    SconvParams x = {};
    x.G = 1; x.TH = 4 * MTv; x.tiles_h = Hp / x.TH; x.tiles_w = Wp / 16;
    x.HRi = (x.TH - 1) * S + (dh_max - dh_min + 1);
    x.HC = 15 * S + (dw_max - dw_min + 1);
    x.PS = p.Ka * 4 + 32;
    x.HCe = (x.HC + 1) / 2;
    x.HCp = x.HC;
    while ((S * x.HCp * x.PS) % 256 != 64) ++x.HCp;                 // one tile row = S halo rows: 64 bytes on in the bank row
    x.dh_min = dh_min; x.dw_min = dw_min;
    x.nk16 = p.Ka / 16; x.inv_nk16 = inv32(x.nk16);
    x.nsteps = ntaps * x.nk16;
    x.c8 = p.Ka / 8; x.inv_c8 = inv32(x.c8); x.inv_hc = inv32(x.HC); x.inv_hri = inv32(x.HRi);
    x.units = x.HRi * x.HC * x.c8;
    if (x.units >= 65536) return 1;
    const size_t lds = std::max((size_t)x.HRi * x.HCp * x.PS, (size_t)16 * 1024);
    if (lds > 80 * 1024) return 1;
    int rc = MV3D_OK;
    const uint4* wf = bconv_get_filter(p, ws, ws_bytes, stream, &x.ntiles, &rc);
    if (!wf) return rc;
    // 64 filters per workgroup (a wave pair = two filter tiles) where that leaves two workgroups per CU; otherwise (and for
    // 32-filter layers) the pair splits the channels
    static int pair_min = -1;
    if (pair_min < 0) { const char* e = getenv("MV3D_S2_PAIR_MIN"); pair_min = e ? atoi(e) : 512; }
    const int tiles = p.N * x.tiles_h * x.tiles_w;
    const bool ksplit = !(p.Cc % 64 == 0 && tiles * (p.Cc / 64) >= pair_min);
    const dim3 grid(tiles, ksplit ? cdiv(p.Cc, 32) : p.Cc / 64, 1);
    const bool hg = p.gact != MV3D_ACT_NONE, k5 = ntaps == 25;
    const int nk = x.nk16;
    using KernelT = void (*)(const IgemmParams, const SconvParams, const uint4*);
    KernelT kern = nullptr;
#define MV3D_S2_PICK(T_, S_, N_, M_) kern = ksplit ? (hg ? &s2conv_kernel<T_, S_, N_, true, true, M_> : &s2conv_kernel<T_, S_, N_, true, false, M_>) \
                                                   : (hg ? &s2conv_kernel<T_, S_, N_, false, true, M_> : &s2conv_kernel<T_, S_, N_, false, false, M_>)
    if (S == 2) { if (k5) MV3D_S2_PICK(25, 2, 2, 1); else MV3D_S2_PICK(9, 2, 2, 1); }
    else if (nk == 2 && MTv == 2) { if (k5) MV3D_S2_PICK(25, 1, 2, 2); else MV3D_S2_PICK(9, 1, 2, 2); }
    else if (nk == 2) { if (k5) MV3D_S2_PICK(25, 1, 2, 1); else MV3D_S2_PICK(9, 1, 2, 1); }
    else { if (k5) MV3D_S2_PICK(25, 1, 4, 1); else MV3D_S2_PICK(9, 1, 4, 1); }
#undef MV3D_S2_PICK
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipGetLastError();            // recording without a device: the attribute call fails and nothing is launched
    if (getenv("MV3D_TRACE"))
        fprintf(stderr, "[mv3d] %-22s s2conv N=%d in %dx%dx%d (stride %d) out %dx%dx%d taps=%d halo %dx%d (pitch %d) lds=%zu grid=%dx%d %s %.2f GFLOP\n",
                who, p.N, p.Ha, p.Wa, p.Ka, S, p.Hc, p.Wc, p.Cc, ntaps, x.HRi, x.HC, x.HCp, lds, grid.x, grid.y, ksplit ? "ksplit" : "N64", flops * 1e-9);
    const IgemmParams pc = p;
    const char* name = intern_label("s2conv<%s,s%d,C%d,%s%s%s>", k5 ? "5x5" : "3x3", S, p.Ka, ksplit ? "N32" : "N64", MTv == 2 ? ",128px" : "", hg ? ",gmask" : "");
    return dispatch(stream, OpInfo{name, flops, bytes}, [=](hipStream_t s) {
        kern<<<grid, 256, lds, s>>>(pc, x, wf);
        return launched(who);
    });
}

}  // namespace mv3d
