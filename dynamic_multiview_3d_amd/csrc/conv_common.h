// Shared between conv.hip (generic implicit GEMM) and hconv.hip (halo-tile kernels).
#pragma once
#include "common.h"

namespace mv3d {

struct IgemmTap { int8_t dh, dw; int16_t widx; };

struct IgemmParams {
    const float* A;      // input activations
    const float* Wt;     // filter [kh*kw][C][K]
    float* Out;          // output activations
    float* Part;         // split-K partials [ksplit][N*Hc*Wc][Cc] (ksplit > 1)
    int N, Ha, Wa, Ca, a_ld;
    int Hc, Wc, Cc, c_ld;
    int sa_h, sa_w;      // input coordinate multiplier
    int so_h, so_w;      // output phase stride
    int Hp[2], Wp[2];    // per-phase output sub-grid
    int tap_begin[5];
    IgemmTap taps[36];
    int fold;            // 1: a tap covers kw*Ca contiguous elements of a dense NHWC row (small Ca)
    int Ka;              // reduction extent per tap
    int w_tap_stride, w_ks, w_ns;
    int ksplit;
    // epilogue
    const float* bias; int act; float leak;
    int gact; float gleak; const float* gref; int g_ld;
};

__device__ __forceinline__ float epilogue_value(const IgemmParams& p, float v, int64_t pix, int col) {
    if (p.bias) v += p.bias[col];
    v = act_apply(v, p.act, p.leak);
    if (p.gact != MV3D_ACT_NONE) v *= act_grad_from_out(p.gref[pix * p.g_ld + col], p.gact, p.gleak);
    return v;
}


// Tile description shared by the halo-tile kernels (hconv.hip fp32, bconv.hip split-bf16).
struct HconvExtra {
    int TH, TW, tw_shift;
    int tiles_h, tiles_w;
    int HR, HC;
    int dh_min, dw_min;
    int chunks, ntaps_total;
    int phase_split;     // 1: grid.z = stride phase; each workgroup computes ONE phase (more workgroups for small layers)
    int G, img_shift;    // small images: a tile is G whole images of 2^img_shift phase-grid pixels (halos stacked vertically)
    int HRi;             // halo rows per image (HR = G * HRi)
    int ksplit;          // channel chunks are split over grid.z; raw partial sums go to p.Part, igemm_splitk_epilogue finishes
    int dbg;             // MV3D_DBG diagnostics: 1 = no halo loads, 8 = skip the tap loop
    int inv_hc, inv_hri; // bconv: ceil(2^20 / HC), ceil(2^20 / HRi) -- divisions by multiply + shift in the staging loops
    int n_tiles;         // bconvu: number of spatial tiles (workgroups walk them with stride gridDim.x)
    int row_bytes;       // bconv (split-bf16) kernels: LDS bytes per halo row (padded for conflict-free 16-byte reads)
};

// hconv.hip: returns MV3D_OK after dispatching, or 1 if the problem is not eligible for the
// halo-tile kernel (caller falls back to the generic igemm).
// returns MV3D_OK after dispatching, 1 when not eligible, 2 when partial sums were written and the caller must
// run igemm_splitk_epilogue with *epi_out
int try_hconv(const IgemmParams& p, void* ws, size_t ws_bytes, void* stream, const char* who, double flops, double bytes, IgemmParams* epi_out);

// fc.hip: weight-streaming linear layers; each returns 1 when not applicable
size_t fc_stream_ws_bytes(int B, int in, int out, bool trans);
int try_fc_stream(bool trans, int B, int in, int out, const void* x, int x_ld, const void* W, void* y, int y_ld,
                  const mv3d_epilogue* epi, void* ws, size_t wsb, void* stream, const char* who,
                  void (*fill_epi)(IgemmParams&, const mv3d_epilogue*), void (*launch_epi)(const IgemmParams&, int, hipStream_t));
// bconv.hip: split-bf16 (hi + lo, three bf16 MFMA products per fp32 product) halo-tile convolution.
//   bconv_filter_bytes: workspace bytes of the fragment-ordered split filter for this problem
//   launch_bconv: filter split + main kernel (2 launches); nph 1|4, MT/NT register blocking, 2 or 4 waves
size_t bconv_filter_bytes(const IgemmParams& p, int NT);
int bconv_lds_bytes(const HconvExtra& x);
size_t bconv_prepared_bytes(const IgemmParams& p);        // 0: this problem never uses a prepared filter
int bconv_cache_bind(const IgemmParams& p, void* prepared, size_t bytes);
void bconv_set_rows(HconvExtra* x, int stride_h = 1);
int launch_bconv(const IgemmParams& p, const HconvExtra& x, int nph_fused, int MT, int NT, int WAVES, dim3 grid,
                 void* wfrag, void* stream, const char* name, const char* who, double flops, double bytes);

const uint4* bconv_get_filter(const IgemmParams& p, void* ws, size_t ws_bytes, void* stream, int* ntiles_out, int* rc);
// sconv.hip: small-image convolution (8 x 8 / 4 x 4 phase grids) with the reduction split inside the workgroup -- no split-K
// partials, no epilogue launch; returns 1 when the problem is not one of its
int try_sconv(const IgemmParams& p, void* ws, size_t ws_bytes, void* stream, const char* who, double flops, double bytes);
int try_s2conv(const IgemmParams& p, void* ws, size_t ws_bytes, void* stream, const char* who, double flops, double bytes);

// thin.hip: row-band kernel for 1..3-channel image sides (stride 2); returns 1 when not applicable
int try_smallc_band(const mv3d_conv_geom* g, const IgemmParams& ep, int pt, int pl, const void* img, const void* w, void* feat,
                    void* stream, const char* who, double flops, double bytes);

// thin.hip: matrix-core filter gradient of the same layers: slabs (0 = not applicable) and launch (partials [slab][filter] + [slab][K])
int try_thin_head(const IgemmParams& p, void* stream, const char* who, double flops, double bytes);
int thin_wgrad_slabs(const mv3d_conv_geom* g);
int thin_wgrad_launch(const mv3d_conv_geom* g, const void* img, const void* feat, void* part, void* bias_part, void* stream,
                      const char* who, double flops, double bytes);

// cconv.hip: software-pipelined split-bf16 convolution (one persistent 8-wave workgroup per CU, LDS-DMA halo staging) for
// single-phase stride-1 5x5 / 3x3 problems on images of at least 16 x 16 pixels
bool cconv_eligible(const IgemmParams& p, int* kw_out, bool* rev_out);
int launch_cconv(const IgemmParams& p, const HconvExtra& x, const void* wf, int ntiles, void* stream, const char* who, double flops, double bytes);

int try_fc_wgrad(int B, int in, int out, const void* x, int x_ld, const void* dy, int dy_ld, void* dM, void* db, void* stream, const char* who);
int try_fc_wgrad_adam(int B, int in, int out, const void* x, int x_ld, const void* dy, int dy_ld, void* P, void* M1, void* V2, void* db,
                      const void* state, void* stream, const char* who);

}  // namespace mv3d
