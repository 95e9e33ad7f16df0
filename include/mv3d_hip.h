/* mv3d_hip.h -- C ABI of libmv3d_hip.so: the MI355X (gfx950) kernels behind the
 * appearance-flow train step of aclike/dynamic_multiview_3d.
 *
 * The reference has no native code: every entry point below replaces one TensorFlow-1.3
 * op call site of dyn_mult_view/mv3d/utils/tf_utils.py (cited per function) plus the
 * reverse-mode ops tf.train.AdamOptimizer.minimize() derives from it
 * (multi_view_model/appearance_flow_model.py:77).
 *
 * Conventions
 *  - plain pointers + sizes, no torch/HIP types: a stream is passed as void* (hipStream_t).
 *  - every pointer is DEVICE memory owned by the caller; the library never allocates or
 *    frees device memory and keeps no global mutable state (except an optional recording plan,
 *    see mv3d_plan_*).  Scratch is passed in as (workspace, workspace_bytes).
 *  - all calls are asynchronous on `stream` and return 0 (MV3D_OK) or a negative MV3D_E_*;
 *    shape/alignment violations are rejected BEFORE anything is launched.
 *    mv3d_last_error() returns a thread-local message for the last failure.
 *  - activations are NHWC; a tensor may be a channel slice of a wider buffer: `*_ld` is the
 *    element stride between consecutive pixels (>= channels).  This makes tf.concat /
 *    tf.split on the channel axis free (main_model.py:94,131).
 *  - conv filters HWIO [kh,kw,Cin,Cout] (tf_utils.py:76), deconv filters [kh,kw,Cout,Cin]
 *    (tf_utils.py:94), fc matrices [in,out] (tf_utils.py:61).  Both filter kinds are
 *    [kh,kw,C_image_side,C_feature_side].
 *  - dtype: MV3D_F32 only in this round (IEEE fp32 in/out, fp32 MFMA accumulate).
 */
#ifndef MV3D_HIP_H
#define MV3D_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { MV3D_OK = 0, MV3D_E_INVAL = -1, MV3D_E_HIP = -2, MV3D_E_WORKSPACE = -3, MV3D_E_UNSUPPORTED = -4 };
enum { MV3D_F32 = 0 };
/* activation kinds: value = f1*x + f2*|x| forms of tf_utils.py:25-33, tf.nn.tanh (main_model.py:79) */
enum { MV3D_ACT_NONE = 0, MV3D_ACT_LRELU = 1, MV3D_ACT_RELU = 2, MV3D_ACT_TANH = 3 };

/* Geometry of one conv2d / conv2d_transpose layer.  The IMAGE side is the strided-into tensor
 * (conv input, deconv output); the FEATURE side is the other one (conv output, deconv input):
 * Ho = ceil(H/sh), Wo = ceil(W/sw), TF 'SAME' padding (pad_before = total/2). */
typedef struct mv3d_conv_geom {
    int32_t N;              /* batch */
    int32_t H, W, C;        /* image side  */
    int32_t Ho, Wo, K;      /* feature side */
    int32_t kh, kw, sh, sw;
    int32_t img_ld, feat_ld;/* pixel strides (elements) of the image-/feature-side tensors */
    int32_t dtype;          /* MV3D_F32 */
} mv3d_conv_geom;

/* Fused epilogue applied to the tensor a kernel produces.
 *   y = act(acc + bias)                  (forward: tf_utils.py:82 '+ b', :25-33 activations)
 *   y = acc * act'(ref)                  (backward: the producing layer's activation gradient,
 *                                         evaluated from its saved OUTPUT `ref`; TF: sign(0)=0)
 * gmask_ref has the shape of the produced tensor, pixel stride gmask_ld. */
typedef struct mv3d_epilogue {
    const void* bias;       /* [channels] or NULL */
    int32_t act;            /* MV3D_ACT_* applied to the result */
    float leak;             /* lrelu leak (tf_utils.py:29: 0.2) */
    int32_t gmask_act;      /* MV3D_ACT_* whose derivative multiplies the result, or NONE */
    float gmask_leak;
    const void* gmask_ref;
    int32_t gmask_ld;
} mv3d_epilogue;

const char* mv3d_version(void);
const char* mv3d_last_error(void);
/* Diagnostics: replace the mask of disabled dispatch rungs (environment MV3D_DISABLE at load; bits in DESIGN.md 4.5);
 * returns the previous mask.  Results never change beyond rounding; 4096 selects the exact fp32-MFMA kernels. */
int mv3d_set_diagnostics(int mask);
/* Tuning: the number of CUs the following conv / deconv filter-gradient calls (and their workspace queries) spread their
 * partial-filter slabs over; 0 restores the default (MV3D_WG_CUS, 128 = half the chip: those launches normally share it with
 * the data-gradient chain).  Returns the previous value.  Results never change beyond summation order. */
int mv3d_set_wgrad_cus(int cus);

/* ---- input side (host): the reference's TFRecord shards, multi_view_model/utils/read_tf_records.py:46-85 -----------------
 * mv3d_tfrecord_read copies feature k of up to max_records records into dst[k] + (first + i) * sizes[k] (host memory, e.g. a
 * pinned batch buffer): kinds[k] = 0 a single bytes value of exactly sizes[k] bytes (raw uint8 image), 1 a float list of
 * sizes[k] / 4 values.  *nread < max_records only at the end of the file.  CRC-32C of every record is checked when the
 * reader was opened with verify_crc != 0.  mv3d_u8_to_unit_f32: device-side uint8 -> float32 / 255 (read_tf_records.py:111). */
typedef struct mv3d_tfrecord_reader mv3d_tfrecord_reader;
int mv3d_tfrecord_open(const char* path, int verify_crc, mv3d_tfrecord_reader** out);
int mv3d_tfrecord_read(mv3d_tfrecord_reader* r, int max_records, int first, int nfeat, const char* const* names, const int* kinds,
                       const size_t* sizes, void* const* dst, int* nread);
void mv3d_tfrecord_close(mv3d_tfrecord_reader* r);
int mv3d_u8_to_unit_f32(int64_t count, const void* src, void* dst, void* stream);

/* CRC-32C of a host buffer: the record checksum of the reference's TFRecord shards (multi_view_model/utils/read_tf_records.py:46-48
 * reads them through tf.TFRecordReader); used by dynamic_multiview_3d_amd/read_tf_records.py */
uint32_t mv3d_crc32c(const void* data, size_t n);

/* ---- conv2d: tf.nn.conv2d(x, w, [1,sh,sw,1], 'SAME') + b  (tf_utils.py:81-82) ------------- */
/* y[N,Ho,Wo,K] = epi( x[N,H,W,C] (*) w[kh,kw,C,K] ) */
int mv3d_conv2d_fwd(const mv3d_conv_geom* g, const void* x, const void* w, void* y,
                    const mv3d_epilogue* epi, void* workspace, size_t workspace_bytes, void* stream);
/* dx[N,H,W,C] = epi( Conv2DBackpropInput(dy[N,Ho,Wo,K], w) ) */
int mv3d_conv2d_dgrad(const mv3d_conv_geom* g, const void* dy, const void* w, void* dx,
                      const mv3d_epilogue* epi, void* workspace, size_t workspace_bytes, void* stream);
/* dw[kh,kw,C,K] = Conv2DBackpropFilter(x, dy);  db[K] = BiasAddGrad(dy) (db may be NULL) */
int mv3d_conv2d_wgrad(const mv3d_conv_geom* g, const void* x, const void* dy, void* dw, void* db,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ---- conv2d_transpose: tf.nn.conv2d_transpose(x, w, output_shape, strides) (tf_utils.py:96-97)
 * x is the FEATURE side [N,Ho,Wo,K], y the IMAGE side [N,H,W,C], w [kh,kw,C,K]; no bias in the
 * reference (epi->bias must be NULL unless the caller wants one). */
int mv3d_deconv2d_fwd(const mv3d_conv_geom* g, const void* x, const void* w, void* y,
                      const mv3d_epilogue* epi, void* workspace, size_t workspace_bytes, void* stream);
int mv3d_deconv2d_dgrad(const mv3d_conv_geom* g, const void* dy, const void* w, void* dx,
                        const mv3d_epilogue* epi, void* workspace, size_t workspace_bytes, void* stream);
int mv3d_deconv2d_wgrad(const mv3d_conv_geom* g, const void* x, const void* dy, void* dw,
                        void* workspace, size_t workspace_bytes, void* stream);
/* bytes of scratch the six calls above may need for this geometry (max over them) */
size_t mv3d_conv_workspace_bytes(const mv3d_conv_geom* g);
/* the part of it the filter-gradient call of this geometry needs (its per-slab partial sums): what a caller that gives every
 * layer its own workspace (mv3d_grad_finalize_*) has to reserve per layer */
size_t mv3d_conv_wgrad_workspace_bytes(const mv3d_conv_geom* g);

/* ---- prepared filters (optional) -------------------------------------------------------------
 * The matrix-core convolution kernels read the filter split into bf16 hi/lo parts in MFMA fragment order.
 * By default each call converts its filter into the workspace (one extra small launch).  A caller that knows
 * its filters stay constant over several calls -- the reference's train step uses every filter in the forward
 * pass and again in the backward-data pass before tf.train.AdamOptimizer updates it (appearance_flow_model.py:77)
 * -- binds one caller-owned buffer per (filter, operation), commits the job table, and calls
 * mv3d_filter_cache_refresh() once after every weight update: ONE launch converts all bound filters and the
 * convolution calls that find their filter pointer in the cache skip their own conversion.
 * The cache is process-global (guarded by a mutex); the library never frees the bound buffers. */
enum { MV3D_FILTER_CONV_FWD = 0, MV3D_FILTER_CONV_DGRAD = 1, MV3D_FILTER_DECONV_FWD = 2, MV3D_FILTER_DECONV_DGRAD = 3 };
/* bytes of the prepared copy for this geometry / operation; 0 = the operation does not take one */
size_t mv3d_filter_prepared_bytes(const mv3d_conv_geom* g, int op);
int mv3d_filter_cache_bind(const mv3d_conv_geom* g, int op, const void* w, void* prepared, size_t prepared_bytes);
/* device bytes of the job table for the current bindings; commit uploads it (synchronises `stream`) */
size_t mv3d_filter_cache_table_bytes(void);
int mv3d_filter_cache_commit(void* table_dev, size_t table_bytes, void* stream);
/* one launch: convert every bound filter (recordable in a plan) */
int mv3d_filter_cache_refresh(void* stream);
int mv3d_filter_cache_clear(void);

/* ---- linear: tf.matmul(x, M) + b  (tf_utils.py:67) ------------------------------------------
 * x [B,in] (row stride x_ld), M [in,out] dense, y [B,out] (row stride y_ld). */
int mv3d_fc_fwd(int B, int in, int out, const void* x, int x_ld, const void* M, void* y, int y_ld,
                const mv3d_epilogue* epi, void* workspace, size_t workspace_bytes, void* stream);
int mv3d_fc_dgrad(int B, int in, int out, const void* dy, int dy_ld, const void* M, void* dx, int dx_ld,
                  const mv3d_epilogue* epi, void* workspace, size_t workspace_bytes, void* stream);
int mv3d_fc_wgrad(int B, int in, int out, const void* x, int x_ld, const void* dy, int dy_ld,
                  void* dM, void* db, void* workspace, size_t workspace_bytes, void* stream);
/* Filter gradient AND data gradient of one small linear layer in ONE launch (both read dy; the angle MLP's a1 / a2,
 * appearance_flow_model.py:101-103): the results are those of mv3d_fc_wgrad followed by mv3d_fc_dgrad, bit for bit.  Layers
 * that are not "small" (the streaming fc kernels take them) are run as those two calls. */
int mv3d_fc_wgrad_dgrad(int B, int in, int out, const void* x, int x_ld, const void* dy, int dy_ld, const void* M,
                        void* dM, void* db, void* dx, int dx_ld, const mv3d_epilogue* dx_epi,
                        void* workspace, size_t workspace_bytes, void* stream);
/* A chain of 2..4 small linear layers (every width <= 64), y_l = act_l(y_{l-1} M_l + b_l), in ONE launch: a workgroup carries
 * four batch rows through all layers (activations stay in LDS, every y_l is also stored: the reverse pass needs it).  The
 * results are those of nlayers mv3d_fc_fwd calls, bit for bit. */
#define MV3D_FC_CHAIN_MAX 4
typedef struct mv3d_fc_chain_layer {
    const void* M;        /* [in, out], in = the previous layer's out (first layer: mv3d_fc_chain.in) */
    const void* bias;     /* [out] or NULL */
    void* y;              /* [B, out], row stride y_ld */
    int32_t y_ld, out;
    int32_t act;          /* MV3D_ACT_* */
    float leak;
} mv3d_fc_chain_layer;
typedef struct mv3d_fc_chain {
    int32_t B, nlayers, in, x_ld;
    const void* x;        /* [B, in], row stride x_ld */
    mv3d_fc_chain_layer l[MV3D_FC_CHAIN_MAX];
} mv3d_fc_chain;
int mv3d_fc_chain_fwd(const mv3d_fc_chain* chain, void* stream);
size_t mv3d_fc_workspace_bytes(int B, int in, int out);

/* ---- element-wise pieces -------------------------------------------------------------------- */
/* y = act(x) on [rows, ch] with row strides (standalone form of tf_utils.py:25-33 / tanh) */
int mv3d_act_fwd(int64_t rows, int ch, const void* x, int x_ld, void* y, int y_ld, int act, float leak, void* stream);
/* dx = dy * act'(ref) with ref = the activation's OUTPUT (sign(0) = 0) */
int mv3d_act_bwd(int64_t rows, int ch, const void* dy, int dy_ld, const void* ref, int ref_ld,
                 void* dx, int dx_ld, int act, float leak, void* stream);
/* strided copy / accumulate of [rows, ch] blocks (tf.concat/tf.split glue that cannot be a view,
 * tf.tile of the angle code over 4x4: multiobject_appflow.py:148-150):
 * dst[r*dst_ld + c] (+)= src[(r / src_row_div) * src_ld + c] */
int mv3d_copy2d(int64_t rows, int ch, const void* src, int64_t src_ld, int64_t src_row_div,
                void* dst, int64_t dst_ld, int accumulate, void* stream);
/* dst[g, c] = sum_{r < group} src[(g*group + r)*src_ld + c]   (gradient of the tile above) */
int mv3d_group_sum(int64_t groups, int group, int ch, const void* src, int64_t src_ld, void* dst, int64_t dst_ld, void* stream);

/* ---- warp + resampler: warp_pts_layer + resample_layer (tf_utils.py:35-52) ------------------
 * flow [N,H,W,2] (pixel stride flow_ld); warp = flow + coords where coords[...,0] = ROW index,
 * coords[...,1] = COLUMN index (tf_utils.py:48-51) and tf.contrib.resampler reads warp[...,0]
 * as x (column) and warp[...,1] as y (row); zero outside.
 * src [N,Hs,Ws,C] dense.  gen [N,H,W,C] dense.  warp_out (optional) [N,H,W,2] dense. */
int mv3d_warp_resample_fwd(int N, int H, int W, int Hs, int Ws, int C, const void* src, const void* flow, int flow_ld,
                           void* warp_out, void* gen, void* stream);
/* dflow[N,H,W,2] (pixel stride dflow_ld) = resampler grad w.r.t. warp (= grad w.r.t. flow) */
int mv3d_warp_resample_bwd(int N, int H, int W, int Hs, int Ws, int C, const void* src, const void* flow, int flow_ld,
                           const void* dgen, void* dflow, int dflow_ld, void* stream);
/* The appearance-flow head in one pass (appearance_flow_model.py:127-130 + tf_utils.py:18-23): gen = resample(src, flow +
 * coords); loss_accum[0] += weight * mean_{n,h,w} sum_c f(gen - target) (kind 2 squared, 1 absolute); dflow (optional) =
 * d(weight * loss)/dflow.  Same arithmetic as mv3d_warp_resample_fwd -> mv3d_pixel_loss -> mv3d_warp_resample_bwd with the
 * loss gradient kept on chip.  target [N,H,W,C] with pixel stride target_ld; C <= 4. */
int mv3d_warp_resample_loss(int N, int H, int W, int Hs, int Ws, int C, const void* src, const void* flow, int flow_ld,
                            const void* target, int target_ld, int kind, float weight, void* warp_out, void* gen,
                            void* dflow, int dflow_ld, void* loss_accum, void* stream);

/* ---- losses: euclidean_loss / l1_loss (tf_utils.py:18-23) -----------------------------------
 * loss_accum[0] += weight * mean_{n,h,w} sum_c f((a-b)*mask);  grad (optional, same shape as a,
 * dense) = d(weight*loss)/da.  kind 2 = squared (euclidean), 1 = absolute (l1).  mask (optional)
 * is [pixels,1] and multiplies the difference (multiobject_appflow.py:239-242).
 * loss_accum must be zeroed by the caller before the first term (mv3d_fill). */
int mv3d_pixel_loss(int64_t pixels, int ch, const void* a, const void* b, const void* mask, int kind, float weight,
                    void* loss_accum, void* grad, void* stream);
/* same on channel-slice views (pixel strides *_ld), with the target read as b * b_scale and a per-pixel mask of stride
 * mask_ld: the mv3d losses slice a 4-channel prediction / target into colour and depth or mask parts
 * (mv3d/nobg_dm.py:85-92, mv3d/bg_nodm.py:85-93: gt_sm * 0.75, tf.multiply(.., sm)) */
int mv3d_pixel_loss_strided(int64_t pixels, int ch, const void* a, int a_ld, const void* b, int b_ld, float b_scale,
                            const void* mask, int mask_ld, int kind, float weight, void* loss_accum, void* grad, int grad_ld,
                            void* stream);
int mv3d_fill(void* dst, int64_t count, float value, void* stream);
/* The NEXT loss call of the calling thread (mv3d_pixel_loss*, mv3d_warp_resample_loss) stores its term into loss_accum instead of
 * adding it: the first term of a recorded step then needs no launch that clears the accumulator (tf.add_n over the terms of
 * appearance_flow_model.py:127-130 starts from the first one). */
int mv3d_loss_overwrite_next(void);

/* ---- Adam: tf.train.AdamOptimizer ApplyAdam (appearance_flow_model.py:77) -------
 *   alpha = lr*sqrt(1-beta2_power)/(1-beta1_power);  m += (g-m)(1-b1);  v += (g*g-v)(1-b2);
 *   p -= m*alpha/(sqrt(v)+eps).   One fused pass over a flat fp32 buffer; grad_scale multiplies g
 *   first (1/world_size after a SUM all-reduce). */
int mv3d_adam_step(int64_t count, void* p, const void* g, void* m, void* v, float lr, float beta1, float beta2,
                   float eps, float beta1_power, float beta2_power, float grad_scale, void* stream);

/* Device-resident optimiser state, so that a RECORDED step replays with the current bias correction:
 * state[MV3D_ADAM_LR .. MV3D_ADAM_GSCALE] floats (8 allocated).  mv3d_adam_step_dev is mv3d_adam_step with the scalars read
 * from it, and leaves up to 8 index ranges [skip_lo, skip_hi) (multiples of 4) untouched; mv3d_adam_advance multiplies the two
 * beta powers by their betas (tf.train.AdamOptimizer._finish, appearance_flow_model.py:77), one launch per step. */
enum { MV3D_ADAM_LR = 0, MV3D_ADAM_BETA1 = 1, MV3D_ADAM_BETA2 = 2, MV3D_ADAM_EPS = 3, MV3D_ADAM_BETA1_POWER = 4, MV3D_ADAM_BETA2_POWER = 5,
       MV3D_ADAM_GSCALE = 6, MV3D_ADAM_STATE_FLOATS = 8 };
int mv3d_adam_step_dev(int64_t count, void* p, const void* g, void* m, void* v, const void* adam_state, int nskip,
                       const int64_t* skip_lo, const int64_t* skip_hi, void* stream);
int mv3d_adam_advance(void* adam_state, void* stream);
/* linear_msra's filter gradient with the ApplyAdam update of that matrix fused into the epilogue (tf_utils.py:54-67 +
 * appearance_flow_model.py:77): M, adam_m, adam_v [in,out] are updated in place from dM = x^T dy, which never goes to memory
 * (24 instead of 32 B of HBM traffic per parameter over the two passes it replaces); db[out] (optional) receives the bias
 * gradient as mv3d_fc_wgrad writes it.  Single-GPU steps only: the data-parallel step needs the gradient itself.  The caller
 * orders this call behind the last reader of M (the layer's own data gradient).  _supported: 1 when the layer is one the fused
 * kernel takes, 0 when mv3d_fc_wgrad + mv3d_adam_step_dev must be used. */
int mv3d_fc_wgrad_adam(int B, int in, int out, const void* x, int x_ld, const void* dy, int dy_ld, void* M, void* adam_m, void* adam_v,
                       void* db, const void* adam_state, void* stream);
int mv3d_fc_wgrad_adam_supported(int B, int in, int out, int x_ld, int dy_ld);

/* ---- gradient finalisation: the slab reductions of ALL filter gradients (+ their optimiser update) in one launch ------------
 * Replaces, on the recorded single-GPU step, the per-layer partial-filter reductions behind tf.gradients' Conv2DBackpropFilter
 * ops and the tf.train.AdamOptimizer ApplyAdam ops of every variable that is not an fc matrix (appearance_flow_model.py:77).
 * Between mv3d_grad_finalize_begin() and _commit() on this thread the filter-gradient entry points (mv3d_conv2d_wgrad,
 * mv3d_deconv2d_wgrad) leave their per-slab partial sums in the workspace they were given -- which the caller must then keep
 * untouched until the committed launch has run, i.e. one workspace per layer -- and record a segment instead of launching a
 * reduction.  mv3d_grad_finalize_add names a gradient range that is already final (fc biases, tiny fc layers) so that the
 * optimiser covers it too.  _table_bytes: device bytes the segment table needs for what has been collected so far.
 * _commit closes the collection, uploads the table (synchronously, at call / record time) and dispatches ONE launch that sums
 * every segment's slabs in the fixed order of the per-layer reduction (same bits) and
 *   adam_state == NULL: stores the gradients where the wgrad calls were told to (grads/params/adam_m/adam_v ignored);
 *   adam_state != NULL: applies ApplyAdam to each element instead (mv3d_adam_step_dev's arithmetic, same bits); every
 *                       segment must lie inside the flat buffer `grads`, whose layout params / adam_m / adam_v share.
 * _abort drops an open collection.  (table may be NULL only while recording a plan without a device, which can never run.) */
int mv3d_grad_finalize_begin(void);
int mv3d_grad_finalize_add(void* grad, int64_t count);
size_t mv3d_grad_finalize_table_bytes(void);
int mv3d_grad_finalize_commit(void* table, size_t table_bytes, void* grads, void* params, void* adam_m, void* adam_v,
                              const void* adam_state, void* stream);
int mv3d_grad_finalize_abort(void);

/* ---- data-parallel exchange: RCCL over xGMI behind the ABI (one process per GPU) -----------------------------------------
 * The reference trains on one device (multi_view_model/train.py:21,35); the batch shards over ranks and the flat fp32 gradient
 * buffer is summed across them.  RCCL is looked up in the process image at run time (the host program's own
 * librccl / HIP runtime); without it every call returns MV3D_E_UNSUPPORTED.  Rank 0 creates a 128-byte id
 * (mv3d_comm_unique_id) and hands it to the other ranks by any side channel (file, environment, TCP store); every rank then
 * calls mv3d_comm_init on its own device.  Collectives are in place on `stream`, fp32, SUM; counts are in elements.
 * reduce_scatter: send holds world * recv_count elements, rank r receives the sum of slice r; allgather is its inverse. */
typedef struct mv3d_comm mv3d_comm;
int mv3d_comm_available(void);
int mv3d_comm_unique_id(void* id128);
int mv3d_comm_init(mv3d_comm** out, int rank, int world, const void* id128);
int mv3d_comm_destroy(mv3d_comm* comm);
int mv3d_comm_allreduce_sum(mv3d_comm* comm, void* buf, int64_t count, void* stream);
int mv3d_comm_reduce_scatter_sum(mv3d_comm* comm, const void* send, void* recv, int64_t recv_count, void* stream);
int mv3d_comm_allgather(mv3d_comm* comm, const void* send, void* recv, int64_t send_count, void* stream);

/* ---- mesh-direct exchange: xGMI is a full mesh of point-to-point links, so a rank that PULLS its slice of every
 * peer's gradient buffer directly uses all seven links at once, where a ring collective is bound by one.  mv3d_ipc_export gives
 * the 64-byte hipIpc handle of the allocation that holds `ptr` and ptr's offset in it; a peer process maps it with
 * mv3d_ipc_open (base of the allocation; add the offset) and unmaps it with mv3d_ipc_close.  mv3d_mesh_reduce_sum writes
 * dst[i] = srcs[0][i] + ... + srcs[nsrc-1][i], added in that order (srcs: host array of device pointers, local or mapped;
 * dst may be one of them); mv3d_mesh_copy is the pull of an updated slice.  Ordering between ranks (nobody reads a buffer its
 * owner is still writing) is the caller's: parallel.MeshComm does it over the control plane. */
#define MV3D_MESH_MAX_RANKS 8
int mv3d_ipc_export(const void* ptr, void* handle64, int64_t* offset);
int mv3d_ipc_open(const void* handle64, void** base);
int mv3d_ipc_close(void* base);
int mv3d_mesh_reduce_sum(const void* const* srcs, int nsrc, void* dst, int64_t count, void* stream);
int mv3d_mesh_copy(void* dst, const void* src, int64_t count, void* stream);

/* ---- recorded plans: native replay of a fixed launch sequence (the step is static) ----------
 * Between mv3d_plan_begin() and mv3d_plan_end() every mv3d_* op call on this thread is RECORDED
 * (validated, not launched).  mv3d_plan_run() launches the recorded sequence on a stream in one
 * native call (no per-op Python/ctypes cost); it is capture-safe (hipGraph). */
typedef struct mv3d_plan mv3d_plan;
mv3d_plan* mv3d_plan_create(void);
void mv3d_plan_destroy(mv3d_plan* p);
int mv3d_plan_begin(mv3d_plan* p);
int mv3d_plan_end(void);
int mv3d_plan_size(const mv3d_plan* p);
int mv3d_plan_run(mv3d_plan* p, void* stream);
/* launches ops [begin, end) only: lets the host interleave other stream work (bucket all-reduces of
 * the data-parallel path) between segments of the recorded backward sequence */
int mv3d_plan_run_range(mv3d_plan* p, int begin, int end, void* stream);
/* Two-stream runs.  Calls recorded between mv3d_plan_side(k > 0) and mv3d_plan_side(0) are "side work": they depend on
 * everything recorded before them and nothing recorded after them in the plan depends on them (the filter / bias
 * gradients of the reverse pass: only the optimiser reads them).  mv3d_plan_run_range2 issues side work on
 * `side_stream` behind an event on `stream` and makes `stream` wait for it at the end of the range, so the filter
 * gradients of a layer run concurrently with the data gradients of the layers below it.  Side work must not share
 * scratch memory with main work.  side_stream == NULL: same as mv3d_plan_run_range. */
#define MV3D_MAX_SIDE 4
int mv3d_plan_side(int side);                  /* 0 = main, 1..MV3D_MAX_SIDE = side work class */
int mv3d_plan_run_range2(mv3d_plan* p, int begin, int end, void* stream, void* side_stream);
/* side class k runs on side_streams[(k-1) % nside]; classes on different streams must not share scratch either */
#define MV3D_RUN_NO_JOIN 1      /* the caller orders `stream` behind the side streams itself */
#define MV3D_RUN_HOLD_CLASS2 2  /* leave side class 2 out: the caller issues it with mv3d_plan_run_side once ITS dependencies are met */
int mv3d_plan_run_range_multi(mv3d_plan* p, int begin, int end, void* stream, void* const* side_streams, int nside, int flags);
int mv3d_plan_run_side(mv3d_plan* p, int cls, void* stream);
/* Per-launch timing for the roofline report: with profiling enabled, mv3d_plan_run brackets every
 * recorded launch with hipEventRecord on the launch stream (no host synchronisation);
 * mv3d_plan_profile_collect() synchronises once and folds all runs into per-op totals.
 * mv3d_plan_op_info returns the kernel label, the algorithmic FLOPs and HBM bytes of launch i
 * (formulas in DESIGN.md), the accumulated milliseconds and the number of timed runs. */
int mv3d_plan_profile(mv3d_plan* p, int enable);
int mv3d_plan_profile_collect(mv3d_plan* p);
/* bracket only the launches whose kernel label matches `name` (NULL: all): one label or a '|'-separated list, each entry
 * optionally ending in '*' for a prefix match -- a handful of events per step, cheap enough to leave on inside a timed region */
int mv3d_plan_profile_select(mv3d_plan* p, const char* name);
int mv3d_plan_profile_reset(mv3d_plan* p);
int mv3d_plan_op_info(const mv3d_plan* p, int i, const char** name, double* flops, double* bytes,
                      double* total_ms, int* runs);

/* Diagnostics: copies the in-kernel clock stamps of the pipelined convolution kernel (csrc/cconv.hip; written only when the
 * environment sets MV3D_DBG bit 32) to host memory: [256 workgroups][8 waves][64 events] uint64 (tools/cconv_stamps.py). */
int mv3d_debug_cconv_stamps(void* host_dst, size_t bytes);
/* the same for the pipelined filter-gradient kernel (cwgrad): [128 slabs][8 waves][64 events] of the workgroups with blockIdx.x == 0 */
int mv3d_debug_cwgrad_stamps(void* host_dst, size_t bytes);
/* the same for the row-band kernel of the thin input layers (thin.hip): [workgroup < 512][wave 4][16] uint64 */
int mv3d_debug_band_stamps(void* host_dst, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif
