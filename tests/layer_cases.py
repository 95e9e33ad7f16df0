"""The layer shapes the benchmarked configurations really run, shared by the GPU parity test
(tests/test_gpu_layers.py) and the CPU label-coverage test (tests/test_label_coverage.py).

The tile planner (csrc/hconv.hip try_hconv, csrc/conv.hip) picks kernels by batch x spatial size, so a shape at
batch 2 does not exercise the kernel the same layer uses at batch 64.  Every row below is one layer of
multi_view_model/appearance_flow_model.py:88-125 at the batch of tensorflowdata/appflow_offset/conf.py:21 (64),
of main_model.py:96-137 at BASELINE config 3's batch (128), or of multiobject_appflow.py:93-153 at 256 x 256
(BASELINE config 5: batch 32 per GPU) -- including the pixel strides of the concat buffers they read and write.

kind, n, h, w, c (image side), k (feature side), ksz, stride, img_ld, feat_ld, need_dx
"""
import ctypes as C

CONV, DECONV = 'conv', 'deconv'

APPFLOW_B64 = [
    (CONV, 64, 128, 128, 3, 32, 5, 2, 3, 32, False),      # e0
    (CONV, 64, 64, 64, 32, 32, 5, 1, 32, 32, True),       # e0_0, d1_0
    (CONV, 64, 64, 64, 32, 32, 5, 2, 32, 32, True),       # e1
    (CONV, 64, 32, 32, 32, 32, 5, 1, 32, 32, True),       # e1_0
    (CONV, 64, 32, 32, 32, 64, 5, 2, 32, 64, True),       # e2
    (CONV, 64, 16, 16, 64, 64, 5, 1, 64, 64, True),       # e2_0, d3_0
    (CONV, 64, 16, 16, 64, 128, 3, 2, 64, 128, True),     # e3
    (CONV, 64, 8, 8, 128, 128, 3, 1, 128, 128, True),     # e3_0, d4_0
    (CONV, 64, 8, 8, 128, 256, 3, 2, 128, 256, True),     # e4
    (CONV, 64, 4, 4, 256, 256, 3, 1, 256, 256, True),     # e4_0
    (DECONV, 64, 8, 8, 128, 256, 3, 2, 128, 256, True),   # d4
    (DECONV, 64, 16, 16, 64, 128, 3, 2, 64, 128, True),   # d3
    (DECONV, 64, 32, 32, 32, 64, 5, 2, 32, 64, True),     # d2
    (CONV, 64, 32, 32, 32, 64, 5, 1, 32, 64, True),       # d2_0
    (DECONV, 64, 64, 64, 32, 64, 5, 2, 32, 64, True),     # d1
    (DECONV, 64, 128, 128, 2, 32, 5, 2, 2, 32, True),     # flow_field
]

# main_model.Base_Prediction_Model colour + depth at batch 128
BASEPRED_B128 = [
    (CONV, 128, 128, 128, 1, 32, 5, 2, 1, 32, False),     # pre_dimage0/e0
    (CONV, 128, 64, 64, 32, 32, 5, 1, 32, 32, True),      # pre_*/e0_0, dec_*/d1_0
    (CONV, 128, 64, 64, 32, 32, 5, 2, 32, 32, True),      # pre_*/e1
    (CONV, 128, 32, 32, 32, 32, 5, 1, 32, 32, True),      # pre_*/e1_0
    (CONV, 128, 32, 32, 32, 64, 5, 2, 32, 128, True),     # pre_*/e2: writes its half of the 128-channel concat buffer
    (CONV, 128, 16, 16, 128, 64, 5, 1, 128, 64, True),    # e2_0 (reads the concat buffer)
    (CONV, 128, 16, 16, 64, 128, 5, 1, 64, 128, True),    # d3_0 with two decoders (main_model.py:128)
    (CONV, 128, 16, 16, 64, 128, 3, 2, 64, 128, True),    # e3
    (CONV, 128, 8, 8, 128, 128, 3, 1, 128, 128, True),    # e3_0, d4_0
    (CONV, 128, 8, 8, 128, 256, 3, 2, 128, 256, True),    # e4
    (CONV, 128, 4, 4, 256, 256, 3, 1, 256, 256, True),    # e4_0
    (DECONV, 128, 8, 8, 128, 256, 3, 2, 128, 256, True),  # d4
    (DECONV, 128, 16, 16, 64, 128, 3, 2, 64, 128, True),  # d3
    (DECONV, 128, 32, 32, 32, 64, 5, 2, 32, 128, True),   # dec_*/d2: reads its half of d3_0's split output
    (CONV, 128, 32, 32, 32, 64, 5, 1, 32, 64, True),      # dec_*/d2_0
    (DECONV, 128, 64, 64, 32, 64, 5, 2, 32, 64, True),    # dec_*/d1
    (DECONV, 128, 128, 128, 3, 32, 5, 2, 3, 32, True),    # dec_image1/d0
    (DECONV, 128, 128, 128, 1, 32, 5, 2, 1, 32, True),    # dec_dimage1/d0
]

# multiobject_appflow.MultiObjectAppFlow at 256 x 256, 'fully_conv', colour + depth + separate images (BASELINE config 5:
# global batch 256 = 8 x 32; the 256 x 256 extension of multiobject_appflow.py:93-153 is SURVEY 8a note 2), batch 32 per GPU
MULTIOBJ_256_B32 = [
    (CONV, 32, 256, 256, 3, 32, 5, 2, 3, 32, False),      # pre_image*/e0
    (CONV, 32, 256, 256, 1, 32, 5, 2, 1, 32, False),      # pre_dimage*/e0
    (CONV, 32, 128, 128, 32, 32, 5, 1, 32, 32, True),     # pre_*/e0_0, dec_*/d1_0
    (CONV, 32, 128, 128, 32, 32, 5, 2, 32, 32, True),     # pre_*/e1
    (CONV, 32, 64, 64, 32, 32, 5, 1, 32, 32, True),       # pre_*/e1_0
    (CONV, 32, 64, 64, 32, 64, 5, 2, 32, 256, True),      # pre_*/e2 into the 4-tower concat buffer
    (CONV, 32, 32, 32, 256, 64, 5, 1, 256, 64, True),     # e2_0
    (CONV, 32, 32, 32, 64, 128, 3, 2, 64, 128, True),     # e3
    (CONV, 32, 16, 16, 128, 128, 3, 1, 128, 128, True),   # e3_0, d4_0
    (CONV, 32, 16, 16, 128, 256, 3, 2, 128, 256, True),   # e4
    (CONV, 32, 8, 8, 256, 256, 3, 1, 256, 320, True),     # e4_0 into the [e4_0, tiled angle code] buffer
    (CONV, 32, 8, 8, 320, 256, 3, 1, 320, 256, True),     # e4_1
    (CONV, 32, 8, 8, 256, 256, 3, 1, 256, 256, True),     # e4_2
    (DECONV, 32, 16, 16, 128, 256, 3, 2, 128, 256, True), # d4
    (DECONV, 32, 32, 32, 64, 128, 3, 2, 64, 128, True),   # d3
    (CONV, 32, 32, 32, 64, 384, 5, 1, 64, 384, True),     # d3_0 for six decoders
    (DECONV, 32, 64, 64, 32, 64, 5, 2, 32, 384, True),    # dec_*/d2 (its slice of d3_0)
    (CONV, 32, 64, 64, 32, 64, 5, 1, 32, 64, True),       # dec_*/d2_0
    (DECONV, 32, 128, 128, 32, 64, 5, 2, 32, 64, True),   # dec_*/d1
    (DECONV, 32, 256, 256, 2, 32, 5, 2, 2, 32, True),     # flow heads
    (DECONV, 32, 256, 256, 1, 32, 5, 2, 1, 32, True),     # depth heads
]
MULTIOBJ_256_CONF = {'use_color': '', 'use_depth': 0.1, 'combination_image': '', 'gen_sep_images': '', 'fully_conv': '', 'image_size': 256}
FC_MULTIOBJ = [(32, 2, 64, 2, 64), (32, 64, 64, 64, 64)]       # the angle MLP (a0, a1 / a2)

# Kernel instances that try_hconv can dispatch but none of the configurations above does.
#   cconv<3x3,256px,N32(,gmask)>: 3x3 stride-1 on >= 512 (16 x 16 tile, 32-filter) tasks -- the 16 x 16-tile instance of the
#   3x3 pipelined kernel, whose data gradient keeps FOUR groups of saved-output loads in flight (with two, the
#   mask of a group was overwritten before its stores left).
EXTRA_KERNEL_CASES = [
    (CONV, 64, 32, 32, 64, 64, 3, 1, 64, 64, True),       # 256 tiles x 2 filter blocks = 512 tasks
    (CONV, 128, 16, 16, 128, 128, 3, 1, 128, 160, True),  # 128 tiles x 4 = 512, output into a wider buffer
]

FC_B64 = [  # B, in, out, x_ld, y_ld
    (64, 4096, 4096, 4096, 4160),     # fc1 (writes into the [fc1, a2] concat buffer)
    (64, 2, 64, 2, 64),               # a0
    (64, 64, 64, 64, 64),             # a1
    (64, 64, 64, 64, 4160),           # a2
    (64, 4160, 4096, 4160, 4096),     # a3
    (64, 4096, 4096, 4096, 4096),     # a4, a5
    (128, 4096, 4096, 4096, 4160),    # Base_Prediction_Model at batch 128
    (128, 4160, 4096, 4160, 4096),
    (128, 64, 64, 64, 4160),
]

FC_HIGHDIM = [(64, 2, 256, 2, 4352), (64, 4352, 4096, 4352, 4096), (64, 2, 19, 2, 19), (64, 2, 128, 2, 128)]   # highdim_angle.py:7-10


def case_id(case):
    return '-'.join(str(c) for c in case)


def record_conv_case(lib_mod, case):
    """Kernel labels the three calls of a layer dispatch (recorded in a plan, nothing is launched; works without a GPU)."""
    kind, n, h, w, c, k, ksz, s, img_ld, feat_ld, need_dx = case
    L = lib_mod.lib()
    g = lib_mod.conv_geom(n, h, w, c, k, ksz, ksz, s, s, img_ld, feat_ld)
    wsb = int(L.conv_workspace_bytes(C.byref(g)))
    ws = 0x40000000
    X, W, Y, B, R = 0x10000000, 0x20000000, 0x30000000, 0x50000000, 0x60000000
    plan = L.plan_create()
    L.plan_begin(plan)
    try:
        if kind == CONV:
            epi = lib_mod.epilogue(B, lib_mod.ACT_LRELU, 0.2)
            L.conv2d_fwd(C.byref(g), X, W, Y, C.byref(epi), ws, wsb, None)
            if need_dx:
                epi = lib_mod.epilogue(None, 0, 0.2, lib_mod.ACT_LRELU, 0.2, R, img_ld)
                L.conv2d_dgrad(C.byref(g), Y, W, X, C.byref(epi), ws, wsb, None)
            L.conv2d_wgrad(C.byref(g), X, Y, W, B, ws, wsb, None)
        else:
            epi = lib_mod.epilogue(None, lib_mod.ACT_LRELU if c > 4 else lib_mod.ACT_NONE, 0.2)
            L.deconv2d_fwd(C.byref(g), X, W, Y, C.byref(epi), ws, wsb, None)
            if need_dx:
                epi = lib_mod.epilogue(None, 0, 0.2, lib_mod.ACT_LRELU, 0.2, R, feat_ld)
                L.deconv2d_dgrad(C.byref(g), Y, W, X, C.byref(epi), ws, wsb, None)
            L.deconv2d_wgrad(C.byref(g), X, Y, W, ws, wsb, None)
    finally:
        L.plan_end()
    labels = [o[0] for o in lib_mod.plan_ops(plan)]
    L.plan_destroy(plan)
    return labels


def record_fc_case(lib_mod, case):
    B, fin, fout, x_ld, y_ld = case
    L = lib_mod.lib()
    wsb = int(L.fc_workspace_bytes(B, fin, fout))
    ws = 0x40000000
    X, W, Y, Bi, R = 0x10000000, 0x20000000, 0x30000000, 0x50000000, 0x60000000
    plan = L.plan_create()
    L.plan_begin(plan)
    try:
        epi = lib_mod.epilogue(Bi, lib_mod.ACT_LRELU, 0.2)
        L.fc_fwd(B, fin, fout, X, x_ld, W, Y, y_ld, C.byref(epi), ws, wsb, None)
        epi = lib_mod.epilogue(None, 0, 0.2, lib_mod.ACT_LRELU, 0.2, R, x_ld)
        L.fc_dgrad(B, fin, fout, Y, y_ld, W, X, x_ld, C.byref(epi), ws, wsb, None)
        L.fc_wgrad(B, fin, fout, X, x_ld, Y, y_ld, W, Bi, ws, wsb, None)
    finally:
        L.plan_end()
    labels = [o[0] for o in lib_mod.plan_ops(plan)]
    L.plan_destroy(plan)
    return labels
