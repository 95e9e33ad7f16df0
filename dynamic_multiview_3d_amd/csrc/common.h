// Internal helpers shared by the HIP translation units of libmv3d_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <functional>
#include "../../include/mv3d_hip.h"

namespace mv3d {

int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
// What one recorded launch is, for the bench's roofline accounting: kernel label, algorithmic
// FLOPs and algorithmic HBM bytes of this launch (DESIGN.md states the formulas).
struct OpInfo {
    const char* name;
    double flops;
    double bytes;
};

// Kernel labels live for the life of the process (OpInfo keeps the pointer): intern_label returns one stable copy per
// distinct text.  A label names ONE kernel instance (template arguments included), so that the label-coverage test
// (tests/test_label_coverage.py) can tie the launches of a recorded step to the kernels the parity tests ran.
const char* intern_label(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

bool recording();
void record(std::function<int(hipStream_t)> fn, const OpInfo& info);

// Launch now, or append to the plan being recorded on this thread (mv3d_plan_begin).
// Every dispatch() is exactly ONE kernel launch.
template <class F>
inline int dispatch(void* stream, const OpInfo& info, F fn) {
    if (recording()) { record(std::function<int(hipStream_t)>(fn), info); return MV3D_OK; }
    return fn(reinterpret_cast<hipStream_t>(stream));
}

inline int launched(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MV3D_E_HIP, "%s: %s", what, hipGetErrorString(e));
    return MV3D_OK;
}

// Deferred slab reductions (mv3d_grad_finalize_*, elem.hip): while a collection is open on this thread the filter-gradient entry
// points hand their per-slab partial sums to finalize_push() instead of launching reduce_slabs_kernel per layer; nslab == 0
// marks a gradient that is already final at `out` (a single slab written in place).
struct FinSegHost { const float* part; int nslab; int64_t count; float* out; };
bool finalize_collecting();
void finalize_push(const float* part, int nslab, int64_t count, float* out);
void finalize_open();
}  // namespace mv3d
#include <vector>
namespace mv3d {
const std::vector<FinSegHost>* finalize_peek();             // nullptr when no collection is open
void finalize_take(std::vector<FinSegHost>* out);          // closes the collection (out may be null)

// Diagnostics mask (dispatch rungs switched off): MV3D_DISABLE at load time, mv3d_set_diagnostics()
// at run time.  One definition for all translation units (core.hip).
int disabled_paths();

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// TF 'SAME' padding: pad_before = total/2
static inline void same_pad(int size, int k, int s, int* out, int* before) {
    int o = (size + s - 1) / s;
    int total = (o - 1) * s + k - size;
    if (total < 0) total = 0;
    *out = o;
    *before = total / 2;
}

// ---- device-side activation helpers -------------------------------------------------------
// Forward: f1*x + f2*|x| exactly as tf_utils.py:25-33 writes it (two products, one sum).
// RELU stores -0.0f for x < 0 so the backward pass can tell x < 0 (slope 0) from x == 0
// (slope 0.5, TF's sign(0) = 0); -0.0f is arithmetically identical to +0.0f downstream.
__device__ __forceinline__ float act_apply(float x, int act, float leak) {
    if (act == MV3D_ACT_LRELU) {
        float f1 = 0.5f * (1.0f + leak), f2 = 0.5f * (1.0f - leak);
        return __fadd_rn(__fmul_rn(f1, x), __fmul_rn(f2, fabsf(x)));
    }
    if (act == MV3D_ACT_RELU) {
        float y = __fadd_rn(__fmul_rn(0.5f, x), __fmul_rn(0.5f, fabsf(x)));
        return x < 0.0f ? -0.0f : y;
    }
    if (act == MV3D_ACT_TANH) return tanhf(x);
    return x;
}

// Derivative of the activation evaluated from its OUTPUT y.
__device__ __forceinline__ float act_grad_from_out(float y, int act, float leak) {
    if (act == MV3D_ACT_LRELU) {
        float f1 = 0.5f * (1.0f + leak), f2 = 0.5f * (1.0f - leak);
        float s = (y > 0.0f) ? 1.0f : ((y < 0.0f) ? -1.0f : 0.0f);   // sign(out) == sign(pre)
        return f1 + f2 * s;
    }
    if (act == MV3D_ACT_RELU) {
        if (y > 0.0f) return 1.0f;
        return (__float_as_uint(y) >> 31) ? 0.0f : 0.5f;               // -0.0 marks pre < 0
    }
    if (act == MV3D_ACT_TANH) return 1.0f - y * y;
    return 1.0f;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

}  // namespace mv3d
