// Error reporting and recorded launch plans of libmv3d_hip.so.
#include "common.h"
#include <stdarg.h>
#include <stdio.h>
#include <vector>

struct mv3d_plan {
    std::vector<std::function<int(hipStream_t)>> ops;
};

namespace mv3d {
static thread_local char g_err[512] = "";
static thread_local mv3d_plan* g_rec = nullptr;

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
bool recording() { return g_rec != nullptr; }
void record(std::function<int(hipStream_t)> fn) { g_rec->ops.push_back(std::move(fn)); }
}  // namespace mv3d

extern "C" {

const char* mv3d_version(void) { return "mv3d_hip 0.1 (gfx950, fp32 MFMA)"; }
const char* mv3d_last_error(void) { return mv3d::g_err; }

mv3d_plan* mv3d_plan_create(void) { return new mv3d_plan(); }
void mv3d_plan_destroy(mv3d_plan* p) {
    if (mv3d::g_rec == p) mv3d::g_rec = nullptr;
    delete p;
}
int mv3d_plan_begin(mv3d_plan* p) {
    if (!p) return mv3d::fail(MV3D_E_INVAL, "mv3d_plan_begin: null plan");
    if (mv3d::g_rec) return mv3d::fail(MV3D_E_INVAL, "mv3d_plan_begin: a plan is already recording on this thread");
    p->ops.clear();
    mv3d::g_rec = p;
    return MV3D_OK;
}
int mv3d_plan_end(void) {
    if (!mv3d::g_rec) return mv3d::fail(MV3D_E_INVAL, "mv3d_plan_end: nothing is recording");
    mv3d::g_rec = nullptr;
    return MV3D_OK;
}
int mv3d_plan_size(const mv3d_plan* p) { return p ? (int)p->ops.size() : 0; }
int mv3d_plan_run(const mv3d_plan* p, void* stream) {
    if (!p) return mv3d::fail(MV3D_E_INVAL, "mv3d_plan_run: null plan");
    if (mv3d::g_rec) return mv3d::fail(MV3D_E_INVAL, "mv3d_plan_run: cannot run while recording");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    for (size_t i = 0; i < p->ops.size(); ++i) {
        int rc = p->ops[i](s);
        if (rc != MV3D_OK) return rc;
    }
    return MV3D_OK;
}

}  // extern "C"
