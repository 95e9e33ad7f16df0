// Error reporting and recorded launch plans of libmv3d_hip.so.
#include "common.h"
#include <atomic>
#include <cstdlib>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include <mutex>
#include <set>
#include <string>

struct PlanOp {
    std::function<int(hipStream_t)> fn;
    mv3d::OpInfo info;
    double total_ms;
    int runs;
    bool selected = true;      // bracketed with events when profiling
    int side = 0;              // 1: may run on the side stream of mv3d_plan_run_range2 (see mv3d_plan_side)
};

struct mv3d_plan {
    std::vector<PlanOp> ops;
    bool profile = false;
    std::vector<hipEvent_t> pool;   // 2 events per op per profiled run, collected in bulk
    size_t used = 0;
    hipEvent_t fork[MV3D_MAX_SIDE] = {}, join[MV3D_MAX_SIDE] = {};      // stream dependencies of multi-stream runs
    bool pass_open = false;         // a profiled pass has been started by a range and not yet closed by the range that ends at the last op
};

namespace mv3d {

static std::atomic<int> g_disabled{-1};
int disabled_paths() {
    int v = g_disabled.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = getenv("MV3D_DISABLE");
        v = e ? atoi(e) : 0;
        g_disabled.store(v, std::memory_order_relaxed);
    }
    return v;
}

static thread_local char g_err[512] = "";
static thread_local mv3d_plan* g_rec = nullptr;
static thread_local int g_side = 0;

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
const char* intern_label(const char* fmt, ...) {
    char buf[128];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    static std::mutex mu;
    static std::set<std::string>* labels = new std::set<std::string>();      // never destroyed: plans may outlive static teardown
    std::lock_guard<std::mutex> lk(mu);
    return labels->insert(buf).first->c_str();
}
static thread_local std::vector<FinSegHost>* g_fin = nullptr;
bool finalize_collecting() { return g_fin != nullptr; }
void finalize_open() { if (!g_fin) g_fin = new std::vector<FinSegHost>(); }
void finalize_push(const float* part, int nslab, int64_t count, float* out) { if (g_fin) g_fin->push_back(FinSegHost{part, nslab, count, out}); }
const std::vector<FinSegHost>* finalize_peek() { return g_fin; }
void finalize_take(std::vector<FinSegHost>* out) {
    if (!g_fin) return;
    if (out) out->swap(*g_fin);
    delete g_fin;
    g_fin = nullptr;
}
bool recording() { return g_rec != nullptr; }
void record(std::function<int(hipStream_t)> fn, const OpInfo& info) { g_rec->ops.push_back(PlanOp{std::move(fn), info, 0.0, 0, true, g_side}); }
}  // namespace mv3d

extern "C" {

const char* mv3d_version(void) { return "mv3d_hip 0.3 (gfx950, split-bf16 / fp32 MFMA)"; }
const char* mv3d_last_error(void) { return mv3d::g_err; }

mv3d_plan* mv3d_plan_create(void) { return new mv3d_plan(); }
void mv3d_plan_destroy(mv3d_plan* p) {
    if (!p) return;
    if (mv3d::g_rec == p) mv3d::g_rec = nullptr;
    for (hipEvent_t e : p->pool) (void)hipEventDestroy(e);
    for (int k = 0; k < MV3D_MAX_SIDE; ++k) {
        if (p->fork[k]) (void)hipEventDestroy(p->fork[k]);
        if (p->join[k]) (void)hipEventDestroy(p->join[k]);
    }
    delete p;
}
int mv3d_plan_begin(mv3d_plan* p) {
    if (!p) return mv3d::fail(MV3D_E_INVAL, "mv3d_plan_begin: null plan");
    if (mv3d::g_rec) return mv3d::fail(MV3D_E_INVAL, "mv3d_plan_begin: a plan is already recording on this thread");
    p->ops.clear();
    p->used = 0;
    mv3d::g_rec = p;
    mv3d::g_side = 0;
    return MV3D_OK;
}
// Launches recorded after mv3d_plan_side(1) (until mv3d_plan_side(0)) are tagged as side work: they depend on
// everything recorded before them, and nothing recorded later in the same plan range depends on them.
int mv3d_plan_side(int side) {
    if (side < 0 || side > MV3D_MAX_SIDE) return mv3d::fail(MV3D_E_INVAL, "mv3d_plan_side: side %d not in [0, %d]", side, MV3D_MAX_SIDE);
    mv3d::g_side = side;
    return MV3D_OK;
}
int mv3d_plan_end(void) {
    if (!mv3d::g_rec) return mv3d::fail(MV3D_E_INVAL, "mv3d_plan_end: nothing is recording");
    mv3d::g_rec = nullptr;
    return MV3D_OK;
}
int mv3d_plan_size(const mv3d_plan* p) { return p ? (int)p->ops.size() : 0; }

// Launches ops [begin, end) of the plan.  A profiled pass over the whole plan may be issued as several
// consecutive ranges (data-parallel training interleaves bucket all-reduces): the event pool slot of
// op i is (pass base + 2i), and the pass is closed when a range ends at the last op.
int mv3d_plan_run_range(mv3d_plan* p, int begin, int end, void* stream) {
    return mv3d_plan_run_range_multi(p, begin, end, stream, nullptr, 0, 0);
}

int mv3d_plan_run_range2(mv3d_plan* p, int begin, int end, void* stream, void* side_stream) {
    void* sides[1] = {side_stream};
    return mv3d_plan_run_range_multi(p, begin, end, stream, sides, side_stream ? 1 : 0, 0);
}

// Multi-stream form: launches tagged side k (1..nside) go to side_streams[(k-1) % nside] behind an event on `stream`
// (fork whenever a side's run of launches begins), `stream` waits for every used side stream at the end of the range.
int mv3d_plan_run_range_multi(mv3d_plan* p, int begin, int end, void* stream, void* const* side_streams, int nside, int flags) {
    if (!p) return mv3d::fail(MV3D_E_INVAL, "mv3d_plan_run_range: null plan");
    if (mv3d::g_rec) return mv3d::fail(MV3D_E_INVAL, "mv3d_plan_run_range: cannot run while recording");
    const int n = (int)p->ops.size();
    if (begin < 0 || end > n || begin > end) return mv3d::fail(MV3D_E_INVAL, "mv3d_plan_run_range: bad range [%d, %d) of %d", begin, end, n);
    if (nside < 0 || nside > MV3D_MAX_SIDE || (nside > 0 && !side_streams)) return mv3d::fail(MV3D_E_INVAL, "mv3d_plan_run_range_multi: bad side stream list");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    for (int k = 0; k < nside; ++k)
        if (!p->fork[k]) {
            if (hipEventCreateWithFlags(&p->fork[k], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&p->join[k], hipEventDisableTiming) != hipSuccess)
                return mv3d::fail(MV3D_E_HIP, "mv3d_plan_run_range_multi: hipEventCreate failed");
        }
    int prev_side = 0;
    bool used[MV3D_MAX_SIDE] = {};
    auto stream_of = [&](int i) -> hipStream_t {
        const int tag = p->ops[i].side;
        const int sd = (nside > 0 && tag > 0) ? 1 + (tag - 1) % nside : 0;
        hipStream_t so = sd ? reinterpret_cast<hipStream_t>(side_streams[sd - 1]) : s;
        if (sd && so == s) { prev_side = 0; return s; }
        if (sd && sd != prev_side) {                         // fork: side work sees everything issued on the main stream so far
            (void)hipEventRecord(p->fork[sd - 1], s);
            (void)hipStreamWaitEvent(so, p->fork[sd - 1], 0);
            used[sd - 1] = true;
        }
        prev_side = sd;
        return so;
    };
    auto join = [&]() {
        if (flags & MV3D_RUN_NO_JOIN) return;
        for (int k = 0; k < nside; ++k)
            if (used[k]) {
                (void)hipEventRecord(p->join[k], reinterpret_cast<hipStream_t>(side_streams[k]));
                (void)hipStreamWaitEvent(s, p->join[k], 0);
            }
    };
    const int held = (flags & MV3D_RUN_HOLD_CLASS2) ? 2 : -1;      // the caller issues that class itself (mv3d_plan_run_side)
    if (!p->profile) {
        for (int i = begin; i < end; ++i) {
            if (p->ops[i].side == held) continue;
            int rc = p->ops[i].fn(stream_of(i));
            if (rc != MV3D_OK) return rc;
        }
        join();
        return MV3D_OK;
    }
    // profiled run: bracket every launch with HIP events on the launch stream; no host sync here
    const size_t need = p->used + 2 * (size_t)n;
    while (p->pool.size() < need) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return mv3d::fail(MV3D_E_HIP, "mv3d_plan_run: hipEventCreate failed");
        p->pool.push_back(e);
    }
    if (!p->pass_open) {
        // a pass that starts behind op 0 (the caller left the first launches out): their events are recorded back to back, so
        // that the collection finds a (zero) duration instead of the stale events of an earlier pass
        for (int i = 0; i < begin; ++i)
            if (p->ops[i].selected) { (void)hipEventRecord(p->pool[p->used + 2 * i], s); (void)hipEventRecord(p->pool[p->used + 2 * i + 1], s); }
        p->pass_open = true;
    }
    for (int i = begin; i < end; ++i) {
        if (p->ops[i].side == held) continue;
        const bool sel = p->ops[i].selected;
        hipStream_t so = stream_of(i);
        if (sel) (void)hipEventRecord(p->pool[p->used + 2 * i], so);
        int rc = p->ops[i].fn(so);
        if (sel) (void)hipEventRecord(p->pool[p->used + 2 * i + 1], so);
        if (rc != MV3D_OK) return rc;
    }
    join();
    if (end == n) { p->used = need; p->pass_open = false; }
    return MV3D_OK;
}

// The launches of side class `cls` that a run with MV3D_RUN_HOLD_CLASS2 left out, in recorded order on `stream` (the caller has
// ordered it behind whatever they depend on).  Profiled plans: the events go into the slots of the pass that run just closed.
int mv3d_plan_run_side(mv3d_plan* p, int cls, void* stream) {
    if (!p) return mv3d::fail(MV3D_E_INVAL, "mv3d_plan_run_side: null plan");
    if (mv3d::g_rec) return mv3d::fail(MV3D_E_INVAL, "mv3d_plan_run_side: cannot run while recording");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const size_t n = p->ops.size();
    const bool prof = p->profile && p->used >= 2 * n;
    for (size_t i = 0; i < n; ++i) {
        if (p->ops[i].side != cls) continue;
        const bool sel = prof && p->ops[i].selected;
        if (sel) (void)hipEventRecord(p->pool[p->used - 2 * n + 2 * i], s);
        int rc = p->ops[i].fn(s);
        if (sel) (void)hipEventRecord(p->pool[p->used - 2 * n + 2 * i + 1], s);
        if (rc != MV3D_OK) return rc;
    }
    return MV3D_OK;
}

int mv3d_plan_run(mv3d_plan* p, void* stream) {
    if (!p) return mv3d::fail(MV3D_E_INVAL, "mv3d_plan_run: null plan");
    return mv3d_plan_run_range(p, 0, (int)p->ops.size(), stream);
}

int mv3d_plan_profile(mv3d_plan* p, int enable) {
    if (!p) return mv3d::fail(MV3D_E_INVAL, "mv3d_plan_profile: null plan");
    p->profile = enable != 0;
    return MV3D_OK;
}

// Host-synchronising: folds all profiled runs since the last collect into per-op totals.
int mv3d_plan_profile_collect(mv3d_plan* p) {
    if (!p) return mv3d::fail(MV3D_E_INVAL, "mv3d_plan_profile_collect: null plan");
    const size_t n = p->ops.size();
    if (n == 0 || p->used == 0) return MV3D_OK;
    size_t last = n;
    for (size_t i = 0; i < n; ++i) if (p->ops[i].selected) last = i;
    if (last == n) { p->used = 0; return MV3D_OK; }
    if (hipEventSynchronize(p->pool[p->used - 2 * n + 2 * last + 1]) != hipSuccess)
        return mv3d::fail(MV3D_E_HIP, "mv3d_plan_profile_collect: hipEventSynchronize failed");
    for (size_t base = 0; base + 2 * n <= p->used; base += 2 * n)
        for (size_t i = 0; i < n; ++i) {
            if (!p->ops[i].selected) continue;
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, p->pool[base + 2 * i], p->pool[base + 2 * i + 1]) == hipSuccess) {
                p->ops[i].total_ms += ms;
                p->ops[i].runs += 1;
            }
        }
    p->used = 0;
    (void)hipGetLastError();      // an unrecorded event pair (held side class) is not an error of the next launch
    return MV3D_OK;
}

// Restrict event bracketing to the launches whose kernel label matches `name` (NULL = all launches): one label, or a
// '|'-separated list of labels, each optionally ending in '*' (prefix match) -- e.g. "bconv*|wgrad_b3*|reduce_slabs" for
// the conv / deconv family.  Timing a subset costs a handful of events per step, so it can stay on inside a timed region.
static bool label_matches(const char* label, const char* pats) {
    const size_t ll = strlen(label);
    for (const char* p = pats; *p;) {
        const char* e = strchr(p, '|');
        size_t n = e ? (size_t)(e - p) : strlen(p);
        if (n > 0 && p[n - 1] == '*') { if (ll >= n - 1 && strncmp(label, p, n - 1) == 0) return true; }
        else if (ll == n && strncmp(label, p, n) == 0) return true;
        p += n + (e ? 1 : 0);
    }
    return false;
}

int mv3d_plan_profile_select(mv3d_plan* p, const char* name) {
    if (!p) return mv3d::fail(MV3D_E_INVAL, "mv3d_plan_profile_select: null plan");
    if (p->used) return mv3d::fail(MV3D_E_INVAL, "mv3d_plan_profile_select: collect pending runs first");
    for (auto& o : p->ops) o.selected = (name == nullptr) || label_matches(o.info.name, name);
    return MV3D_OK;
}

int mv3d_plan_profile_reset(mv3d_plan* p) {
    if (!p) return mv3d::fail(MV3D_E_INVAL, "mv3d_plan_profile_reset: null plan");
    for (auto& o : p->ops) { o.total_ms = 0.0; o.runs = 0; }
    p->used = 0;
    return MV3D_OK;
}

// Replaces the diagnostics mask (initially MV3D_DISABLE); returns the previous one.  Affects dispatch decisions taken
// afterwards: plans recorded earlier keep the kernels they were recorded with.
int mv3d_set_diagnostics(int mask) {
    const int old = mv3d::disabled_paths();
    mv3d::g_disabled.store(mask < 0 ? 0 : mask, std::memory_order_relaxed);
    return old;
}

// CRC-32C (Castagnoli), the checksum of the TFRecord framing (host only; slicing-by-8 tables built on first use).
uint32_t mv3d_crc32c(const void* data, size_t n) {
    static uint32_t T[8][256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            T[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int t = 1; t < 8; ++t) T[t][i] = (T[t - 1][i] >> 8) ^ T[0][T[t - 1][i] & 0xFF];
        ready = true;
    }
    const unsigned char* p = static_cast<const unsigned char*>(data);
    uint32_t c = 0xFFFFFFFFu;
    while (n >= 8) {
        uint32_t lo, hi;
        memcpy(&lo, p, 4); memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = T[7][lo & 0xFF] ^ T[6][(lo >> 8) & 0xFF] ^ T[5][(lo >> 16) & 0xFF] ^ T[4][lo >> 24] ^
            T[3][hi & 0xFF] ^ T[2][(hi >> 8) & 0xFF] ^ T[1][(hi >> 16) & 0xFF] ^ T[0][hi >> 24];
        p += 8; n -= 8;
    }
    while (n--) c = (c >> 8) ^ T[0][(c ^ *p++) & 0xFF];
    return c ^ 0xFFFFFFFFu;
}

int mv3d_plan_op_info(const mv3d_plan* p, int i, const char** name, double* flops, double* bytes, double* total_ms, int* runs) {
    if (!p || i < 0 || i >= (int)p->ops.size()) return mv3d::fail(MV3D_E_INVAL, "mv3d_plan_op_info: bad index");
    const PlanOp& o = p->ops[i];
    if (name) *name = o.info.name;
    if (flops) *flops = o.info.flops;
    if (bytes) *bytes = o.info.bytes;
    if (total_ms) *total_ms = o.total_ms;
    if (runs) *runs = o.runs;
    return MV3D_OK;
}

}  // extern "C"
