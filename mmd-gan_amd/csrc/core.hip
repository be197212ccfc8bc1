// error state, version, device probe, handles (workspace / prezeroed mode / launch plans / events)
#include "common.h"
#include "slab_reduce.h"
#include "tuning.h"

#include <cxxabi.h>

#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace mmdgan {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// A launch plan: the recorded work of one static step, cut into segments at mmdgan_plan_mark() so that a caller can run
// work the library does not issue (an RCCL collective) between two segments.
struct Plan {
    std::vector<std::function<void()>> nodes;
    std::vector<size_t> segment_end;          // node count at the end of each closed segment
    std::vector<hipEvent_t> events;           // one per recorded stream_wait node (re-used on every replay), freed with the plan
    struct KernelNote { const void *fn; unsigned grid, block; hipStream_t st; };
    std::vector<KernelNote> kernels;          // every kernel launch of the plan in issue order (mmdgan_plan_describe)
    unsigned comm_generation = 0;             // of the library's communicator when a collective was recorded (0: none recorded)
    ~Plan() {
        for (hipEvent_t e : events)
            if (e) (void)hipEventDestroy(e);
    }
};
unsigned comm_generation();                   // comm.hip: bumped by every mmdgan_comm_init / _destroy

}  // namespace mmdgan

// One handle per engine: the state that used to be process-global.  A handle is used by one thread at a time; distinct
// handles are independent.  Entry points act on the calling thread's CURRENT handle (mmdgan_make_current); a thread that
// never made one current uses the process default handle, which keeps single-engine callers as simple as before.
struct mmdgan_handle {
    void *ws = nullptr;
    size_t ws_bytes = 0;
    bool prezeroed = false;
    struct WsSlot { bool used = false; hipStream_t st = nullptr; } ws_slot[2];   // halves of the workspace: their latest users
    // mmdgan_wgrad_defer: the slab reduction the last weight-gradient launch left for the next one on its stream
    bool defer = false;
    struct Pending {
        bool active = false;
        mmdgan::SlabReduceArgs args{};
        hipStream_t st = nullptr;
        int half = 0, sub = 0;         // where its slabs lie: half of the workspace, ping-pong part of that half
    } pending;
    int acq_half = 0, acq_sub = -1;    // what the last wgrad_slabs_acquire handed out (sub -1: an ordinary workspace_acquire)
    std::unique_ptr<mmdgan::Plan> recording;
    std::vector<std::unique_ptr<mmdgan::Plan>> plans;      // plan id = index (destroyed plans leave a null slot)
    std::vector<hipEvent_t> pool;                          // round-robin pool of the un-recorded stream_wait calls
    size_t pool_next = 0;
    std::vector<hipEvent_t> slots;                         // named slots of mmdgan_event_record / _wait
    ~mmdgan_handle() {
        for (auto *v : {&pool, &slots})
            for (hipEvent_t e : *v)
                if (e) (void)hipEventDestroy(e);
    }
};

namespace mmdgan {
static mmdgan_handle g_default_handle;
static thread_local mmdgan_handle *t_current = nullptr;
static inline mmdgan_handle &cur() { return t_current ? *t_current : g_default_handle; }

bool outputs_prezeroed() { return cur().prezeroed; }
void *workspace(size_t need) {
    mmdgan_handle &h = cur();
    return (h.ws && need <= h.ws_bytes) ? h.ws : nullptr;
}
// The workspace is used as TWO halves (or whole, by a request that does not fit a half).  A stream keeps the half it used
// last; another stream takes the other one - the two launch chains of a training step (input-gradients on the main stream,
// weight gradients beside them) then never touch each other's partial sums and need no ordering.  Whoever must take a half
// (or the whole buffer) after ANOTHER stream is ordered behind that stream first (event record + wait, recorded into a
// plan like any dependency), so concurrent chains serialise there instead of corrupting each other.
static int ws_half_for(const mmdgan_handle &h, hipStream_t st) {
    return (h.ws_slot[0].used && h.ws_slot[0].st == st) ? 0 : (h.ws_slot[1].used && h.ws_slot[1].st == st) ? 1
         : !h.ws_slot[0].used ? 0 : !h.ws_slot[1].used ? 1 : 0;
}
static bool ws_take(mmdgan_handle &h, int i, hipStream_t st) {           // slot i goes to `st`, behind its previous user
    mmdgan_handle::WsSlot &s = h.ws_slot[i];
    if (s.used && s.st != st && mmdgan_stream_wait((void *)st, (void *)s.st) != MMDGAN_OK) return false;
    s.used = true;
    s.st = st;
    return true;
}
void *workspace_acquire(size_t need, hipStream_t st) {
    mmdgan_handle &h = cur();
    if (!h.ws || need > h.ws_bytes) return nullptr;
    const size_t half = (h.ws_bytes / 2) & ~(size_t)255;
    h.acq_sub = -1;
    if (need > half) {
        if (h.pending.active && wgrad_flush_pending() != 0) return nullptr;
        return (ws_take(h, 0, st) && ws_take(h, 1, st)) ? h.ws : nullptr;
    }
    const int i = ws_half_for(h, st);
    // un-summed slabs in the half this request gets: their stand-alone pass goes first (on their own stream; ws_take orders
    // `st` behind that stream if it is another one)
    if (h.pending.active && h.pending.half == i && wgrad_flush_pending() != 0) return nullptr;
    h.acq_half = i;
    return ws_take(h, i, st) ? (char *)h.ws + (size_t)i * half : nullptr;
}
bool wgrad_deferred() { return cur().defer; }
int wgrad_flush_pending() {
    mmdgan_handle &h = cur();
    if (!h.pending.active) return 0;
    h.pending.active = false;
    slab_reduce_launch(h.pending.args, h.pending.st);
    return check_launch("slab reduction (deferred)");
}
void *wgrad_slabs_acquire(size_t need, hipStream_t st, SlabReduceArgs *prev) {
    mmdgan_handle &h = cur();
    *prev = SlabReduceArgs{};
    if (!h.ws) return nullptr;
    const size_t half = (h.ws_bytes / 2) & ~(size_t)255, part = (half / 2) & ~(size_t)255;
    if (!h.defer || need > part) return workspace_acquire(need, st);       // (flushes what is pending in its way)
    if (h.pending.active && h.pending.st != st && wgrad_flush_pending() != 0) return nullptr;
    const int i = ws_half_for(h, st);
    if (h.pending.active && h.pending.half != i && wgrad_flush_pending() != 0) return nullptr;   // (a stream keeps its half: not expected)
    // take the half FIRST: if that fails (a stream_wait error) the pending reduction stays pending - the caller falls through to
    // a non-slab kernel, whose own workspace request / the next flush still sums the previous layer's slabs
    if (!ws_take(h, i, st)) return nullptr;
    int sub = 0;
    if (h.pending.active) {                                // same stream, same half: the caller's prologue sums it
        *prev = h.pending.args;
        sub = 1 - h.pending.sub;
        h.pending.active = false;
    }
    h.acq_half = i;
    h.acq_sub = sub;
    return (char *)h.ws + (size_t)i * half + (size_t)sub * part;
}
int wgrad_slabs_release(const SlabReduceArgs &mine, hipStream_t st) {
    mmdgan_handle &h = cur();
    if (h.defer && h.acq_sub >= 0) {
        h.pending.active = true;
        h.pending.args = mine;
        h.pending.st = st;
        h.pending.half = h.acq_half;
        h.pending.sub = h.acq_sub;
        return 0;
    }
    slab_reduce_launch(mine, st);
    return check_launch("slab reduction");
}
// Dependency events: hipEventDisableTiming.  (Round 4 A/B on the CIFAR step, eager / plan ms: plain 1.895 / 1.887,
// + hipEventDisableSystemFence 1.888 / 1.885, + hipEventReleaseToDevice 1.892 / 1.891 - inside the run-to-run spread, so the
// flag with the documented visibility guarantees stays.)
constexpr unsigned kEventFlags = hipEventDisableTiming;
bool plan_recording() { return cur().recording != nullptr; }
void plan_push(std::function<void()> &&node) { cur().recording->nodes.emplace_back(std::move(node)); }
void plan_note_kernel(const void *fn, dim3 grid, dim3 block, hipStream_t st) {
    cur().recording->kernels.push_back({fn, grid.x * grid.y * grid.z, block.x * block.y * block.z, st});
}
void plan_note_collective() { cur().recording->comm_generation = comm_generation(); }

hipError_t memset_async(void *p, int value, size_t bytes, hipStream_t st) {
    hipError_t e = hipMemsetAsync(p, value, bytes, st);
    if (e == hipSuccess && plan_recording()) plan_push([=]() { (void)hipMemsetAsync(p, value, bytes, st); });
    return e;
}
}  // namespace mmdgan

using namespace mmdgan;

extern "C" const char *mmdgan_last_error(void) { return mmdgan::g_err; }
extern "C" int mmdgan_version(void) { return MMDGAN_VERSION; }
// "name=value ..." of every kernel-selection switch of this process (csrc/tuning.h), the ones off their default marked '*'
extern "C" long mmdgan_tuning_describe(char *buf, size_t cap) {
    const Tuning &t = tuning(), &d = tuning_defaults();
    std::string out;
    auto add = [&](const char *name, long v, long dv) {
        char tmp[96];
        snprintf(tmp, sizeof(tmp), "%s%s=%ld%s", out.empty() ? "" : " ", name, v, v != dv ? "*" : "");
        out += tmp;
    };
    add("force_direct", t.force_direct, d.force_direct); add("thin_valu", t.thin_valu, d.thin_valu);
    add("wino", t.wino, d.wino); add("wino_min_tiles", t.wino_min_tiles, d.wino_min_tiles);
    add("wino_ksplit_below", t.wino_ksplit_below, d.wino_ksplit_below); add("wino_wgrad", t.wino_wgrad, d.wino_wgrad);
    add("wino_wgrad_slab", t.wino_wgrad_slab, d.wino_wgrad_slab); add("wino2", t.wino2, d.wino2);
    add("wino2_ksplit", t.wino2_ksplit, d.wino2_ksplit); add("wino2_ksplit_below", t.wino2_ksplit_below, d.wino2_ksplit_below);
    add("wino2_wgrad", t.wino2_wgrad, d.wino2_wgrad); add("wino2_wgrad_min_tiles", t.wino2_wgrad_min_tiles, d.wino2_wgrad_min_tiles);
    add("wgrad_cus", t.wgrad_cus, d.wgrad_cus); add("gemm_skinny", t.gemm_skinny, d.gemm_skinny); add("gemm_panel", t.gemm_panel, d.gemm_panel); add("mmd_d16", t.mmd_d16, d.mmd_d16);
    add("wino43", t.wino43, d.wino43); add("wino43_min_tiles", t.wino43_min_tiles, d.wino43_min_tiles);
    add("wino43_ksplit_below", t.wino43_ksplit_below, d.wino43_ksplit_below);
    add("wino43_wgrad", t.wino43_wgrad, d.wino43_wgrad); add("wino43_wgrad_min_tiles", t.wino43_wgrad_min_tiles, d.wino43_wgrad_min_tiles);
    add("wino43_wgrad_cus", t.wino43_wgrad_cus, d.wino43_wgrad_cus);
    if (buf && cap > 0) {
        const size_t n = out.size() < cap - 1 ? out.size() : cap - 1;
        memcpy(buf, out.data(), n);
        buf[n] = 0;
    }
    return (long)out.size() + 1;
}
extern "C" int mmdgan_device_ok(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { (void)hipGetLastError(); return 0; }
    hipDeviceProp_t p;
    if (hipGetDeviceProperties(&p, 0) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

extern "C" int mmdgan_create(mmdgan_handle **out) {
    MMDGAN_REQUIRE(out, "create: null pointer");
    *out = new mmdgan_handle();
    return MMDGAN_OK;
}
extern "C" int mmdgan_destroy(mmdgan_handle *h) {
    if (!h) return MMDGAN_OK;
    MMDGAN_REQUIRE(h != &g_default_handle, "destroy: not a handle of mmdgan_create");
    if (t_current == h) t_current = nullptr;
    delete h;
    return MMDGAN_OK;
}
extern "C" int mmdgan_make_current(mmdgan_handle *h) {
    t_current = h;                     // NULL: back to the process default handle
    return MMDGAN_OK;
}
extern "C" int mmdgan_set_workspace(void *ptr, size_t bytes) {
    cur().ws = ptr;
    cur().ws_bytes = ptr ? bytes : 0;
    cur().ws_slot[0] = cur().ws_slot[1] = mmdgan_handle::WsSlot();
    cur().pending.active = false;      // (slabs of a buffer that is being replaced or re-registered: nothing to sum)
    return MMDGAN_OK;
}
// Deferred slab reduction of the Winograd-domain weight gradients (slab_reduce.h).  on = 1: a slab weight-gradient launch no
// longer appends its reduction pass; the NEXT such launch on the same stream sums the slabs in its prologue.  dw / dbias /
// dot of a call are therefore complete only after the stream's next weight-gradient call, after mmdgan_wgrad_flush(), after
// any other workspace user of that stream, or after mmdgan_wgrad_defer(0) - all of which issue what is pending.
extern "C" int mmdgan_wgrad_defer(int on) {
    cur().defer = on != 0;                     // (issue-time state, like the workspace bookkeeping: a recorded step replays the
    return on ? MMDGAN_OK : (wgrad_flush_pending() ? MMDGAN_E_LAUNCH : MMDGAN_OK);   // launches that were chosen under it)
}
extern "C" int mmdgan_wgrad_flush(void) { return wgrad_flush_pending() ? MMDGAN_E_LAUNCH : MMDGAN_OK; }

extern "C" int mmdgan_set_outputs_prezeroed(int on) {
    const bool v = on != 0;
    if (plan_recording()) {            // a mode switch inside a recorded step is part of the step
        mmdgan_handle *h = &cur();
        plan_push([h, v]() { h->prezeroed = v; });
    }
    cur().prezeroed = v;
    return MMDGAN_OK;
}

// ---- stream plumbing a recorded step needs (what torch's wait_stream / Event / zero_ / copy_ do, as library calls) ----
extern "C" int mmdgan_stream_wait(void *waiting_stream, void *signalling_stream) {
    // everything issued so far on `signalling_stream` completes before anything issued later on `waiting_stream` starts
    hipStream_t w = (hipStream_t)waiting_stream, s = (hipStream_t)signalling_stream;
    if (w == s) return MMDGAN_OK;
    mmdgan_handle &h = cur();
    hipEvent_t ev = nullptr;
    if (plan_recording()) {                    // the node keeps an event of its own
        if (hipEventCreateWithFlags(&ev, kEventFlags) != hipSuccess) return check_launch("stream_wait event");
        h.recording->events.push_back(ev);
    } else {                                   // a wait captures the event's state when it is issued: re-use is safe
        constexpr size_t kPool = 64;
        if (h.pool.size() < kPool) {
            if (hipEventCreateWithFlags(&ev, kEventFlags) != hipSuccess) return check_launch("stream_wait event");
            h.pool.push_back(ev);
        } else {
            ev = h.pool[h.pool_next++ % kPool];
        }
    }
    if (hipEventRecord(ev, s) != hipSuccess || hipStreamWaitEvent(w, ev, 0) != hipSuccess) return check_launch("stream_wait");
    if (plan_recording()) plan_push([=]() { (void)hipEventRecord(ev, s); (void)hipStreamWaitEvent(w, ev, 0); });
    return MMDGAN_OK;
}

// named events: record at one point of a stream, wait for that point later from another stream
extern "C" int mmdgan_event_record(int slot, void *stream) {
    MMDGAN_REQUIRE(slot >= 0 && slot < 64, "event_record: slot %d outside [0,64)", slot);
    std::vector<hipEvent_t> &t = cur().slots;
    if ((int)t.size() <= slot) t.resize(slot + 1, nullptr);
    if (!t[slot] && hipEventCreateWithFlags(&t[slot], kEventFlags) != hipSuccess) return check_launch("event_record");
    hipEvent_t ev = t[slot];
    hipStream_t s = (hipStream_t)stream;
    if (hipEventRecord(ev, s) != hipSuccess) return check_launch("event_record");
    if (plan_recording()) plan_push([=]() { (void)hipEventRecord(ev, s); });
    return MMDGAN_OK;
}
extern "C" int mmdgan_event_wait(int slot, void *stream) {
    std::vector<hipEvent_t> &t = cur().slots;
    MMDGAN_REQUIRE(slot >= 0 && slot < (int)t.size() && t[slot], "event_wait: slot %d was never recorded", slot);
    hipEvent_t ev = t[slot];
    hipStream_t s = (hipStream_t)stream;
    if (hipStreamWaitEvent(s, ev, 0) != hipSuccess) return check_launch("event_wait");
    if (plan_recording()) plan_push([=]() { (void)hipStreamWaitEvent(s, ev, 0); });
    return MMDGAN_OK;
}
extern "C" int mmdgan_memset_zero(void *ptr, size_t bytes, void *stream) {
    MMDGAN_REQUIRE(ptr || bytes == 0, "memset_zero: null pointer");
    if (bytes == 0) return MMDGAN_OK;
    if (memset_async(ptr, 0, bytes, (hipStream_t)stream) != hipSuccess) return check_launch("memset_zero");
    return MMDGAN_OK;
}
// several small buffers in ONE launch (a step zeroes half a dozen scratch buffers before its first kernel: six memset
// dispatches of 4-11 us each on the critical path, or one of these)
namespace mmdgan {
constexpr int kZeroMultiMax = 16;
struct ZeroTable {
    uint4 *ptr[kZeroMultiMax];
    unsigned first_block[kZeroMultiMax + 1];     // blocks of 256 threads x 16 bytes, prefix sums
    unsigned long long vec16[kZeroMultiMax];     // 16-byte words per buffer
    int n;
};
__global__ __launch_bounds__(256) void zero_multi_kernel(ZeroTable t) {
    int i = 0;
    while (i + 1 < t.n && blockIdx.x >= t.first_block[i + 1]) ++i;
    const unsigned long long e = (unsigned long long)(blockIdx.x - t.first_block[i]) * 256 + threadIdx.x;
    if (e < t.vec16[i]) t.ptr[i][e] = make_uint4(0u, 0u, 0u, 0u);
}
}  // namespace mmdgan
extern "C" int mmdgan_memset_zero_multi(void *const *ptrs, const size_t *bytes, int n, void *stream) {
    MMDGAN_REQUIRE(n >= 0 && (n == 0 || (ptrs && bytes)), "memset_zero_multi: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    int i = 0;
    while (i < n) {
        ZeroTable t;
        t.n = 0;
        unsigned blocks = 0;
        for (; i < n && t.n < kZeroMultiMax; ++i) {
            if (bytes[i] == 0) continue;
            MMDGAN_REQUIRE(ptrs[i], "memset_zero_multi: null pointer");
            if (((uintptr_t)ptrs[i] & 15) || (bytes[i] & 15) || bytes[i] > ((size_t)1 << 30)) {      // odd or huge: a plain memset
                if (memset_async(ptrs[i], 0, bytes[i], st) != hipSuccess) return check_launch("memset_zero_multi");
                continue;
            }
            t.ptr[t.n] = (uint4 *)ptrs[i];
            t.vec16[t.n] = bytes[i] / 16;
            t.first_block[t.n] = blocks;
            blocks += (unsigned)((bytes[i] / 16 + 255) / 256);
            ++t.n;
        }
        if (t.n == 0) continue;
        t.first_block[t.n] = blocks;
        hipLaunchKernelGGL(zero_multi_kernel, dim3(blocks), dim3(256), 0, st, t);
        if (int rc = check_launch("memset_zero_multi")) return rc;
    }
    return MMDGAN_OK;
}
extern "C" int mmdgan_copy(void *dst, const void *src, size_t bytes, void *stream) {
    MMDGAN_REQUIRE((dst && src) || bytes == 0, "copy: null pointer");
    if (bytes == 0) return MMDGAN_OK;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return check_launch("copy");
    if (plan_recording()) plan_push([=]() { (void)hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st); });
    return MMDGAN_OK;
}

// ---- launch plans ----------------------------------------------------------------------------------------------
extern "C" int mmdgan_plan_begin(void) {
    MMDGAN_REQUIRE(!plan_recording(), "plan_begin: already recording");
    cur().recording.reset(new Plan());
    return MMDGAN_OK;
}
extern "C" int mmdgan_plan_mark(void) {
    MMDGAN_REQUIRE(plan_recording(), "plan_mark: not recording");
    Plan &p = *cur().recording;
    p.segment_end.push_back(p.nodes.size());
    return (int)p.segment_end.size();          // index of the segment that starts here
}
extern "C" int mmdgan_plan_end(int *plan_id) {
    MMDGAN_REQUIRE(plan_recording() && plan_id, "plan_end: not recording");
    mmdgan_handle &h = cur();
    h.recording->segment_end.push_back(h.recording->nodes.size());
    h.plans.emplace_back(std::move(h.recording));
    *plan_id = (int)h.plans.size() - 1;
    return MMDGAN_OK;
}
extern "C" int mmdgan_plan_abort(void) {
    cur().recording.reset();
    return MMDGAN_OK;
}
extern "C" int mmdgan_plan_segments(int plan_id) {
    mmdgan_handle &h = cur();
    if (plan_id < 0 || plan_id >= (int)h.plans.size() || !h.plans[plan_id]) return MMDGAN_E_ARG;
    return (int)h.plans[plan_id]->segment_end.size();
}
extern "C" long mmdgan_plan_nodes(int plan_id) {
    mmdgan_handle &h = cur();
    if (plan_id < 0 || plan_id >= (int)h.plans.size() || !h.plans[plan_id]) return MMDGAN_E_ARG;
    return (long)h.plans[plan_id]->nodes.size();
}
extern "C" long mmdgan_plan_describe(int plan_id, char *buf, size_t cap) {
    mmdgan_handle &h = cur();
    if (plan_id < 0 || plan_id >= (int)h.plans.size() || !h.plans[plan_id]) return MMDGAN_E_ARG;
    Plan &p = *h.plans[plan_id];
    std::vector<hipStream_t> streams;                     // streams numbered in order of first use
    std::string out;
    for (const Plan::KernelNote &k : p.kernels) {
        size_t si = 0;
        while (si < streams.size() && streams[si] != k.st) ++si;
        if (si == streams.size()) streams.push_back(k.st);
        const char *mangled = hipKernelNameRefByPtr(k.fn, nullptr);      // (the null stream: the recorded one may be gone by now)
        (void)hipGetLastError();
        int status = 1;
        char *dem = mangled ? abi::__cxa_demangle(mangled, nullptr, nullptr, &status) : nullptr;
        out += (status == 0 && dem) ? dem : (mangled ? mangled : "?");
        if (dem) free(dem);
        char tail[64];
        snprintf(tail, sizeof(tail), "\t%u\t%u\t%zu\n", k.grid, k.block, si);
        out += tail;
    }
    if (buf && cap > 0) {
        const size_t n = out.size() < cap - 1 ? out.size() : cap - 1;
        memcpy(buf, out.data(), n);
        buf[n] = 0;
    }
    return (long)out.size() + 1;
}
extern "C" int mmdgan_plan_replay(int plan_id, int segment) {
    mmdgan_handle &h = cur();
    MMDGAN_REQUIRE(!plan_recording(), "plan_replay: a plan is being recorded");
    MMDGAN_REQUIRE(plan_id >= 0 && plan_id < (int)h.plans.size() && h.plans[plan_id], "plan_replay: no plan %d", plan_id);
    Plan &p = *h.plans[plan_id];
    MMDGAN_REQUIRE(p.comm_generation == 0 || p.comm_generation == comm_generation(),
                   "plan_replay: plan %d holds collectives of a communicator that has been destroyed since (record it again)", plan_id);
    const int nseg = (int)p.segment_end.size();
    MMDGAN_REQUIRE(segment >= -1 && segment < nseg, "plan_replay: plan %d has %d segments (asked for %d)", plan_id, nseg, segment);
    const size_t lo = segment <= 0 ? 0 : p.segment_end[segment - 1];
    const size_t hi = segment < 0 ? p.nodes.size() : p.segment_end[segment];
    for (size_t i = lo; i < hi; ++i) p.nodes[i]();
    return check_launch("plan_replay");
}
extern "C" int mmdgan_plan_destroy(int plan_id) {
    mmdgan_handle &h = cur();
    MMDGAN_REQUIRE(plan_id >= 0 && plan_id < (int)h.plans.size(), "plan_destroy: no plan %d", plan_id);
    h.plans[plan_id].reset();
    return MMDGAN_OK;
}
