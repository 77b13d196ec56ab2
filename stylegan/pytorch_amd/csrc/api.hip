// Error plumbing + version of the C ABI (include/sgx.h).
#include "common.h"
#include <atomic>
#include <mutex>

#define SGX_VERSION 100   // 0.1.0

static thread_local char g_err[512] = "";

void sgx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Hardware self-test of the LDS transpose read the bf16 weight-gradient kernel relies on: LDS holds lds[e] = e, lane l
// passes the address of elements [4l, 4l+4); out[l*4 + j] = what lane l received in element j.
__global__ void selftest_tr16_kernel(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[256];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (short)i;
    __syncthreads();
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + threadIdx.x * 4));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
extern "C" int sgx_selftest_tr16(void* out256, void* stream) {
    hipLaunchKernelGGL(selftest_tr16_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (short*)out256);
    SGX_LAUNCH_CHECK("selftest_tr16");
    return 0;
}

// ---------------------------------------------------------------- per-launch profiler
#include <cxxabi.h>
#include <map>
#include <mutex>
#include <string>
#include <vector>
int sgx_prof_mode = 0;
const void* sgx_prof_only_fn = nullptr;
namespace {
struct ProfRec { const void* fn; hipEvent_t e0, e1; double flops, bytes; char desc[56]; };
struct ProfNote { bool set; double flops, bytes; char desc[56]; };
std::mutex g_prof_mu;                                      // launches come from the caller's and autograd's threads
std::vector<ProfRec> g_recs;
std::vector<hipEvent_t> g_event_pool;
std::map<const void*, std::string> g_names;
thread_local ProfNote g_note = {false, 0.0, 0.0, ""};
hipEvent_t prof_event() {
    if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}
}  // namespace
void sgx_prof_note(double flops, double bytes, const char* fmt, ...) {
    g_note.set = true; g_note.flops = flops; g_note.bytes = bytes;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_note.desc, sizeof(g_note.desc), fmt, ap);
    va_end(ap);
}
void sgx_prof_begin(const void* fn, hipStream_t st, int* slot) {
    ProfNote note = g_note;
    g_note.set = false;                                    // a note describes exactly one launch
    if (sgx_prof_mode == 2 && fn != sgx_prof_only_fn) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r;
    r.fn = fn; r.e0 = prof_event(); r.e1 = prof_event();
    r.flops = note.set ? note.flops : 0.0; r.bytes = note.set ? note.bytes : 0.0;
    snprintf(r.desc, sizeof(r.desc), "%s", note.set ? note.desc : "");
    if (!r.e0 || !r.e1) return;
    (void)hipEventRecord(r.e0, st);
    g_recs.push_back(r);
    *slot = (int)g_recs.size() - 1;
}
void sgx_prof_end(int slot, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (slot < (int)g_recs.size()) (void)hipEventRecord(g_recs[slot].e1, st);
}
// mode 0: stop.  1: record every launch.  2: record only launches of the kernel record `only_of` (an index valid before
// this call) belongs to.  Starting (mode != 0) clears earlier records.
extern "C" int sgx_prof_start(int mode, int only_of) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    SGX_REQUIRE(mode >= 0 && mode <= 2, SGX_EINVAL, "prof_start: mode %d", mode);
    const void* only = nullptr;
    if (mode == 2) {
        SGX_REQUIRE(only_of >= 0 && only_of < (int)g_recs.size(), SGX_EINVAL, "prof_start: record %d", only_of);
        only = g_recs[only_of].fn;
    }
    if (mode != 0) {
        for (auto& r : g_recs) { g_event_pool.push_back(r.e0); g_event_pool.push_back(r.e1); }
        g_recs.clear();
    }
    sgx_prof_only_fn = only;
    sgx_prof_mode = mode;
    return 0;
}
extern "C" int sgx_prof_count(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    return (int)g_recs.size();
}
// Record i: demangled kernel name (as rocprofv3 prints it), milliseconds between its two events (waits for them),
// and the flops / algorithmic bytes / layer description the launching entry point attached (0 / "" if none).
extern "C" int sgx_prof_get(int i, char* name, int name_cap, float* ms, double* flops, double* bytes, char* desc, int desc_cap) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    SGX_REQUIRE(i >= 0 && i < (int)g_recs.size(), SGX_EINVAL, "prof_get: record %d of %d", i, (int)g_recs.size());
    ProfRec& r = g_recs[i];
    auto it = g_names.find(r.fn);
    if (it == g_names.end()) {
        const char* m = hipKernelNameRefByPtr(r.fn, nullptr);
        std::string n = m ? m : "?";
        int status = 0;
        char* d = m ? abi::__cxa_demangle(m, nullptr, nullptr, &status) : nullptr;
        if (d && status == 0) n = d;
        free(d);
        it = g_names.emplace(r.fn, n).first;
    }
    if (name && name_cap > 0) snprintf(name, name_cap, "%s", it->second.c_str());
    if (desc && desc_cap > 0) snprintf(desc, desc_cap, "%s", r.desc);
    if (flops) *flops = r.flops;
    if (bytes) *bytes = r.bytes;
    if (ms) {
        hipError_t e = hipEventSynchronize(r.e1);
        if (e == hipSuccess) e = hipEventElapsedTime(ms, r.e0, r.e1);
        SGX_REQUIRE(e == hipSuccess, (int)e, "prof_get: %s", hipGetErrorString(e));
    }
    return 0;
}

extern "C" int sgx_version(void) { return SGX_VERSION; }
extern "C" const char* sgx_last_error(void) { return g_err; }
extern "C" int sgx_clear_error(void) { g_err[0] = 0; return (int)hipGetLastError(); }   // also resets HIP's sticky last-error

// waiter waits for everything enqueued on signaler so far: one event record + one stream wait on an event from a small
// round-robin pool (a wait binds to the record that precedes it, so an event may be re-recorded while older waits are
// still pending).  The step forks its weight-gradient launches to a side stream ~70 times per iteration; through
// torch.cuda.Stream.wait_stream that is an event object + several Python-level device queries each time.
extern "C" int sgx_stream_wait_stream(void* waiter, void* signaler) {
    constexpr int NEV = 256;
    static hipEvent_t pool[NEV];
    static std::atomic<unsigned> next{0};
    static std::once_flag once;
    static int create_rc = 0;
    std::call_once(once, [] {
        for (int i = 0; i < NEV && !create_rc; ++i) create_rc = (int)hipEventCreateWithFlags(&pool[i], hipEventDisableTiming);
    });
    SGX_REQUIRE(create_rc == 0, create_rc, "stream_wait_stream: hipEventCreateWithFlags failed (%d)", create_rc);
    if (waiter == signaler) return 0;
    hipEvent_t ev = pool[next.fetch_add(1) % NEV];
    hipError_t e = hipEventRecord(ev, (hipStream_t)signaler);
    if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)waiter, ev, 0);
    SGX_REQUIRE(e == hipSuccess, (int)e, "stream_wait_stream: %s", hipGetErrorString(e));
    return 0;
}

// Compute units of the CURRENT device, cached per device id (a process that drives several GPUs sizes each launch for the device it
// is made on; round 4 cached the first device's count per process).
int sgx_ncu() {
    static int cache[32] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) dev = 0;
    if (!cache[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
        cache[dev] = n;
    }
    return cache[dev];
}
