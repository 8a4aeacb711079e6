// Library plumbing: version, error strings, device query, optional hipEvent kernel timing.
#include "common.h"
#include "knobs.h"
#include <cstdlib>
#include <mutex>
#include <vector>

namespace {
struct DevInfo {
    int cu_count = 0;
    int lds_per_cu = 0;
    bool ok = false;
};
DevInfo g_dev[64];

struct ProfRec {
    hipEvent_t a, b;
    int id;
};
bool g_prof_on = false;
std::vector<ProfRec> g_pending;
double g_total_ms[PROF_NUM];
int64_t g_launches[PROF_NUM];
std::mutex g_mu;
hipEvent_t g_cur[PROF_NUM];
}  // namespace

// ---- knobs: the environment is read exactly once (knobs.h) ---------------------------------------
namespace {
AsrkKnobs g_knobs;
std::once_flag g_knobs_once;
int env_int(const char *name) {
    const char *e = getenv(name);
    return e ? atoi(e) : AsrkKnobs::UNSET;
}
int env_present(const char *name) { return getenv(name) ? 1 : AsrkKnobs::UNSET; }
void read_knobs() {
    AsrkKnobs &k = g_knobs;
    k.gemm_noskinny = env_present("ASRK_GEMM_NOSKINNY");
    k.skinny_sk = env_int("ASRK_SKINNY_SK");
    k.gemm_dbg = env_int("ASRK_GEMM_DBG");
    k.gemm_nofast = env_present("ASRK_GEMM_NOFAST");
    k.split_w256 = env_int("ASRK_SPLIT_W256");
    k.split_tail = env_int("ASRK_SPLIT_TAIL");
    k.fwd_mt = env_int("ASRK_FWD_MT");
    k.fwd_nt = env_int("ASRK_FWD_NT");
    k.rec_bf_mt4 = env_int("ASRK_REC_BF_MT4");
    k.bwd_ub = env_int("ASRK_BWD_UB");
    k.bwd_nt = env_int("ASRK_BWD_NT");
    k.fwd_poll = env_int("ASRK_FWD_POLL");
    k.fwd_presleep = env_int("ASRK_FWD_PRESLEEP");
    k.bwd_poll = env_int("ASRK_BWD_POLL");
    k.bwd_presleep = env_int("ASRK_BWD_PRESLEEP");
    k.dbg_noload = env_present("ASRK_DBG_NOLOAD");
    k.deterministic = env_int("ASRK_DETERMINISTIC");
}
}  // namespace

const AsrkKnobs &asrk_knobs_() {
    std::call_once(g_knobs_once, read_knobs);
    return g_knobs;
}

extern "C" int asrk_version(void) { return 200; }

extern "C" const char *asrk_strerror(int rc) {
    switch (rc) {
        case ASRK_OK: return "ok";
        case ASRK_EINVAL: return "asrk: invalid argument";
        case ASRK_ESHAPE: return "asrk: shape not supported by the persistent gfx950 kernels";
        case ASRK_EWORKSPACE: return "asrk: workspace too small";
        case ASRK_EDEVICE: return "asrk: device lacks the CUs/LDS the persistent kernel needs";
        case ASRK_ETIMEOUT: return "asrk: in-kernel grid synchronisation timed out";
        default: break;
    }
    if (rc > 0) return hipGetErrorString((hipError_t)rc);
    return "asrk: unknown error";
}

extern "C" int asrk_init(int device) {
    if (device < 0 || device >= 64) return ASRK_EINVAL;
    (void)asrk_knobs_();    // the one and only read of the ASRK_* tuning environment
    if (!g_dev[device].ok) {
        hipDeviceProp_t prop;
        hipError_t e = hipGetDeviceProperties(&prop, device);
        if (e != hipSuccess) return (int)e > 0 ? -(1000 + (int)e) : ASRK_EDEVICE;
        g_dev[device].cu_count = prop.multiProcessorCount;
        g_dev[device].lds_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
        g_dev[device].ok = true;
    }
    return g_dev[device].cu_count;
}

// current-device CU count for the persistent kernels (lazy init)
extern "C" int asrk_cu_count_(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (dev < 0 || dev >= 64) return 0;
    if (!g_dev[dev].ok) {
        if (asrk_init(dev) <= 0) return 0;
    }
    return g_dev[dev].cu_count;
}

// algorithmic work (flops) a family has been asked to do while profiling is on (GEMM: 2*M*N*K)
static double g_work[PROF_NUM] = {0};

static unsigned g_prof_mask = 0xffffffffu;      // families that record events / count work while profiling is on
extern "C" void asrk_profile_enable(int on) { g_prof_on = on != 0; }
extern "C" void asrk_profile_families(unsigned mask) { g_prof_mask = mask; }
static inline bool prof_live(int id) { return g_prof_on && id >= 0 && id < PROF_NUM && ((g_prof_mask >> id) & 1u); }

extern "C" void asrk_profile_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &r : g_pending) {
        hipEventDestroy(r.a);
        hipEventDestroy(r.b);
    }
    g_pending.clear();
    for (int i = 0; i < PROF_NUM; ++i) {
        g_total_ms[i] = 0.0;
        g_launches[i] = 0;
        g_work[i] = 0.0;
    }
}

extern "C" void asrk_prof_work_(int id, double flops) {
    if (prof_live(id)) g_work[id] += flops;
}
extern "C" int asrk_profile_get_work(int id, double *flops) {
    if (id < 0 || id >= PROF_NUM || !flops) return ASRK_EINVAL;
    *flops = g_work[id];
    return ASRK_OK;
}

extern "C" void asrk_prof_begin_(int id, hipStream_t s) {
    if (!prof_live(id)) return;
    hipEvent_t a;
    if (hipEventCreate(&a) != hipSuccess) return;
    hipEventRecord(a, s);
    g_cur[id] = a;
}

extern "C" void asrk_prof_end_(int id, hipStream_t s) {
    if (!prof_live(id)) return;
    hipEvent_t b;
    if (hipEventCreate(&b) != hipSuccess) return;
    hipEventRecord(b, s);
    std::lock_guard<std::mutex> lk(g_mu);
    g_pending.push_back({g_cur[id], b, id});
}

// a call that enqueued several kernels under one event pair reports the extra launches here
extern "C" void asrk_prof_launches_(int id, int64_t n) {
    if (!prof_live(id)) return;
    std::lock_guard<std::mutex> lk(g_mu);
    g_launches[id] += n;
}

extern "C" int asrk_profile_get(int id, double *total_ms, int64_t *launches) {
    if (id < 0 || id >= PROF_NUM) return ASRK_EINVAL;
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &r : g_pending) {
        hipEventSynchronize(r.b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            g_total_ms[r.id] += ms;
            g_launches[r.id] += 1;
        }
        hipEventDestroy(r.a);
        hipEventDestroy(r.b);
    }
    g_pending.clear();
    if (total_ms) *total_ms = g_total_ms[id];
    if (launches) *launches = g_launches[id];
    return ASRK_OK;
}
