// sl_aql.hip -- user-mode AQL queues of the library's own, next to HIP's streams: the launcher of sliced stepping.
//
// Why: a step of the batch is issued as one launch per slice so that one slice's kernel boundary (~2.2 us between its
// last acknowledged store and the first wave of its next launch) runs under the other slices' kernels.  Through HIP a
// launch costs the host 2.4-3 us (argument copy, device-side argument buffer with its read-back, stream bookkeeping),
// so a single stepping thread can feed two slices per ~8 us step and no more -- and launching from several threads
// contends inside the runtime.  Here a slice is an HSA queue created with the runtime HIP itself sits on:
// a dispatch is one 64-byte packet and one doorbell, the argument blocks of all slices of a step are written into a
// device-memory ring and flushed ONCE, and three to six slices become affordable.
//
// Ordering is the queue's own, exactly what a HIP stream gives: every dispatch carries the barrier bit (it starts when
// its queue's previous packet has completed) with agent-scope acquire and release fences -- in that (default) mode
// nothing here relies on where a workgroup runs.  The one exception is OPT-IN (AqlLaunch::release_free, asked for
// through slhip_queues_open's SL_QUEUES_RELEASE_FREE): steps without a release fence, valid only while a workgroup
// index keeps running on the same XCD, which the step kernels themselves verify at every step (sl_rowlane.hip).
// The kernel is the one HIP loaded: its descriptor is looked up in the runtime's executables (HSA loader extension)
// under the name HIP reports for the hipFunction_t, so streams and queues run the same code object with the same
// argument block.
//
// What stays with the caller (vector_env.py): a first step after work of HIP streams (resets, the policy's actions) is
// dispatched with a system-scope acquire once the caller has synchronised those streams; before anything outside the
// queues reads what the steps wrote, aql_fence() puts a barrier packet with a system-scope release behind them and
// waits for its completion signal.
//
// The HSA entry points are taken from the libhsa-runtime64 that is ALREADY in the process (the one HIP loaded: torch
// ships its own copy, and a second runtime would see neither HIP's allocations nor its executables).
#include <dlfcn.h>
#include <link.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <hsa/hsa_ven_amd_loader.h>
#include <immintrin.h>

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>
#include <cstdio>

#include "sl_kernels.h"

namespace sl {
namespace {

constexpr int MAX_QUEUES = 8;
constexpr uint32_t QUEUE_PACKETS = 1024;
constexpr size_t KARG_SLOT = 1024;              // bytes per argument block
// Argument blocks per queue: a short ring.  A slot is written again only when the packet that used it last has
// COMPLETED (the read index has passed the packet behind it: every step packet carries the barrier bit), so the host
// runs at most KARG_SLOTS - 2 dispatches ahead of a queue.  A slot that already holds the block of the same slice of
// the same batch (AqlPatch; aql_warm() writes it into every idle slot when a batch first steps) only has the words that
// change from step to step sent across the PCIe aperture again: two pointers instead of ~600 bytes, i.e. 0.6 us of a
// dispatch's 0.7 us of host time.
constexpr size_t KARG_SLOTS = 64;

struct Hsa {
    bool ok = false;
    std::string why;
#define SL_HSA_FNS(X)                                                                                                   \
    X(hsa_init) X(hsa_iterate_agents) X(hsa_agent_get_info) X(hsa_queue_create) X(hsa_signal_create)                    \
    X(hsa_signal_store_screlease) X(hsa_signal_store_relaxed) X(hsa_signal_wait_scacquire)                              \
    X(hsa_signal_load_relaxed)                                                                                          \
    X(hsa_queue_add_write_index_relaxed) X(hsa_queue_load_read_index_scacquire) X(hsa_system_get_major_extension_table) \
    X(hsa_executable_get_symbol_by_name) X(hsa_executable_symbol_get_info) X(hsa_amd_agent_iterate_memory_pools)        \
    X(hsa_amd_memory_pool_get_info) X(hsa_amd_memory_pool_allocate) X(hsa_amd_agents_allow_access)                      \
    X(hsa_amd_agent_memory_pool_get_info) X(hsa_status_string) X(hsa_system_get_info)                                   \
    X(hsa_amd_profiling_set_profiler_enabled) X(hsa_amd_profiling_get_dispatch_time)                                   \
    X(hsa_amd_profiling_convert_tick_to_system_domain)
#define X(n) decltype(&::n) n = nullptr;
    SL_HSA_FNS(X)
#undef X
};

int find_loaded_hsa(struct dl_phdr_info *info, size_t, void *data) {
    if (info->dlpi_name && strstr(info->dlpi_name, "libhsa-runtime64")) {
        *(std::string *)data = info->dlpi_name;
        return 1;
    }
    return 0;
}

const Hsa &hsa() {
    static Hsa h;
    static std::once_flag once;
    std::call_once(once, [] {
        std::string path;
        dl_iterate_phdr(find_loaded_hsa, &path);
        if (path.empty()) {
            h.why = "libhsa-runtime64 is not loaded in this process (HIP not initialised?)";
            return;
        }
        void *lib = dlopen(path.c_str(), RTLD_NOW | RTLD_NOLOAD);
        if (!lib) {
            h.why = std::string("dlopen(") + path + ", RTLD_NOLOAD): " + dlerror();
            return;
        }
#define X(n)                                                                                                            \
    h.n = (decltype(h.n))dlsym(lib, #n);                                                                                \
    if (!h.n) {                                                                                                         \
        h.why = "libhsa-runtime64 lacks " #n;                                                                           \
        return;                                                                                                         \
    }
        SL_HSA_FNS(X)
#undef X
        h.ok = true;
    });
    return h;
}

std::string hsa_err(const char *what, hsa_status_t st) {
    const char *s = nullptr;
    if (hsa().hsa_status_string) hsa().hsa_status_string(st, &s);
    return std::string(what) + ": " + (s ? s : "HSA error") + " (" + std::to_string((int)st) + ")";
}

struct KernelInfo {
    uint64_t object = 0;
    uint32_t kernarg = 0, group = 0, priv = 0;
};

struct Queue {
    hsa_queue_t *q = nullptr;
    hsa_signal_t fence{0};
    unsigned char *karg = nullptr;
    uint64_t karg_next = 0;
    uint64_t read_seen = 0;         // the queue's read index when it was last looked at (it lives in memory the device
                                    // writes: every look is a cache miss, so it is only looked at when it matters)
    struct Slot {
        uint64_t packet = 0;        // index of the packet that used the slot last
        bool used = false;
        uint32_t owner = 0, version = 0;    // whose argument block it holds (AqlPatch), 0 = nobody's
    } slots[KARG_SLOTS];
    bool dirty = false;             // dispatched since the last marker
};

// A marker: one barrier packet per queue (system-scope release, completion signal) behind everything dispatched so
// far.  Issued without waiting; whoever needs the results waits for its signals (the stepping thread in
// slhip_queues_sync, the gather's worker thread before it hands a window to RCCL).
constexpr int MARKERS = 16;
struct Marker {
    hsa_signal_t sig[MAX_QUEUES];
    bool have[MAX_QUEUES] = {};     // signal created
    bool waiting[MAX_QUEUES] = {};  // a barrier packet of this marker will set sig[i] to 0
    long long ticket = -1;
};

struct Device {
    bool ok = false;
    std::string why;
    hsa_agent_t gpu{0}, cpu{0};
    hsa_amd_memory_pool_t vram{0}, host_karg{0};
    bool have_vram = false, have_host_karg = false, karg_in_vram = false;
    Queue queues[MAX_QUEUES];
    int n_queues = 0;
    std::unordered_map<hipFunction_t, KernelInfo> kernels;
    hsa_ven_amd_loader_1_03_pfn_t loader{};
    int readback = 1;       // SL_AQL_READBACK=0 (A/B only): no read-back after writing arguments into device memory
    int step_acquire = HSA_FENCE_SCOPE_AGENT, step_release = HSA_FENCE_SCOPE_AGENT;   // SL_AQL_STEP_ACQUIRE / _RELEASE
                            // (A/B only: anything below agent scope makes the results depend on where workgroups run)
    std::mutex mu;
    // dispatches of one step, written but not yet published (aql_begin ... aql_commit): their argument blocks are
    // flushed together, then the packets go out
    struct Pending {
        int queue;
        uint32_t *packet;
        uint32_t head_word;
        uint64_t index;
    };
    static constexpr int MAX_PENDING = 512;     // (a staged region: up to KARG_SLOTS - 2 steps of up to eight slices)
    Pending pending[MAX_PENDING];
    int n_pending = 0;
    bool batching = false;
    const volatile unsigned char *last_tail = nullptr;
    Marker markers[MARKERS];
    long long next_ticket = 0;
    bool poisoned = false;          // a marker timed out: work may still be in flight, nothing of the queues' is freed
    // SL_AQL_TIMELINE=1 (debugging): the queues run with dispatch profiling on, every dispatch carries a completion
    // signal, and aql_timeline_dump() prints when the host rang which doorbell and when the device started and ended
    // what, all on the HSA system clock
    bool timeline = false;
    struct TlEvent {
        const char *what;
        int queue;
        uint64_t host_tick;
        hsa_signal_t sig;       // {0}: a host-side mark
    };
    std::vector<TlEvent> tl;
};

Device g_dev[64];
std::mutex g_open_mu;

struct AgentSearch {
    uint32_t bdf, domain;
    hsa_agent_t gpu, cpu;
    bool have_gpu, have_cpu;
};

hsa_status_t agent_cb(hsa_agent_t a, void *data) {
    AgentSearch *s = (AgentSearch *)data;
    hsa_device_type_t type;
    if (hsa().hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &type) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    if (type == HSA_DEVICE_TYPE_CPU) {
        if (!s->have_cpu) {
            s->cpu = a;
            s->have_cpu = true;
        }
    } else if (type == HSA_DEVICE_TYPE_GPU) {
        uint32_t bdf = 0, domain = 0;
        hsa().hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf);
        hsa().hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &domain);
        if (bdf == s->bdf && domain == s->domain && !s->have_gpu) {
            s->gpu = a;
            s->have_gpu = true;
        }
    }
    return HSA_STATUS_SUCCESS;
}

struct PoolSearch {
    Device *d;
    bool gpu;
};

hsa_status_t pool_cb(hsa_amd_memory_pool_t pool, void *data) {
    PoolSearch *s = (PoolSearch *)data;
    hsa_amd_segment_t seg;
    uint32_t flags = 0;
    bool alloc = false;
    hsa().hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
    hsa().hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    hsa().hsa_amd_memory_pool_get_info(pool, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
    if (!alloc) return HSA_STATUS_SUCCESS;
    if (s->gpu) {
        if ((flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !s->d->have_vram) {
            s->d->vram = pool;
            s->d->have_vram = true;
        }
    } else {
        if ((flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !s->d->have_host_karg) {
            s->d->host_karg = pool;
            s->d->have_host_karg = true;
        }
    }
    return HSA_STATUS_SUCCESS;
}

int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}

// everything per device that does not depend on the number of queues
void open_device(Device &d, int dev) {
    const Hsa &h = hsa();
    if (!h.ok) {
        d.why = h.why;
        return;
    }
    hsa_status_t st = h.hsa_init();         // (reference-counted: HIP holds the runtime open already)
    if (st != HSA_STATUS_SUCCESS) {
        d.why = hsa_err("hsa_init", st);
        return;
    }
    int bus = 0, device = 0, domain = 0;
    if (hipDeviceGetAttribute(&bus, hipDeviceAttributePciBusId, dev) != hipSuccess ||
        hipDeviceGetAttribute(&device, hipDeviceAttributePciDeviceId, dev) != hipSuccess ||
        hipDeviceGetAttribute(&domain, hipDeviceAttributePciDomainID, dev) != hipSuccess) {
        (void)hipGetLastError();
        d.why = "no PCI address for the HIP device";
        return;
    }
    AgentSearch s{};
    s.bdf = ((uint32_t)bus << 8) | ((uint32_t)device << 3);
    s.domain = (uint32_t)domain;
    h.hsa_iterate_agents(agent_cb, &s);
    if (!s.have_gpu || !s.have_cpu) {
        d.why = "no HSA agent at the HIP device's PCI address";
        return;
    }
    d.gpu = s.gpu;
    d.cpu = s.cpu;
    PoolSearch pg{&d, true}, pc{&d, false};
    h.hsa_amd_agent_iterate_memory_pools(d.gpu, pool_cb, &pg);
    h.hsa_amd_agent_iterate_memory_pools(d.cpu, pool_cb, &pc);
    st = h.hsa_system_get_major_extension_table(HSA_EXTENSION_AMD_LOADER, 1, sizeof(d.loader), &d.loader);
    if (st != HSA_STATUS_SUCCESS || !d.loader.hsa_ven_amd_loader_iterate_executables) {
        d.why = "the HSA loader extension cannot enumerate executables";
        return;
    }
    // argument blocks: device memory the host writes through the PCIe aperture (what HIP_FORCE_DEV_KERNARG does:
    // arguments fetched across PCIe at every wave start cost ~1.5 us per launch), if the CPU may map it
    d.karg_in_vram = false;
    if (d.have_vram && env_int("SL_AQL_KERNARG_HOST", 0) == 0) {
        hsa_amd_memory_pool_access_t acc = HSA_AMD_MEMORY_POOL_ACCESS_NEVER_ALLOWED;
        h.hsa_amd_agent_memory_pool_get_info(d.cpu, d.vram, HSA_AMD_AGENT_MEMORY_POOL_INFO_ACCESS, &acc);
        d.karg_in_vram = acc != HSA_AMD_MEMORY_POOL_ACCESS_NEVER_ALLOWED;
    }
    if (!d.karg_in_vram && !d.have_host_karg) {
        d.why = "no memory pool for kernel arguments";
        return;
    }
    d.timeline = env_int("SL_AQL_TIMELINE", 0) != 0;
    d.readback = env_int("SL_AQL_READBACK", 1);
    d.step_acquire = env_int("SL_AQL_STEP_ACQUIRE", HSA_FENCE_SCOPE_AGENT);
    d.step_release = env_int("SL_AQL_STEP_RELEASE", HSA_FENCE_SCOPE_AGENT);
    d.ok = true;
}

bool open_queue(Device &d, Queue &q) {
    const Hsa &h = hsa();
    hsa_status_t st = h.hsa_queue_create(d.gpu, QUEUE_PACKETS, HSA_QUEUE_TYPE_MULTI, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q.q);
    if (st != HSA_STATUS_SUCCESS) {
        d.why = hsa_err("hsa_queue_create", st);
        return false;
    }
    if (d.timeline) (void)h.hsa_amd_profiling_set_profiler_enabled(q.q, 1);
    // SL_AQL_PRIORITY=1 (an experiment's knob): the step queues at HIGH priority (optional entry point, looked up on
    // its own) -- for runs that put long-lived kernels of other queues beside the steps (the episode-end pass of C5
    // under them: no gain measured, profiles/round5_p_c5_overlap.txt)
    if (env_int("SL_AQL_PRIORITY", 0) > 0) {
        static auto set_priority = [] {
            std::string path;
            dl_iterate_phdr(find_loaded_hsa, &path);
            void *lib = path.empty() ? nullptr : dlopen(path.c_str(), RTLD_NOW | RTLD_NOLOAD);
            return lib ? (decltype(&::hsa_amd_queue_set_priority))dlsym(lib, "hsa_amd_queue_set_priority") : nullptr;
        }();
        if (set_priority) (void)set_priority(q.q, HSA_AMD_QUEUE_PRIORITY_HIGH);
    }
    st = h.hsa_signal_create(0, 0, nullptr, &q.fence);
    if (st != HSA_STATUS_SUCCESS) {
        d.why = hsa_err("hsa_signal_create", st);
        return false;
    }
    void *p = nullptr;
    if (d.karg_in_vram) {
        st = h.hsa_amd_memory_pool_allocate(d.vram, KARG_SLOT * KARG_SLOTS, 0, &p);
        if (st == HSA_STATUS_SUCCESS) st = h.hsa_amd_agents_allow_access(1, &d.cpu, nullptr, p);
        if (st != HSA_STATUS_SUCCESS) {
            d.karg_in_vram = false;          // (no large BAR: arguments stay in host memory)
            p = nullptr;
        }
    }
    if (!p) {
        if (!d.have_host_karg) {
            d.why = "no host pool for kernel arguments";
            return false;
        }
        st = h.hsa_amd_memory_pool_allocate(d.host_karg, KARG_SLOT * KARG_SLOTS, 0, &p);
        if (st == HSA_STATUS_SUCCESS) st = h.hsa_amd_agents_allow_access(1, &d.gpu, nullptr, p);
        if (st != HSA_STATUS_SUCCESS) {
            d.why = hsa_err("kernel-argument ring", st);
            return false;
        }
    }
    q.karg = (unsigned char *)p;
    return true;
}

struct SymbolSearch {
    Device *d;
    const char *name;
    KernelInfo info;
    bool found;
};

hsa_status_t exec_cb(hsa_executable_t exec, void *data) {
    SymbolSearch *s = (SymbolSearch *)data;
    const Hsa &h = hsa();
    hsa_executable_symbol_t sym;
    if (h.hsa_executable_get_symbol_by_name(exec, s->name, &s->d->gpu, &sym) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    KernelInfo k;
    if (h.hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object) != HSA_STATUS_SUCCESS ||
        h.hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.kernarg) != HSA_STATUS_SUCCESS ||
        h.hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group) != HSA_STATUS_SUCCESS ||
        h.hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.priv) != HSA_STATUS_SUCCESS)
        return HSA_STATUS_SUCCESS;
    s->info = k;
    s->found = true;
    return HSA_STATUS_INFO_BREAK;
}

const KernelInfo *kernel_info(Device &d, hipFunction_t f) {
    auto it = d.kernels.find(f);
    if (it != d.kernels.end()) return it->second.object ? &it->second : nullptr;
    KernelInfo k;
    const char *name = hipKernelNameRef(f);
    if (name) {
        const std::string kd = std::string(name) + ".kd";
        SymbolSearch s{&d, kd.c_str(), {}, false};
        d.loader.hsa_ven_amd_loader_iterate_executables(exec_cb, &s);
        if (s.found) k = s.info;
    }
    auto &slot = d.kernels[f];
    slot = k;
    return k.object ? &slot : nullptr;
}

uint16_t header(uint16_t type, bool barrier, int acquire, int release) {
    return (uint16_t)((type << HSA_PACKET_HEADER_TYPE) | ((barrier ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                      (acquire << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (release << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
}

// the next packet slot of the queue (waits while the ring is full)
void *claim(const Hsa &h, Queue &qq, uint64_t *index) {
    hsa_queue_t *q = qq.q;
    const uint64_t idx = h.hsa_queue_add_write_index_relaxed(q, 1);
    while (idx - qq.read_seen >= q->size) {
        qq.read_seen = h.hsa_queue_load_read_index_scacquire(q);
        if (idx - qq.read_seen >= q->size) _mm_pause();
    }
    *index = idx;
    return (unsigned char *)q->base_address + (idx & (q->size - 1)) * 64;
}

uint64_t sys_tick(const Hsa &h) {
    uint64_t t = 0;
    h.hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP, &t);
    return t;
}
void tl_mark(Device &d, const Hsa &h, const char *what, int queue = -1, hsa_signal_t sig = hsa_signal_t{0}) {
    if (d.timeline && d.tl.size() < 4096) d.tl.push_back({what, queue, sys_tick(h), sig});
}

// Argument blocks written since the last call have landed in device memory (posted writes through the aperture:
// drained from the CPU's write-combining buffers, then -- as the runtime does for device-resident arguments -- one
// byte read back: a read does not pass the writes in front of it); then the pending packets are handed to the queues.
void publish(Device &d, const Hsa &h) {
    if (d.karg_in_vram && d.last_tail) {
        _mm_sfence();
        if (d.readback) (void)*d.last_tail;
    }
    d.last_tail = nullptr;
    if (d.n_pending) tl_mark(d, h, "doorbells");
    // every pending packet becomes valid, then ONE doorbell per queue with the index of its youngest packet (the packet
    // processor takes everything up to it); queues in the order of their first pending packet
    uint64_t last[MAX_QUEUES];
    int order[MAX_QUEUES], n_order = 0;
    bool seen[MAX_QUEUES] = {};
    for (int i = 0; i < d.n_pending; ++i) {
        const Device::Pending &p = d.pending[i];
        __atomic_store_n(p.packet, p.head_word, __ATOMIC_RELEASE);
        if (!seen[p.queue]) {
            seen[p.queue] = true;
            order[n_order++] = p.queue;
        }
        last[p.queue] = p.index;
    }
    for (int i = 0; i < n_order; ++i)
        h.hsa_signal_store_screlease(d.queues[order[i]].q->doorbell_signal, (hsa_signal_value_t)last[order[i]]);
    d.n_pending = 0;
}

Device *current(int *dev_out = nullptr) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    if (dev_out) *dev_out = dev;
    return &g_dev[dev];
}

}  // namespace

const char *aql_open(int n_queues) {
    static thread_local std::string why;
    if (n_queues < 1 || n_queues > MAX_QUEUES) return "between 1 and 8 queues";
    int dev = 0;
    Device *d = current(&dev);
    if (!d) return "no current HIP device";
    std::lock_guard<std::mutex> lock(g_open_mu);
    if (!d->ok && d->why.empty()) open_device(*d, dev);
    if (!d->ok) {
        why = d->why;
        return why.c_str();
    }
    while (d->n_queues < n_queues) {
        if (!open_queue(*d, d->queues[d->n_queues])) {
            why = d->why;
            return why.c_str();
        }
        ++d->n_queues;
    }
    return nullptr;
}

const char *aql_probe(hipFunction_t f) {
    Device *d = current();
    if (!d || !d->ok) return "AQL queues are not open";
    if (!f) return "no HIP function handle to look up";
    std::lock_guard<std::mutex> lock(d->mu);
    return kernel_info(*d, f) ? nullptr : "HIP's kernels were not found in the HSA runtime's executables";
}

namespace {
// one kernel dispatch packet on `queue` (the caller holds d.mu); hd: the packet header, completion: its signal or {0}
hipError_t emit(Device &d, const Hsa &h, int queue, hipFunction_t f, unsigned grid, unsigned threads, unsigned lds,
                const void *args, size_t arg_bytes, uint16_t hd, hsa_signal_t completion, bool may_batch,
                const AqlPatch *patch = nullptr) {
    const KernelInfo *k = kernel_info(d, f);
    if (!k) return hipErrorNotFound;
#ifndef SL_AQL_SCRATCH
#define SL_AQL_SCRATCH 0        /* experiment: 1 = kernels with a private segment are dispatched too (the queue's scratch is the runtime's to set up) */
#endif
    if (arg_bytes > k->kernarg || k->kernarg > KARG_SLOT || (k->priv != 0 && !SL_AQL_SCRATCH)) {
        // (never expected: say which -- a kernel that spills has private memory, which these packets do not set up)
        fprintf(stderr, "libsafelife_hip: AQL dispatch refused: argument block %zu bytes, kernel takes %u (slot %zu), private "
                        "segment %u bytes\n", arg_bytes, (unsigned)k->kernarg, KARG_SLOT, (unsigned)k->priv);
        return hipErrorInvalidValue;
    }
    Queue &q = d.queues[queue];
    // the argument block: explicit arguments, zeros where hidden ones would follow (the step kernels have none)
    const unsigned si = (unsigned)(q.karg_next++ % KARG_SLOTS);
    Queue::Slot &sl = q.slots[si];
    unsigned char *slot = q.karg + si * KARG_SLOT;
    if (sl.used && q.read_seen < sl.packet + 2 && (q.read_seen = h.hsa_queue_load_read_index_scacquire(q.q)) < sl.packet + 2) {
        // the slot's last dispatch may still be reading it: hand over what is pending (it may be that very packet) and
        // wait until the packet BEHIND it has been taken off the ring
        publish(d, h);
        while ((q.read_seen = h.hsa_queue_load_read_index_scacquire(q.q)) < sl.packet + 2) _mm_pause();
    }
    const size_t total = (k->kernarg + 63) & ~(size_t)63;
    if (patch && patch->owner && sl.owner == patch->owner && sl.version == patch->version) {
        for (int i = 0; i < patch->n; ++i)
            *(volatile uint64_t *)(slot + patch->offset[i]) = *(const uint64_t *)((const unsigned char *)args + patch->offset[i]);
        d.last_tail = slot + patch->offset[patch->n ? patch->n - 1 : 0];
    } else {
        memcpy(slot, args, arg_bytes);
        if (total > arg_bytes) memset(slot + arg_bytes, 0, total - arg_bytes);
        d.last_tail = slot + total - 1;
        sl.owner = patch ? patch->owner : 0;
        sl.version = patch ? patch->version : 0;
    }
    uint64_t idx;
    hsa_kernel_dispatch_packet_t *p = (hsa_kernel_dispatch_packet_t *)claim(h, q, &idx);
    sl.packet = idx;
    sl.used = true;
    p->workgroup_size_x = (uint16_t)threads;
    p->workgroup_size_y = 1;
    p->workgroup_size_z = 1;
    p->reserved0 = 0;
    p->grid_size_x = grid * threads;
    p->grid_size_y = 1;
    p->grid_size_z = 1;
    p->private_segment_size = SL_AQL_SCRATCH ? k->priv : 0;
    p->group_segment_size = k->group + lds;
    p->kernel_object = k->object;
    p->kernarg_address = slot;
    p->reserved2 = 0;
    if (d.timeline && !completion.handle && d.tl.size() < 4096 &&
        h.hsa_signal_create(1, 0, nullptr, &completion) == HSA_STATUS_SUCCESS)
        tl_mark(d, h, "dispatch", queue, completion);
    p->completion_signal = completion;
    const uint16_t setup = 3 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
    q.dirty = true;
    const uint32_t head_word = (uint32_t)hd | ((uint32_t)setup << 16);
    if (may_batch && d.batching) {
        if (d.n_pending == Device::MAX_PENDING) publish(d, h);
        d.pending[d.n_pending++] = {queue, (uint32_t *)p, head_word, idx};
        return hipSuccess;
    }
    publish(d, h);
    __atomic_store_n((uint32_t *)p, head_word, __ATOMIC_RELEASE);
    h.hsa_signal_store_screlease(q.q->doorbell_signal, (hsa_signal_value_t)idx);
    return hipSuccess;
}

// a barrier-AND packet on `queue`: waits for the queue's earlier packets (barrier bit) and for `deps`
void barrier(Device &d, const Hsa &h, int queue, const hsa_signal_t *deps, int n_deps, int release, hsa_signal_t completion) {
    Queue &q = d.queues[queue];
    uint64_t idx;
    hsa_barrier_and_packet_t *p = (hsa_barrier_and_packet_t *)claim(h, q, &idx);
    memset((unsigned char *)p + 4, 0, 60);
    for (int i = 0; i < n_deps && i < 5; ++i) p->dep_signal[i] = deps[i];
    p->completion_signal = completion;
    const uint16_t hd = header(HSA_PACKET_TYPE_BARRIER_AND, true, HSA_FENCE_SCOPE_NONE, release);
    __atomic_store_n((uint32_t *)p, (uint32_t)hd, __ATOMIC_RELEASE);
    h.hsa_signal_store_screlease(q.q->doorbell_signal, (hsa_signal_value_t)idx);
}
}  // namespace

hipError_t aql_dispatch(const AqlLaunch &a, hipFunction_t f, unsigned grid, unsigned threads, unsigned lds,
                        const void *args, size_t arg_bytes, const AqlPatch *patch) {
    Device *d = current();
    if (!d || !d->ok || a.queue < 0 || a.queue >= d->n_queues) return hipErrorNotInitialized;
    const Hsa &h = hsa();
    std::lock_guard<std::mutex> lock(d->mu);
    // release-free (opt-in): the step goes without a release fence, so what it wrote stays in the L2 of the XCD its
    // workgroups ran on, which is where the next step's workgroups of the same index read it -- valid as long as
    // workgroup i of this queue keeps its XCD, which every step verifies itself (sl_rowlane.hip: xcd_base / xcd_flag).
    // Otherwise agent scope on both sides, as a HIP stream.
    const int release = a.release_free ? HSA_FENCE_SCOPE_NONE : d->step_release;
    const uint16_t hd = header(HSA_PACKET_TYPE_KERNEL_DISPATCH, true, a.head ? HSA_FENCE_SCOPE_SYSTEM : d->step_acquire, release);
    return emit(*d, h, a.queue, f, grid, threads, lds, args, arg_bytes, hd, hsa_signal_t{0}, true, patch);
}

void aql_warm(int queue, hipFunction_t f, const void *args, size_t arg_bytes, const AqlPatch &patch) {
    Device *d = current();
    if (!d || !d->ok || queue < 0 || queue >= d->n_queues || !patch.owner) return;
    const Hsa &h = hsa();
    std::lock_guard<std::mutex> lock(d->mu);
    const KernelInfo *k = kernel_info(*d, f);
    if (!k || arg_bytes > k->kernarg || k->kernarg > KARG_SLOT) return;
    Queue &q = d->queues[queue];
    const size_t total = (k->kernarg + 63) & ~(size_t)63;
    const uint64_t read = q.read_seen = h.hsa_queue_load_read_index_scacquire(q.q);
    for (size_t si = 0; si < KARG_SLOTS; ++si) {
        Queue::Slot &sl = q.slots[si];
        if (sl.used && read < sl.packet + 2) continue;          // still (possibly) being read: rewritten when its turn comes
        if (sl.owner == patch.owner && sl.version == patch.version) continue;
        unsigned char *slot = q.karg + si * KARG_SLOT;
        memcpy(slot, args, arg_bytes);
        if (total > arg_bytes) memset(slot + arg_bytes, 0, total - arg_bytes);
        d->last_tail = slot + total - 1;
        sl.owner = patch.owner;
        sl.version = patch.version;
    }
}

void aql_begin() {
    Device *d = current();
    if (!d || !d->ok) return;
    std::lock_guard<std::mutex> lock(d->mu);
    d->batching = true;
}

void aql_commit() {
    Device *d = current();
    if (!d || !d->ok) return;
    const Hsa &h = hsa();
    std::lock_guard<std::mutex> lock(d->mu);
    publish(*d, h);
    d->batching = false;
}

void aql_flush() {
    Device *d = current();
    if (!d || !d->ok) return;
    const Hsa &h = hsa();
    std::lock_guard<std::mutex> lock(d->mu);
    publish(*d, h);
}

hipError_t aql_marker(int n_queues, bool force, long long *ticket) {
    Device *d = current();
    if (!d || !d->ok) return hipErrorNotInitialized;
    const Hsa &h = hsa();
    std::lock_guard<std::mutex> lock(d->mu);
    publish(*d, h);
    if (n_queues > d->n_queues) n_queues = d->n_queues;
    *ticket = -1;
    bool any = force;
    for (int i = 0; i < n_queues; ++i) any = any || d->queues[i].dirty;
    if (!any) return hipSuccess;                    // nothing dispatched since the last marker: nothing to wait for
    Marker &m = d->markers[d->next_ticket % MARKERS];
    // (a slot comes round after MARKERS - 1 younger markers; every queue is in-order, so its packets retired long ago)
    for (int i = 0; i < MAX_QUEUES; ++i)
        if (m.waiting[i] && h.hsa_signal_load_relaxed(m.sig[i]) != 0) return hipErrorNotReady;
    m.ticket = d->next_ticket++;
    for (int i = 0; i < MAX_QUEUES; ++i) m.waiting[i] = false;
    for (int i = 0; i < n_queues; ++i) {
        Queue &q = d->queues[i];
        if (!q.dirty && !force) continue;
        if (!m.have[i]) {
            if (h.hsa_signal_create(0, 0, nullptr, &m.sig[i]) != HSA_STATUS_SUCCESS) return hipErrorOutOfMemory;
            m.have[i] = true;
        }
        h.hsa_signal_store_relaxed(m.sig[i], 1);
        m.waiting[i] = true;
        q.dirty = false;
        barrier(*d, h, i, nullptr, 0, HSA_FENCE_SCOPE_SYSTEM, m.sig[i]);
        tl_mark(*d, h, "marker", i, m.sig[i]);
    }
    *ticket = m.ticket;
    return hipSuccess;
}

hipError_t aql_wait(long long ticket) {
    if (ticket < 0) return hipSuccess;
    Device *d = current();
    if (!d || !d->ok) return hipErrorNotInitialized;
    const Hsa &h = hsa();
    hsa_signal_t sig[MAX_QUEUES];
    int n = 0;
    {
        std::lock_guard<std::mutex> lock(d->mu);
        if (d->poisoned) return hipErrorLaunchTimeOut;
        Marker &m = d->markers[ticket % MARKERS];
        if (m.ticket != ticket) return hipSuccess;          // the slot has been reused: that marker was passed long ago
        for (int i = 0; i < MAX_QUEUES; ++i)
            if (m.waiting[i]) sig[n++] = m.sig[i];
    }
    // (outside the lock: the stepping thread keeps dispatching while a worker thread waits here; bounded: a queue
    //  that cannot finish must not take the calling thread with it)
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i)
        while (h.hsa_signal_wait_scacquire(sig[i], HSA_SIGNAL_CONDITION_EQ, 0, 1000000, HSA_WAIT_STATE_ACTIVE) != 0)
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) {
                std::lock_guard<std::mutex> lock(d->mu);
                d->poisoned = true;
                return hipErrorLaunchTimeOut;
            }
    return hipSuccess;
}

// every queue idle, WITHOUT any cache action (self-test only: orders dispatches across queues from the host)
hipError_t aql_drain(int n_queues) {
    Device *d = current();
    if (!d || !d->ok) return hipErrorNotInitialized;
    const Hsa &h = hsa();
    std::lock_guard<std::mutex> lock(d->mu);
    publish(*d, h);
    if (n_queues > d->n_queues) n_queues = d->n_queues;
    for (int i = 0; i < n_queues; ++i) {
        h.hsa_signal_store_relaxed(d->queues[i].fence, 1);
        barrier(*d, h, i, nullptr, 0, HSA_FENCE_SCOPE_NONE, d->queues[i].fence);
    }
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n_queues; ++i)
        while (h.hsa_signal_wait_scacquire(d->queues[i].fence, HSA_SIGNAL_CONDITION_EQ, 0, 1000000, HSA_WAIT_STATE_ACTIVE) != 0)
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) {
                d->poisoned = true;
                return hipErrorLaunchTimeOut;
            }
    return hipSuccess;
}

hipError_t aql_fence(int n_queues) {
    long long ticket = -1;
    const hipError_t err = aql_marker(n_queues, false, &ticket);
    return err != hipSuccess ? err : aql_wait(ticket);
}

void aql_timeline_mark(const char *what) {
    Device *d = current();
    if (!d || !d->ok || !d->timeline) return;
    std::lock_guard<std::mutex> lock(d->mu);
    tl_mark(*d, hsa(), what);
}

void aql_timeline_dump() {
    Device *d = current();
    if (!d || !d->ok || !d->timeline) return;
    const Hsa &h = hsa();
    std::lock_guard<std::mutex> lock(d->mu);
    if (d->tl.empty()) return;
    uint64_t freq = 0;
    h.hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &freq);
    const double us = freq ? 1e6 / (double)freq : 0.0;
    const uint64_t t0 = d->tl[0].host_tick;
    fprintf(stderr, "aql timeline (us since the first mark; host = when the host did it, device = start .. end on the device)\n");
    for (const Device::TlEvent &e : d->tl) {
        if (!e.sig.handle) {
            fprintf(stderr, "  host %9.2f  %s\n", (double)(int64_t)(e.host_tick - t0) * us, e.what);
            continue;
        }
        hsa_amd_profiling_dispatch_time_t dt{};
        uint64_t a = 0, b = 0;
        const bool ok = h.hsa_amd_profiling_get_dispatch_time(d->gpu, e.sig, &dt) == HSA_STATUS_SUCCESS &&
                        h.hsa_amd_profiling_convert_tick_to_system_domain(d->gpu, dt.start, &a) == HSA_STATUS_SUCCESS &&
                        h.hsa_amd_profiling_convert_tick_to_system_domain(d->gpu, dt.end, &b) == HSA_STATUS_SUCCESS;
        fprintf(stderr, "  host %9.2f  %-8s queue %d   device %9.2f .. %9.2f%s\n", (double)(int64_t)(e.host_tick - t0) * us, e.what,
                e.queue, (double)(int64_t)(a - t0) * us, (double)(int64_t)(b - t0) * us, ok ? "" : "  (no timestamps)");
    }
    d->tl.clear();          // (the dispatches' signals are leaked: debugging only)
}

bool aql_poisoned() {
    Device *d = current();
    return d && d->poisoned;
}

}  // namespace sl
