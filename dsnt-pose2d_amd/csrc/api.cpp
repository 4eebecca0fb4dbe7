// Error reporting, version and launch lists for libdsnt_hip.so.  State: a thread-local error string and a thread-local
// "recording" pointer; launch lists are objects the caller creates and destroys.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

int dsnt_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int dsnt_device_id(void) {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0) d = 0;
    return d < DSNT_MAX_DEVICES ? d : DSNT_MAX_DEVICES - 1;
}

int dsnt_device_cus(void) {
    static std::atomic<int> cus[DSNT_MAX_DEVICES];
    const int d = dsnt_device_id();
    int c = cus[d].load(std::memory_order_acquire);
    if (!c) {
        hipDeviceProp_t prop;
        int dev = 0;
        c = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                ? prop.multiProcessorCount : 256;
        cus[d].store(c, std::memory_order_release);
    }
    return c;
}

extern "C" int dsnt_version(void) { return 113; }      // 113 (additive): dsnt_conv_fwd_stream_form      // 112 (additive): dsnt_conv_dgrad_f16x3_stream_apply      // 110: dsnt_f16_prep_weights takes its row width; dsnt_conv1x1_bwd_*; 111 (additive): dsnt_conv1x1_fwd_*, DSNT_BN_FROZEN, dsnt_maxpool2_bwd_add
extern "C" const char* dsnt_last_error(void) { return g_err; }

// ------------------------------------------------------------------ launch lists
#include <vector>
#include <new>

struct dsnt_list {
    struct Entry {
        int kind;                    // 0 launch, 1 lane synchronisation
        int lane, src, dst;
        hipEvent_t ev;
        std::function<void(hipStream_t)> fn;
    };
    std::vector<Entry> entries;
    std::vector<size_t> seg_start;   // first entry of every segment
    dsnt_list() { seg_start.push_back(0); }
};

static thread_local dsnt_list* g_rec = nullptr;

dsnt_list* dsnt_recording(void) { return g_rec; }

void dsnt_record_launch(dsnt_list* l, int lane, std::function<void(hipStream_t)>&& fn) {
    dsnt_list::Entry e;
    e.kind = 0; e.lane = lane; e.src = e.dst = 0; e.ev = nullptr; e.fn = std::move(fn);
    l->entries.push_back(std::move(e));
}

extern "C" dsnt_list* dsnt_list_create(void) { return new (std::nothrow) dsnt_list(); }

extern "C" void dsnt_list_destroy(dsnt_list* l) {
    if (!l) return;
    if (g_rec == l) g_rec = nullptr;
    for (auto& e : l->entries)
        if (e.ev) (void)hipEventDestroy(e.ev);
    delete l;
}

extern "C" int dsnt_list_begin(dsnt_list* l) {
    DSNT_REQUIRE(l && !g_rec, DSNT_ERR_ARG, "dsnt_list_begin: null list, or this thread is already recording");
    g_rec = l;
    return DSNT_OK;
}

extern "C" int dsnt_list_end(void) {
    DSNT_REQUIRE(g_rec, DSNT_ERR_ARG, "dsnt_list_end: this thread is not recording");
    g_rec = nullptr;
    return DSNT_OK;
}

extern "C" int dsnt_list_sync(dsnt_list* l, int src_lane, int dst_lane) {
    DSNT_REQUIRE(l && src_lane >= 0 && dst_lane >= 0 && src_lane < DSNT_MAX_LANES && dst_lane < DSNT_MAX_LANES, DSNT_ERR_ARG,
                 "dsnt_list_sync: bad argument");
    dsnt_list::Entry e;
    e.kind = 1; e.lane = 0; e.src = src_lane; e.dst = dst_lane; e.ev = nullptr;
    if (hipEventCreateWithFlags(&e.ev, hipEventDisableTiming) != hipSuccess) return dsnt_set_error(DSNT_ERR_HIP, "dsnt_list_sync: hipEventCreate");
    l->entries.push_back(std::move(e));
    return DSNT_OK;
}

extern "C" int dsnt_list_mark(dsnt_list* l) {
    DSNT_REQUIRE(l, DSNT_ERR_ARG, "dsnt_list_mark: null list");
    l->seg_start.push_back(l->entries.size());
    return (int)l->seg_start.size() - 1;
}

extern "C" int dsnt_list_segments(const dsnt_list* l) { return l ? (int)l->seg_start.size() : 0; }
extern "C" int dsnt_list_size(const dsnt_list* l) { return l ? (int)l->entries.size() : 0; }

extern "C" int dsnt_list_replay(const dsnt_list* l, int segment, void* const* streams, int nstreams) {
    DSNT_REQUIRE(l && streams && nstreams > 0 && nstreams <= DSNT_MAX_LANES, DSNT_ERR_ARG, "dsnt_list_replay: bad argument");
    DSNT_REQUIRE(!g_rec, DSNT_ERR_ARG, "dsnt_list_replay: this thread is recording");
    const int nseg = (int)l->seg_start.size();
    DSNT_REQUIRE(segment >= -1 && segment < nseg, DSNT_ERR_ARG, "dsnt_list_replay: segment %d of %d", segment, nseg);
    const size_t a = segment < 0 ? 0 : l->seg_start[segment];
    const size_t b = (segment < 0 || segment + 1 == nseg) ? l->entries.size() : l->seg_start[segment + 1];
    for (size_t i = a; i < b; ++i) {
        const dsnt_list::Entry& e = l->entries[i];
        if (e.kind == 0) {
            DSNT_REQUIRE(e.lane >= 0 && e.lane < nstreams, DSNT_ERR_ARG, "dsnt_list_replay: entry %zu wants lane %d of %d", i, e.lane, nstreams);
            e.fn((hipStream_t)streams[e.lane]);
        } else {
            DSNT_REQUIRE(e.src < nstreams && e.dst < nstreams, DSNT_ERR_ARG, "dsnt_list_replay: lane out of range");
            if (hipEventRecord(e.ev, (hipStream_t)streams[e.src]) != hipSuccess ||
                hipStreamWaitEvent((hipStream_t)streams[e.dst], e.ev, 0) != hipSuccess)
                return dsnt_set_error(DSNT_ERR_HIP, "dsnt_list_replay: event record / wait failed");
        }
    }
    hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) return dsnt_set_error(DSNT_ERR_HIP, "dsnt_list_replay: %s", hipGetErrorString(e_));
    return DSNT_OK;
}
