// Error reporting, version and launch lists for libdsnt_hip.so.  State: a thread-local error string and a thread-local
// "recording" pointer; launch lists are objects the caller creates and destroys.
#include "common.h"
#include "stage.h"
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

extern "C" int dsnt_version(void) { return 114; }      // 114 (additive): dsnt_list_fuse*, dsnt_list_stages      // 113 (additive): dsnt_conv_fwd_stream_form      // 112 (additive): dsnt_conv_dgrad_f16x3_stream_apply      // 110: dsnt_f16_prep_weights takes its row width; dsnt_conv1x1_bwd_*; 111 (additive): dsnt_conv1x1_fwd_*, DSNT_BN_FROZEN, dsnt_maxpool2_bwd_add
extern "C" const char* dsnt_last_error(void) { return g_err; }

// ------------------------------------------------------------------ launch lists
#include <vector>
#include <new>

struct dsnt_list {
    struct Entry {
        int kind;                    // 0 launch, 1 lane synchronisation
        int lane, src, dst;
        mutable hipEvent_t ev;       // (created at the first replay: recording needs no device)
        std::function<void(hipStream_t)> fn;
        // a launch that can join a persistent stage (stage.h): its code, grid, workgroup size and parameter block
        int st_code = 0, gx = 0, gy = 0, nt = 0;
        std::vector<unsigned char> params;
        int fused = 0;               // > 0: this entry IS a stage launch over that many recorded launches
        unsigned* st_sync = nullptr; // ... and this is its counter line (device)
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

void dsnt_record_launch_op(dsnt_list* l, int lane, std::function<void(hipStream_t)>&& fn, int code, int gx, int gy, int nt,
                           const void* params, size_t bytes) {
    dsnt_list::Entry e;
    e.kind = 0; e.lane = lane; e.src = e.dst = 0; e.ev = nullptr; e.fn = std::move(fn);
    if (code > 0 && code < DSNT_ST_CODES && bytes <= DSNT_STAGE_PARAM_BYTES && gx > 0 && gy > 0 && (nt == 256 || nt == DSNT_STAGE_NT || nt == 1024)) {
        e.st_code = code; e.gx = gx; e.gy = gy; e.nt = nt;
        e.params.assign((const unsigned char*)params, (const unsigned char*)params + bytes);
    }
    l->entries.push_back(std::move(e));
}

// ---- dsnt_list_fuse: runs of stage-able launches -> persistent stage launches (stage.h)
// A run belongs to ONE lane: consecutive (in that lane's order) stage-able launches with no lane synchronisation that names the
// lane and no bucket mark between the first and the last of them (entries of OTHER lanes may lie between: the lanes only meet at
// synchronisations).  The run's launches are replaced by one entry at the position of the first.
struct StageRun { std::vector<size_t> idx; int lane; int max_vb; };

static void stage_runs(const dsnt_list* l, int min_run, int max_vgrid, std::vector<StageRun>& runs) {
    std::vector<StageRun> open(DSNT_MAX_LANES);
    for (int i = 0; i < DSNT_MAX_LANES; ++i) { open[i].lane = i; open[i].max_vb = 0; }
    auto close = [&](int lane) {
        if ((int)open[lane].idx.size() >= min_run) runs.push_back(open[lane]);
        open[lane].idx.clear(); open[lane].max_vb = 0;
    };
    size_t next_seg = 1;
    for (size_t i = 0; i < l->entries.size(); ++i) {
        while (next_seg < l->seg_start.size() && l->seg_start[next_seg] == i) {       // a bucket mark: every lane's run ends
            for (int k = 0; k < DSNT_MAX_LANES; ++k) close(k);
            ++next_seg;
        }
        const dsnt_list::Entry& e = l->entries[i];
        if (e.kind == 1) { close(e.src); close(e.dst); continue; }
        if (e.lane < 0 || e.lane >= DSNT_MAX_LANES) continue;
        const int nvb = e.gx * e.gy;
        if (e.st_code && !e.fused && nvb <= max_vgrid) {
            open[e.lane].idx.push_back(i);
            if (nvb > open[e.lane].max_vb) open[e.lane].max_vb = nvb;
        } else close(e.lane);
    }
    for (int k = 0; k < DSNT_MAX_LANES; ++k) close(k);
}

extern "C" int64_t dsnt_list_fuse_bytes(const dsnt_list* l, int min_run, int max_vgrid) {
    if (!l) return 0;
    std::vector<StageRun> runs;
    stage_runs(l, min_run < 2 ? 2 : min_run, max_vgrid, runs);
    size_t bytes = 0;
    for (auto& r : runs) bytes += r.idx.size() * sizeof(DsntStageOp) + 64;       // the table + one 64-byte line of counters
    return (int64_t)bytes;
}

// what dsnt_list_fuse would do: stages it would create, (launches) recorded launches they would replace; needs no device
extern "C" int dsnt_list_fuse_plan(const dsnt_list* l, int min_run, int max_vgrid, int* launches) {
    int n = 0;
    std::vector<StageRun> runs;
    if (l) stage_runs(l, min_run < 2 ? 2 : min_run, max_vgrid, runs);
    for (auto& r : runs) n += (int)r.idx.size();
    if (launches) *launches = n;
    return (int)runs.size();
}

extern "C" int dsnt_list_fuse(dsnt_list* l, void* workspace, int64_t bytes, int min_run, int max_vgrid, int grid_cap) {
    DSNT_REQUIRE(l && !g_rec, DSNT_ERR_ARG, "dsnt_list_fuse: null list, or this thread is recording");
    DSNT_REQUIRE(grid_cap > 0 && grid_cap <= 256 && max_vgrid > 0, DSNT_ERR_ARG, "dsnt_list_fuse: grid_cap must be 1..256 (co-resident workgroups)");
    if (min_run < 2) min_run = 2;
    std::vector<StageRun> runs;
    stage_runs(l, min_run, max_vgrid, runs);
    if (runs.empty()) return 0;
    size_t need = 0;
    for (auto& r : runs) need += r.idx.size() * sizeof(DsntStageOp) + 64;
    DSNT_REQUIRE(workspace && (((uintptr_t)workspace) & 63u) == 0 && (size_t)bytes >= need, DSNT_ERR_ARG,
                 "dsnt_list_fuse: workspace of %zu bytes (64-byte aligned) needed, %lld given", need, (long long)bytes);
    std::vector<unsigned char> host(need, 0);
    std::vector<char> drop(l->entries.size(), 0);
    size_t off = 0;
    for (auto& r : runs) {
        DsntStageOp* tab = reinterpret_cast<DsntStageOp*>(host.data() + off);
        for (size_t k = 0; k < r.idx.size(); ++k) {
            const dsnt_list::Entry& e = l->entries[r.idx[k]];
            tab[k].code = e.st_code; tab[k].gx = e.gx; tab[k].gy = e.gy; tab[k].nt = e.nt;
            memcpy(tab[k].params, e.params.data(), e.params.size());
            if (k) drop[r.idx[k]] = 1;
        }
        const DsntStageOp* tab_dev = reinterpret_cast<const DsntStageOp*>((unsigned char*)workspace + off);
        unsigned* sync_dev = reinterpret_cast<unsigned*>((unsigned char*)workspace + off + r.idx.size() * sizeof(DsntStageOp));
        const int nops = (int)r.idx.size();
        const int grid = r.max_vb < grid_cap ? r.max_vb : grid_cap;
        dsnt_list::Entry& first = l->entries[r.idx[0]];
        first.fn = [=](hipStream_t s) { dsnt_stage_launch(tab_dev, nops, sync_dev, grid, s); };
        first.fused = nops; first.st_code = 0; first.params.clear(); first.st_sync = sync_dev;
        off += r.idx.size() * sizeof(DsntStageOp) + 64;
    }
    // the tables and zeroed counters, once (a blocking copy: list building is not on any hot path)
    if (hipMemcpy(workspace, host.data(), need, hipMemcpyHostToDevice) != hipSuccess)
        return dsnt_set_error(DSNT_ERR_HIP, "dsnt_list_fuse: hipMemcpy of the stage tables failed");
    // compact the list (segment starts move with it)
    std::vector<dsnt_list::Entry> kept;
    kept.reserve(l->entries.size());
    std::vector<size_t> new_pos(l->entries.size() + 1, 0);
    for (size_t i = 0; i < l->entries.size(); ++i) {
        new_pos[i] = kept.size();
        if (!drop[i]) kept.push_back(std::move(l->entries[i]));
    }
    new_pos[l->entries.size()] = kept.size();
    for (auto& ss : l->seg_start) ss = new_pos[ss];
    l->entries.swap(kept);
    return (int)runs.size();
}

// Diagnostics: stages whose barrier gave up (word 2 of the counter line; blocking copies).
extern "C" int dsnt_list_stage_errors(const dsnt_list* l) {
    int bad = 0;
    if (l) for (auto& e : l->entries) if (e.fused && e.st_sync) {
        unsigned w = 0;
        if (hipMemcpy(&w, e.st_sync + 2, sizeof(w), hipMemcpyDeviceToHost) != hipSuccess)
            return dsnt_set_error(DSNT_ERR_HIP, "dsnt_list_stage_errors: hipMemcpy failed");
        bad += w != 0;
    }
    return bad;
}

// Diagnostics: how many entries of the list are stage launches, and how many recorded launches they stand for.
extern "C" int dsnt_list_stages(const dsnt_list* l, int* launches_inside) {
    int n = 0, m = 0;
    if (l) for (auto& e : l->entries) if (e.fused) { ++n; m += e.fused; }
    if (launches_inside) *launches_inside = m;
    return n;
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
            if (!e.ev && hipEventCreateWithFlags(&e.ev, hipEventDisableTiming) != hipSuccess)
                return dsnt_set_error(DSNT_ERR_HIP, "dsnt_list_replay: hipEventCreate");
            if (hipEventRecord(e.ev, (hipStream_t)streams[e.src]) != hipSuccess ||
                hipStreamWaitEvent((hipStream_t)streams[e.dst], e.ev, 0) != hipSuccess)
                return dsnt_set_error(DSNT_ERR_HIP, "dsnt_list_replay: event record / wait failed");
        }
    }
    hipError_t e_ = hipGetLastError();
    if (e_ != hipSuccess) return dsnt_set_error(DSNT_ERR_HIP, "dsnt_list_replay: %s", hipGetErrorString(e_));
    return DSNT_OK;
}
