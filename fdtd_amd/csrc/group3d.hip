// group3d.hip -- several z-slab engines of one 3-D problem driven from ONE host thread: the slab
// choreography (edge planes first, ghost planes copied device to device while the bulk is computed)
// on HIP streams and events, written against the library's own C ABI.  The reference steps its whole
// grid from one thread (FDTD3D::run, fdtd3d/src/FDTD3D.cpp:1085-1086); this is that loop for a grid
// spread over the GPUs of one node, for callers that do not want one process per GPU.
#include "common.hpp"

#include <dlfcn.h>

#include <cstdlib>
#include <memory>
#include <unordered_map>
#include <utility>

using namespace fdtd;

struct fdtd3d_group {
    fdtd3d_desc d{};
    bool fused = false;
    // Two steps per launch: the slabs then OVERLAP by one layer either side of every cut (slab r stores the layers
    // [lo[r]-1, hi[r]+1) and answers for [lo[r], hi[r])), because the two-step kernel recomputes its halos from what a
    // slab stores and is exact only two planes away from a cut; after every launch the two planes beyond each cut are
    // overwritten with the neighbour's exact ones (step_pairs below).
    bool pairs = false;
    std::vector<int> lo, hi;                    // the layers each slab answers for
    std::vector<int> xchg_ids;                  // arrays whose planes cross a cut after a launch
    std::vector<fdtd3d_handle> member;          // bottom slab first
    std::vector<int> device;
    // per member: the stream its kernels run on, and the streams that carry its planes to the slab
    // above (up) and below (down); all blocking streams, so that calls on a member's device through
    // the null stream (downloads, observables) order themselves behind the step
    std::vector<hipStream_t> run, up, down;
    // Events.  HIP only records an event on a stream of the device the event was created on (waiting
    // on it from another device's stream is fine), so every event belongs to the slab whose stream
    // RECORDS it -- all four of slab r live on device[r]:
    //   edge_done[r]  on run[r]:  lowest planes done (also: r's low ghost has been consumed)
    //   top_done[r]   on run[r]:  top layer done (also: r's high ghost has been consumed)
    //   sent_down[r]  on down[r]: r's lowest D/E planes have arrived in the high ghost of slab r-1
    //   sent_up[r]    on up[r]:   r's top B/H planes have arrived in the low ghost of slab r+1
    // Slab r-1 waits for sent_down[r], slab r+1 for sent_up[r].
    std::vector<hipEvent_t> edge_done, top_done, sent_down, sent_up;
    bool exchanged_once = false;                // overlapping slabs: the arrival events have been recorded at least once
    // Transport of the planes: device-to-device copies issued by the sender's copy stream (default), or RCCL
    // ncclSend / ncclRecv (one communicator per distinct device, created with ncclCommInitAll; slabs that share a device
    // share its rank and exchange with themselves).  With RCCL the receiving side has streams and arrival events of its
    // own: rx_lo[r] / got_lo[r] for what comes from the slab below, rx_hi[r] / got_hi[r] for what comes from above.
    bool rccl = false;
    std::vector<void*> comm;                    // ncclComm_t per rank
    std::vector<int> rank_of;                   // slab -> rank
    std::vector<hipStream_t> rx_lo, rx_hi;
    std::vector<hipEvent_t> got_lo, got_hi;
    // The slab every stream and event was created for, so that record() can refuse a pair that would
    // sit on different devices on a multi-GPU node -- also when all slabs share one device, where HIP
    // itself would accept it.
    std::unordered_map<void*, int> owner;
    long records_checked = 0;
    int owner_of(void* handle) const {
        const auto it = owner.find(handle);
        return it == owner.end() ? -1 : it->second;
    }

    ~fdtd3d_group() {
        for (size_t r = 0; r < member.size(); ++r) {
            DeviceScope scope(device[r]);
            (void)hipDeviceSynchronize();
        }
        for (size_t r = 0; r < member.size(); ++r) {
            DeviceScope scope(device[r]);
            for (auto* v : {&run, &up, &down, &rx_lo, &rx_hi}) if (r < v->size() && (*v)[r]) (void)hipStreamDestroy((*v)[r]);
            for (auto* v : {&edge_done, &top_done, &sent_down, &sent_up, &got_lo, &got_hi}) if (r < v->size() && (*v)[r]) (void)hipEventDestroy((*v)[r]);
            if (member[r]) fdtd3d_destroy(member[r]);
        }
        destroy_comms();
    }
    void destroy_comms();
};

namespace {

bool g_misown_events = false;       // see fdtd3d_group_debug_misown_events

// hipEventRecord with the ownership rule enforced: event and stream must have been created for the
// same slab (hence on the same device).
int record(fdtd3d_group* g, hipEvent_t ev, hipStream_t stream) {
    const int eo = g->owner_of((void*)ev), so = g->owner_of((void*)stream);
    ++g->records_checked;
    if (eo < 0 || eo != so)
        return fail(FDTD_ERR_STATE, "group: event of slab %d recorded on a stream of slab %d (an event must be recorded on its own device)", eo, so);
    FDTD_HIP(hipEventRecord(ev, stream));
    return FDTD_OK;
}

// ---- RCCL, bound at run time ------------------------------------------------------------------------------------
// The library is only looked for when a group asks for this transport, so nothing links against it: a copy the process
// has loaded already (PyTorch brings one) is used if there is one, else the ROCm installation's.
struct RcclApi {
    void* lib = nullptr;
    int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void* buf, size_t count, int datatype, int peer, void* comm, hipStream_t stream) = nullptr;
    int (*Recv)(void* buf, size_t count, int datatype, int peer, void* comm, hipStream_t stream) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
constexpr int kNcclInt8 = 0;            // ncclInt8 / ncclChar (rccl.h): planes travel as bytes
RcclApi g_rccl;
int g_default_transport = -1;           // -1: FDTD_GROUP_TRANSPORT in the environment decides; 0 copies, 1 RCCL

int rccl_load() {
    if (g_rccl.lib) return FDTD_OK;
    void* h = nullptr;
    for (const char* name : {"librccl.so.1", "librccl.so"}) if (!h) h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) if (!h) h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(FDTD_ERR_STATE, "group: RCCL transport asked for, but librccl cannot be loaded (%s)", dlerror());
    RcclApi a;
    a.lib = h;
    bool ok = true;
    auto sym = [&](auto& fn, const char* name) { fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(h, name)); ok = ok && fn != nullptr; };
    sym(a.CommInitAll, "ncclCommInitAll"); sym(a.CommDestroy, "ncclCommDestroy"); sym(a.GroupStart, "ncclGroupStart");
    sym(a.GroupEnd, "ncclGroupEnd"); sym(a.Send, "ncclSend"); sym(a.Recv, "ncclRecv"); sym(a.GetErrorString, "ncclGetErrorString");
    if (!ok) return fail(FDTD_ERR_STATE, "group: librccl lacks one of ncclCommInitAll / ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd");
    g_rccl = a;
    return FDTD_OK;
}
#define FDTD_NCCL(expr)                                                                                       \
    do {                                                                                                      \
        const int r_ = (expr);                                                                                \
        if (r_ != 0) return fail(FDTD_ERR_HIP, "%s failed: %s", #expr, g_rccl.GetErrorString(r_));           \
    } while (0)

// ---- moving planes between neighbouring slabs -----------------------------------------------------------------
struct Piece { const void* from; void* to; size_t nb; };

// The ghost-plane pair (components 0 and 1) of the one-step choreographies.
int halo_pieces(fdtd3d_group* g, int src, int which_src, int dst, int which_dst, int parity, std::vector<Piece>& out) {
    for (int c = 0; c < 2; ++c) {
        void *from = nullptr, *to = nullptr;
        size_t nb = 0, nb2 = 0;
        int rc = g->fused ? fdtd3d_halo_at(g->member[src], parity, which_src, c, &from, &nb) : fdtd3d_halo(g->member[src], which_src, c, &from, &nb);
        if (rc) return rc;
        rc = g->fused ? fdtd3d_halo_at(g->member[dst], parity, which_dst, c, &to, &nb2) : fdtd3d_halo(g->member[dst], which_dst, c, &to, &nb2);
        if (rc) return rc;
        if (nb != nb2) return fail(FDTD_ERR_STATE, "group: ghost planes of slabs %d and %d differ in size", src, dst);
        out.push_back(Piece{from, to, nb});
    }
    return FDTD_OK;
}
// Two planes of every exchanged array from global plane k (two steps per launch: the planes beyond a cut), current
// buffers of both slabs.
int cut_pieces(fdtd3d_group* g, int src, int dst, int k, std::vector<Piece>& out) {
    for (int id : g->xchg_ids) {
        void *from = nullptr, *to = nullptr;
        size_t nb = 0, nb2 = 0;
        if (int rc = fdtd3d_planes(g->member[src], id, k, 2, &from, &nb)) return rc;
        if (int rc = fdtd3d_planes(g->member[dst], id, k, 2, &to, &nb2)) return rc;
        if (nb != nb2) return fail(FDTD_ERR_STATE, "group: planes of array %d differ in size between slabs %d and %d", id, src, dst);
        out.push_back(Piece{from, to, nb});
    }
    return FDTD_OK;
}
// Moves `pieces` from slab src to the slab above (`up`) or below it, once `src_ready` (the planes hold what is to be
// sent) and `dst_free` (what the destination planes held has been used).  Copies: the sender's copy stream waits for
// both and copies, over xGMI between different devices.  RCCL: ncclSend on the sender's copy stream after src_ready,
// ncclRecv on the receiver's stream after dst_free, one group per call.  sent_up / sent_down[src] is recorded on the
// sender's stream either way; a consumer on the receiving slab waits for arrival().
int transfer(fdtd3d_group* g, bool up, int src, hipEvent_t src_ready, hipEvent_t dst_free, const std::vector<Piece>& pieces) {
    const int dst = up ? src + 1 : src - 1;
    hipStream_t tx = up ? g->up[src] : g->down[src];
    hipEvent_t sent = up ? g->sent_up[src] : g->sent_down[src];
    if (!g->rccl) {
        DeviceScope scope(g->device[src]);
        FDTD_HIP(hipStreamWaitEvent(tx, src_ready, 0));
        FDTD_HIP(hipStreamWaitEvent(tx, dst_free, 0));
        for (const Piece& p : pieces) {
            if (g->device[src] == g->device[dst]) FDTD_HIP(hipMemcpyAsync(p.to, p.from, p.nb, hipMemcpyDeviceToDevice, tx));
            else FDTD_HIP(hipMemcpyPeerAsync(p.to, g->device[dst], p.from, g->device[src], p.nb, tx));
        }
        return record(g, sent, tx);
    }
    hipStream_t rx = up ? g->rx_lo[dst] : g->rx_hi[dst];
    hipEvent_t got = up ? g->got_lo[dst] : g->got_hi[dst];
    { DeviceScope scope(g->device[src]); FDTD_HIP(hipStreamWaitEvent(tx, src_ready, 0)); }
    { DeviceScope scope(g->device[dst]); FDTD_HIP(hipStreamWaitEvent(rx, dst_free, 0)); }
    FDTD_NCCL(g_rccl.GroupStart());
    for (const Piece& p : pieces) {
        { DeviceScope scope(g->device[src]); FDTD_NCCL(g_rccl.Send(p.from, p.nb, kNcclInt8, g->rank_of[dst], g->comm[g->rank_of[src]], tx)); }
        { DeviceScope scope(g->device[dst]); FDTD_NCCL(g_rccl.Recv(p.to, p.nb, kNcclInt8, g->rank_of[src], g->comm[g->rank_of[dst]], rx)); }
    }
    FDTD_NCCL(g_rccl.GroupEnd());
    { DeviceScope scope(g->device[src]); if (int rc = record(g, sent, tx)) return rc; }
    { DeviceScope scope(g->device[dst]); if (int rc = record(g, got, rx)) return rc; }
    return FDTD_OK;
}
// The event after which the planes that slab `src` sent up (down) are in its neighbour's memory.
hipEvent_t arrival(fdtd3d_group* g, bool up, int src) {
    if (!g->rccl) return up ? g->sent_up[src] : g->sent_down[src];
    return up ? g->got_lo[src + 1] : g->got_hi[src - 1];
}

// ---- two steps per launch ------------------------------------------------------------------------------------
// After the launches that produced the planes next to the cuts (recorded as edge_done): the two planes below every cut
// go up, the two above it go down.  The copies run on the copy streams, beside whatever the launch streams do next
// (the bulk of a two-step launch neither reads the buffers the planes land in nor writes those planes).
int exchange_cut_planes(fdtd3d_group* g) {
    const int R = (int)g->member.size();
    int rc;
    for (int r = 0; r + 1 < R; ++r) {
        const int cut = g->hi[r];
        std::vector<Piece> going_up, going_down;
        if ((rc = cut_pieces(g, r, r + 1, cut - 2, going_up))) return rc;
        if ((rc = cut_pieces(g, r + 1, r, cut, going_down))) return rc;
        if ((rc = transfer(g, true, r, g->edge_done[r], g->edge_done[r + 1], going_up))) return rc;
        if ((rc = transfer(g, false, r + 1, g->edge_done[r + 1], g->edge_done[r], going_down))) return rc;
    }
    return FDTD_OK;
}
// A slab's next launch reads the planes that arrived and overwrites (two launches on) the ones that left.
int wait_for_cut_planes(fdtd3d_group* g, int r) {
    const int R = (int)g->member.size();
    if (!g->exchanged_once) return FDTD_OK;
    if (r > 0) { FDTD_HIP(hipStreamWaitEvent(g->run[r], arrival(g, true, r - 1), 0)); FDTD_HIP(hipStreamWaitEvent(g->run[r], g->sent_down[r], 0)); }
    if (r + 1 < R) { FDTD_HIP(hipStreamWaitEvent(g->run[r], arrival(g, false, r + 1), 0)); FDTD_HIP(hipStreamWaitEvent(g->run[r], g->sent_up[r], 0)); }
    return FDTD_OK;
}
// Steps n and n+1 in one launch per slab, cut in two (fdtd3d_pair_part): the three planes at either end first, the
// exchange is started, the bulk runs while the planes travel.
int step_pair(fdtd3d_group* g, int n) {
    const int R = (int)g->member.size();
    int rc;
    for (int r = 0; r < R; ++r) {
        DeviceScope scope(g->device[r]);
        if ((rc = wait_for_cut_planes(g, r))) return rc;
        if ((rc = fdtd3d_pair_part(g->member[r], n, FDTD3D_PART_EDGE, g->run[r]))) return rc;
        if ((rc = record(g, g->edge_done[r], g->run[r]))) return rc;
    }
    if ((rc = exchange_cut_planes(g))) return rc;
    g->exchanged_once = true;
    for (int r = 0; r < R; ++r) {
        DeviceScope scope(g->device[r]);
        if ((rc = fdtd3d_pair_part(g->member[r], n, FDTD3D_PART_INTERIOR, g->run[r]))) return rc;
    }
    return FDTD_OK;
}
// The same with the RMS window of FDTD3D::run advanced inside the launches (fdtd3d_pair_part_observed).  A slab below
// another one leaves the second step's contribution of its top layer out -- it needs E one layer further up, which is
// exact only once the planes from above have arrived -- and adds it afterwards with a one-layer launch of the observe
// kernel, queued behind the bulk on its launch stream.
int step_pair_observed(fdtd3d_group* g, int n, double weight, int window) {
    const int R = (int)g->member.size();
    int rc;
    for (int r = 0; r < R; ++r) {
        DeviceScope scope(g->device[r]);
        if ((rc = wait_for_cut_planes(g, r))) return rc;
        if ((rc = fdtd3d_pair_part_observed(g->member[r], n, FDTD3D_PART_EDGE, weight, window, r + 1 < R, g->run[r]))) return rc;
        if ((rc = record(g, g->edge_done[r], g->run[r]))) return rc;
    }
    if ((rc = exchange_cut_planes(g))) return rc;
    g->exchanged_once = true;
    for (int r = 0; r < R; ++r) {
        DeviceScope scope(g->device[r]);
        if ((rc = fdtd3d_pair_part_observed(g->member[r], n, FDTD3D_PART_INTERIOR, weight, window, r + 1 < R, g->run[r]))) return rc;
    }
    for (int r = 0; r < R; ++r) {
        DeviceScope scope(g->device[r]);
        if (r + 1 < R) FDTD_HIP(hipStreamWaitEvent(g->run[r], arrival(g, false, r + 1), 0));
        if ((rc = fdtd3d_observe_finish_pair(g->member[r], n, weight, window, r + 1 < R, g->run[r]))) return rc;
    }
    return FDTD_OK;
}
// One step on overlapping slabs: the one-step kernel over everything a slab stores, then the same exchange.
int step_single_overlapping(fdtd3d_group* g, int n) {
    const int R = (int)g->member.size();
    int rc;
    for (int r = 0; r < R; ++r) {
        DeviceScope scope(g->device[r]);
        if ((rc = wait_for_cut_planes(g, r))) return rc;
        if ((rc = fdtd3d_fused(g->member[r], n, FDTD3D_PART_ALL, g->run[r]))) return rc;
        if ((rc = record(g, g->edge_done[r], g->run[r]))) return rc;
    }
    if ((rc = exchange_cut_planes(g))) return rc;
    g->exchanged_once = true;
    // probes sample the layers a slab answers for, which the exchange does not touch
    for (int r = 0; r < R; ++r) {
        DeviceScope scope(g->device[r]);
        if ((rc = fdtd3d_probe_record(g->member[r], g->run[r]))) return rc;
    }
    return FDTD_OK;
}

// Step n with the fused kernel.  Order per slab: lowest planes (their D starts travelling down), highest
// planes, top layer's B once the D plane from above is there (it starts travelling up), then the bulk.
int step_fused(fdtd3d_group* g, int n) {
    const int R = (int)g->member.size();
    int rc;
    for (int r = 0; r < R; ++r) {
        DeviceScope scope(g->device[r]);
        if (r > 0) FDTD_HIP(hipStreamWaitEvent(g->run[r], arrival(g, true, r - 1), 0));
        if ((rc = fdtd3d_fused(g->member[r], n, FDTD3D_PART_EDGE, g->run[r]))) return rc;
        if ((rc = record(g, g->edge_done[r], g->run[r]))) return rc;
    }
    const int parity = fdtd3d_buffer_parity(g->member[0]);       // all members change buffers together
    for (int r = 1; r < R; ++r) {
        std::vector<Piece> pieces;
        if ((rc = halo_pieces(g, r, FDTD3D_HALO_SEND_DOWN, r - 1, FDTD3D_HALO_RECV_HI, parity, pieces))) return rc;
        if ((rc = transfer(g, false, r, g->edge_done[r], g->top_done[r - 1], pieces))) return rc;      // (top_done: the ghost's previous content has been used)
    }
    for (int r = 0; r < R; ++r) {
        DeviceScope scope(g->device[r]);
        if ((rc = fdtd3d_fused(g->member[r], n, FDTD3D_PART_TOP, g->run[r]))) return rc;
        if (r + 1 < R) FDTD_HIP(hipStreamWaitEvent(g->run[r], arrival(g, false, r + 1), 0));
        if ((rc = fdtd3d_fused_top(g->member[r], g->run[r]))) return rc;
        if ((rc = record(g, g->top_done[r], g->run[r]))) return rc;
    }
    for (int r = 0; r + 1 < R; ++r) {
        std::vector<Piece> pieces;
        if ((rc = halo_pieces(g, r, FDTD3D_HALO_SEND_UP, r + 1, FDTD3D_HALO_RECV_LO, parity, pieces))) return rc;
        if ((rc = transfer(g, true, r, g->top_done[r], g->edge_done[r + 1], pieces))) return rc;        // (edge_done: the ghost's previous content has been used)
    }
    for (int r = 0; r < R; ++r) {
        DeviceScope scope(g->device[r]);
        if ((rc = fdtd3d_fused(g->member[r], n, FDTD3D_PART_INTERIOR, g->run[r]))) return rc;
        if ((rc = fdtd3d_probe_record(g->member[r], g->run[r]))) return rc;
    }
    return FDTD_OK;
}

// Step n with the two half-step kernels (FULL and LEAN storage): the bulk of each half never looks at
// a ghost plane, so it runs while the plane the edge layer needs is still travelling.
int step_halves(fdtd3d_group* g, int n) {
    const int R = (int)g->member.size();
    int rc;
    for (int r = 0; r < R; ++r) {
        DeviceScope scope(g->device[r]);
        if ((rc = fdtd3d_half_e(g->member[r], n, FDTD3D_PART_INTERIOR, g->run[r]))) return rc;
        if (r > 0) FDTD_HIP(hipStreamWaitEvent(g->run[r], arrival(g, true, r - 1), 0));
        if ((rc = fdtd3d_half_e(g->member[r], n, FDTD3D_PART_EDGE, g->run[r]))) return rc;
        if ((rc = record(g, g->edge_done[r], g->run[r]))) return rc;
    }
    for (int r = 1; r < R; ++r) {
        std::vector<Piece> pieces;
        if ((rc = halo_pieces(g, r, FDTD3D_HALO_SEND_DOWN, r - 1, FDTD3D_HALO_RECV_HI, 0, pieces))) return rc;
        if ((rc = transfer(g, false, r, g->edge_done[r], g->top_done[r - 1], pieces))) return rc;
    }
    for (int r = 0; r < R; ++r) {
        DeviceScope scope(g->device[r]);
        if ((rc = fdtd3d_half_h(g->member[r], FDTD3D_PART_INTERIOR, g->run[r]))) return rc;
        if (r + 1 < R) FDTD_HIP(hipStreamWaitEvent(g->run[r], arrival(g, false, r + 1), 0));
        if ((rc = fdtd3d_half_h(g->member[r], FDTD3D_PART_EDGE, g->run[r]))) return rc;
        if ((rc = record(g, g->top_done[r], g->run[r]))) return rc;
    }
    for (int r = 0; r + 1 < R; ++r) {
        std::vector<Piece> pieces;
        if ((rc = halo_pieces(g, r, FDTD3D_HALO_SEND_UP, r + 1, FDTD3D_HALO_RECV_LO, 0, pieces))) return rc;
        if ((rc = transfer(g, true, r, g->top_done[r], g->edge_done[r + 1], pieces))) return rc;
    }
    for (int r = 0; r < R; ++r) {
        DeviceScope scope(g->device[r]);
        if ((rc = fdtd3d_probe_record(g->member[r], g->run[r]))) return rc;
    }
    return FDTD_OK;
}

int build_group(fdtd3d_group* g, const fdtd3d_desc* desc, int n_slabs, const int* devices) {
    g->d = *desc;
    g->fused = desc->mode == FDTD3D_MODE_FUSED;
    int ndev = 0;
    FDTD_HIP(hipGetDeviceCount(&ndev));
    if (ndev < 1) return fail(FDTD_ERR_HIP, "no HIP device visible");
    const int base = desc->Nz / n_slabs, extra = desc->Nz % n_slabs;
    for (int r = 0, lo = 0; r < n_slabs; ++r) {
        g->lo.push_back(lo);
        lo += base + (r < extra ? 1 : 0);
        g->hi.push_back(lo);
    }
    // Two steps per launch where they pay (as fdtd3d_step decides for a whole grid: enough cells per slab, or FDTD_PAIR=1
    // in the environment; FDTD_PAIR=0: never) and where the slabs allow them: FUSED storage, every slab at least two
    // layers thick, every cut clear of the z-face CPML layers by the planes that cross it (their state is not exchanged).
    bool pml = false;
    for (int q = 0; q < 6; ++q) pml = pml || desc->n_pml[q] != 0;
    const char* env_pair = getenv("FDTD_PAIR");
    const double slab_cells = (double)desc->Nx * desc->Ny * base, edge = pml ? 448.0 : 384.0;
    bool want_pairs = g->fused && n_slabs > 1 && (env_pair ? atoi(env_pair) != 0 : slab_cells >= edge * edge * edge);
    for (int r = 0; r < n_slabs && want_pairs; ++r) {
        if (g->hi[r] - g->lo[r] < 2) want_pairs = false;
        const int cut = g->lo[r];
        if (r > 0 && !(cut - 2 > desc->n_pml[4] + 1 && cut + 2 < desc->Nz - desc->n_pml[5] - 1)) want_pairs = false;
    }
    for (int r = 0; r < n_slabs; ++r) {
        const int dev = devices ? devices[r] : r % ndev;
        if (dev < 0 || dev >= ndev) return fail(FDTD_ERR_ARG, "group: slab %d asks for device %d, %d visible", r, dev, ndev);
        fdtd3d_desc md = *desc;
        md.z_lo = want_pairs ? std::max(0, g->lo[r] - 1) : g->lo[r];
        md.z_hi = want_pairs ? std::min(desc->Nz, g->hi[r] + 1) : g->hi[r];
        md.device = dev;
        g->member.push_back(nullptr);
        g->device.push_back(dev);
        if (int rc = fdtd3d_create(&md, &g->member.back())) return rc;
        if (want_pairs && !fdtd3d_pair_capable(g->member.back()))
            return fail(FDTD_ERR_STATE, "group: slab %d cannot take two steps per launch (CPML state or scratch memory beyond 32-bit indexing); set FDTD_PAIR=0", r);
    }
    g->pairs = want_pairs;
    if (g->pairs) {
        for (int r = 0; r < n_slabs; ++r)       // observables are reported for the layers a slab answers for
            if (int rc = fdtd3d_observe_set_range(g->member[r], g->lo[r], g->hi[r])) return rc;
        // what crosses a cut: D, B and the x- and y-face CPML state of both families
        for (int c = 0; c < 3; ++c) g->xchg_ids.push_back(FDTD3D_DX + c);
        for (int c = 0; c < 3; ++c) g->xchg_ids.push_back(FDTD3D_BX + c);
        for (int base_id : {(int)FDTD3D_PHI_E, (int)FDTD3D_PHI_M})
            for (int comp = 0; comp < 3; ++comp) for (int axis = 0; axis < 2; ++axis) for (int side = 0; side < 2; ++side) {
                const int id = base_id + (comp * 3 + axis) * 2 + side;
                if (axis != comp && fdtd3d_array_size(g->member[0], id) > 0) g->xchg_ids.push_back(id);
            }
    }
    // direct copies between the devices of neighbouring slabs (over xGMI); where the driver refuses,
    // hipMemcpyPeerAsync still works, staged by the runtime
    for (int r = 0; r + 1 < n_slabs; ++r) {
        const int a = g->device[r], b = g->device[r + 1];
        if (a == b) continue;
        for (int dir = 0; dir < 2; ++dir) {
            const int from = dir ? b : a, to = dir ? a : b;
            DeviceScope scope(from);
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, from, to) == hipSuccess && can && hipDeviceEnablePeerAccess(to, 0) != hipSuccess)
                (void)hipGetLastError();                 // already enabled by someone else: fine
        }
    }
    // an arena backed through the virtual-memory API is only visible to other devices once they have been granted access
    for (int r = 0; r < n_slabs; ++r)
        for (int nb : {r - 1, r + 1})
            if (nb >= 0 && nb < n_slabs && g->device[nb] != g->device[r])
                if (int rc = fdtd3d_grant_peer_access(g->member[r], g->device[nb])) return rc;
    const size_t R = (size_t)n_slabs;
    // transport of the planes: FDTD_GROUP_TRANSPORT=rccl in the environment, or fdtd3d_group_default_transport(1)
    int transport = g_default_transport;
    if (transport < 0) { const char* e = getenv("FDTD_GROUP_TRANSPORT"); transport = (e && std::string(e) == "rccl") ? 1 : 0; }
    if (transport == 1 && n_slabs > 1) {
        if (int rc = rccl_load()) return rc;
        std::vector<int> devs;
        for (int r = 0; r < n_slabs; ++r) {
            size_t q = 0;
            while (q < devs.size() && devs[q] != g->device[r]) ++q;
            if (q == devs.size()) devs.push_back(g->device[r]);
            g->rank_of.push_back((int)q);
        }
        g->comm.assign(devs.size(), nullptr);
        FDTD_NCCL(g_rccl.CommInitAll(g->comm.data(), (int)devs.size(), devs.data()));
        g->rccl = true;
        g->rx_lo.assign(R, nullptr); g->rx_hi.assign(R, nullptr); g->got_lo.assign(R, nullptr); g->got_hi.assign(R, nullptr);
    }
    g->run.assign(R, nullptr); g->up.assign(R, nullptr); g->down.assign(R, nullptr);
    g->edge_done.assign(R, nullptr); g->top_done.assign(R, nullptr); g->sent_down.assign(R, nullptr); g->sent_up.assign(R, nullptr);
    // (tests only, fdtd3d_group_debug_misown_events) the two arrival events filed under the RECEIVING slab, the
    // mistake the ownership rule exists to catch: the first step of such a group must be refused
    const bool misown = g_misown_events;
    for (size_t r = 0; r < R; ++r) {
        DeviceScope scope(g->device[r]);
        for (auto* v : {&g->run, &g->up, &g->down}) {
            FDTD_HIP(hipStreamCreate(&(*v)[r]));
            g->owner[(void*)(*v)[r]] = (int)r;
        }
        for (auto* v : {&g->edge_done, &g->top_done, &g->sent_down, &g->sent_up}) {
            FDTD_HIP(hipEventCreateWithFlags(&(*v)[r], hipEventDisableTiming));
            int own = (int)r;
            if (misown && v == &g->sent_down && r > 0) own = (int)r - 1;
            if (misown && v == &g->sent_up && r + 1 < R) own = (int)r + 1;
            g->owner[(void*)(*v)[r]] = own;
        }
        if (g->rccl) {
            for (auto* v : {&g->rx_lo, &g->rx_hi}) { FDTD_HIP(hipStreamCreate(&(*v)[r])); g->owner[(void*)(*v)[r]] = (int)r; }
            for (auto* v : {&g->got_lo, &g->got_hi}) { FDTD_HIP(hipEventCreateWithFlags(&(*v)[r], hipEventDisableTiming)); g->owner[(void*)(*v)[r]] = (int)r; }
        }
    }
    return FDTD_OK;
}

template <typename F>
int each_member(fdtd3d_group_handle g, F&& f) {
    if (!g) return fail(FDTD_ERR_ARG, "null group");
    for (size_t r = 0; r < g->member.size(); ++r)
        if (int rc = f(g->member[r])) return rc;
    return FDTD_OK;
}

}  // namespace

void fdtd3d_group::destroy_comms() {
    for (void* c : comm) if (c && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c);
    comm.clear();
}

extern "C" {

void fdtd3d_group_default_transport(int transport) { g_default_transport = (transport == 0 || transport == 1) ? transport : -1; }
int fdtd3d_group_transport(fdtd3d_group_handle g) { return (g && g->rccl) ? 1 : 0; }

int fdtd3d_group_create(const fdtd3d_desc* desc, int n_slabs, const int* devices, fdtd3d_group_handle* out) {
    if (!out) return fail(FDTD_ERR_ARG, "null output handle");
    *out = nullptr;
    if (!desc) return fail(FDTD_ERR_ARG, "null descriptor");
    if (desc->struct_size != (int32_t)sizeof(fdtd3d_desc)) return fail(FDTD_ERR_ARG, "descriptor size %d, library expects %zu", desc->struct_size, sizeof(fdtd3d_desc));
    if (n_slabs < 1 || n_slabs > desc->Nz) return fail(FDTD_ERR_ARG, "cannot split %d layers into %d slabs", desc->Nz, n_slabs);
    if (desc->z_lo != 0 || desc->z_hi != desc->Nz) return fail(FDTD_ERR_ARG, "group: the descriptor must describe the whole grid (z_lo = 0, z_hi = Nz)");
    try {
        auto g = std::make_unique<fdtd3d_group>();
        if (int rc = build_group(g.get(), desc, n_slabs, devices)) return rc;
        *out = g.release();
    } catch (const std::bad_alloc&) {
        return fail(FDTD_ERR_HOST, "out of host memory");
    }
    return FDTD_OK;
}

int fdtd3d_group_destroy(fdtd3d_group_handle g) {
    delete g;
    return FDTD_OK;
}

int fdtd3d_group_slabs(fdtd3d_group_handle g) { return g ? (int)g->member.size() : 0; }
// Test hooks, deliberately absent from the public header: how many event records have passed the ownership check,
// and a switch that makes groups created afterwards file their arrival events under the wrong slab.
long fdtd3d_group_records_checked(fdtd3d_group_handle g) { return g ? g->records_checked : 0; }
void fdtd3d_group_debug_misown_events(int on) { g_misown_events = on != 0; }

fdtd3d_handle fdtd3d_group_member(fdtd3d_group_handle g, int r) {
    return (g && r >= 0 && r < (int)g->member.size()) ? g->member[r] : nullptr;
}

int fdtd3d_group_step(fdtd3d_group_handle g, int n0, int count) {
    if (!g) return fail(FDTD_ERR_ARG, "null group");
    if (g->member.size() == 1) {
        DeviceScope scope(g->device[0]);
        return fdtd3d_step(g->member[0], n0, count, g->run[0]);        // (pairs its steps by itself where that pays)
    }
    if (g->pairs) {
        // a probe is sampled after every single step, and the values after the first step of a two-step launch never
        // reach memory: groups with probes take single steps
        bool probed = false;
        for (fdtd3d_handle m : g->member) {
            int ns = 0, np = 0;
            if (int rc = fdtd3d_probe_read(m, nullptr, 0, &ns, &np)) return rc;
            probed = probed || np > 0;
        }
        for (int n = n0; n < n0 + count;) {
            if (!probed && n + 2 <= n0 + count) { if (int rc = step_pair(g, n)) return rc; n += 2; }
            else { if (int rc = step_single_overlapping(g, n)) return rc; ++n; }
        }
        return FDTD_OK;
    }
    for (int n = n0; n < n0 + count; ++n)
        if (int rc = g->fused ? step_fused(g, n) : step_halves(g, n)) return rc;
    return FDTD_OK;
}
int fdtd3d_group_pairs(fdtd3d_group_handle g) { return (g && g->pairs) ? 1 : 0; }
int fdtd3d_group_bounds(fdtd3d_group_handle g, int r, int* lo, int* hi) {
    if (!g || r < 0 || r >= (int)g->member.size() || !lo || !hi) return fail(FDTD_ERR_ARG, "group_bounds: bad argument");
    *lo = g->lo[r]; *hi = g->hi[r];
    return FDTD_OK;
}

int fdtd3d_group_sync(fdtd3d_group_handle g) {
    if (!g) return fail(FDTD_ERR_ARG, "null group");
    for (size_t r = 0; r < g->member.size(); ++r) {
        DeviceScope scope(g->device[r]);
        FDTD_HIP(hipStreamSynchronize(g->run[r]));
        FDTD_HIP(hipStreamSynchronize(g->up[r]));
        FDTD_HIP(hipStreamSynchronize(g->down[r]));
        if (g->rccl) { FDTD_HIP(hipStreamSynchronize(g->rx_lo[r])); FDTD_HIP(hipStreamSynchronize(g->rx_hi[r])); }
    }
    return FDTD_OK;
}

// Whole-problem calls: every member takes (or reports) its own layers of the global host arrays.
int fdtd3d_group_set_array(fdtd3d_group_handle g, int id, const double* host, size_t n) {
    if (int rc = g ? fdtd3d_group_sync(g) : FDTD_OK) return rc;
    return each_member(g, [&](fdtd3d_handle h) { return fdtd3d_set_array(h, id, host, n); });
}
int fdtd3d_group_get_array(fdtd3d_group_handle g, int id, double* host, size_t n) {
    if (int rc = g ? fdtd3d_group_sync(g) : FDTD_OK) return rc;
    return each_member(g, [&](fdtd3d_handle h) { return fdtd3d_get_array(h, id, host, n); });
}
int fdtd3d_group_fill_materials(fdtd3d_group_handle g, uint64_t seed, double elo, double ehi, double mlo, double mhi) {
    return each_member(g, [&](fdtd3d_handle h) { return fdtd3d_fill_materials(h, seed, elo, ehi, mlo, mhi); });
}
int fdtd3d_group_fill_reference_materials(fdtd3d_group_handle g) {
    return each_member(g, [&](fdtd3d_handle h) { return fdtd3d_fill_reference_materials(h); });
}
int fdtd3d_group_observe_accumulate(fdtd3d_group_handle g, double weight, int reset) {
    if (!g) return fail(FDTD_ERR_ARG, "null group");
    for (size_t r = 0; r < g->member.size(); ++r) {
        if (g->pairs) { DeviceScope scope(g->device[r]); if (int rc = wait_for_cut_planes(g, (int)r)) return rc; }
        if (int rc = fdtd3d_observe_accumulate(g->member[r], weight, reset, g->run[r])) return rc;
    }
    return FDTD_OK;
}
int fdtd3d_group_observe_rms(fdtd3d_group_handle g, double* host, size_t n) {
    return each_member(g, [&](fdtd3d_handle h) { return fdtd3d_observe_rms(h, host, n); });
}
int fdtd3d_group_observe_rms_window(fdtd3d_group_handle g, double* host, size_t n) {
    return each_member(g, [&](fdtd3d_handle h) { return fdtd3d_observe_rms_window(h, host, n); });
}
// Steps with the RMS window of FDTD3D::run (FDTD3D.cpp:1089-1103) advanced after every one of them.  One slab: the
// engine's own fdtd3d_step_observed (two steps per launch where that pays).  Several: on overlapping slabs two steps per
// launch with the window advanced inside (step_pair_observed); else step, accumulate on every slab, and where a step
// completes a window every slab keeps it.
int fdtd3d_group_step_observed(fdtd3d_group_handle g, int n0, int count, double weight, int window) {
    if (!g) return fail(FDTD_ERR_ARG, "null group");
    if (window < 1 || n0 < 0) return fail(FDTD_ERR_ARG, "group_step_observed: bad window or step number");
    if (g->member.size() == 1) {
        DeviceScope scope(g->device[0]);
        return fdtd3d_step_observed(g->member[0], n0, count, weight, window, g->run[0]);
    }
    bool probed = false;
    for (fdtd3d_handle m : g->member) {
        int ns = 0, np = 0;
        if (int rc = fdtd3d_probe_read(m, nullptr, 0, &ns, &np)) return rc;
        probed = probed || np > 0;
    }
    for (int n = n0; n < n0 + count; ++n) {
        if (g->pairs && !probed && n + 2 <= n0 + count) {
            if (int rc = step_pair_observed(g, n, weight, window)) return rc;
            ++n;
            continue;
        }
        if (int rc = g->pairs ? step_single_overlapping(g, n) : (g->fused ? step_fused(g, n) : step_halves(g, n))) return rc;
        for (size_t r = 0; r < g->member.size(); ++r) {
            if (g->pairs) { DeviceScope scope(g->device[r]); if (int rc = wait_for_cut_planes(g, (int)r)) return rc; }      // the top layer's cells look at the plane above
            if (int rc = fdtd3d_observe_accumulate(g->member[r], weight, n % window == 0, g->run[r])) return rc;
            if ((n + 1) % window == 0) if (int rc = fdtd3d_observe_keep_window(g->member[r])) return rc;
        }
    }
    return FDTD_OK;
}
int fdtd3d_group_observe_frame(fdtd3d_group_handle g, double* ex_center, double* e_mag, size_t n) {
    if (!g) return fail(FDTD_ERR_ARG, "null group");
    for (size_t r = 0; r < g->member.size(); ++r) {
        if (g->pairs) { DeviceScope scope(g->device[r]); if (int rc = wait_for_cut_planes(g, (int)r)) return rc; }
        if (int rc = fdtd3d_observe_frame(g->member[r], ex_center, e_mag, n, g->run[r])) return rc;
    }
    return FDTD_OK;
}

}  // extern "C"
