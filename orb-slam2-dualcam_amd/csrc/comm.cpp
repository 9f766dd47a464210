// comm.cpp -- the one exchange step of the path behind the C ABI: all-gather of the newest dual frame's feature slots
// over RCCL / xGMI, so that every GPU can match its cameras against features extracted on the other GPUs (SURVEY.md 8(e);
// the reference's analogue is SearchByBoWCrossCam(curFrame, camS, KF, CAP), src/Tracking.cc:822, with the key frame held by
// another process). One communicator per process (one process per GPU).
//
// RCCL is bound at run time (dlopen "librccl.so.1"): a host that already carries an RCCL (PyTorch does) shares that instance,
// and a single-GPU host never loads it. Slots are fixed-capacity and contiguous per rank, so the keypoint, descriptor and
// count arrays are gathered in place by three grouped ncclAllGather calls -- no pack / unpack pass.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <string>

#include "common.h"

namespace {

struct Rccl {
    void* so = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string why;                                 // why the library is unusable (written once, inside the call_once)
};

Rccl& rccl_state() { static Rccl r; return r; }
std::string rccl_why() { const Rccl& r = rccl_state(); return r.why.empty() ? std::string("unknown reason") : r.why; }

Rccl* rccl()
{
    Rccl& r = rccl_state();
    static std::once_flag once;
    std::call_once(once, [&r] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.so) break;
            if (const char* e = dlerror()) r.why = e;           // dlerror() clears itself: read ONCE, here, on the loading thread
        }
        if (!r.so) return;
        r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.so, "ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.so, "ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.so, "ncclCommDestroy");
        r.AllGather = (decltype(r.AllGather))dlsym(r.so, "ncclAllGather");
        r.GroupStart = (decltype(r.GroupStart))dlsym(r.so, "ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))dlsym(r.so, "ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.so, "ncclGetErrorString");
        if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GroupStart || !r.GroupEnd) {
            r.why = "a required nccl* symbol is missing";
            dlclose(r.so);
            r.so = nullptr;
        }
    });
    return r.so ? &r : nullptr;
}

int need_rccl(Rccl*& r)
{
    r = rccl();
    if (!r) { dcs::set_error("librccl.so.1 could not be loaded: %s", rccl_why().c_str()); return DCS_ERR_HIP; }
    return DCS_OK;
}

#define DCS_NCCL(r, expr)                                                                                            \
    do {                                                                                                             \
        ncclResult_t _e = (expr);                                                                                    \
        if (_e != ncclSuccess) {                                                                                     \
            dcs::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, (r)->GetErrorString ? (r)->GetErrorString(_e) : "rccl error"); \
            return DCS_ERR_HIP;                                                                                      \
        }                                                                                                            \
    } while (0)

}  // namespace

struct dcs_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

extern "C" {

int dcs_comm_unique_id(uint8_t id[DCS_COMM_ID_BYTES])
{
    if (!id) { dcs::set_error("null id"); return DCS_ERR_INVALID; }
    static_assert(DCS_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    Rccl* r;
    int rc = need_rccl(r);
    if (rc) return rc;
    ncclUniqueId u;
    DCS_NCCL(r, r->GetUniqueId(&u));
    memcpy(id, u.internal, DCS_COMM_ID_BYTES);
    return DCS_OK;
}

int dcs_comm_create(const uint8_t id[DCS_COMM_ID_BYTES], int rank, int world, dcs_comm** out)
{
    if (!id || !out || world < 1 || rank < 0 || rank >= world) { dcs::set_error("bad communicator arguments (rank %d of %d)", rank, world); return DCS_ERR_INVALID; }
    int rc = dcs::ensure_device();
    if (rc) return rc;
    Rccl* r;
    if ((rc = need_rccl(r))) return rc;
    ncclUniqueId u;
    memcpy(u.internal, id, DCS_COMM_ID_BYTES);
    dcs_comm* c = new dcs_comm;
    c->rank = rank; c->world = world;
    ncclResult_t e = r->CommInitRank(&c->comm, world, u, rank);
    if (e != ncclSuccess) {
        dcs::set_error("ncclCommInitRank(rank %d of %d) -> %s", rank, world, r->GetErrorString ? r->GetErrorString(e) : "rccl error");
        delete c;
        return DCS_ERR_HIP;
    }
    *out = c;
    return DCS_OK;
}

void dcs_comm_destroy(dcs_comm* c)
{
    if (!c) return;
    Rccl* r = rccl();
    if (r && c->comm) (void)r->CommDestroy(c->comm);
    delete c;
}

int dcs_comm_info(const dcs_comm* c, int* rank, int* world)
{
    if (!c) { dcs::set_error("null communicator"); return DCS_ERR_INVALID; }
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return DCS_OK;
}

int dcs_features_allgather(dcs_comm* c, const dcs_keypoint* d_kp, const uint8_t* d_desc, const int32_t* d_n, int n_slots, int cap,
                           dcs_keypoint* d_kp_all, uint8_t* d_desc_all, int32_t* d_n_all, void* stream)
{
    if (!c || !d_kp || !d_desc || !d_n || !d_kp_all || !d_desc_all || !d_n_all || n_slots < 1 || cap < 1) {
        dcs::set_error("bad all-gather arguments"); return DCS_ERR_INVALID;
    }
    Rccl* r;
    int rc = need_rccl(r);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const size_t nk = (size_t)n_slots * cap * sizeof(dcs_keypoint), nd = (size_t)n_slots * cap * 32;
    DCS_NCCL(r, r->GroupStart());
    ncclResult_t e1 = r->AllGather(d_kp, d_kp_all, nk, ncclUint8, c->comm, st);
    ncclResult_t e2 = r->AllGather(d_desc, d_desc_all, nd, ncclUint8, c->comm, st);
    ncclResult_t e3 = r->AllGather(d_n, d_n_all, (size_t)n_slots, ncclInt32, c->comm, st);
    ncclResult_t e4 = r->GroupEnd();
    for (ncclResult_t e : {e1, e2, e3, e4})
        if (e != ncclSuccess) { dcs::set_error("ncclAllGather -> %s", r->GetErrorString ? r->GetErrorString(e) : "rccl error"); return DCS_ERR_HIP; }
    return DCS_OK;
}

}  // extern "C"
