// comm.hip — the one exchange of the frame-sharded paths in their process-per-GPU form (batched rfft + peak,
// STFT and the fused generator shard by frames with NO data-path collective): the gather of the 8-byte per-frame peak
// records to one rank, natively over RCCL (grouped ncclSend / ncclRecv on one communicator per process).  The reference is
// a single process (main.py:448); this is the north-star's multi-GPU form.  Round 2 gathered through torch.distributed —
// the one place PyTorch sat on a product path; torch is now needed by nobody but the bench harness' barrier.
//
// Rendezvous: rank 0 takes a unique id from RCCL (sa_comm_unique_id, 128 bytes) and hands it to the other ranks by
// whatever channel the launcher has (fft_amd/dist.py: a file); every rank then calls sa_comm_create on its own device.
#include <string.h>

#include "rccl_api.h"
#include "runtime.h"

struct sa_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, dev = 0;
};

namespace sa {
#define SA_NCCL(call)                                                                                     \
    do {                                                                                                  \
        ncclResult_t n_ = (call);                                                                         \
        if (n_ != ncclSuccess) {                                                                          \
            set_error("%s failed: %s (%s:%d)", #call, rccl_api()->GetErrorString(n_), __FILE__, __LINE__);  \
            return SA_EHIP;                                                                               \
        }                                                                                                 \
    } while (0)
static int need_rccl() {
    if (rccl_api()) return SA_OK;
    const char* why = dlerror();
    set_error("librccl.so could not be loaded (%s)", why ? why : "missing symbols");
    return SA_EUNSUP;
}
}  // namespace sa

extern "C" {

int sa_comm_unique_id(void* id) {
    using namespace sa;
    SA_ENTER();
    static_assert(sizeof(ncclUniqueId) == SA_COMM_ID_BYTES, "SA_COMM_ID_BYTES follows ncclUniqueId");
    if (!id) { set_error("sa_comm_unique_id: id is NULL"); return SA_EINVAL; }
    SA_TRY(need_rccl());
    ncclUniqueId u;
    SA_NCCL(rccl_api()->GetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return SA_OK;
}

int sa_comm_create(sa_comm** comm, const void* id, int32_t rank, int32_t world) {
    using namespace sa;
    SA_ENTER();
    if (!comm || !id || world < 1 || rank < 0 || rank >= world) { set_error("sa_comm_create: bad arguments (rank %d of %d)", rank, world); return SA_EINVAL; }
    *comm = nullptr;
    SA_TRY(need_rccl());
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    sa_comm* c = new sa_comm;
    c->rank = rank; c->world = world;
    (void)hipGetDevice(&c->dev);
    const ncclResult_t r = rccl_api()->CommInitRank(&c->comm, world, u, rank);
    if (r != ncclSuccess) {
        set_error("ncclCommInitRank(rank %d of %d, device %d): %s", rank, world, c->dev, rccl_api()->GetErrorString(r));
        delete c;
        return SA_EHIP;
    }
    *comm = c;
    return SA_OK;
}

int sa_comm_destroy(sa_comm* c) {
    using namespace sa;
    SA_ENTER();
    if (!c) return SA_OK;
    if (c->comm && rccl_api()) (void)rccl_api()->CommDestroy(c->comm);
    delete c;
    return SA_OK;
}

int sa_comm_gather_peaks_dev(sa_comm* c, const int32_t* bins, const float* mags, int64_t count, const int64_t* counts, int32_t root,
                             int32_t* all_bins, float* all_mags, void* stream) {
    using namespace sa;
    SA_ENTER();
    if (!c || !counts || root < 0 || root >= c->world || count < 0) { set_error("sa_comm_gather_peaks_dev: bad arguments"); return SA_EINVAL; }
    if (counts[c->rank] != count) { set_error("sa_comm_gather_peaks_dev: counts[%d] = %lld but count = %lld", c->rank, (long long)counts[c->rank], (long long)count); return SA_EINVAL; }
    if (count > 0 && (!bins || !mags)) { set_error("sa_comm_gather_peaks_dev: NULL records"); return SA_EINVAL; }
    int64_t total = 0;
    for (int r = 0; r < c->world; ++r) {
        if (counts[r] < 0) { set_error("sa_comm_gather_peaks_dev: counts[%d] < 0", r); return SA_EINVAL; }
        total += counts[r];
    }
    if (c->rank == root && total > 0 && (!all_bins || !all_mags)) { set_error("sa_comm_gather_peaks_dev: NULL output on the root"); return SA_EINVAL; }
    hipStream_t st = stream ? (hipStream_t)stream : thread_stream();
    const RcclApi* nc = rccl_api();
    // one group: every rank sends its two arrays to the root, the root receives every rank's (its own included: a
    // send to self matched by a receive from self inside one group is a local copy in RCCL)
    SA_NCCL(nc->GroupStart());
    ncclResult_t r_ = ncclSuccess;
    if (count > 0) {
        r_ = nc->Send(bins, (size_t)count, ncclInt32, root, c->comm, st);
        if (r_ == ncclSuccess) r_ = nc->Send(mags, (size_t)count, ncclFloat32, root, c->comm, st);
    }
    if (c->rank == root) {
        int64_t off = 0;
        for (int r = 0; r < c->world && r_ == ncclSuccess; ++r) {
            if (counts[r] > 0) {
                r_ = nc->Recv(all_bins + off, (size_t)counts[r], ncclInt32, r, c->comm, st);
                if (r_ == ncclSuccess) r_ = nc->Recv(all_mags + off, (size_t)counts[r], ncclFloat32, r, c->comm, st);
            }
            off += counts[r];
        }
    }
    const ncclResult_t e_ = nc->GroupEnd();
    SA_NCCL(r_);
    SA_NCCL(e_);
    return SA_OK;
}

}  // extern "C"
