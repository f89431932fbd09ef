// dist_fft.hip — BASELINE config 5: ONE fp64 complex FFT of n = 2^L points spread over P ranks, inside the library.
//
// The reference has nothing here (single process, CPU) — this is the north-star's distributed four-step.
// One host process drives every rank: a rank is a (device, streams, buffers) triple, several ranks may share a device
// ("virtual ranks": the same code above the exchange runs on a one-GPU box), ranks on different devices
// exchange over peer-mapped pointers (xGMI).
//
//   n = R1 * S,  x viewed as [R1][S] row-major (index i S + s);  P ranks,  Sp = S / P,  K1L = R1 / P
//   input  (default)  rank r holds columns s in [r Sp, (r+1) Sp):  in[i * Sp + sl]            "column blocks"
//   pass 1            R1-point transforms down the owned columns, times w_n^{s k1}            (fft_c2c.hip)
//   exchange          row k1 goes to rank k1 / K1L — the ONE all-to-all: n/P^2 points per ordered pair
//   phase 2           every owner finishes its K1L rows as S-point transforms                 (fft_c2c.hip)
//   output (default)  rank q holds rows k1 in [q K1L, (q+1) K1L):  out[k1l * S + k'] = X[k1 + R1 k']   "row blocks"
// Natural-order slabs (x[r n/P ...], X[q n/P ...]) cost one more exchange on that side (flags).
//
// Exchange modes (measured against each other by tools/bench_dist_c5.py):
//   DMA     pass 1 writes a send buffer; one hipMemcpyAsync per (source, owner, chunk) on a stream per pair
//   KERNEL  same, copies done by a copy kernel on the source device storing through the peer mapping
//   FUSED   pass 1's store epilogue writes straight into the owners' receive buffers: no send buffer, no copy pass
//   RCCL    pass 1 writes a send buffer; per chunk one RCCL group over all ranks' communicators: ncclAllToAll when the
//           exchange is a single chunk, grouped ncclSend / ncclRecv per (source, owner) otherwise.  One
//           communicator per rank from ncclCommInitAll, so every rank needs a device of its own (RCCL refuses two ranks on
//           one device).  librccl is loaded with dlopen when this mode is first asked for: no link-time dependency.
// DMA / KERNEL deliver the rows in chunks and an owner starts the transforms of a chunk as soon as all P parts of it
// have arrived, so the rest of the exchange overlaps them.  The receive layout is always "one block per
// source": recv[p][k1l][sl]; phase 2's first pass reads it in place (blocked loads), nothing is unpacked.
#include <chrono>
#include <dlfcn.h>
#include <string.h>

#include "rccl_api.h"

#include <vector>

#include "runtime.h"

namespace sa {

int dist_pass1(const double* in, double* out, double* const* push_tab, int push_shift, int64_t n, int l1, int64_t Sp,
               int64_t col0, int nt, hipStream_t st);
int dist_phase2(const double* in_blocked, double* work, double* out, int64_t S, int64_t rows, int64_t Sp, int64_t blk_pitch,
                int nt, hipStream_t st);

using Cd = double2;

// dst[r * dpitch + c] = src[r * spitch + c]  (16-byte elements; rows are long, so a flat grid-stride loop coalesces)
__global__ void __launch_bounds__(256) dist_copy2d_kernel(Cd* __restrict__ dst, int64_t dpitch, const Cd* __restrict__ src,
                                                          int64_t spitch, int64_t width, int64_t height) {
    const int64_t total = width * height;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / width, c = i - r * width;
        dst[r * dpitch + c] = src[r * spitch + c];
    }
}

// dst[c * dpitch + r] = src[r * spitch + c], 32 x 32 tiles through LDS (natural-order output slabs)
__global__ void __launch_bounds__(256) dist_transpose_kernel(Cd* __restrict__ dst, int64_t dpitch, const Cd* __restrict__ src,
                                                             int64_t spitch, int64_t rows, int64_t cols) {
    __shared__ Cd tile[32][33];
    const int64_t tiles_c = (cols + 31) / 32, tiles_r = (rows + 31) / 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 32 x 8
    for (int64_t t = blockIdx.x; t < tiles_c * tiles_r; t += gridDim.x) {
        const int64_t r0 = (t / tiles_c) * 32, c0 = (t % tiles_c) * 32;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t r = r0 + ty + 8 * j, c = c0 + tx;
            if (r < rows && c < cols) tile[ty + 8 * j][tx] = src[r * spitch + c];
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t c = c0 + ty + 8 * j, r = r0 + tx;
            if (r < rows && c < cols) dst[c * dpitch + r] = tile[tx][ty + 8 * j];
        }
        __syncthreads();
    }
}

// ---- analytic inputs / checks for sizes no host can hold twice (SURVEY §8(d) C5 KATs) ---------------------------------
// Global index of local element o of a rank's slab.
struct DistLayout {
    int64_t n, S, Sp, R1, K1L;
    int natural;
    int rank;
};
__device__ __forceinline__ int64_t in_index(const DistLayout& l, int64_t o) {
    if (l.natural) return (int64_t)l.rank * (l.n / (l.R1 / l.K1L)) + o;
    const int64_t i = o / l.Sp, sl = o - i * l.Sp;
    return i * l.S + (int64_t)l.rank * l.Sp + sl;
}
__device__ __forceinline__ int64_t out_index(const DistLayout& l, int64_t o) {
    if (l.natural) return (int64_t)l.rank * (l.n / (l.R1 / l.K1L)) + o;
    const int64_t k1l = o / l.S, kp = o - k1l * l.S;
    return ((int64_t)l.rank * l.K1L + k1l) + l.R1 * kp;
}
__device__ __forceinline__ Cd unit_phase(int64_t num, int64_t n, double sign) {   // exp(sign 2 pi i num / n), num < n = 2^L: exact fraction
    double s, c;
    sincospi(2.0 * (double)num / (double)n, &s, &c);
    return Cd{c, sign * s};
}
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ void __launch_bounds__(256) dist_fill_kernel(Cd* x, int64_t count, DistLayout l, int kind, int64_t param, uint64_t seed) {
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < count; o += (int64_t)gridDim.x * 256) {
        const int64_t j = in_index(l, o);
        Cd v{0.0, 0.0};
        if (kind == SA_DIST_KAT_IMPULSE) v.x = j == param ? 1.0 : 0.0;
        else if (kind == SA_DIST_KAT_TONE) v = unit_phase((int64_t)(((unsigned __int128)j * (uint64_t)param) % (uint64_t)l.n), l.n, 1.0);
        else {                                           // counter-based N(0,1) pair per point (Box-Muller), keyed by (seed, j)
            const uint64_t a = mix64(seed ^ (uint64_t)j * 2), b = mix64(seed ^ ((uint64_t)j * 2 + 1));
            const double u1 = ((double)(a >> 11) + 1.0) * (1.0 / 9007199254740992.0), u2 = (double)(b >> 11) * (1.0 / 9007199254740992.0);
            const double r = sqrt(-2.0 * log(u1));
            double s, c;
            sincospi(2.0 * u2, &s, &c);
            v = Cd{r * c, r * s};
        }
        x[o] = v;
    }
}
// max |X[k] - expected| over the slab -> *err_bits (non-negative doubles order like their bit patterns)
__global__ void __launch_bounds__(256) dist_check_kernel(const Cd* X, int64_t count, DistLayout l, int kind, int64_t param,
                                                         unsigned long long* err_bits) {
    double worst = 0.0;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < count; o += (int64_t)gridDim.x * 256) {
        const int64_t k = out_index(l, o);
        Cd e{0.0, 0.0};
        if (kind == SA_DIST_KAT_IMPULSE) e = unit_phase((int64_t)(((unsigned __int128)k * (uint64_t)param) % (uint64_t)l.n), l.n, -1.0);
        else if (k == param) e.x = (double)l.n;
        const double dx = X[o].x - e.x, dy = X[o].y - e.y;
        double d = sqrt(dx * dx + dy * dy);
        if (!(d == d)) d = INFINITY;                     // NaN counts as the worst error
        worst = fmax(worst, d);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) worst = fmax(worst, __shfl_xor(worst, off));
    if ((threadIdx.x & 63) == 0) atomicMax(err_bits, (unsigned long long)__double_as_longlong(worst));
}

}  // namespace sa

// ---- the plan -----------------------------------------------------------------------------------------------------------
struct sa_dist_rank {
    int dev = 0;
    hipStream_t compute = nullptr;
    std::vector<hipStream_t> copy;                       // [P]: stream towards owner q (on this rank's device)
    double *in = nullptr, *out = nullptr, *send = nullptr, *recv = nullptr, *work = nullptr;
    double** push_tab = nullptr;                         // device array [P]: recv_q + this rank's block offset
    unsigned long long* err = nullptr;                   // device scalar for sa_dist_check
    hipEvent_t e_start = nullptr, e_pre = nullptr, e_p1s = nullptr, e_p1e = nullptr, e_p2s = nullptr, e_p2e = nullptr, e_end = nullptr;
    std::vector<hipEvent_t> e_cs, e_ce;                  // [P] first copy start / last copy end towards q
    std::vector<std::vector<hipEvent_t>> e_chunk;        // [P][J] chunk j has left for q
    std::vector<hipEvent_t> e_prex, e_post;              // [P] natural-order exchanges towards q
};
struct sa_dist_plan {
    int64_t n = 0, R1 = 0, S = 0, Sp = 0, K1L = 0, CH = 0;
    int P = 0, L = 0, l1 = 0, J = 1, mode = 0, flags = 0, nt = 0, n_devices = 0;
    bool nat_in = false, nat_out = false;
    std::vector<sa_dist_rank> r;
    std::vector<ncclComm_t> comms;                       // RCCL mode: one communicator per rank (ncclCommInitAll)
};

namespace sa {

#define SA_HIPD(call)                                                                                             \
    do {                                                                                                          \
        hipError_t e_ = (call);                                                                                   \
        if (e_ != hipSuccess) {                                                                                   \
            set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__);                 \
            rc = e_ == hipErrorOutOfMemory ? SA_ENOMEM : SA_EHIP;                                                 \
            goto done;                                                                                            \
        }                                                                                                         \
    } while (0)
#define SA_TRYD(expr)                     \
    do {                                  \
        rc = (expr);                      \
        if (rc != SA_OK) goto done;       \
    } while (0)
#define SA_NCCLD(call)                                                                                            \
    do {                                                                                                          \
        ncclResult_t n_ = (call);                                                                                 \
        if (n_ != ncclSuccess) {                                                                                  \
            set_error("%s failed: %s (%s:%d)", #call, rccl_api()->GetErrorString(n_), __FILE__, __LINE__);        \
            rc = SA_EHIP;                                                                                         \
            goto done;                                                                                            \
        }                                                                                                         \
    } while (0)

struct DeviceRestore {
    int dev = 0;
    DeviceRestore() { (void)hipGetDevice(&dev); }
    ~DeviceRestore() { (void)hipSetDevice(dev); }
};

static void dist_free(sa_dist_plan* p) {
    if (!p) return;
    DeviceRestore keep;
    for (size_t i = 0; i < p->comms.size(); ++i)
        if (p->comms[i]) {
            (void)hipSetDevice(p->r[i].dev);
            if (p->r[i].compute) (void)hipStreamSynchronize(p->r[i].compute);
            for (hipStream_t s : p->r[i].copy) if (s) (void)hipStreamSynchronize(s);
            (void)rccl_api()->CommDestroy(p->comms[i]);
        }
    for (sa_dist_rank& k : p->r) {
        (void)hipSetDevice(k.dev);
        if (k.compute) (void)hipStreamSynchronize(k.compute);
        for (hipStream_t s : k.copy) if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
        if (k.compute) (void)hipStreamDestroy(k.compute);
        for (double* b : {k.in, k.out, k.send, k.recv, k.work}) if (b) (void)hipFree(b);
        if (k.push_tab) (void)hipFree(k.push_tab);
        if (k.err) (void)hipFree(k.err);
        for (hipEvent_t e : {k.e_start, k.e_pre, k.e_p1s, k.e_p1e, k.e_p2s, k.e_p2e, k.e_end}) if (e) (void)hipEventDestroy(e);
        for (auto* v : {&k.e_cs, &k.e_ce, &k.e_prex, &k.e_post}) for (hipEvent_t e : *v) if (e) (void)hipEventDestroy(e);
        for (auto& v : k.e_chunk) for (hipEvent_t e : v) if (e) (void)hipEventDestroy(e);
    }
    delete p;
}

static DistLayout layout_of(const sa_dist_plan* p, int rank, bool natural) {
    DistLayout l;
    l.n = p->n; l.S = p->S; l.Sp = p->Sp; l.R1 = p->R1; l.K1L = p->K1L; l.natural = natural; l.rank = rank;
    return l;
}

static unsigned copy_grid(int64_t elems) {
    const int64_t nb = (elems + 255) / 256, cap = (int64_t)num_cus() * 8;
    return (unsigned)(nb < cap ? (nb < 1 ? 1 : nb) : cap);
}

}  // namespace sa

extern "C" {

int sa_dist_plan_create(sa_dist_plan** plan_out, int64_t n, int32_t n_ranks, const int32_t* devices, int32_t flags) {
    using namespace sa;
    SA_ENTER();
    if (!plan_out) { set_error("sa_dist_plan_create: plan is NULL"); return SA_EINVAL; }
    *plan_out = nullptr;
    int L = 0;
    while ((1LL << L) < n) ++L;
    int lp = 0;
    while ((1 << lp) < n_ranks) ++lp;
    if (n_ranks < 1 || n_ranks > SA_DIST_MAX_RANKS || (1 << lp) != n_ranks) {
        set_error("dist c2c: n_ranks=%d must be a power of two in [1, %d]", n_ranks, SA_DIST_MAX_RANKS);
        return SA_EINVAL;
    }
    if (n < 2 || (1LL << L) != n || L > 30) { set_error("dist c2c: n=%lld must be a power of two <= 2^30", (long long)n); return SA_EUNSUP; }
    const int mode_req = (flags >> 4) & 15, chunks_req = (flags >> 8) & 255;
    if ((flags & ~0xFFF3) || mode_req > SA_DIST_XCHG_RCCL) { set_error("dist c2c: unknown flag bits 0x%x", flags); return SA_EINVAL; }
    // first factor: 2^6 .. 2^10, at least one row per rank, leaving rows of >= 2^13 points where n allows (so the owners'
    // first pass is the blocked strided pass at every size worth measuring) and columns per rank in groups of 8
    int l1 = L - 14;
    if (l1 < 6) l1 = 6;
    if (l1 > 10) l1 = 10;
    if (l1 < lp) l1 = lp;
    if (L - l1 - lp < 3 || L - l1 > 20 || l1 > 10) {
        set_error("dist c2c: n=2^%d is too short for %d ranks (needs n >= %lld)", L, n_ranks, (long long)(512LL * n_ranks));
        return SA_EUNSUP;
    }
    int ndev = 0;
    SA_HIP(hipGetDeviceCount(&ndev));
    sa_dist_plan* p = new sa_dist_plan;
    p->n = n; p->L = L; p->l1 = l1; p->P = n_ranks; p->flags = flags;
    p->R1 = 1LL << l1; p->S = n >> l1; p->Sp = p->S / n_ranks; p->K1L = p->R1 / n_ranks;
    p->nat_in = flags & SA_DIST_NATURAL_IN; p->nat_out = flags & SA_DIST_NATURAL_OUT;
    p->r.resize(n_ranks);
    bool one_device = true;
    std::vector<int> devs;
    for (int i = 0; i < n_ranks; ++i) {
        const int d = devices ? devices[i] : i % ndev;
        if (d < 0 || d >= ndev) { set_error("dist c2c: rank %d on device %d, but %d device(s) are visible", i, d, ndev); delete p; return SA_EINVAL; }
        p->r[i].dev = d;
        one_device = one_device && d == p->r[0].dev;
        bool seen = false;
        for (int x : devs) seen = seen || x == d;
        if (!seen) devs.push_back(d);
    }
    p->n_devices = (int)devs.size();
    p->mode = mode_req ? mode_req : (one_device ? SA_DIST_XCHG_FUSED : SA_DIST_XCHG_DMA);
    if (p->mode == SA_DIST_XCHG_RCCL) {
        if (p->n_devices != n_ranks) {
            set_error("dist c2c: the RCCL exchange needs one device per rank (%d ranks on %d device(s)); RCCL refuses two ranks "
                      "on one device - use DMA, KERNEL or FUSED for virtual ranks", n_ranks, p->n_devices);
            delete p;
            return SA_EUNSUP;
        }
        if (!rccl_api()) {
            const char* why = dlerror();
            set_error("dist c2c: librccl.so could not be loaded (%s)", why ? why : "missing symbols");
            delete p;
            return SA_EUNSUP;
        }
    }
    int J = chunks_req ? chunks_req : 4;
    if (p->mode == SA_DIST_XCHG_FUSED) J = 1;
    while (J > 1 && (p->K1L % J || p->K1L / J < 1)) --J;
    if (J > p->K1L) J = (int)p->K1L;
    p->J = J; p->CH = p->K1L / J;
    const size_t slab = (size_t)(n / n_ranks) * 16;
    p->nt = slab >= ((size_t)1 << 30);                       // like the single-GPU passes: stream once far beyond the caches
    int rc = SA_OK;
    DeviceRestore keep;
    // peer mappings between every pair of distinct devices
    for (int a : devs)
        for (int b : devs) {
            if (a == b) continue;
            int can = 0;
            SA_HIPD(hipDeviceCanAccessPeer(&can, a, b));
            if (!can) { set_error("dist c2c: device %d cannot map device %d's memory (no P2P)", a, b); rc = SA_EUNSUP; goto done; }
            SA_HIPD(hipSetDevice(a));
            hipError_t e = hipDeviceEnablePeerAccess(b, 0);
            if (e == hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            else if (e != hipSuccess) { set_error("hipDeviceEnablePeerAccess(%d -> %d): %s", a, b, hipGetErrorString(e)); rc = SA_EHIP; goto done; }
        }
    for (int i = 0; i < n_ranks; ++i) {
        sa_dist_rank& k = p->r[i];
        SA_HIPD(hipSetDevice(k.dev));
        SA_HIPD(hipStreamCreateWithFlags(&k.compute, hipStreamNonBlocking));
        k.copy.assign(n_ranks, nullptr);
        for (int q = 0; q < n_ranks; ++q) SA_HIPD(hipStreamCreateWithFlags(&k.copy[q], hipStreamNonBlocking));
        SA_HIPD(hipMalloc(reinterpret_cast<void**>(&k.in), slab));
        SA_HIPD(hipMalloc(reinterpret_cast<void**>(&k.out), slab));
        SA_HIPD(hipMalloc(reinterpret_cast<void**>(&k.recv), slab));
        SA_HIPD(hipMalloc(reinterpret_cast<void**>(&k.work), slab));
        if (p->mode != SA_DIST_XCHG_FUSED) SA_HIPD(hipMalloc(reinterpret_cast<void**>(&k.send), slab));
        SA_HIPD(hipMalloc(reinterpret_cast<void**>(&k.push_tab), sizeof(double*) * n_ranks));
        SA_HIPD(hipMalloc(reinterpret_cast<void**>(&k.err), 8));
        for (hipEvent_t* e : {&k.e_start, &k.e_pre, &k.e_p1s, &k.e_p1e, &k.e_p2s, &k.e_p2e, &k.e_end}) SA_HIPD(hipEventCreate(e));
        for (auto* v : {&k.e_cs, &k.e_ce, &k.e_prex, &k.e_post}) {
            v->assign(n_ranks, nullptr);
            for (int q = 0; q < n_ranks; ++q) SA_HIPD(hipEventCreate(&(*v)[q]));
        }
        k.e_chunk.assign(n_ranks, std::vector<hipEvent_t>(J, nullptr));
        for (int q = 0; q < n_ranks; ++q)
            for (int j = 0; j < J; ++j) SA_HIPD(hipEventCreate(&k.e_chunk[q][j]));
    }
    if (p->mode == SA_DIST_XCHG_RCCL) {
        std::vector<int> devlist(n_ranks);
        for (int i = 0; i < n_ranks; ++i) devlist[i] = p->r[i].dev;
        p->comms.assign(n_ranks, nullptr);
        SA_NCCLD(rccl_api()->CommInitAll(p->comms.data(), n_ranks, devlist.data()));
    }
    for (int i = 0; i < n_ranks; ++i) {                      // push table of source i: owner q's block for source i
        std::vector<double*> tab(n_ranks);
        for (int q = 0; q < n_ranks; ++q) tab[q] = p->r[q].recv + 2 * ((int64_t)i * p->K1L * p->Sp);
        SA_HIPD(hipSetDevice(p->r[i].dev));
        SA_HIPD(hipMemcpy(p->r[i].push_tab, tab.data(), sizeof(double*) * n_ranks, hipMemcpyHostToDevice));
    }
done:
    if (rc != SA_OK) { dist_free(p); return rc; }
    *plan_out = p;
    return SA_OK;
}

int sa_dist_plan_destroy(sa_dist_plan* plan) {
    using namespace sa;
    SA_ENTER();                                              // a concurrent sa_shutdown waits for us
    dist_free(plan);
    return SA_OK;
}

int sa_dist_plan_info(const sa_dist_plan* p, int32_t rank, sa_dist_info* info) {
    using namespace sa;
    SA_ENTER();
    if (!p || !info || rank < 0 || rank >= p->P) { set_error("sa_dist_plan_info: bad arguments"); return SA_EINVAL; }
    const sa_dist_rank& k = p->r[rank];
    memset(info, 0, sizeof(*info));
    info->n = p->n; info->R1 = p->R1; info->S = p->S; info->cols_per_rank = p->Sp; info->rows_per_rank = p->K1L;
    info->n_ranks = p->P; info->n_devices = p->n_devices; info->device = k.dev; info->exchange_mode = p->mode; info->n_chunks = p->J;
    info->slab_elems = p->n / p->P;
    info->in = k.in;
    info->out = p->nat_out ? k.recv : k.out;                 // the natural-order exchange lands in the receive buffer
    return SA_OK;
}

int sa_dist_fill(sa_dist_plan* p, int32_t kind, int64_t param, uint64_t seed) {
    using namespace sa;
    SA_ENTER();
    if (!p) { set_error("sa_dist_fill: plan is NULL"); return SA_EINVAL; }
    if (kind < SA_DIST_KAT_IMPULSE || kind > SA_DIST_KAT_NOISE || ((kind != SA_DIST_KAT_NOISE) && (param < 0 || param >= p->n))) {
        set_error("sa_dist_fill: kind %d / index %lld out of range", kind, (long long)param);
        return SA_EINVAL;
    }
    DeviceRestore keep;
    const int64_t cnt = p->n / p->P;
    for (int i = 0; i < p->P; ++i) {
        SA_HIP(hipSetDevice(p->r[i].dev));
        hipLaunchKernelGGL(dist_fill_kernel, dim3(copy_grid(cnt)), dim3(256), 0, p->r[i].compute, reinterpret_cast<Cd*>(p->r[i].in), cnt,
                           layout_of(p, i, p->nat_in), kind, param, seed);
        SA_HIP(hipGetLastError());
    }
    for (int i = 0; i < p->P; ++i) { SA_HIP(hipSetDevice(p->r[i].dev)); SA_HIP(hipStreamSynchronize(p->r[i].compute)); }
    return SA_OK;
}

int sa_dist_check(sa_dist_plan* p, int32_t kind, int64_t param, double* max_abs_err) {
    using namespace sa;
    SA_ENTER();
    if (!p || !max_abs_err) { set_error("sa_dist_check: NULL argument"); return SA_EINVAL; }
    if ((kind != SA_DIST_KAT_IMPULSE && kind != SA_DIST_KAT_TONE) || param < 0 || param >= p->n) {
        set_error("sa_dist_check: only the impulse and tone inputs have closed-form spectra");
        return SA_EINVAL;
    }
    DeviceRestore keep;
    const int64_t cnt = p->n / p->P;
    double worst = 0;
    for (int i = 0; i < p->P; ++i) {
        sa_dist_rank& k = p->r[i];
        SA_HIP(hipSetDevice(k.dev));
        SA_HIP(hipMemsetAsync(k.err, 0, 8, k.compute));
        hipLaunchKernelGGL(dist_check_kernel, dim3(copy_grid(cnt)), dim3(256), 0, k.compute,
                           reinterpret_cast<const Cd*>(p->nat_out ? k.recv : k.out), cnt, layout_of(p, i, p->nat_out), kind, param, k.err);
        SA_HIP(hipGetLastError());
    }
    for (int i = 0; i < p->P; ++i) {
        sa_dist_rank& k = p->r[i];
        SA_HIP(hipSetDevice(k.dev));
        unsigned long long bits = 0;
        SA_HIP(hipMemcpyAsync(&bits, k.err, 8, hipMemcpyDeviceToHost, k.compute));
        SA_HIP(hipStreamSynchronize(k.compute));
        double d;
        memcpy(&d, &bits, 8);
        if (d > worst) worst = d;
    }
    *max_abs_err = worst;
    return SA_OK;
}

// The same analytic inputs / checks for ONE natural-order vector on the current device (the single-GPU engine's C5 KATs).
static sa::DistLayout single_layout(int64_t n) {
    sa::DistLayout l;
    l.n = n; l.S = n; l.Sp = n; l.R1 = 1; l.K1L = 1; l.natural = 1; l.rank = 0;
    return l;
}
int sa_c2c_kat_fill_dev(double* x, int64_t n, int32_t kind, int64_t param, uint64_t seed, void* stream) {
    using namespace sa;
    SA_ENTER();
    if (!x || n < 1 || kind < SA_DIST_KAT_IMPULSE || kind > SA_DIST_KAT_NOISE || (kind != SA_DIST_KAT_NOISE && (param < 0 || param >= n))) {
        set_error("sa_c2c_kat_fill_dev: bad arguments (n=%lld kind=%d param=%lld)", (long long)n, kind, (long long)param);
        return SA_EINVAL;
    }
    hipStream_t st = stream ? (hipStream_t)stream : thread_stream();
    hipLaunchKernelGGL(dist_fill_kernel, dim3(copy_grid(n)), dim3(256), 0, st, reinterpret_cast<Cd*>(x), n, single_layout(n), kind, param, seed);
    SA_HIP(hipGetLastError());
    return SA_OK;
}
int sa_c2c_kat_check_dev(const double* X, int64_t n, int32_t kind, int64_t param, double* max_abs_err, void* stream) {
    using namespace sa;
    SA_ENTER();
    if (!X || !max_abs_err || n < 1 || (kind != SA_DIST_KAT_IMPULSE && kind != SA_DIST_KAT_TONE) || param < 0 || param >= n) {
        set_error("sa_c2c_kat_check_dev: bad arguments (only the impulse and tone inputs have closed-form spectra)");
        return SA_EINVAL;
    }
    hipStream_t st = stream ? (hipStream_t)stream : thread_stream();
    Temp tmp;
    unsigned long long* derr = static_cast<unsigned long long*>(tmp.dev(8));
    if (!derr) return SA_ENOMEM;
    SA_HIP(hipMemsetAsync(derr, 0, 8, st));
    hipLaunchKernelGGL(dist_check_kernel, dim3(copy_grid(n)), dim3(256), 0, st, reinterpret_cast<const Cd*>(X), n, single_layout(n), kind, param, derr);
    SA_HIP(hipGetLastError());
    unsigned long long bits = 0;
    SA_HIP(hipMemcpyAsync(&bits, derr, 8, hipMemcpyDeviceToHost, st));
    SA_HIP(hipStreamSynchronize(st));
    memcpy(max_abs_err, &bits, 8);
    return SA_OK;
}

int sa_dist_execute(sa_dist_plan* p, sa_dist_stats* stats) {
    using namespace sa;
    // the guard keeps sa_shutdown from freeing the cached twiddle tables (get_cached_table) under the kernels queued below
    SA_ENTER();
    if (!p) { set_error("sa_dist_execute: plan is NULL"); return SA_EINVAL; }
    const int P = p->P, J = p->J;
    const int64_t Sp = p->Sp, K1L = p->K1L, CH = p->CH, S = p->S;
    const int64_t blk = K1L * Sp;                            // points per (source, owner) pair
    int rc = SA_OK;
    DeviceRestore keep;
    const auto t0 = std::chrono::steady_clock::now();
    // --- natural-order input: x[r n/P ...] = rows [r K1L, (r+1) K1L) of the [R1][S] view -> column blocks (in `work`)
    for (int i = 0; i < P; ++i) {
        sa_dist_rank& k = p->r[i];
        SA_HIPD(hipSetDevice(k.dev));
        SA_HIPD(hipEventRecord(k.e_start, k.compute));
        if (!p->nat_in) continue;
        for (int q = 0; q < P; ++q) {
            SA_HIPD(hipStreamWaitEvent(k.copy[q], k.e_start, 0));
            hipLaunchKernelGGL(dist_copy2d_kernel, dim3(copy_grid(blk)), dim3(256), 0, k.copy[q],
                               reinterpret_cast<Cd*>(p->r[q].work) + (int64_t)i * blk, Sp, reinterpret_cast<const Cd*>(k.in) + (int64_t)q * Sp, S, Sp, K1L);
            SA_HIPD(hipGetLastError());
            SA_HIPD(hipEventRecord(k.e_prex[q], k.copy[q]));
        }
    }
    // --- pass 1 (+ the fused exchange)
    for (int i = 0; i < P; ++i) {
        sa_dist_rank& k = p->r[i];
        SA_HIPD(hipSetDevice(k.dev));
        if (p->nat_in)
            for (int s = 0; s < P; ++s) SA_HIPD(hipStreamWaitEvent(k.compute, p->r[s].e_prex[i], 0));
        SA_HIPD(hipEventRecord(k.e_p1s, k.compute));
        const double* src = p->nat_in ? k.work : k.in;
        if (p->mode == SA_DIST_XCHG_FUSED) {
            int sh = 0;
            while ((1LL << sh) < K1L) ++sh;
            SA_TRYD(dist_pass1(src, nullptr, k.push_tab, sh, p->n, p->l1, Sp, (int64_t)i * Sp, p->nt, k.compute));
        } else {
            SA_TRYD(dist_pass1(src, k.send, nullptr, 0, p->n, p->l1, Sp, (int64_t)i * Sp, p->nt, k.compute));
        }
        SA_HIPD(hipEventRecord(k.e_p1e, k.compute));
    }
    // --- the exchange: (source i, owner q, chunk j) = rows [q K1L + j CH, + CH) of i's send buffer -> block i of q's recv
    if (p->mode == SA_DIST_XCHG_RCCL) {
        // one RCCL group per chunk over every rank's communicator, on the rank's stream copy[0]; the chunk has arrived at
        // owner q when q's own stream passes the event recorded after the group (its receives are ordered before it)
        const RcclApi* nc = rccl_api();
        for (int j = 0; j < J; ++j) {
            for (int i = 0; i < P; ++i) {
                sa_dist_rank& k = p->r[i];
                if (j == 0) {
                    SA_HIPD(hipSetDevice(k.dev));
                    SA_HIPD(hipStreamWaitEvent(k.copy[0], k.e_p1e, 0));
                    SA_HIPD(hipEventRecord(k.e_cs[0], k.copy[0]));   // ONE window per rank: RCCL does not time pairs apart
                }
            }
            SA_NCCLD(nc->GroupStart());
            for (int i = 0; i < P; ++i) {
                sa_dist_rank& k = p->r[i];
                if (J == 1) {                                // blocks of blk points at stride blk on both sides: the all-to-all itself
                    ncclResult_t r_ = nc->AllToAll(k.send, k.recv, (size_t)blk * 2, ncclDouble, p->comms[i], k.copy[0]);
                    if (r_ != ncclSuccess) { (void)nc->GroupEnd(); SA_NCCLD(r_); }
                    continue;
                }
                for (int dq = 0; dq < P; ++dq) {
                    const int q = (i + dq) % P;
                    ncclResult_t r_ = nc->Send(k.send + 2 * (((int64_t)q * K1L + (int64_t)j * CH) * Sp), (size_t)CH * Sp * 2, ncclDouble, q, p->comms[i], k.copy[0]);
                    if (r_ == ncclSuccess)
                        r_ = nc->Recv(k.recv + 2 * (((int64_t)q * K1L + (int64_t)j * CH) * Sp), (size_t)CH * Sp * 2, ncclDouble, q, p->comms[i], k.copy[0]);
                    if (r_ != ncclSuccess) { (void)nc->GroupEnd(); SA_NCCLD(r_); }
                }
            }
            SA_NCCLD(nc->GroupEnd());
            for (int i = 0; i < P; ++i) {
                sa_dist_rank& k = p->r[i];
                SA_HIPD(hipSetDevice(k.dev));
                // chunk j has arrived at THIS rank (its receives precede the event on its stream): slot [0][j] only
                SA_HIPD(hipEventRecord(k.e_chunk[0][j], k.copy[0]));
                if (j == J - 1) SA_HIPD(hipEventRecord(k.e_ce[0], k.copy[0]));
            }
        }
    } else if (p->mode != SA_DIST_XCHG_FUSED) {
        for (int j = 0; j < J; ++j)
            for (int i = 0; i < P; ++i) {
                sa_dist_rank& k = p->r[i];
                SA_HIPD(hipSetDevice(k.dev));
                for (int dq = 0; dq < P; ++dq) {
                    const int q = (i + dq) % P;              // every source starts with a different owner
                    hipStream_t cs = k.copy[q];
                    if (j == 0) {
                        SA_HIPD(hipStreamWaitEvent(cs, k.e_p1e, 0));
                        SA_HIPD(hipEventRecord(k.e_cs[q], cs));
                    }
                    const double* src = k.send + 2 * (((int64_t)q * K1L + (int64_t)j * CH) * Sp);
                    double* dst = p->r[q].recv + 2 * (((int64_t)i * K1L + (int64_t)j * CH) * Sp);
                    if (p->mode == SA_DIST_XCHG_DMA) {
                        SA_HIPD(hipMemcpyAsync(dst, src, (size_t)CH * Sp * 16, hipMemcpyDefault, cs));
                    } else {
                        hipLaunchKernelGGL(dist_copy2d_kernel, dim3(copy_grid(CH * Sp)), dim3(256), 0, cs, reinterpret_cast<Cd*>(dst), CH * Sp,
                                           reinterpret_cast<const Cd*>(src), CH * Sp, CH * Sp, (int64_t)1);
                        SA_HIPD(hipGetLastError());
                    }
                    SA_HIPD(hipEventRecord(k.e_chunk[q][j], cs));
                    if (j == J - 1) SA_HIPD(hipEventRecord(k.e_ce[q], cs));
                }
            }
    }
    // --- phase 2: chunk j of owner q as soon as its P parts are there
    for (int j = 0; j < J; ++j)
        for (int q = 0; q < P; ++q) {
            sa_dist_rank& k = p->r[q];
            SA_HIPD(hipSetDevice(k.dev));
            if (p->mode == SA_DIST_XCHG_RCCL) {
                SA_HIPD(hipStreamWaitEvent(k.compute, k.e_chunk[0][j], 0));      // the owner's own stream has passed group j
            } else {
                for (int s = 0; s < P; ++s)
                    SA_HIPD(hipStreamWaitEvent(k.compute, p->mode == SA_DIST_XCHG_FUSED ? p->r[s].e_p1e : p->r[s].e_chunk[q][j], 0));
            }
            if (j == 0) SA_HIPD(hipEventRecord(k.e_p2s, k.compute));
            SA_TRYD(dist_phase2(k.recv + 2 * ((int64_t)j * CH * Sp), k.work, k.out + 2 * ((int64_t)j * CH * S), S, CH, Sp, blk, p->nt, k.compute));
            if (j == J - 1) SA_HIPD(hipEventRecord(k.e_p2e, k.compute));
        }
    // --- natural-order output: X[k1 + R1 k'] of owner q -> rank t = k' / Sp at (k' % Sp) R1 + k1, into t's recv
    for (int q = 0; q < P && p->nat_out; ++q) {
        sa_dist_rank& k = p->r[q];
        SA_HIPD(hipSetDevice(k.dev));
        for (int t = 0; t < P; ++t) {
            hipStream_t cs = k.copy[t];
            for (int s = 0; s < P; ++s) SA_HIPD(hipStreamWaitEvent(cs, p->r[s].e_p2e, 0));   // nobody still reads its recv
            const int64_t tiles = ((K1L + 31) / 32) * ((Sp + 31) / 32), cap = (int64_t)num_cus() * 8;
            hipLaunchKernelGGL(dist_transpose_kernel, dim3((unsigned)(tiles < cap ? tiles : cap)), dim3(256), 0, cs,
                               reinterpret_cast<Cd*>(p->r[t].recv) + (int64_t)q * K1L, p->R1, reinterpret_cast<const Cd*>(k.out) + (int64_t)t * Sp, S, K1L, Sp);
            SA_HIPD(hipGetLastError());
            SA_HIPD(hipEventRecord(k.e_post[t], cs));
        }
    }
    for (int i = 0; i < P; ++i) {
        sa_dist_rank& k = p->r[i];
        SA_HIPD(hipSetDevice(k.dev));
        if (p->nat_out)
            for (int s = 0; s < P; ++s) SA_HIPD(hipStreamWaitEvent(k.compute, p->r[s].e_post[i], 0));
        SA_HIPD(hipEventRecord(k.e_end, k.compute));
    }
done:
    // synchronous by design: the stats below need every event, and the next call may overwrite the receive buffers
    for (int i = 0; i < P; ++i) {
        sa_dist_rank& k = p->r[i];
        (void)hipSetDevice(k.dev);
        for (hipStream_t s : k.copy) {
            hipError_t e = hipStreamSynchronize(s);
            if (e != hipSuccess && rc == SA_OK) { set_error("dist c2c: copy stream of rank %d: %s", i, hipGetErrorString(e)); rc = SA_EHIP; }
        }
        hipError_t e = hipStreamSynchronize(k.compute);
        if (e != hipSuccess && rc == SA_OK) { set_error("dist c2c: compute stream of rank %d: %s", i, hipGetErrorString(e)); rc = SA_EHIP; }
    }
    if (rc != SA_OK || !stats) return rc;
    const double total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    memset(stats, 0, sizeof(*stats));
    stats->n = p->n; stats->n_ranks = P; stats->n_devices = p->n_devices; stats->exchange_mode = p->mode; stats->n_chunks = J;
    stats->exchanges = 1 + (p->nat_in ? 1 : 0) + (p->nat_out ? 1 : 0);
    stats->bytes_per_peer = blk * 16;
    stats->total_ms = total_ms;
    double lmin = 1e300, lmax = 0, lsum = 0, xmax = 0;
    int links = 0;
    for (int i = 0; i < P; ++i) {
        sa_dist_rank& k = p->r[i];
        (void)hipSetDevice(k.dev);
        float ms = 0;
        if (hipEventElapsedTime(&ms, k.e_p1s, k.e_p1e) == hipSuccess && ms > stats->pass1_ms) stats->pass1_ms = ms;
        if (hipEventElapsedTime(&ms, k.e_p2s, k.e_p2e) == hipSuccess && ms > stats->phase2_ms) stats->phase2_ms = ms;
        if (hipEventElapsedTime(&ms, k.e_start, k.e_end) == hipSuccess && ms > stats->device_ms) stats->device_ms = ms;
        if (p->mode == SA_DIST_XCHG_RCCL) {
            // one window per rank (first group start -> last group end) and the time inside each chunk's group; the pairs
            // of a rank are inside one RCCL kernel and cannot be timed apart: report the egress, spread evenly per link
            double win = 0;
            if (hipEventElapsedTime(&ms, k.e_cs[0], k.e_ce[0]) == hipSuccess) win = ms;
            stats->rank_exchange_ms[i] = win;
            if (win > xmax) xmax = win;
            for (int j = 0; j < J && j < SA_DIST_MAX_RANKS; ++j)
                if (hipEventElapsedTime(&ms, j ? k.e_chunk[0][j - 1] : k.e_cs[0], k.e_chunk[0][j]) == hipSuccess && ms > stats->chunk_ms[j])
                    stats->chunk_ms[j] = ms;
            const double egress = win > 0 && P > 1 ? (double)blk * 16 * (P - 1) / (win * 1e-3) / 1e9 : 0.0;
            stats->rank_egress_gbps[i] = egress;
            for (int q = 0; q < P; ++q) {
                const double gbps = q == i ? (win > 0 ? (double)blk * 16 / (win * 1e-3) / 1e9 : 0.0) : egress / (P - 1);
                stats->link_gbps[i][q] = gbps;
                if (q != i && gbps > 0) { lmin = gbps < lmin ? gbps : lmin; lmax = gbps > lmax ? gbps : lmax; lsum += gbps; ++links; }
            }
            continue;
        }
        double first_ms = 1e300, last_ms = 0;                 // this rank's window over its inter-rank pairs, relative to e_p1e
        for (int q = 0; q < P; ++q) {
            double gbps = 0;
            if (p->mode != SA_DIST_XCHG_FUSED && q != i) {
                float a = 0, b = 0;
                if (hipEventElapsedTime(&a, k.e_p1e, k.e_cs[q]) == hipSuccess && hipEventElapsedTime(&b, k.e_p1e, k.e_ce[q]) == hipSuccess) {
                    if (a < first_ms) first_ms = a;
                    if (b > last_ms) last_ms = b;
                }
            }
            if (p->mode != SA_DIST_XCHG_FUSED) {
                if (hipEventElapsedTime(&ms, k.e_cs[q], k.e_ce[q]) == hipSuccess && ms > 0) {
                    gbps = (double)blk * 16 / (ms * 1e-3) / 1e9;
                    if (q != i && ms > xmax) xmax = ms;
                }
            } else if (hipEventElapsedTime(&ms, k.e_p1s, k.e_p1e) == hipSuccess && ms > 0) {
                gbps = (double)blk * 16 / (ms * 1e-3) / 1e9;   // the rows left during pass 1: a lower bound on the link rate
            }
            stats->link_gbps[i][q] = gbps;
            if (q != i && gbps > 0) { lmin = gbps < lmin ? gbps : lmin; lmax = gbps > lmax ? gbps : lmax; lsum += gbps; ++links; }
        }
        if (p->mode == SA_DIST_XCHG_FUSED) {
            if (hipEventElapsedTime(&ms, k.e_p1s, k.e_p1e) == hipSuccess) stats->rank_exchange_ms[i] = ms;
        } else if (last_ms > first_ms) {
            stats->rank_exchange_ms[i] = last_ms - first_ms;
        }
        if (stats->rank_exchange_ms[i] > 0 && P > 1)
            stats->rank_egress_gbps[i] = (double)blk * 16 * (P - 1) / (stats->rank_exchange_ms[i] * 1e-3) / 1e9;
    }
    stats->pairs_timed = p->mode == SA_DIST_XCHG_DMA || p->mode == SA_DIST_XCHG_KERNEL;
    stats->exchange_ms = p->mode == SA_DIST_XCHG_FUSED ? 0.0 : xmax;
    if (links) {
        stats->link_gbps_min = lmin; stats->link_gbps_max = lmax; stats->link_gbps_mean = lsum / links;
        const double window = p->mode == SA_DIST_XCHG_FUSED ? stats->pass1_ms : xmax;
        if (window > 0) stats->aggregate_gbps = (double)blk * 16 * P * (P - 1) / (window * 1e-3) / 1e9;
    }
    return SA_OK;
}

int sa_fft_c2c_f64_dist(const double* in_ri, double* out_ri, int64_t n, int32_t n_ranks, int32_t flags, sa_dist_stats* stats) {
    using namespace sa;
    SA_ENTER();
    if (!in_ri || !out_ri) { set_error("sa_fft_c2c_f64_dist: NULL buffer"); return SA_EINVAL; }
    if (n_ranks == 0) {                                      // 0 = one rank per visible device (rounded down to a power of two)
        int nd = 0;
        SA_HIP(hipGetDeviceCount(&nd));
        n_ranks = 1;
        while (n_ranks * 2 <= nd && n_ranks * 2 <= SA_DIST_MAX_RANKS) n_ranks *= 2;
    }
    sa_dist_plan* p = nullptr;
    // host arrays are natural order on both sides
    SA_TRY(sa_dist_plan_create(&p, n, n_ranks, nullptr, flags | SA_DIST_NATURAL_IN | SA_DIST_NATURAL_OUT));
    int rc = SA_OK;
    DeviceRestore keep;
    const size_t slab = (size_t)(n / n_ranks) * 16;
    for (int i = 0; i < n_ranks; ++i) {
        SA_HIPD(hipSetDevice(p->r[i].dev));
        SA_HIPD(hipMemcpyAsync(p->r[i].in, reinterpret_cast<const char*>(in_ri) + slab * i, slab, hipMemcpyHostToDevice, p->r[i].compute));
    }
    for (int i = 0; i < n_ranks; ++i) { SA_HIPD(hipSetDevice(p->r[i].dev)); SA_HIPD(hipStreamSynchronize(p->r[i].compute)); }
    SA_TRYD(sa_dist_execute(p, stats));
    for (int i = 0; i < n_ranks; ++i) {
        SA_HIPD(hipSetDevice(p->r[i].dev));
        SA_HIPD(hipMemcpyAsync(reinterpret_cast<char*>(out_ri) + slab * i, p->r[i].recv, slab, hipMemcpyDeviceToHost, p->r[i].compute));
    }
    for (int i = 0; i < n_ranks; ++i) { SA_HIPD(hipSetDevice(p->r[i].dev)); SA_HIPD(hipStreamSynchronize(p->r[i].compute)); }
done:
    dist_free(p);
    return rc;
}

}  // extern "C"
