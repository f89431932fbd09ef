// metrics_v15.hip — the scalar read-outs of versão15.py's analysis panel that its own formulas take over the FULL
// fftshift-ordered spectrum (versão15.py:1081, 1101, 1126-1131, 1145-1148, 1163-1166), on the device:
//     fundamental          np.argmax(Y) (first maximum: the negative-frequency twin), Y[argmax]
//     noise floor          np.median(Y)                      — exact order statistics by radix select on the bit patterns
//     half-power band      np.where(Y > A / np.sqrt(2))[0][[0, -1]]
//     FM side band         np.max(Y[(f > 0.5 f0) & (f < 1.5 f0) & (f != f0)])
//     duty cycle           np.sum(y > np.max(y) * 0.5)       — on the time samples
// One 256-thread workgroup per record; the record is swept from L2 a dozen times (a panel record is a few thousand
// points; 10^6 at most, sab10.py:43).  fp64 throughout, every comparison is the reference's own, so the integers it
// returns are exact and the doubles are elements of the input (the median of an even count: their mean, as numpy).
#include <string.h>

#include "runtime.h"

namespace sa {

namespace {

constexpr int VT = 256;

__device__ __forceinline__ unsigned long long sort_key(double v) {      // order-preserving map double -> uint64
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_value(unsigned long long k) {
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}

// k-th smallest (0-based) of x[0..n) by 8 passes of an 8-bit histogram over the keys; all threads return it
__device__ double select_kth(const double* __restrict__ x, int64_t n, int64_t k, unsigned* hist /*[256]*/, unsigned long long* sh /*[2]*/) {
    unsigned long long prefix = 0, himask = 0;
    for (int shift = 56; shift >= 0; shift -= 8) {
        for (int i = threadIdx.x; i < 256; i += VT) hist[i] = 0;
        __syncthreads();
        for (int64_t i = threadIdx.x; i < n; i += VT) {
            const unsigned long long key = sort_key(x[i]);
            if ((key & himask) == prefix) atomicAdd(&hist[(key >> shift) & 255], 1u);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int64_t acc = 0;
            int b = 0;
            for (; b < 255; ++b) {
                if (acc + hist[b] > k) break;
                acc += hist[b];
            }
            sh[0] = prefix | ((unsigned long long)b << shift);
            sh[1] = (unsigned long long)(k - acc);
        }
        __syncthreads();
        prefix = sh[0];
        k = (int64_t)sh[1];
        himask |= 0xffull << shift;
        __syncthreads();
    }
    return key_value(prefix);
}

template <typename V> __device__ V block_reduce(V v, V* scratch /*[VT/64]*/, V (*op)(V, V)) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = op(v, __shfl_xor(v, off));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    V r = scratch[0];
    for (int w = 1; w < VT / 64; ++w) r = op(r, scratch[w]);
    return r;
}
__device__ long long op_min_ll(long long a, long long b) { return a < b ? a : b; }
__device__ long long op_max_ll(long long a, long long b) { return a > b ? a : b; }
__device__ long long op_add_ll(long long a, long long b) { return a + b; }
__device__ double op_max_d(double a, double b) { return a > b ? a : b; }     // NaN-free inputs by construction (see below)

__global__ void __launch_bounds__(VT) v15_readouts_kernel(const double* __restrict__ y, int64_t ny, const double* __restrict__ f,
                                                          const double* __restrict__ Y, int64_t n, int64_t batch, sa_v15_readouts* out) {
    __shared__ unsigned hist[256];
    __shared__ unsigned long long sh[2];
    __shared__ long long sll[VT / 64];
    __shared__ double sd[VT / 64];
    for (int64_t rec = blockIdx.x; rec < batch; rec += gridDim.x) {
        const double* Yr = Y + rec * n;
        sa_v15_readouts r;
        memset(&r, 0, sizeof(r));
        // ---- np.argmax(Y) / Y[argmax]: first index of the maximum; a NaN is the maximum (numpy: the first NaN wins) ----------
        double best = -INFINITY;
        long long bi = n, nan_i = n;
        for (int64_t i = threadIdx.x; i < n; i += VT) {
            const double v = Yr[i];
            if (v != v) { nan_i = nan_i < i ? nan_i : (long long)i; continue; }
            if (v > best) { best = v; bi = i; }
        }
        const long long first_nan = block_reduce<long long>(nan_i, sll, op_min_ll);
        const double A0 = block_reduce<double>(best, sd, op_max_d);
        const long long amax = block_reduce<long long>(best == A0 ? bi : (long long)n, sll, op_min_ll);
        const bool has_nan = first_nan < n;
        r.argmax = has_nan ? first_nan : (amax < n ? amax : 0);
        const double A = has_nan ? NAN : A0;
        r.max = A;
        r.nan_count = has_nan ? 1 : 0;
        // ---- half-power band: first / last index with Y > A / sqrt(2) (false everywhere when A is NaN) ---------------------------
        const double half = A / sqrt(2.0);
        long long lo = n, hi = -1;
        const double f0 = f ? f[r.argmax] : 0.0;
        double sb = -INFINITY;
        long long sbc = 0;
        const double flo = f0 * 0.5, fhi = f0 * 1.5;
        for (int64_t i = threadIdx.x; i < n; i += VT) {
            const double v = Yr[i];
            if (v > half) { lo = lo < i ? lo : (long long)i; hi = i; }
            if (f) {
                const double fi = f[i];
                if (fi > flo && fi < fhi && fi != f0) {                  // the reference's mask, literally (:1163-1164)
                    ++sbc;
                    if (v != v) sb = NAN; else if (!(sb != sb) && v > sb) sb = v;
                }
            }
        }
        r.bw_first = block_reduce<long long>(lo, sll, op_min_ll);
        r.bw_last = block_reduce<long long>(hi, sll, op_max_ll);
        if (r.bw_first >= n) r.bw_first = -1;
        r.sideband_count = block_reduce<long long>(sbc, sll, op_add_ll);
        {   // np.max over the mask: NaN if any member is NaN
            const long long anynan = block_reduce<long long>(sb != sb ? 1 : 0, sll, op_max_ll);
            const double m = block_reduce<double>(sb != sb ? -INFINITY : sb, sd, op_max_d);
            r.sideband_max = anynan ? NAN : m;
        }
        // ---- np.median(Y): NaN if any NaN, else the middle order statistic(s) --------------------------------------------------------
        if (has_nan) {
            r.median = NAN;
        } else {
            const int64_t k1 = (n - 1) / 2, k2 = n / 2;
            const double m1 = select_kth(Yr, n, k1, hist, sh);
            double m2 = m1;
            if (k2 != k1) {
                // the next order statistic: m1 again when enough elements are <= m1, else the smallest element above it
                long long le = 0;
                double above = INFINITY;
                for (int64_t i = threadIdx.x; i < n; i += VT) {
                    const double v = Yr[i];
                    if (v <= m1) ++le; else above = v < above ? v : above;
                }
                const long long nle = block_reduce<long long>(le, sll, op_add_ll);
                const double mab = -block_reduce<double>(-above, sd, op_max_d);
                m2 = nle > k2 ? m1 : mab;
            }
            r.median = k2 == k1 ? m1 : 0.5 * (m1 + m2);                  // numpy: mean of the two middle values
        }
        // ---- duty cycle of the time samples: #{ y > max(y) * 0.5 } (versão15.py:1081) ---------------------------------------------------
        r.above_half_max = -1;
        if (y && ny > 0) {
            const double* yr = y + rec * ny;
            double ym = -INFINITY;
            long long ynan = 0;
            for (int64_t i = threadIdx.x; i < ny; i += VT) {
                const double v = yr[i];
                if (v != v) ynan = 1; else ym = v > ym ? v : ym;
            }
            const long long anyn = block_reduce<long long>(ynan, sll, op_max_ll);
            const double ymax = anyn ? NAN : block_reduce<double>(ym, sd, op_max_d);
            const double thr = ymax * 0.5;
            long long c = 0;
            for (int64_t i = threadIdx.x; i < ny; i += VT) c += yr[i] > thr ? 1 : 0;
            r.above_half_max = block_reduce<long long>(c, sll, op_add_ll);
            r.y_max = ymax;
        }
        if (threadIdx.x == 0) out[rec] = r;
        __syncthreads();
    }
}

}  // namespace

int v15_readouts_dev(const double* y, int64_t ny, const double* f, const double* Y, int64_t n, int64_t batch, sa_v15_readouts* out,
                     hipStream_t st) {
    const int64_t cap = (int64_t)num_cus() * 8;
    hipLaunchKernelGGL(v15_readouts_kernel, dim3((unsigned)(batch < cap ? batch : cap)), dim3(VT), 0, st, y, ny, f, Y, n, batch, out);
    SA_HIP(hipGetLastError());
    return SA_OK;
}

}  // namespace sa

extern "C" {

int sa_v15_readouts_f64_dev(const double* y, int64_t ny, const double* f, const double* Y, int64_t n, int64_t batch,
                            sa_v15_readouts* out, void* stream) {
    using namespace sa;
    SA_ENTER();
    if (batch < 0 || n < 1 || ny < 0) { set_error("v15 readouts: need batch >= 0, n >= 1, ny >= 0"); return SA_EINVAL; }
    if (batch == 0) return SA_OK;
    if (!Y || !out) { set_error("v15 readouts: NULL buffer"); return SA_EINVAL; }
    return v15_readouts_dev(y, ny, f, Y, n, batch, out, stream ? (hipStream_t)stream : thread_stream());
}

int sa_v15_readouts_f64(const double* y, int64_t ny, const double* f, const double* Y, int64_t n, int64_t batch, sa_v15_readouts* out) {
    using namespace sa;
    SA_ENTER();
    if (batch < 0 || n < 1 || ny < 0) { set_error("v15 readouts: need batch >= 0, n >= 1, ny >= 0"); return SA_EINVAL; }
    if (batch == 0) return SA_OK;
    if (!Y || !out) { set_error("v15 readouts: NULL buffer"); return SA_EINVAL; }
    hipStream_t st = thread_stream();
    const size_t yb = y ? (size_t)batch * ny * 8 : 0, fb = f ? (size_t)n * 8 : 0, Yb = (size_t)batch * n * 8, ob = (size_t)batch * sizeof(sa_v15_readouts);
    Temp tmp;
    char* d = static_cast<char*>(tmp.dev(yb + fb + Yb + ob));
    char* h = static_cast<char*>(tmp.pinned(yb + fb + Yb + ob));
    if (!d || !h) return SA_ENOMEM;
    if (yb) memcpy(h, y, yb);
    if (fb) memcpy(h + yb, f, fb);
    memcpy(h + yb + fb, Y, Yb);
    SA_HIP(hipMemcpyAsync(d, h, yb + fb + Yb, hipMemcpyHostToDevice, st));
    SA_TRY(v15_readouts_dev(yb ? reinterpret_cast<const double*>(d) : nullptr, ny, fb ? reinterpret_cast<const double*>(d + yb) : nullptr,
                            reinterpret_cast<const double*>(d + yb + fb), n, batch, reinterpret_cast<sa_v15_readouts*>(d + yb + fb + Yb), st));
    SA_HIP(hipMemcpyAsync(h + yb + fb + Yb, d + yb + fb + Yb, ob, hipMemcpyDeviceToHost, st));
    SA_HIP(hipStreamSynchronize(st));
    memcpy(out, h + yb + fb + Yb, ob);
    return SA_OK;
}

}  // extern "C"
