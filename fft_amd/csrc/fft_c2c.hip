// fft_c2c.hip — power-of-two complex-to-complex FFT in fp64 for any length 2 .. 2^30.
//
// Serves (i) the GUI drop-in path for frame lengths the batched rfft kernels do not cover
// (reference main.py:475, 963, 1080: scipy.fft.fft of arbitrary N, through Bluestein in
// bluestein.hip) and (ii) BASELINE config 5, the single huge fp64 transform.
//
//   n <= 4096      one LDS-resident pass;
//   n <= 2^20      two passes   n = R1 * R2      (each factor 2^6 .. 2^10);
//   n <= 2^30      three passes n = R1 * R2 * R3.
// fp64 caps an LDS-resident transform at 8192 points (128 KiB), so 2^30 cannot be done in the two
// passes a 2^15 x 2^15 split would need: three passes, 3 x (16 B read + 16 B write) per point.
//
// Every pass is a decimation-in-frequency step: a strided pass transforms dimension R of the view
// [A][R][S] (8 adjacent columns per workgroup, 128-byte segments) and multiplies by w_{RS}^{s k}
// from a two-level table; the last pass transforms contiguous rows and scatters them to natural
// order with 128-byte coalesced segments.
#include <math.h>
#include <string.h>

#include <string>
#include <vector>

#include "lds_fft.h"
#include "runtime.h"

namespace sa {

using Cd = double2;
#ifndef C2C_PP
#define C2C_PP 8
#endif

// Streaming accesses of the long transforms (every pass reads and writes far more than the caches hold, each element
// once): nontemporal, as in rfft4096.h.
typedef double v2d __attribute__((ext_vector_type(2)));
template <bool NT> __device__ __forceinline__ Cd ld_c(const Cd* p) {
    if (NT) { const v2d v = __builtin_nontemporal_load(reinterpret_cast<const v2d*>(p)); return Cd{v.x, v.y}; }
    return *p;
}
template <bool NT> __device__ __forceinline__ void st_c(Cd* p, Cd z) {
    if (NT) { v2d v; v.x = z.x; v.y = z.y; __builtin_nontemporal_store(v, reinterpret_cast<v2d*>(p)); }
    else *p = z;
}

struct C2cArgs {
    const Cd* in;
    Cd* out;
    int64_t n;        // transform length
    int64_t batch;
    int64_t A;        // outer count of the [A][R][S] view (strided) / number of rows (final)
    int64_t S;        // inner stride (strided pass)
    int64_t R1, R2;   // final pass: row index a = k1 * R2 + k2, natural index = k1 + R1 k2 + R1 R2 k3
    const Cd* twR;    // w_R^i
    const Cd* tlo;    // w_Mp^b,        b < 2^h
    const Cd* thi;    // w_Mp^{a 2^h}
    int h;
    int conj_in;      // inverse transform: conjugate on load (first pass)
    int conj_out;     // inverse transform: conjugate on store (last pass)
    double scale;     // 1/n for the inverse, else 1
    int natural;      // final pass: 1 = natural order, 0 = leave digit-reversed
    // final pass on a SLICE of k1 (rows k1 in [slOff, slOff + R1) of slR1; in points at the slice): the natural index of
    // local row (k1', k2) is slOff + k1' + slR1 k2 + Aglob k3.  slR1 = 0: no slicing.
    int64_t slR1, slOff, Aglob;
    // ---- distributed four-step (dist_fft.hip) ----
    int64_t tw_col0;  // strided pass: global column index of local column 0 (the twiddle is w^{(tw_col0 + s) k})
    int nt_hint;      // -1: streaming passes chosen from n * batch; 0 / 1: forced
    // XF == 1 (blocked loads): element s of batch row b lives at in + b * blk_row + (s >> blk_shift) * blk_pitch +
    // (s & blk_mask) — the layout the exchange delivers: one block of columns per source rank
    int blk_shift;
    int64_t blk_mask, blk_pitch, blk_row;
    // XF == 2 (push stores): output row k goes to push_tab[k >> push_shift] + (k & push_mask) * push_row + column —
    // straight into the receive buffers of the ranks that own those rows (peer-mapped pointers)
    int push_shift;
    int64_t push_mask, push_row;
    Cd* const* push_tab;
};
enum { XF_PLAIN = 0, XF_BLK = 1, XF_PUSH = 2 };
constexpr int PUSH_TAB_MAX = 16;      // owners a push pass can address (= SA_DIST_MAX_RANKS)

// ---- strided pass ------------------------------------------------------------------------------
// FAR = false: the next column group's loads are issued before the twiddle/store epilogue, into the butterfly
//   registers, which are dead by then (no extra registers).
// FAR = true (rows far apart, first pass of a 3-pass transform): the loads go out a whole transform earlier, into a
//   second register set, and w_R sits in LDS (TwLds) so that nothing inside the transform waits on the vector-memory
//   counter.  With one 128 KiB workgroup per CU nothing else covers the latency of 1024 x 128-byte reads 16 MiB
//   apart: timing-only variants put that pass at 9.1 ms without the transform and 12.3 ms with it, while the
//   second pass (rows 16 KiB apart) hides the same transform completely.
// STREAM: bit 0 = nontemporal loads, bit 1 = nontemporal stores
template <int LR, int B, bool FAR, int STREAM = 0, int XF = XF_PLAIN>
__global__ void __launch_bounds__((FftPlan<LR, C2C_PP>::TPF * B)) c2c_strided_pass(C2cArgs a) {
    using PL = FftPlan<LR, C2C_PP>;
    constexpr int R = PL::R, P = PL::P, TPF = PL::TPF, R0 = PL::radix(0);
    constexpr int LDSF = lds_fft_elems<double, LR>();
    constexpr int S = LdsPadShift<double>::value;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Cd* smem = reinterpret_cast<Cd*>(smem_raw);
    const int col = threadIdx.x % B, tl = threadIdx.x / B;
    Cd* buf = smem + col * LDSF;
    Cd* twl = smem + B * LDSF;                       // FAR: w_R^i, i < R
    // XF_PUSH: the owners' base pointers sit in LDS too — a global table look-up in the store epilogue would queue behind
    // the early loads of the next group on the in-order vector-memory counter (measured 2^30, pass 1: 13.2 vs 10.0 ms)
    Cd** ptab = reinterpret_cast<Cd**>(twl + (FAR ? R : 0));
    if (XF == XF_PUSH && threadIdx.x < PUSH_TAB_MAX) {
        const int owners = (R >> a.push_shift);
        if ((int)threadIdx.x < owners) ptab[threadIdx.x] = a.push_tab[threadIdx.x];
    }
    if (FAR) {
        for (int i = threadIdx.x; i < R; i += TPF * B) twl[i] = a.twR[i];
    }
    if (FAR || XF == XF_PUSH) __syncthreads();
    const int64_t groups_per_a = a.S / B;
    const int64_t ngroups = a.batch * a.A * groups_per_a;
    Cd v[P], nx[FAR ? P : 1];
    auto load_group = [&](int64_t g, Cd* d) {
        const int64_t sg = g % groups_per_a, ab = g / groups_per_a;
        if (XF == XF_BLK) {                              // A == 1: ab is the batch row; a column group never straddles a block
            const Cd* src = a.in + ab * a.blk_row + sg * B + col;
#pragma unroll
            for (int u = 0; u < P / R0; ++u)
#pragma unroll
                for (int n1 = 0; n1 < R0; ++n1) {
                    const int64_t rs = (int64_t)(n1 * (R / R0) + tl + TPF * u) * a.S;
                    d[u * R0 + n1] = ld_c<(STREAM & 1) != 0>(&src[(rs >> a.blk_shift) * a.blk_pitch + (rs & a.blk_mask)]);
                }
            return;
        }
        const int64_t base = (ab / a.A) * a.n + (ab % a.A) * (R * a.S) + sg * B + col;
#pragma unroll
        for (int u = 0; u < P / R0; ++u)
#pragma unroll
            for (int n1 = 0; n1 < R0; ++n1) d[u * R0 + n1] = ld_c<(STREAM & 1) != 0>(&a.in[base + (int64_t)(n1 * (R / R0) + tl + TPF * u) * a.S]);
    };
    if ((int64_t)blockIdx.x < ngroups) load_group(blockIdx.x, v);
    for (int64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const int64_t sg = g % groups_per_a;
        const int64_t ab = g / groups_per_a;            // = batch * A + a
        const bool more = g + gridDim.x < ngroups;
        // unconditional (the last iteration re-reads its own group): a load under a branch would be waited for at the
        // join, which is right here
        if (FAR) load_group(more ? g + gridDim.x : g, nx);
        if (a.conj_in) {
#pragma unroll
            for (int i = 0; i < P; ++i) v[i].y = -v[i].y;
        }
        if (FAR) lds_fft_tw<double, LR, C2C_PP, false>(v, buf, tl, TwLds<double>{twl});
        else lds_fft<double, LR, C2C_PP, false>(v, buf, tl, a.twR);
        // twiddle w^{s k} (w = w_{RS}) and store, B contiguous columns per k.  A thread's outputs are column c
        // (s = sc), rows k = k0 + TPF u:  w^(sc k) = w^(sc k0) * (w^(sc TPF))^u.  Two table look-ups (requested
        // before the barrier that publishes the transform) and a product tree of depth <= log2 P replace one
        // look-up per output (2^30: 33.7 -> 31.0 ms).  Splitting w^(sc k0) further into wave- and column-uniform
        // factors measured slower (32.6 ms): HBM traffic per pass is already the algorithmic 2 x 17.18 GB.
        constexpr int NT = TPF * B;
        constexpr int NU = (R * B) / NT;                 // outputs per thread (= P)
        const int c = threadIdx.x % B, k0 = threadIdx.x / B;
        const int64_t sc = sg * B + c;
        const int64_t sct = sc + a.tw_col0;              // column index inside the whole (possibly distributed) view
        const int64_t hm = ((int64_t)1 << a.h) - 1;
        auto lookup = [&](int64_t j) { return cmul(a.thi[j >> a.h], a.tlo[j & hm]); };   // j < R * S
        const Cd w0 = lookup(sct * k0);
        Cd pw[5];                                        // d^(2^i), d = w^(sc TPF)
        pw[0] = lookup(sct * TPF);
        __syncthreads();
        if (!FAR && more) load_group(g + gridDim.x, v);
#pragma unroll
        for (int i = 1; (1 << i) < NU; ++i) pw[i] = cmul(pw[i - 1], pw[i - 1]);
        Cd* outp = a.out + (ab / a.A) * a.n + (ab % a.A) * (R * a.S) + sc;
        Cd bs[5];                                        // bs[t]: factor of the latest u with ctz(u) == t
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            Cd w;
            if (u == 0) {
                w = w0;
            } else {
                const int t = __builtin_ctz(u), parent = u & (u - 1);
                w = cmul(parent == 0 ? w0 : bs[__builtin_ctz(parent)], pw[t]);
                bs[t] = w;
            }
            const int k = k0 + TPF * u;
            const Cd z = cmul(smem[c * LDSF + lds_pad<S>(k)], w);
            if (XF == XF_PUSH) st_c<(STREAM & 2) != 0>(ptab[k >> a.push_shift] + (int64_t)(k & a.push_mask) * a.push_row + sc, z);
            else st_c<(STREAM & 2) != 0>(&outp[(int64_t)k * a.S], z);
        }
        if (FAR) {
#pragma unroll
            for (int i = 0; i < P; ++i) v[i] = nx[i];
        }
        __syncthreads();
    }
}

// ---- final (contiguous rows) pass ------------------------------------------------------------------
template <int LR, int B, int PP = C2C_PP, int STREAM = 0>
__global__ void __launch_bounds__((FftPlan<LR, PP>::TPF * B)) c2c_final_pass(C2cArgs a) {
    using PL = FftPlan<LR, PP>;
    constexpr int R = PL::R, P = PL::P, TPF = PL::TPF, R0 = PL::radix(0);
    constexpr int LDSF = lds_fft_elems<double, LR>();
    constexpr int S = LdsPadShift<double>::value;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    Cd* smem = reinterpret_cast<Cd*>(smem_raw);
    const int tl = threadIdx.x % TPF, col = threadIdx.x / TPF;
    Cd* buf = smem + col * LDSF;
    const int64_t groups_per_t = a.A / B;                // A is a multiple of B (or A == B == 1)
    const int64_t ngroups = a.batch * groups_per_t;
    Cd v[P];                                             // next group's loads fly during the store epilogue (see above)
    auto load_group = [&](int64_t g) {
        const int64_t b = g / groups_per_t;
        const int64_t rev = (g % groups_per_t) * B + col;   // natural-order row index: rev = k1 + R1 k2  ->  a = k1 R2 + k2
        const int64_t arow = a.natural ? (rev % a.R1) * a.R2 + rev / a.R1 : rev;
        const Cd* src = a.in + b * a.n + arow * R;
#pragma unroll
        for (int u = 0; u < P / R0; ++u)
#pragma unroll
            for (int n1 = 0; n1 < R0; ++n1) v[u * R0 + n1] = ld_c<(STREAM & 1) != 0>(&src[n1 * (R / R0) + tl + TPF * u]);
    };
    if ((int64_t)blockIdx.x < ngroups) load_group(blockIdx.x);
    for (int64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const int64_t b = g / groups_per_t;
        const int64_t r0 = (g % groups_per_t) * B;      // first natural-order row index of the group
        if (a.conj_in) {
#pragma unroll
            for (int i = 0; i < P; ++i) v[i].y = -v[i].y;
        }
        lds_fft<double, LR, PP>(v, buf, tl, a.twR);
        if (g + gridDim.x < ngroups) load_group(g + gridDim.x);
        constexpr int NT = TPF * B;
#pragma unroll
        for (int u = 0; u < (R * B) / NT; ++u) {
            const int idx = threadIdx.x + NT * u;
            const int c = idx % B, k = idx / B;
            Cd z = smem[c * LDSF + lds_pad<S>(k)];
            z.x *= a.scale; z.y *= a.scale;
            if (a.conj_out) z.y = -z.y;
            int64_t o = a.natural ? (r0 + c) + a.A * (int64_t)k : (r0 + c) * R + k;
            if (a.slR1) o = a.slOff + (r0 % a.R1) + c + a.slR1 * (r0 / a.R1) + a.Aglob * (int64_t)k;   // B divides the slice's R1
            st_c<(STREAM & 2) != 0>(&a.out[b * a.n + o], z);
        }
        __syncthreads();
    }
}

// ---- host side ---------------------------------------------------------------------------------------
struct TwCtx { int64_t n; int64_t count; int64_t stride; };
static void fill_tw(void* host, void* ctx) {
    TwCtx* c = static_cast<TwCtx*>(ctx);
    std::vector<double> t;
    host_twiddle(c->n, c->count, c->stride, t);
    memcpy(host, t.data(), t.size() * sizeof(double));
}
static const Cd* tw_table(int64_t n, int64_t count, int64_t stride) {
    TwCtx c{n, count, stride};
    const std::string key = "tw64:" + std::to_string(n) + ":" + std::to_string(count) + ":" + std::to_string(stride);
    return static_cast<const Cd*>(get_cached_table(key, (size_t)count * 16, fill_tw, &c));
}

template <int LR, int B, bool FAR = false, int NT = 0, int XF = XF_PLAIN>
static int launch_strided(C2cArgs a, hipStream_t st) {
    constexpr int lds = lds_fft_elems<double, LR>() * B * 16 + (FAR ? (16 << LR) : 0) + (XF == XF_PUSH ? PUSH_TAB_MAX * 8 : 0);
    static_assert(lds <= 160 * 1024, "strided pass does not fit LDS");
    auto kern = c2c_strided_pass<LR, B, FAR, NT, XF>;
    static thread_local uint64_t attr_devs = 0;            // bit d: attribute set for device d (it is per device)
    SA_TRY(allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds, &attr_devs));
    const int64_t ngroups = a.batch * a.A * (a.S / B);
    const int64_t cap = grid_cap(16, "C2C");
    hipLaunchKernelGGL(kern, dim3((unsigned)(ngroups < cap ? ngroups : cap)), dim3(FftPlan<LR, C2C_PP>::TPF * B), lds, st, a);
    SA_HIP(hipGetLastError());
    return SA_OK;
}
template <int LR, int B, int PP = C2C_PP, int NT = 0>
static int launch_final(C2cArgs a, hipStream_t st) {
    constexpr int lds = lds_fft_elems<double, LR>() * B * 16;
    auto kern = c2c_final_pass<LR, B, PP, NT>;
    static thread_local uint64_t attr_devs = 0;            // bit d: attribute set for device d (it is per device)
    SA_TRY(allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds, &attr_devs));
    const int64_t ngroups = a.batch * (a.A / B);
    const int64_t cap = grid_cap(16, "C2C");
    hipLaunchKernelGGL(kern, dim3((unsigned)(ngroups < cap ? ngroups : cap)), dim3(FftPlan<LR, PP>::TPF * B), lds, st, a);
    SA_HIP(hipGetLastError());
    return SA_OK;
}

// Nontemporal passes once the data set is far beyond the caches (>= SA_C2C_NT_MIB MiB per buffer, default 1024;
// 0 = never): below that the next pass still finds part of this one's output in L2 / MALL.
static bool c2c_stream_nt(const C2cArgs& a) {
    if (a.nt_hint >= 0) return a.nt_hint != 0;
    static const long mib = getenv("SA_C2C_NT_MIB") ? atol(getenv("SA_C2C_NT_MIB")) : 1024;
    return mib > 0 && a.n * a.batch * 16 >= mib * (1L << 20);
}

// FAR variant for the radix-1024 and radix-512 passes of long transforms (every workgroup then loops over several
// column groups, which is what the early loads pay off on; measured 2^27: 3.24 -> 3.06 ms, 2^29: 14.7 -> 13.9 ms,
// first two passes of 2^30: 13.2 + 9.1 -> 12.5 + 8.4 ms; short transforms lose a few percent to its set-up).
// SA_C2C_FAR (tuning aid): bit 0 = radix 1024, bit 1 = radix 512; default both.
static int run_strided(int lr, const C2cArgs& a, hipStream_t st) {
    static const int far_env = getenv("SA_C2C_FAR") ? atoi(getenv("SA_C2C_FAR")) : 3;
    const int far = a.batch * a.A * (a.S / 8) >= 8 * (int64_t)num_cus() ? far_env : 0;
    if (c2c_stream_nt(a)) {
        if (lr == 10 && (far & 1)) return launch_strided<10, 8, true, 3>(a, st);
        // SA_C2C_B16 (A/B aid, profiles/r03_c5_split_ab.txt): radix-512 passes take 16 columns per workgroup (256-byte segments)
        static const int b16 = getenv("SA_C2C_B16") ? atoi(getenv("SA_C2C_B16")) : 0;
        if (lr == 9 && (far & 2) && b16 && a.S % 16 == 0) return launch_strided<9, 16, true, 3>(a, st);
        if (lr == 9 && (far & 2)) return launch_strided<9, 8, true, 3>(a, st);
    }
    if (lr == 10 && (far & 1)) return launch_strided<10, 8, true>(a, st);
    if (lr == 9 && (far & 2)) return launch_strided<9, 8, true>(a, st);
    switch (lr) {
        case 6: return launch_strided<6, 8>(a, st);
        case 7: return launch_strided<7, 8>(a, st);
        case 8: return launch_strided<8, 8>(a, st);
        case 9: return launch_strided<9, 8>(a, st);
        case 10: return launch_strided<10, 8>(a, st);
    }
    set_error("c2c: internal factor 2^%d unsupported", lr);
    return SA_EUNSUP;
}
// Strided passes of the distributed transform: XF_BLK reads the exchange's block layout, XF_PUSH stores into the
// owners' receive buffers.  Long passes (every workgroup loops over several column groups) take the early-load
// form, streaming once the rank's slab is far beyond the caches, like the plain passes above.
template <int XF>
static int run_strided_x(int lr, const C2cArgs& a, hipStream_t st) {
    const bool far = a.batch * a.A * (a.S / 8) >= 8 * (int64_t)num_cus();
    if (far && c2c_stream_nt(a)) {
        if (lr == 10) return launch_strided<10, 8, true, 3, XF>(a, st);
        if (lr == 9) return launch_strided<9, 8, true, 3, XF>(a, st);
    }
    switch (lr) {
        case 6: return launch_strided<6, 8, false, 0, XF>(a, st);
        case 7: return launch_strided<7, 8, false, 0, XF>(a, st);
        case 8: return launch_strided<8, 8, false, 0, XF>(a, st);
        case 9: return launch_strided<9, 8, false, 0, XF>(a, st);
        case 10: return launch_strided<10, 8, false, 0, XF>(a, st);
    }
    set_error("c2c: internal factor 2^%d unsupported", lr);
    return SA_EUNSUP;
}
static int run_final_multi(int lr, const C2cArgs& a, hipStream_t st) {
    // 1024-point rows: 16 points per thread (512 threads, 208 VGPRs) measured 8.7 -> 7.8 ms at 2^30; 512-point rows
    // gain nothing from it.  SA_C2C_FINAL_PP16=0 (tuning aid) turns it off.
    static const int pp16 = getenv("SA_C2C_FINAL_PP16") ? atoi(getenv("SA_C2C_FINAL_PP16")) : 1;
    if (c2c_stream_nt(a)) {
        if (lr == 10 && (pp16 & 1)) return launch_final<10, 8, 16, 3>(a, st);
        if (lr == 9) return launch_final<9, 8, C2C_PP, 3>(a, st);
        if (lr == 11) return launch_final<11, 4, 16, 3>(a, st);        // 2048-point rows (SA_C2C_SPLIT experiments only)
    }
    if (lr == 10 && (pp16 & 1) && a.batch * (a.A / 8) >= 8 * (int64_t)num_cus()) return launch_final<10, 8, 16>(a, st);
    switch (lr) {
        case 6: return launch_final<6, 8>(a, st);
        case 7: return launch_final<7, 8>(a, st);
        case 8: return launch_final<8, 8>(a, st);
        case 9: return launch_final<9, 8>(a, st);
        case 10: return launch_final<10, 8>(a, st);
    }
    set_error("c2c: internal factor 2^%d unsupported", lr);
    return SA_EUNSUP;
}
static int run_final_single(int lr, const C2cArgs& a, hipStream_t st) {
    switch (lr) {
        case 1: return launch_final<1, 1>(a, st);
        case 2: return launch_final<2, 1>(a, st);
        case 3: return launch_final<3, 1>(a, st);
        case 4: return launch_final<4, 1>(a, st);
        case 5: return launch_final<5, 1>(a, st);
        case 6: return launch_final<6, 1>(a, st);
        case 7: return launch_final<7, 1>(a, st);
        case 8: return launch_final<8, 1>(a, st);
        case 9: return launch_final<9, 1>(a, st);
        case 10: return launch_final<10, 1>(a, st);
        case 11: return launch_final<11, 1>(a, st);
        case 12: return launch_final<12, 1>(a, st);
    }
    set_error("c2c: single-pass length 2^%d unsupported", lr);
    return SA_EUNSUP;
}

// Fill the two-level twiddle members for a strided pass over a view with Mp = R * S.
static int set_pass_twiddles(C2cArgs& a, int64_t mp) {
    int l = 0;
    while ((1LL << l) < mp) ++l;
    a.h = (l + 1) / 2;
    a.tlo = tw_table(mp, 1LL << a.h, 1);
    a.thi = tw_table(mp, mp >> a.h, 1LL << a.h);
    return (a.tlo && a.thi) ? SA_OK : SA_EHIP;
}

int c2c_f64_dev(const double* in, double* out, double* work, int64_t n, int64_t batch, int32_t flags, hipStream_t st) {
    SA_ENTER();
    int L = 0;
    while ((1LL << L) < n) ++L;
    if (n < 2 || (1LL << L) != n || L > 30) { set_error("c2c f64: n=%lld must be a power of two in [2, 2^30]", (long long)n); return SA_EUNSUP; }
    if (batch <= 0) return batch == 0 ? SA_OK : (set_error("batch < 0"), SA_EINVAL);
    const bool inverse = flags & 1;
    C2cArgs a{};
    a.nt_hint = -1;
    a.n = n; a.batch = batch;
    a.scale = inverse ? 1.0 / (double)n : 1.0;
    a.natural = (flags & 2) ? 0 : 1;
    if (L <= 12) {
        a.in = reinterpret_cast<const Cd*>(in); a.out = reinterpret_cast<Cd*>(out);
        a.A = 1; a.R1 = 1; a.R2 = 1;
        a.conj_in = inverse; a.conj_out = inverse;
        a.twR = tw_table(n, n, 1);
        if (!a.twR) return SA_EHIP;
        return run_final_single(L, a, st);
    }
    if (!work) { set_error("c2c f64: n > 4096 needs a work buffer of n*batch complex values"); return SA_EINVAL; }
    int l1, l2, l3 = 0;
    if (L <= 20) { l1 = (L + 1) / 2; l2 = L - l1; }
    else { l1 = (L + 2) / 3; l2 = (L - l1 + 1) / 2; l3 = L - l1 - l2; }
    // SA_C2C_SPLIT="l1,l2,l3" (A/B aid): another factorisation of a three-pass transform, e.g. 9,10,11 at 2^30
    if (l3 && getenv("SA_C2C_SPLIT")) {
        int e1 = 0, e2 = 0, e3 = 0;
        if (sscanf(getenv("SA_C2C_SPLIT"), "%d,%d,%d", &e1, &e2, &e3) == 3 && e1 + e2 + e3 == L && e1 >= 6 && e1 <= 10 && e2 >= 6 && e2 <= 10 &&
            e3 >= 6 && e3 <= 11 && (e3 < 11 || c2c_stream_nt(a))) { l1 = e1; l2 = e2; l3 = e3; }
    }
    const int64_t R1 = 1LL << l1, R2 = 1LL << l2, R3 = 1LL << l3;
    // pass 1: [1][R1][n/R1], in -> work
    a.in = reinterpret_cast<const Cd*>(in); a.out = reinterpret_cast<Cd*>(work);
    a.A = 1; a.S = n / R1; a.conj_in = inverse; a.conj_out = 0;
    a.twR = tw_table(R1, R1, 1);
    if (!a.twR) return SA_EHIP;
    SA_TRY(set_pass_twiddles(a, n));
    SA_TRY(run_strided(l1, a, st));
    a.conj_in = 0;
    if (l3 == 0) {
        // final: rows of R2, A = R1 rows, natural index = k1 + R1 k2
        a.in = reinterpret_cast<const Cd*>(work); a.out = reinterpret_cast<Cd*>(out);
        a.A = R1; a.R1 = R1; a.R2 = 1; a.conj_out = inverse;
        a.twR = tw_table(R2, R2, 1);
        if (!a.twR) return SA_EHIP;
        return run_final_multi(l2, a, st);
    }
    // Sliced form (streaming sizes, radix-1024 passes): passes 2 and 3 run slice by slice over k1, SA_C2C_SLICE_MIB of the
    // work buffer at a time, so that pass 3 reads what pass 2 just wrote out of the 256 MiB MALL instead of HBM: pass 2
    // then stores with the default policy and pass 3 loads with it (the other two streams stay nontemporal).
    static const long slice_mib = getenv("SA_C2C_SLICE_MIB") ? atol(getenv("SA_C2C_SLICE_MIB")) : 0;
    if (slice_mib > 0 && c2c_stream_nt(a) && l2 == 10 && l3 == 10 && batch == 1 && a.natural) {
        int64_t G = ((int64_t)slice_mib << 20) / (R2 * R3 * 16);
        if (G < 8) G = 8;
        while (G & (G - 1)) G &= G - 1;                       // power of two
        if (G > R1) G = R1;
        const Cd* tw2 = tw_table(R2, R2, 1);
        const Cd* tw3 = tw_table(R3, R3, 1);
        if (!tw2 || !tw3) return SA_EHIP;
        C2cArgs p2 = a, p3 = a;
        p2.A = G; p2.S = R3; p2.twR = tw2;
        SA_TRY(set_pass_twiddles(p2, R2 * R3));
        p3.A = G * R2; p3.R1 = G; p3.R2 = R2; p3.conj_out = inverse; p3.twR = tw3;
        p3.slR1 = R1; p3.Aglob = R1 * R2;
        for (int64_t a0 = 0; a0 < R1; a0 += G) {
            Cd* sl = reinterpret_cast<Cd*>(work) + a0 * R2 * R3;
            p2.in = sl; p2.out = sl;
            SA_TRY((launch_strided<10, 8, true, 1>(p2, st)));          // loads nontemporal, stores cached
            p3.in = sl; p3.out = reinterpret_cast<Cd*>(out); p3.slOff = a0;
            SA_TRY((launch_final<10, 8, 16, 2>(p3, st)));              // loads cached, stores nontemporal
        }
        return SA_OK;
    }
    // pass 2: [R1][R2][R3], work -> work (in place: a workgroup owns its columns)
    a.in = reinterpret_cast<const Cd*>(work); a.out = reinterpret_cast<Cd*>(work);
    a.A = R1; a.S = R3;
    a.twR = tw_table(R2, R2, 1);
    if (!a.twR) return SA_EHIP;
    SA_TRY(set_pass_twiddles(a, R2 * R3));
    SA_TRY(run_strided(l2, a, st));
    // final: rows of R3, A = R1 R2 rows
    a.in = reinterpret_cast<const Cd*>(work); a.out = reinterpret_cast<Cd*>(out);
    a.A = R1 * R2; a.R1 = R1; a.R2 = R2; a.conj_out = inverse;
    a.twR = tw_table(R3, R3, 1);
    if (!a.twR) return SA_EHIP;
    return run_final_multi(l3, a, st);
}

// ---- building blocks of the distributed four-step (dist_fft.hip) ---------------------------------------------------
// The distributed transform of n = R1 * S points IS the multi-pass transform above with the exchange put between its
// first pass and the rest: rank r owns columns [r Sp, (r+1) Sp) of the [R1][S] view (Sp = S / P), runs the first
// strided pass on them (twiddle w_n^{s k1} with the GLOBAL column s), rows k1 travel to their owners, and every owner
// finishes its R1 / P rows as a batch of independent S-point transforms.

// Pass 1 of rank r: in [R1][Sp] (local columns) -> [R1][Sp] rows k1 (times w_n^{(col0 + s) k1}); with push_tab the
// rows go straight to their owners: row k -> push_tab[k >> push_shift] + (k & push_mask) * Sp + s.
int dist_pass1(const double* in, double* out, double* const* push_tab, int push_shift, int64_t n, int l1, int64_t Sp,
               int64_t col0, int nt, hipStream_t st) {
    C2cArgs a{};
    a.nt_hint = nt;
    a.n = (int64_t)Sp << l1; a.batch = 1;               // per-rank extent (addresses); the twiddle tables cover n
    a.in = reinterpret_cast<const Cd*>(in); a.out = reinterpret_cast<Cd*>(out);
    a.A = 1; a.S = Sp; a.tw_col0 = col0; a.scale = 1.0;
    a.twR = tw_table(1LL << l1, 1LL << l1, 1);
    if (!a.twR) return SA_EHIP;
    SA_TRY(set_pass_twiddles(a, n));
    if (push_tab) {
        a.push_tab = reinterpret_cast<Cd* const*>(push_tab);
        a.push_shift = push_shift; a.push_mask = ((int64_t)1 << push_shift) - 1; a.push_row = Sp;
        return run_strided_x<XF_PUSH>(l1, a, st);
    }
    return run_strided_x<XF_PLAIN>(l1, a, st);
}

// [P][rows][Sp] -> [rows][S]: only for transforms so short that the owner finishes a row in one LDS pass
__global__ void unblock_rows_kernel(const Cd* __restrict__ in, Cd* __restrict__ out, int64_t rows, int64_t S, int blk_shift,
                                    int64_t blk_pitch, int64_t blk_row) {
    const int64_t total = rows * S;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / S, s = i % S;
        out[i] = in[b * blk_row + (s >> blk_shift) * blk_pitch + (s & (((int64_t)1 << blk_shift) - 1))];
    }
}

// The owner's part: `rows` transforms of S points whose input sits in the exchange's block layout (element s of row b
// at in + b Sp + (s / Sp) blk_pitch + s % Sp); out [rows][S] natural order, work [rows][S].
int dist_phase2(const double* in_blocked, double* work, double* out, int64_t S, int64_t rows, int64_t Sp, int64_t blk_pitch,
                int nt, hipStream_t st) {
    int LS = 0, lsp = 0;
    while ((1LL << LS) < S) ++LS;
    while ((1LL << lsp) < Sp) ++lsp;
    if (rows <= 0) return SA_OK;
    C2cArgs a{};
    a.nt_hint = nt;
    a.n = S; a.batch = rows; a.scale = 1.0; a.natural = 1;
    if (LS <= 12) {
        const int64_t total = rows * S;
        const int64_t nb = (total + 255) / 256;
        hipLaunchKernelGGL(unblock_rows_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, st,
                           reinterpret_cast<const Cd*>(in_blocked), reinterpret_cast<Cd*>(work), rows, S, lsp, blk_pitch, Sp);
        SA_HIP(hipGetLastError());
        a.in = reinterpret_cast<const Cd*>(work); a.out = reinterpret_cast<Cd*>(out);
        a.A = 1; a.R1 = 1; a.R2 = 1;
        a.twR = tw_table(S, S, 1);
        if (!a.twR) return SA_EHIP;
        return run_final_single(LS, a, st);
    }
    if (LS > 20) { set_error("dist c2c: per-row length 2^%d beyond two passes", LS); return SA_EUNSUP; }
    const int l1 = (LS + 1) / 2, l2 = LS - l1;
    const int64_t R1 = 1LL << l1, R2 = 1LL << l2;
    if (R2 > Sp) { set_error("dist c2c: too many ranks for this length (columns per rank %lld < %lld)", (long long)Sp, (long long)R2); return SA_EUNSUP; }
    a.in = reinterpret_cast<const Cd*>(in_blocked); a.out = reinterpret_cast<Cd*>(work);
    a.A = 1; a.S = S / R1;
    a.blk_shift = lsp; a.blk_mask = Sp - 1; a.blk_pitch = blk_pitch; a.blk_row = Sp;
    a.twR = tw_table(R1, R1, 1);
    if (!a.twR) return SA_EHIP;
    SA_TRY(set_pass_twiddles(a, S));
    SA_TRY(run_strided_x<XF_BLK>(l1, a, st));
    a.in = reinterpret_cast<const Cd*>(work); a.out = reinterpret_cast<Cd*>(out);
    a.A = R1; a.R1 = R1; a.R2 = 1;
    a.twR = tw_table(R2, R2, 1);
    if (!a.twR) return SA_EHIP;
    return run_final_multi(l2, a, st);
}

}  // namespace sa

extern "C" {

int sa_fft_c2c_f64_dev(const double* in_ri, double* out_ri, double* work, int64_t n, int64_t batch, int32_t flags,
                       void* stream) {
    SA_ENTER();
    if (!in_ri || !out_ri) { sa::set_error("sa_fft_c2c_f64_dev: NULL buffer"); return SA_EINVAL; }
    if (flags & ~3) { sa::set_error("sa_fft_c2c_f64_dev: unknown flag bits"); return SA_EINVAL; }
    hipStream_t st = stream ? (hipStream_t)stream : sa::thread_stream();
    return sa::c2c_f64_dev(in_ri, out_ri, work, n, batch, flags, st);
}

int sa_fft_c2c_f64(const double* in_ri, double* out_ri, int64_t n, int64_t batch, int32_t flags) {
    SA_ENTER();
    if (!in_ri || !out_ri) { sa::set_error("sa_fft_c2c_f64: NULL buffer"); return SA_EINVAL; }
    if (flags & ~3) { sa::set_error("sa_fft_c2c_f64: unknown flag bits"); return SA_EINVAL; }
    if (batch <= 0 || n <= 0) { if (batch == 0) return SA_OK; sa::set_error("sa_fft_c2c_f64: bad n/batch"); return SA_EINVAL; }
    hipStream_t st = sa::thread_stream();
    const size_t bytes = (size_t)n * batch * 16;
    sa::Temp tmp;                                   // per-thread arena; falls back to hipMalloc beyond its cache limit
    double* din = static_cast<double*>(tmp.dev(bytes));
    double* dout = din ? static_cast<double*>(tmp.dev(bytes)) : nullptr;
    double* dwork = (dout && n > 4096) ? static_cast<double*>(tmp.dev(bytes)) : nullptr;
    int rc = (!din || !dout || (n > 4096 && !dwork)) ? SA_ENOMEM : SA_OK;
    hipError_t e = hipSuccess;
    if (rc == SA_OK) {
        e = hipMemcpyAsync(din, in_ri, bytes, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) { sa::set_error("H2D copy failed: %s", hipGetErrorString(e)); rc = SA_EHIP; }
    }
    if (rc == SA_OK) rc = sa::c2c_f64_dev(din, dout, dwork, n, batch, flags, st);
    if (rc == SA_OK) {
        e = hipMemcpyAsync(out_ri, dout, bytes, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { sa::set_error("D2H copy failed: %s", hipGetErrorString(e)); rc = SA_EHIP; }
    }
    if (rc != SA_OK) (void)hipStreamSynchronize(st);   // the arena goes back with nothing in flight
    return rc;
}

}  // extern "C"
