// synth_pipeline.hip — fused generator pipeline (BASELINE config 4), fp32 spectra, zero HBM input:
//   params -> synthesis + AM/FM (main.py:460-472, 536-629) at n_base points
//          -> U x not-a-knot cubic-spline upsample (scipy CubicSpline + np.linspace, main.py:1390-1392)
//          -> window -> rfft(n = U * n_base) -> |Y| (-> dB) -> peak       (main.py:475-479, 1029)
// One 1024-thread (n/16-thread) workgroup per frame; the only HBM traffic is the 64-byte parameter
// record in and the n/2+1 magnitudes + peak out.
//
// The spline is a GLOBAL tridiagonal solve (knot derivatives s = A^-1 b).  For uniform knots A is the
// constant [1 4 1] matrix with scipy's two not-a-knot rows, so A^-1 B (B maps samples to right-hand
// sides) is frame independent and its entries decay like (2 - sqrt 3)^|i-j| ~ 0.268^|i-j|: the solve
// becomes a 41-tap stencil in the interior plus an exact table for the first / last 24 rows, both
// computed once on the host in fp64 (truncation < 1e-11, far below fp32 rounding).
#include <stdlib.h>
#include <string.h>

#include <string>
#include <type_traits>
#include <vector>

#include "lds_fft.h"
#include "lds_fft_tail2.h"
#include "runtime.h"
#include "synth_gen.h"

#pragma clang fp contract(fast)

namespace sa {

constexpr int SPL_W = 20;     // interior stencil half width
constexpr int SPL_RB = 24;    // rows treated with the exact boundary table on each side
constexpr int SPL_BC = SPL_RB + SPL_W;   // columns per boundary row

struct PipeArgs {
    const sa_gen_params* params;  // [batch]
    int64_t batch;
    const float* stencil;         // [2 W + 1] interior derivative stencil (per knot spacing)
    const float* brows;           // [RB][BC] exact first rows of A^-1 B (last rows by antisymmetry)
    const float* win;             // [n] or nullptr
    const float2* twM;            // w_M^i, M = n/2
    const float2* twN;            // w_n^i, i <= M/2
    int flags;
    float* mag;                   // [batch][n/2+1] or nullptr
    int32_t* peak_bin;
    float* peak_mag;
    float* y_out;                 // [batch][n] upsampled time signal, or nullptr
};

// CLS: the generator class this instantiation serves (synth_gen.h).  The launch runs one kernel per class, each skipping
// the other's frames: the GEN_LIGHT kernel (every name but the eight that need fp64 exp / log / pow / tan / sinh / j0) fits
// 128 VGPRs, so two of its 512-thread workgroups share a CU; the GEN_HEAVY kernel keeps the 256-register budget.
template <int LOG2M, int U, int PP, int CLS>
__global__ void __launch_bounds__((FftPlan<LOG2M, PP>::TPF), (CLS == GEN_LIGHT && FftPlan<LOG2M, PP>::TPF >= 512 ? 4 : 1))
synth_pipeline_kernel(PipeArgs a) {
    using PL = FftPlan<LOG2M, PP>;
    constexpr int M = PL::R, P = PL::P, TPF = PL::TPF, R0 = PL::radix(0);
    constexpr int N = 2 * M;
    constexpr int NB = N / U;                          // base points
    constexpr int S = LdsPadShift<float>::value;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2* buf = reinterpret_cast<float2*>(smem_raw);
    // The base samples, knot derivatives and per-interval coefficients are dead once the upsampled frame sits in the
    // registers of FFT pass 0, so they live INSIDE the transform buffer (24 NB = 6 M bytes of its 8 M): the workgroup
    // needs 70 KiB instead of 116 and two of them share a CU — one's barriers hide behind the other's arithmetic.
    float* yb = reinterpret_cast<float*>(smem_raw);                               // [NB] base samples
    float* sb = yb + NB;                                                          // [NB] knot derivatives
    // per-interval cubic coefficients (y0, s0, slope - s0 - tt, tt): one 16-byte read per evaluated sample
    float4* cf = reinterpret_cast<float4*>(smem_raw + (size_t)NB * 8);
    static_assert((U > 1 ? 24 : 4) * NB <= 8 * M, "the staging arrays fit inside the transform buffer");
    constexpr int NW = (TPF + 63) / 64;
    float* red_m = reinterpret_cast<float*>(buf + lds_fft_elems<float, LOG2M>());
    int* red_k = reinterpret_cast<int*>(red_m + NW);
    const int tl = threadIdx.x;
    const bool excl = a.flags & SA_F_EXCLUDE_DC_NYQ;
    const bool wmag = (a.flags & SA_F_WRITE_MAG) && a.mag;
    const bool wdb = a.flags & SA_F_DB;
    // Launch constants of the thread (it handles the same positions of every frame): the FFT twiddles always stay in
    // registers; window samples and the Hermitian factors w_n^q too in the small workgroups.  The 512-thread workgroups
    // (n >= 8192) reload those 48 registers from L2 per frame instead: with the fp64 generator's footprint that is what
    // lets the kernel fit 128 VGPRs, i.e. two workgroups per CU.
    constexpr bool HOIST = TPF < 512;
    constexpr int NIT = (M / 2) / TPF;
    float2 w[HOIST ? P : 1];
    float2 tn[HOIST ? NIT : 1];
    auto load_window = [&](float2* wd, int lane) {
        if (a.win) {
#pragma unroll
            for (int u = 0; u < P / R0; ++u)
#pragma unroll
                for (int n1 = 0; n1 < R0; ++n1)
                    wd[u * R0 + n1] = reinterpret_cast<const float2*>(a.win)[n1 * (M / R0) + lane + TPF * u];
        } else {
#pragma unroll
            for (int i = 0; i < P; ++i) wd[i] = mk<float>(1.f, 1.f);
        }
#pragma unroll
        for (int i = 0; i < P; ++i) { wd[i].x *= 0.5f; wd[i].y *= 0.5f; }     // the 1/2 of the Hermitian split (exact)
    };
    if constexpr (HOIST) {
        load_window(w, tl);
#pragma unroll
        for (int u = 0; u < NIT; ++u) tn[u] = a.twN[tl + TPF * u];
    }
    // n = 16384 (8192 complex points, 512 threads): the transform's last radix-2 pass is folded into the Hermitian
    // post-pass (lds_fft_tail2.h)
    constexpr bool T2 = LOG2M == 13 && PP == 16;
    TwRegs<float, LOG2M, PP> tw;
    FftTail2_8192 tw2;
    if constexpr (T2) tw2.load(a.twM, tl, reinterpret_cast<float2*>(red_k + NW)); else tw.load(a.twM, tl);

    // The workgroup's frames are f = blockIdx.x + j gridDim.x.  Which of them belong to this instantiation's class is
    // looked up 64 at a time (lane j of every wave reads record j's waveform id; the ballot is the same in every wave), so
    // a launch with no frame of its class costs each workgroup ONE round of loads instead of one dependent load per frame
    // (65 536 light-class records: the heavy-class launch took 0.14 ms of 3.07).
    for (int64_t f0 = blockIdx.x; f0 < a.batch; f0 += (int64_t)gridDim.x * 64) {
      const int64_t fl = f0 + (int64_t)(tl & 63) * gridDim.x;
      const bool mine = fl < a.batch && is_heavy_waveform(a.params[fl < a.batch ? fl : 0].waveform) == (CLS == GEN_HEAVY);
      unsigned long long todo = __builtin_amdgcn_ballot_w64(mine);
      while (todo) {
        const int64_t f = f0 + (int64_t)__builtin_ctzll(todo) * gridDim.x;
        todo &= todo - 1;
        sa_gen_params p = a.params[f];
        p.N = NB;
        int tq = tl;                                    // opaque per frame (keeps address math out of the loop-invariant set)
        asm volatile("" : "+v"(tq));
        tq &= TPF - 1;
        // ---- 1. synthesis at the base rate (fp64 phase, fp32 sample) ---------------------------------
        for (int i = tq; i < NB; i += TPF) yb[i] = (float)modulated_sample<TrigF32, CLS>(p, i);
        __syncthreads();
        // ---- 2. knot derivatives (per knot spacing) ---------------------------------------------------
        if constexpr (U > 1) {
            for (int i = tq; i < NB; i += TPF) {
                float s = 0.f;
                if (i < SPL_RB) {
                    const float* r = a.brows + i * SPL_BC;
#pragma unroll 4
                    for (int j = 0; j < SPL_BC; ++j) s += r[j] * yb[j];
                } else if (i >= NB - SPL_RB) {
                    const float* r = a.brows + (NB - 1 - i) * SPL_BC;     // antisymmetric mirror
#pragma unroll 4
                    for (int j = 0; j < SPL_BC; ++j) s -= r[j] * yb[NB - 1 - j];
                } else {
#pragma unroll
                    for (int k = -SPL_W; k <= SPL_W; ++k) s += a.stencil[k + SPL_W] * yb[i + k];
                }
                sb[i] = s;
            }
            __syncthreads();
            for (int i = tq; i < NB - 1; i += TPF) {          // scipy PPoly coefficients of interval i, dx = 1
                const float y0 = yb[i], y1 = yb[i + 1], s0 = sb[i], s1 = sb[i + 1];
                const float slope = y1 - y0, tt = s0 + s1 - 2.f * slope;
                cf[i] = make_float4(y0, s0, slope - s0 - tt, tt);
            }
            __syncthreads();
        }
        // ---- 3. upsample + window straight into the registers of FFT pass 0 ---------------------------
        // samples 2c and 2c+1 of np.linspace(t0, t_last, N): position j (NB-1)/(N-1) = i + r/(N-1); the second sample's
        // (i, r) follow from the first by one add and a wrap, so the pair shares the division
        auto sample_pair = [&](unsigned c) -> float2 {
            if constexpr (U == 1) {
                return mk<float>(yb[2 * c], yb[2 * c + 1]);
            } else {
                constexpr float INV = 1.0f / (float)(N - 1);
                const unsigned num = 2 * c * (unsigned)(NB - 1);           // < 2^25
                unsigned i0 = num / (unsigned)(N - 1);
                unsigned r0 = num - i0 * (unsigned)(N - 1);
                unsigned r1 = r0 + (unsigned)(NB - 1);
                const bool wrap = r1 >= (unsigned)(N - 1);
                unsigned i1 = i0 + (wrap ? 1u : 0u);
                r1 -= wrap ? (unsigned)(N - 1) : 0u;
                if (i1 > NB - 2) { i1 = NB - 2; r1 += (unsigned)(N - 1); }  // the very last sample: u = 1 on the last interval
                const float4 c0 = cf[i0], c1 = cf[i1];
                const float u0 = (float)r0 * INV, u1 = (float)r1 * INV;
                return mk<float>(((c0.w * u0 + c0.z) * u0 + c0.y) * u0 + c0.x, ((c1.w * u1 + c1.z) * u1 + c1.y) * u1 + c1.x);
            }
        };
        float2 v[P];
#pragma unroll
        for (int u = 0; u < P / R0; ++u)
#pragma unroll
            for (int n1 = 0; n1 < R0; ++n1) v[u * R0 + n1] = sample_pair(n1 * (M / R0) + tq + TPF * u);
        if (a.y_out) {
            float2* yo = reinterpret_cast<float2*>(a.y_out + f * N);
#pragma unroll
            for (int u = 0; u < P / R0; ++u)
#pragma unroll
                for (int n1 = 0; n1 < R0; ++n1) yo[n1 * (M / R0) + tq + TPF * u] = v[u * R0 + n1];
        }
        if constexpr (HOIST) {
#pragma unroll
            for (int i = 0; i < P; ++i) { v[i].x *= w[i].x; v[i].y *= w[i].y; }
        } else {
            float2 wl[P];
            load_window(wl, tq);
#pragma unroll
            for (int i = 0; i < P; ++i) { v[i].x *= wl[i].x; v[i].y *= wl[i].y; }
        }
        __syncthreads();                                // every thread has its samples: the staging arrays may be overwritten
        if constexpr (T2) { tw2.run(v, buf, tl); __syncthreads(); }
        else lds_fft_tw<float, LOG2M, PP>(v, buf, tl, tw);
        // ---- 4. Hermitian post-pass, magnitudes, peak ---------------------------------------------------
        asm volatile("" : "+v"(tq));
        tq &= TPF - 1;
        float best = -1.f;
        int bk = 0x7fffffff;
        const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(
            wmag ? a.mag + f * (M + 1) : nullptr, 0, wmag ? (M + 1) * 4 : 0, 0x00020000);   // zero-length: stores dropped
        // dB is a uniform run-time choice: two instantiations of the epilogue behind one scalar branch
        // Peak trackers.  A thread's bins arrive as monotone sequences (q ascending, M - q descending, ...): one tracker per
        // sequence then needs ONE compare per bin — '>' keeps the first of equal maxima in an ascending sequence, '>=' takes
        // the last in a descending one, both the lowest bin — instead of the general (m > best || (m == best && k < bk)) with
        // its DC / Nyquist test, eleven instructions per bin, a tenth of this kernel's arithmetic.  The trackers are merged
        // with the general rule once per frame.  NaN never enters (every comparison fails), as before.
        struct Trk {
            float b = -1.f;
            int k = 0x7fffffff;
            __device__ __forceinline__ void asc(float m, int kk, bool cnt) { const bool g = cnt & (m > b); b = g ? m : b; k = g ? kk : k; }
            __device__ __forceinline__ void desc(float m, int kk, bool cnt) { const bool g = cnt & (m >= b); b = g ? m : b; k = g ? kk : k; }
            __device__ __forceinline__ void any(float m, int kk) { const bool g = (m > b) | ((m == b) & (kk < k)); b = g ? m : b; k = g ? kk : k; }
        };
        Trk t1, t2, t3, t4;
        auto epilogue = [&](auto DBT) {
            constexpr bool DB = decltype(DBT)::value;
            auto put = [&](float m, int voff, int soff) {
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(DB ? 20.f * log10f(fmaxf(m, SA_DB_FLOOR_F32)) : m), mrs, voff, soff, 0);
            };
            // (zq, zm) = (Z[q], Z[M - q]) -> bins q and M - q; w = w_n^q
            // SELF0: for lane tq == 0 this call is the pair (Z[M/2], Z[M/2]) — both results are bin M/2, from two formulas that
            // may round one ulp apart; one value for the row and for the peak (rfft4096.h)
            // QASC: this call's q belongs to an ascending sequence of the thread (tracker tq_; its M - q then descends, tracker
            // tm_), else the other way round.  cnt: false drops both bins from the peak search (DC / Nyquist excluded).
            auto pair = [&](float2 zq, float2 zm, float2 wq, int q, int voff_q, int soff_q, int voff_m, int soff_m, auto SELF0,
                            auto QASC, Trk& tq_, Trk& tm_, bool cnt) {
                const float2 e = mk<float>(zq.x + zm.x, zq.y - zm.y);          // the 1/2 is in the window
                const float2 d = mk<float>(zq.x - zm.x, zq.y + zm.y);
                const float2 wo = cmul(mk<float>(d.y, -d.x), wq);
                const float2 xa = cadd(e, wo), xb = csub(e, wo);
                float ma = mag_sqrt(xa.x * xa.x + xa.y * xa.y);
                const float mb = mag_sqrt(xb.x * xb.x + xb.y * xb.y);
                if constexpr (decltype(SELF0)::value) ma = tq == 0 ? mb : ma;
                put(ma, voff_q, soff_q);
                put(mb, voff_m, soff_m);
                if constexpr (decltype(QASC)::value) { tq_.asc(ma, q, cnt); tm_.desc(mb, M - q, cnt); }
                else { tq_.desc(ma, q, cnt); tm_.asc(mb, M - q, cnt); }
            };
            const bool c0 = !(excl && tq == 0);                                 // bins 0 and M are lane 0's first pair
            if constexpr (T2) {
                // Legs of the final radix-2 butterflies: Z[j] = E[j] + O[j], Z[j + M/2] = E[j] - O[j] (j < M/2).  A thread
                // takes butterflies j = tq + TPF u < M/4 and their mirrors M/2 - j: pair A is q = j with
                // Z[M - j] = E[M/2 - j] - O[M/2 - j], pair B is q = M/2 - j with Z[M/2 + j] = E[j] - O[j] and the factor
                // w_n^(M/2 - j) = -i conj(w_n^j).  Lane 0's j = 0 is its own mirror: (Z[0], Z[0]) and (Z[M/2], Z[M/2]).
                constexpr int H = M / 2, PLANE = FftTail2_8192::PLANE;
                const float2* pe = buf + lds_pad<S>(tq);
                const int mlow = lds_pad<S>(TPF - tq);
                float2 tj[H / 2 / TPF];
#pragma unroll
                for (int u = 0; u < H / 2 / TPF; ++u) tj[u] = a.twN[tq + TPF * u];
#pragma unroll
                for (int u = 0; u < H / 2 / TPF; ++u) {
                    const int j = tq + TPF * u;
                    const int moff = lds_pad_c<S>(H - TPF * (u + 1));
                    const int m0 = lds_pad_c<S>((H - TPF * u) & (H - 1)) - moff;
                    const int ml = (tq == 0 ? m0 : mlow) + moff;
                    const float2 ej = pe[lds_pad_c<S>(TPF * u)], oj = pe[PLANE + lds_pad_c<S>(TPF * u)];
                    const float2 em = buf[ml], om = buf[PLANE + ml];
                    float2 za = cadd(ej, oj), zam = csub(em, om), zb = cadd(em, om), zbm = csub(ej, oj);
                    if (u == 0) {                                   // j = 0: (S, D, S, D) -> (S, S), (D, D)
                        const bool z0 = tq == 0;
                        const float2 t = zam;
                        zam.x = z0 ? zb.x : zam.x; zam.y = z0 ? zb.y : zam.y;
                        zb.x = z0 ? t.x : zb.x;    zb.y = z0 ? t.y : zb.y;
                    }
                    // bins j (ascending in u) and M - j (descending); H - j (descending) and H + j (ascending)
                    pair(za, zam, tj[u], j, tq * 4, TPF * u * 4, (TPF - tq) * 4, (M - TPF * (u + 1)) * 4, std::false_type{},
                         std::true_type{}, t1, t2, u == 0 ? c0 : true);
                    if (u == 0)
                        pair(zb, zbm, mk<float>(-tj[u].y, -tj[u].x), H - j, (TPF - tq) * 4, (H - TPF * (u + 1)) * 4,
                             tq * 4, (H + TPF * u) * 4, std::true_type{}, std::false_type{}, t3, t4, true);
                    else
                        pair(zb, zbm, mk<float>(-tj[u].y, -tj[u].x), H - j, (TPF - tq) * 4, (H - TPF * (u + 1)) * 4,
                             tq * 4, (H + TPF * u) * 4, std::false_type{}, std::false_type{}, t3, t4, true);
                }
                if (tq == 0) {                                      // butterfly M/4 is its own mirror: bins M/4 and 3M/4
                    const float2 e = buf[lds_pad_c<S>(H / 2)], o = buf[PLANE + lds_pad_c<S>(H / 2)];
                    Trk x1, x2;                                     // out of sequence: merged with the general rule below
                    pair(cadd(e, o), csub(e, o), a.twN[H / 2], H / 2, 0, (H / 2) * 4, 0, (M - H / 2) * 4, std::false_type{},
                         std::true_type{}, x1, x2, true);
                    t1.any(x1.b, x1.k); t1.any(x2.b, x2.k);
                }
                return;
            }
            // additive padded addresses: Z[q] at pad(tq) + pad(TPF u), its mirror at pad(TPF - tq) + a constant for
            // every lane but tq = 0 (see rfft_generic.h)
            const float2* pq = buf + lds_pad<S>(tq);
            const int mlow = lds_pad<S>(TPF - tq);
            float2 tl_[HOIST ? 1 : NIT];
            if constexpr (!HOIST) {
#pragma unroll
                for (int u = 0; u < NIT; ++u) tl_[u] = a.twN[tq + TPF * u];
            }
#pragma unroll
            for (int u = 0; u < NIT; ++u) {                 // pair q < M/2 -> bins q and M - q
                const int q = tq + TPF * u;
                const int moff = lds_pad_c<S>(M - TPF * (u + 1));
                const int m0 = lds_pad_c<S>((M - TPF * u) & (M - 1)) - moff;
                const float2 zq = pq[lds_pad_c<S>(TPF * u)], zm = buf[(tq == 0 ? m0 : mlow) + moff];
                pair(zq, zm, HOIST ? tn[u] : tl_[u], q, tq * 4, TPF * u * 4, (TPF - tq) * 4, (M - TPF * (u + 1)) * 4, std::false_type{},
                     std::true_type{}, t1, t2, u == 0 ? c0 : true);
            }
            if (tq == 0) {                                  // bin M/2: X = conj(Z[M/2])
                const float2 z = buf[lds_pad_c<S>(M / 2)];
                const float m = 2.0f * mag_sqrt(z.x * z.x + z.y * z.y);
                put(m, 0, (M / 2) * 4);
                t1.any(m, M / 2);
            }
        };
        if (wdb) epilogue(std::true_type{}); else epilogue(std::false_type{});
        t1.any(t2.b, t2.k); t3.any(t4.b, t4.k); t1.any(t3.b, t3.k);
        best = t1.b; bk = t1.k;
        if constexpr (TPF >= 64) {
            // whole waves per frame: wave maximum, then the lowest bin among the lanes that hold it, on the DPP path.  A lane
            // whose bins were all NaN still carries -1: it enters as 0 with no bin (rfft_generic.h)
            const float wb = wave_max_f32(fmaxf(best, 0.0f));
            bk = wave_min_u32(best == wb ? (unsigned)bk : 0x7fffffffu);
            best = bk == 0x7fffffff ? -1.f : wb;
        } else {
            constexpr int WRED = TPF;
#pragma unroll
            for (int off = WRED / 2; off > 0; off >>= 1) {
                const float om = __shfl_xor(best, off);
                const int ok = __shfl_xor(bk, off);
                const bool tk = (om > best) | ((om == best) & (ok < bk));
                best = tk ? om : best; bk = tk ? ok : bk;
            }
        }
        if constexpr (TPF > 64) {
            if ((threadIdx.x & 63) == 0) { red_m[threadIdx.x >> 6] = best; red_k[threadIdx.x >> 6] = bk; }
            __syncthreads();
            if (tl == 0)
                for (int w = 1; w < NW; ++w)
                    if (red_m[w] > best || (red_m[w] == best && red_k[w] < bk)) { best = red_m[w]; bk = red_k[w]; }
        }
        if (tl == 0) { a.peak_bin[f] = bk; a.peak_mag[f] = best; }
        __syncthreads();
      }
    }
}

// ---- host: spline operator tables -----------------------------------------------------------------------
// Solve the not-a-knot system of scipy's CubicSpline for uniform unit spacing on n = 160 knots for each
// unit sample vector; rows far from the ends are the shift-invariant stencil, the first RB rows the table.
static void build_spline_tables(std::vector<float>& stencil, std::vector<float>& brows) {
    const int n = 160;
    std::vector<double> C((size_t)n * n, 0.0);     // C[i][j] = d s_i / d y_j
    std::vector<double> lo(n), di(n), up(n), rhs(n), cp(n), dp(n);
    for (int j = 0; j < n; ++j) {
        std::vector<double> y(n, 0.0);
        y[j] = 1.0;
        auto slope = [&](int i) { return y[i + 1] - y[i]; };
        for (int i = 1; i < n - 1; ++i) { lo[i] = 1; di[i] = 4; up[i] = 1; rhs[i] = 3.0 * (slope(i - 1) + slope(i)); }
        // scipy not-a-knot rows (dx = 1): s0 + 2 s1 = (5 slope0 + slope1)/2 ;  2 s_{n-2} + s_{n-1} = (slope_{n-3} + 5 slope_{n-2})/2
        lo[0] = 0; di[0] = 1; up[0] = 2; rhs[0] = 0.5 * (5.0 * slope(0) + slope(1));
        lo[n - 1] = 2; di[n - 1] = 1; up[n - 1] = 0; rhs[n - 1] = 0.5 * (slope(n - 3) + 5.0 * slope(n - 2));
        cp[0] = up[0] / di[0]; dp[0] = rhs[0] / di[0];
        for (int i = 1; i < n; ++i) {
            const double m = di[i] - lo[i] * cp[i - 1];
            cp[i] = up[i] / m;
            dp[i] = (rhs[i] - lo[i] * dp[i - 1]) / m;
        }
        std::vector<double> s(n);
        s[n - 1] = dp[n - 1];
        for (int i = n - 2; i >= 0; --i) s[i] = dp[i] - cp[i] * s[i + 1];
        for (int i = 0; i < n; ++i) C[(size_t)i * n + j] = s[i];
    }
    stencil.resize(2 * SPL_W + 1);
    for (int k = -SPL_W; k <= SPL_W; ++k) stencil[k + SPL_W] = (float)C[(size_t)(n / 2) * n + n / 2 + k];
    brows.assign((size_t)SPL_RB * SPL_BC, 0.f);
    for (int i = 0; i < SPL_RB; ++i)
        for (int j = 0; j < SPL_BC; ++j) brows[(size_t)i * SPL_BC + j] = (float)C[(size_t)i * n + j];
}
static void fill_stencil(void* host, void*) {
    std::vector<float> s, b;
    build_spline_tables(s, b);
    memcpy(host, s.data(), s.size() * 4);
}
static void fill_brows(void* host, void*) {
    std::vector<float> s, b;
    build_spline_tables(s, b);
    memcpy(host, b.data(), b.size() * 4);
}

template <int LOG2M, int U, int PP>
static int launch_pipe_pp(const PipeArgs& a, hipStream_t st) {
    constexpr int NBASE = (2 << LOG2M) / U;
    constexpr int NW = (FftPlan<LOG2M, PP>::TPF + 63) / 64;
    constexpr int lds = lds_fft_elems<float, LOG2M>() * 8 + NW * 8 + 32 +                // staging arrays alias the transform buffer
                        (LOG2M == 13 && PP == 16 ? FftTail2_8192::TAB * 8 : 0);          // + the second pass' twiddle table
    (void)NBASE;
    auto kern = synth_pipeline_kernel<LOG2M, U, PP, GEN_LIGHT>;
    auto kern_heavy = synth_pipeline_kernel<LOG2M, U, PP, GEN_HEAVY>;
    static thread_local uint64_t attr_devs = 0, attr_devs_heavy = 0;   // bit d: attribute set for device d (it is per device)
    SA_TRY(allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds, &attr_devs));
    SA_TRY(allow_dynamic_lds(reinterpret_cast<const void*>(kern_heavy), (int)lds, &attr_devs_heavy));
    const int64_t cap = grid_cap(4, "PIPE");
    hipLaunchKernelGGL(kern, dim3((unsigned)(a.batch < cap ? a.batch : cap)), dim3(FftPlan<LOG2M, PP>::TPF), lds, st, a);
    SA_HIP(hipGetLastError());
    // the frames of the eight libm-heavy names (a workgroup of the other class reads a frame's 64-byte record and moves on)
    hipLaunchKernelGGL(kern_heavy, dim3((unsigned)(a.batch < cap ? a.batch : cap)), dim3(FftPlan<LOG2M, PP>::TPF), lds, st, a);
    SA_HIP(hipGetLastError());
    return SA_OK;
}

// SA_PIPE_PP (tuning aid): points per thread of the fused generator kernel (8 or 16)
template <int LOG2M, int U>
static int launch_pipe(const PipeArgs& a, hipStream_t st) {
    static const int pp = getenv("SA_PIPE_PP") ? atoi(getenv("SA_PIPE_PP")) : 16;   // 16: the fp64 generator needs the 256-VGPR budget
    if (pp == 16) return launch_pipe_pp<LOG2M, U, 16>(a, st);
    return launch_pipe_pp<LOG2M, U, 8>(a, st);
}

int synth_analyze_dev(const sa_gen_params* dparams, int64_t batch, int32_t n_base, int32_t upsample, int32_t window,
                      double win_param, int32_t flags, float* mag, int32_t* peak_bin, float* peak_mag, float* y_out,
                      hipStream_t st) {
    SA_ENTER();
    if (batch <= 0) return batch == 0 ? SA_OK : (set_error("batch < 0"), SA_EINVAL);
    if (upsample != 1 && upsample != 8) { set_error("synth_analyze: upsample must be 1 or 8"); return SA_EUNSUP; }
    const int64_t n = (int64_t)n_base * upsample;
    int l = 0;
    while ((1LL << l) < n) ++l;
    if ((1LL << l) != n || n < 2048 || n > 16384 || n_base < 256) {
        set_error("synth_analyze: n_base*upsample=%lld must be a power of two in [2048, 16384] (n_base >= 256)", (long long)n);
        return SA_EUNSUP;
    }
    if (window < SA_WIN_NONE || window > SA_WIN_LAST) { set_error("unknown window id %d", window); return SA_EINVAL; }
    if (flags & ~7) { set_error("synth_analyze: unknown flag bits"); return SA_EINVAL; }
    if (!dparams || !peak_bin || !peak_mag || ((flags & SA_F_WRITE_MAG) && !mag)) { set_error("synth_analyze: NULL buffer"); return SA_EINVAL; }
    const RfftPlan* plan = get_rfft_plan(false, (int)n, window, win_param);
    if (!plan) return SA_EHIP;
    PipeArgs a{};
    a.params = dparams; a.batch = batch; a.flags = flags; a.mag = mag; a.peak_bin = peak_bin; a.peak_mag = peak_mag; a.y_out = y_out;
    a.win = static_cast<const float*>(plan->d_win);
    a.twM = static_cast<const float2*>(plan->d_twM);
    a.twN = static_cast<const float2*>(plan->d_twN);
    a.stencil = static_cast<const float*>(get_cached_table("spl_stencil", (2 * SPL_W + 1) * 4, fill_stencil, nullptr));
    a.brows = static_cast<const float*>(get_cached_table("spl_brows", (size_t)SPL_RB * SPL_BC * 4, fill_brows, nullptr));
    if (!a.stencil || !a.brows) return SA_EHIP;
    const int lm = l - 1;
    if (upsample == 8) {
        switch (lm) {
            case 10: return launch_pipe<10, 8>(a, st);
            case 11: return launch_pipe<11, 8>(a, st);
            case 12: return launch_pipe<12, 8>(a, st);
            case 13: return launch_pipe<13, 8>(a, st);
        }
    } else {
        switch (lm) {
            case 10: return launch_pipe<10, 1>(a, st);
            case 11: return launch_pipe<11, 1>(a, st);
            case 12: return launch_pipe<12, 1>(a, st);
            case 13: return launch_pipe<13, 1>(a, st);
        }
    }
    set_error("synth_analyze: unsupported size");
    return SA_EUNSUP;
}

}  // namespace sa

static int synth_analyze_host(const sa_gen_params* params, int64_t batch, int32_t n_base, int32_t upsample, int32_t window,
                              double win_param, int32_t flags, float* mag, int32_t* peak_bin, float* peak_mag, float* y_out);

extern "C" {

int sa_synth_analyze_dev(const sa_gen_params* params_dev, int64_t batch, int32_t n_base, int32_t upsample, int32_t window,
                         double win_param, int32_t flags, float* mag, int32_t* peak_bin, float* peak_mag, float* y_out,
                         void* stream) {
    SA_ENTER();
    hipStream_t st = stream ? (hipStream_t)stream : sa::thread_stream();
    return sa::synth_analyze_dev(params_dev, batch, n_base, upsample, window, win_param, flags, mag, peak_bin, peak_mag, y_out, st);
}

int sa_synth_analyze(const sa_gen_params* params, int64_t batch, int32_t n_base, int32_t upsample, int32_t window,
                     double win_param, int32_t flags, float* mag, int32_t* peak_bin, float* peak_mag, float* y_out) {
    if (sa::shard_count() >= 2 && batch >= 2 && params && peak_bin && peak_mag && n_base > 0 && upsample > 0) {
        // parameter records fan out as contiguous blocks over the shard devices (sa_shard_devices)
        const int64_t n = (int64_t)n_base * upsample, nb = n / 2 + 1;
        return sa::run_sharded(batch, [&](int64_t lo, int64_t hi) {
            return synth_analyze_host(params + lo, hi - lo, n_base, upsample, window, win_param, flags, mag ? mag + lo * nb : nullptr,
                                      peak_bin + lo, peak_mag + lo, y_out ? y_out + lo * n : nullptr);
        });
    }
    return synth_analyze_host(params, batch, n_base, upsample, window, win_param, flags, mag, peak_bin, peak_mag, y_out);
}

}  // extern "C"

static int synth_analyze_host(const sa_gen_params* params, int64_t batch, int32_t n_base, int32_t upsample, int32_t window,
                              double win_param, int32_t flags, float* mag, int32_t* peak_bin, float* peak_mag, float* y_out) {
    using namespace sa;
    SA_ENTER();
    if (batch < 0) { set_error("batch < 0"); return SA_EINVAL; }
    if (batch == 0) return SA_OK;
    if (!params) { set_error("synth_analyze: params is NULL"); return SA_EINVAL; }
    hipStream_t st = thread_stream();
    const int64_t n = (int64_t)n_base * upsample, nb = n / 2 + 1;
    int64_t chunk = (int64_t)(1ull << 30) / (n * 4);
    if (chunk < 1) chunk = 1;
    if (chunk > batch) chunk = batch;
    const bool wmag = flags & SA_F_WRITE_MAG;
    Temp tmp;
    sa_gen_params* dp = static_cast<sa_gen_params*>(tmp.dev((size_t)chunk * sizeof(sa_gen_params)));
    float* dmag = wmag ? static_cast<float*>(tmp.dev((size_t)chunk * nb * 4)) : nullptr;
    float* dy = y_out ? static_cast<float*>(tmp.dev((size_t)chunk * n * 4)) : nullptr;
    float* dpm = static_cast<float*>(tmp.dev((size_t)chunk * 4));
    int32_t* dpb = static_cast<int32_t*>(tmp.dev((size_t)chunk * 4));
    int rc = (!dp || (wmag && !dmag) || (y_out && !dy) || !dpm || !dpb) ? SA_ENOMEM : SA_OK;
    hipError_t e = hipSuccess;
    for (int64_t b0 = 0; rc == SA_OK && b0 < batch; b0 += chunk) {
        const int64_t cnt = batch - b0 < chunk ? batch - b0 : chunk;
        e = hipMemcpyAsync(dp, params + b0, (size_t)cnt * sizeof(sa_gen_params), hipMemcpyHostToDevice, st);
        if (e != hipSuccess) { set_error("synth_analyze: H2D failed: %s", hipGetErrorString(e)); rc = SA_EHIP; break; }
        rc = synth_analyze_dev(dp, cnt, n_base, upsample, window, win_param, flags, dmag, dpb, dpm, dy, st);
        if (rc != SA_OK) break;
        if (wmag) e = hipMemcpyAsync(mag + b0 * nb, dmag, (size_t)cnt * nb * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && y_out) e = hipMemcpyAsync(y_out + b0 * n, dy, (size_t)cnt * n * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(peak_bin + b0, dpb, (size_t)cnt * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(peak_mag + b0, dpm, (size_t)cnt * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { set_error("synth_analyze: D2H failed: %s", hipGetErrorString(e)); rc = SA_EHIP; }
    }
    if (rc != SA_OK) (void)hipStreamSynchronize(st);      // nothing may still be using the arena when it is handed back
    return rc;
}
