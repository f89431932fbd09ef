// stft.hip — short-time |FFT| + per-frame peak of interleaved PCM16 audio (BASELINE config 3).
//
// The reference has no STFT; it is the north-star extension of the same hot path
// (window -> real FFT -> |Y| -> peak, main.py:475-479, 1029) to overlapping frames of a long record,
// with the int16 -> float scaling of the reference's own import path (int16 / 32768, versão15.py:637).
//
// Stereo (channels == 2): frame f of BOTH channels is ONE complex transform of length nfft,
//   z[j] = w[j] (L[j] + i R[j]) — each interleaved (L,R) int16 pair is one 4-byte coalesced load —
//   and the channels are separated by Hermitian symmetry in the epilogue:
//   L^[k] = (Z[k] + conj Z[N-k]) / 2,   R^[k] = (Z[k] - conj Z[N-k]) / (2i).
//   Every loaded byte is used; the 75 %-overlap re-reads of neighbouring frames are served by L2 / MALL.
// Other channel counts: a de-interleave + convert pass, then the batched rfft kernels with
//   frame stride = hop.
// Output layout: mag[channel][frame][nfft/2+1], peak_bin / peak_mag[channel][frame].
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <vector>

#include "lds_fft.h"
#include "lds_fft_tail2.h"
#include "runtime.h"

namespace sa {

template <typename T>
int rfft_peak_strided_dev(const T* x, int64_t batch, int64_t frame_stride, int32_t n, int32_t window, double win_param,
                          int32_t flags, T* mag, int32_t* peak_bin, T* peak_mag, hipStream_t st);

constexpr int STFT_PP(int lr) { return lr >= 10 ? 16 : 8; }   // points per thread

struct StftArgs {
    const int16_t* pcm;     // interleaved stereo
    int64_t nframes;        // STFT frames per channel
    int64_t hop;
    const float* win;       // [nfft] or nullptr
    const float* winq;      // [nfft], window x 1/2 x 1/32768 (never null; stft8192q.h only)
    const float2* twR;      // w_N^i
    int flags;
    float* mag;             // [2][nframes][N/2+1]
    int32_t* peak_bin;      // [2][nframes]
    float* peak_mag;        // [2][nframes]
};

}  // namespace sa
#include "stft8192.h"      // nfft = 8192: 32 points per thread, two exchanges (uses StftArgs)
#include "stft8192q.h"     // the same plan as four 35 KiB workgroups per CU (half exchanges, constants from L2)
namespace sa {

// frames per workgroup: small transforms share a workgroup of >= 256 threads
template <int LR, int PP> constexpr int stft_fpw() {
    constexpr int tpf = FftPlan<LR, PP>::TPF;
    return tpf >= 256 ? 1 : 256 / tpf;
}

template <int LR, int PP>
__global__ void __launch_bounds__((FftPlan<LR, PP>::TPF * stft_fpw<LR, PP>()), 4)   // >= 4 waves per SIMD: two 512-thread workgroups per CU
stft_stereo_kernel(StftArgs a) {
    using PL = FftPlan<LR, PP>;
    constexpr int N = PL::R, P = PL::P, TPF = PL::TPF, R0 = PL::radix(0);
    constexpr int FPW = stft_fpw<LR, PP>();
    constexpr int S = LdsPadShift<float>::value;
    constexpr int LDSF = lds_fft_elems<float, LR>();
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float2* smem = reinterpret_cast<float2*>(smem_raw);
    constexpr int NW = (TPF * FPW + 63) / 64;
    float* red_m = reinterpret_cast<float*>(smem + FPW * LDSF);
    int* red_k = reinterpret_cast<int*>(red_m + 2 * NW);
    // index inside the frame's thread group / frame slot inside the workgroup (slot 0 is a compile-time
    // constant for one frame per workgroup, so the frame index stays uniform and lives in SGPRs)
    const int tl = FPW == 1 ? (int)threadIdx.x : (int)threadIdx.x % TPF;
    const int g = FPW == 1 ? 0 : (int)threadIdx.x / TPF;
    float2* buf = smem + g * LDSF;
    const bool wmag = (a.flags & SA_F_WRITE_MAG) && a.mag;
    const bool wdb = a.flags & SA_F_DB;
    const bool excl = a.flags & SA_F_EXCLUDE_DC_NYQ;
    constexpr int NB = N / 2 + 1;
    // Per-thread constants of the whole launch: every thread handles the same positions of every frame,
    // so its window samples (pre-scaled by 1/32768, and by the 1/2 of the channel split — exact, a power of two)
    // and FFT twiddles live in registers; inside the frame
    // loop the only global loads are the PCM words, and those of the NEXT frame are in flight while this
    // frame is transformed.
    float w[P];
    if (a.win) {
#pragma unroll
        for (int u = 0; u < P / R0; ++u)
#pragma unroll
            for (int n1 = 0; n1 < R0; ++n1) w[u * R0 + n1] = a.win[n1 * (N / R0) + TPF * u + tl] * (0.5f / 32768.0f);
    } else {
#pragma unroll
        for (int i = 0; i < P; ++i) w[i] = 0.5f / 32768.0f;
    }
    // nfft 8192, one frame per workgroup: the last radix-2 pass is folded into the epilogue (lds_fft_tail2.h)
    constexpr bool T2 = LR == 13 && PP == 16 && FPW == 1;
    TwRegs<float, LR, PP> tw;
    FftTail2_8192 tw2;
    if constexpr (T2) tw2.load(a.twR, tl, reinterpret_cast<float2*>(red_k + 2 * NW)); else tw.load(a.twR, tl);
    const uint32_t* pcm = reinterpret_cast<const uint32_t*>(a.pcm);   // one (L,R) pair per dword
    uint32_t raw[P];
    // Frame loads / spectrum stores go through buffer descriptors: one 32-bit lane offset per access family plus
    // scalar offsets, range-checked by the hardware.  With one frame per workgroup the frame base is uniform and
    // lives in the descriptor; with several, the descriptor spans the record and the frame offset is per lane.
    const int64_t last = a.nframes - 1;
    auto load_frame = [&](int64_t f, int lane) {
        if constexpr (FPW == 1) {
            const __amdgpu_buffer_rsrc_t rs =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(pcm + f * a.hop), 0, N * 4, 0x00020000);
#pragma unroll
            for (int u = 0; u < P / R0; ++u)
#pragma unroll
                for (int n1 = 0; n1 < R0; ++n1)
                    raw[u * R0 + n1] = __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, (n1 * (N / R0) + TPF * u) * 4, 0);
        } else {
            const uint32_t* src = pcm + f * a.hop + lane;
#pragma unroll
            for (int u = 0; u < P / R0; ++u)
#pragma unroll
                for (int n1 = 0; n1 < R0; ++n1) raw[u * R0 + n1] = src[n1 * (N / R0) + TPF * u];
        }
    };
    // frames past the end repeat the last frame (every thread reaches the barriers); their stores are dropped.
    // Frame -> workgroup map: blocks b and b + 8 share an XCD (and its L2; observed placement, used for speed only), so
    // each residue class b % 8 walks its own CONTIGUOUS eighth of the frames: the workgroups running on an XCD at any
    // time then work on neighbouring frames, whose (nfft - hop) / nfft overlap is fetched into that L2 once instead
    // of once per XCD (hop = nfft / 4: the same samples used to be fetched by four different L2s).
    const int64_t units = (a.nframes + FPW - 1) / FPW, per_xcd = (units + 7) / 8;
    const int64_t xcd_first = (int64_t)(blockIdx.x & 7) * per_xcd, slot_step = gridDim.x >> 3;   // gridDim.x % 8 == 0 (host)
    const int64_t slot0 = blockIdx.x >> 3;
    const int64_t first = (xcd_first + slot0) * FPW;
    if (slot0 < per_xcd) load_frame(first + g <= last ? first + g : last, tl);
    for (int64_t slot = slot0; slot < per_xcd; slot += slot_step) {
        const int64_t base = (xcd_first + slot) * FPW;
        const bool valid = base + g <= last;
        const int64_t f = valid ? base + g : last;
        int tq = tl;                             // opaque per frame: nothing below is hoisted out of the loop
        asm volatile("" : "+v"(tq));
        tq &= TPF - 1;
        float2 v[P];
#pragma unroll
        for (int i = 0; i < P; ++i)
            v[i] = mk<float>((float)(int16_t)(raw[i] & 0xFFFF) * w[i], (float)(int16_t)(raw[i] >> 16) * w[i]);
        {
            const int64_t nb_ = base + slot_step * FPW + g;                     // next round's frame (clamped, L2-hot, unused at the end)
            load_frame(nb_ <= last ? nb_ : last, tq);
        }
        if constexpr (T2) { tw2.run(v, buf, tl); __syncthreads(); }
        else lds_fft_tw<float, LR, PP>(v, buf, tl, tw);
        // ---- split channels, magnitudes, peaks ------------------------------------------------------
        // pair q < N/2 ... thread 0 adds bin N/2 (rows are NB floats; stores past a row are dropped by
        // the descriptor's range check, a zero-length descriptor drops them all)
        asm volatile("" : "+v"(tq));
        tq &= TPF - 1;
        float bl = -1.f, br = -1.f;
        int kl = 0x7fffffff, kr = 0x7fffffff;
        const bool st_ok = wmag && valid;
        float* rowl = a.mag + f * NB;
        float* rowr = a.mag + (a.nframes + f) * NB;
        const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(
            FPW == 1 && st_ok ? rowl : nullptr, 0, FPW == 1 && st_ok ? NB * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(
            FPW == 1 && st_ok ? rowr : nullptr, 0, FPW == 1 && st_ok ? NB * 4 : 0, 0x00020000);
        // The dB conversion is a uniform run-time choice: two instantiations of the epilogue behind one scalar
        // branch (as a select, every output would pay the logarithm whether it is wanted or not).
        float bl2 = -1.f, br2 = -1.f;                   // T2: second pair of trackers, bins visited in DEscending order
        int kl2 = 0x7fffffff, kr2 = 0x7fffffff;
        auto epilogue = [&](auto DBT) {
            constexpr bool DB = decltype(DBT)::value;
            auto emit = [&](int q, int voff, int soff, float2 zq, float2 zm, auto DESC) {
                const float2 L = mk<float>(zq.x + zm.x, zq.y - zm.y);          // the 1/2 is in the window
                const float2 Rr = mk<float>(zq.y + zm.y, zm.x - zq.x);
                const float mL = mag_sqrt(L.x * L.x + L.y * L.y), mR = mag_sqrt(Rr.x * Rr.x + Rr.y * Rr.y);
                const float oL = DB ? 20.0f * log10f(fmaxf(mL, SA_DB_FLOOR_F32)) : mL;
                const float oR = DB ? 20.0f * log10f(fmaxf(mR, SA_DB_FLOOR_F32)) : mR;
                if constexpr (FPW == 1) {
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(oL), rl, voff, soff, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(oR), rr, voff, soff, 0);
                } else {
                    if (st_ok) { rowl[q] = oL; rowr[q] = oR; }
                }
                const bool ok = !(excl && (q == 0 || q == N / 2));
                if constexpr (decltype(DESC)::value) {                   // q descending: the lowest of equal maxima with '>='
                    const bool gl = ok & (mL >= bl2), gr = ok & (mR >= br2);
                    bl2 = gl ? mL : bl2; kl2 = gl ? q : kl2;
                    br2 = gr ? mR : br2; kr2 = gr ? q : kr2;
                } else {
                    const bool gl = ok & (mL > bl), gr = ok & (mR > br);     // q ascending per thread: first max kept with '>'
                    bl = gl ? mL : bl; kl = gl ? q : kl;
                    br = gr ? mR : br; kr = gr ? q : kr;
                }
            };
            if constexpr (T2) {
                // Legs of the final radix-2 butterflies: Z[q] = E[q] + O[q], Z[q + N/2] = E[q] - O[q] (q < N/2).  A thread
                // takes butterflies q = tq + TPF u < N/4 and their mirrors N/2 - q: bin q pairs Z[q] with
                // Z[N - q] = E[N/2 - q] - O[N/2 - q], bin N/2 - q pairs Z[N/2 - q] = E + O (mirror) with Z[N/2 + q] = E[q] - O[q].
                // Mirror address as below: pad(TPF - tq) + a constant, lane 0 a constant of its own; lane 0's q = 0 is its
                // own mirror and yields bins 0 (Z[0], Z[0]) and N/2 (Z[N/2], Z[N/2]).
                constexpr int H = N / 2, PLANE = FftTail2_8192::PLANE;
                const float2* pe = buf + lds_pad<S>(tq);
                const int mlow = lds_pad<S>(TPF - tq);
#pragma unroll
                for (int u = 0; u < (H / 2) / TPF; ++u) {
                    const int q = tq + TPF * u;
                    const int moff = lds_pad_c<S>(H - TPF * (u + 1));
                    const int m0 = lds_pad_c<S>((H - TPF * u) & (H - 1)) - moff;
                    const int ml = (tq == 0 ? m0 : mlow) + moff;
                    const float2 eq = pe[lds_pad_c<S>(TPF * u)], oq = pe[PLANE + lds_pad_c<S>(TPF * u)];
                    const float2 em = buf[ml], om = buf[PLANE + ml];
                    float2 zq = cadd(eq, oq), zm = csub(em, om), yq = cadd(em, om), ym = csub(eq, oq);
                    if (u == 0) {                                   // q = 0: (zq, zm, yq, ym) = (S, D, S, D) -> (S, S), (D, D)
                        const bool z0 = tq == 0;
                        const float2 t = zm;
                        zm.x = z0 ? yq.x : zm.x; zm.y = z0 ? yq.y : zm.y;
                        yq.x = z0 ? t.x : yq.x;  yq.y = z0 ? t.y : yq.y;
                    }
                    emit(q, tq * 4, TPF * u * 4, zq, zm, std::false_type{});
                    emit(H - q, (TPF - tq) * 4, (H - TPF * (u + 1)) * 4, yq, ym, std::true_type{});
                }
                if (tq == 0) {                                      // bin N/4: butterfly N/4 is its own mirror
                    const float2 e = buf[lds_pad_c<S>(H / 2)], o = buf[PLANE + lds_pad_c<S>(H / 2)];
                    emit(H / 2, 0, (H / 2) * 4, cadd(e, o), csub(e, o), std::true_type{});
                }
                // the ascending trackers hold bins < N/4, the descending ones bins >= N/4
                const bool tl_ = bl2 > bl, tr_ = br2 > br;
                bl = tl_ ? bl2 : bl; kl = tl_ ? kl2 : kl;
                br = tr_ ? br2 : br; kr = tr_ ? kr2 : kr;
                return;
            }
            // Padded LDS addresses are additive over disjoint bit fields (lds_fft.h): Z[q], q = tq + TPF u, sits at
            // pad(tq) + pad(TPF u); its mirror N - q = (N - TPF (u+1)) + (TPF - tq) at pad(TPF - tq) + a constant for
            // every lane but tq = 0 (whose mirror N - TPF u is a constant itself) — one select instead of the
            // subtract / mask / shift / add chain per access.
            const float2* pq = buf + lds_pad<S>(tq);
            const int mlow = lds_pad<S>(TPF - tq);
#pragma unroll
            for (int u = 0; u < (N / 2) / TPF; ++u) {
                const int q = tq + TPF * u;
                const int moff = lds_pad_c<S>(N - TPF * (u + 1));
                const int m0 = lds_pad_c<S>((N - TPF * u) & (N - 1)) - moff;       // what tq = 0 needs in place of mlow
                const int ml = tq == 0 ? m0 : mlow;
                emit(q, tq * 4, TPF * u * 4, pq[lds_pad_c<S>(TPF * u)], buf[ml + moff], std::false_type{});
            }
            if (tq == 0) {
                const float2 z = buf[lds_pad_c<S>(N / 2)];
                emit(N / 2, 0, (N / 2) * 4, z, z, std::false_type{});
            }
        };
        if (wdb) epilogue(std::true_type{}); else epilogue(std::false_type{});
        if constexpr (TPF >= 64) {
            // whole waves per frame: wave maximum and the lowest bin among the lanes that hold it, on the DPP path
            // (magnitudes are >= 0 and never NaN here: int16 samples times a finite window)
            const float wl = wave_max_f32(bl), wr = wave_max_f32(br);
            kl = wave_min_u32(bl == wl ? (unsigned)kl : 0x7fffffffu);
            kr = wave_min_u32(br == wr ? (unsigned)kr : 0x7fffffffu);
            bl = wl; br = wr;
        } else {
            constexpr int WRED = TPF;
#pragma unroll
            for (int off = WRED / 2; off > 0; off >>= 1) {
                const float ol = __shfl_xor(bl, off), orr = __shfl_xor(br, off);
                const int okl = __shfl_xor(kl, off), okr = __shfl_xor(kr, off);
                const bool tl_ = (ol > bl) | ((ol == bl) & (okl < kl)), tr_ = (orr > br) | ((orr == br) & (okr < kr));
                bl = tl_ ? ol : bl; kl = tl_ ? okl : kl;
                br = tr_ ? orr : br; kr = tr_ ? okr : kr;
            }
        }
        if constexpr (TPF > 64) {
            constexpr int WPF = TPF / 64;           // waves per frame
            const int wave = threadIdx.x >> 6;
            if ((threadIdx.x & 63) == 0) { red_m[2 * wave] = bl; red_m[2 * wave + 1] = br; red_k[2 * wave] = kl; red_k[2 * wave + 1] = kr; }
            __syncthreads();
            if (tl == 0)
                for (int w = g * WPF + 1; w < (g + 1) * WPF; ++w) {
                    const float ol = red_m[2 * w], orr = red_m[2 * w + 1];
                    const int okl = red_k[2 * w], okr = red_k[2 * w + 1];
                    if (ol > bl || (ol == bl && okl < kl)) { bl = ol; kl = okl; }
                    if (orr > br || (orr == br && okr < kr)) { br = orr; kr = okr; }
                }
        }
        if (tl == 0 && valid) {
            a.peak_bin[f] = kl; a.peak_mag[f] = bl;
            a.peak_bin[a.nframes + f] = kr; a.peak_mag[a.nframes + f] = br;
        }
        __syncthreads();
    }
}

__global__ void deinterleave_pcm16(const int16_t* __restrict__ pcm, float* __restrict__ out, int64_t n, int channels, int ch) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (float)pcm[i * channels + ch] * (1.0f / 32768.0f);
}

struct TwF32Ctx { int64_t n; };
static void fill_tw32(void* host, void* ctx) {
    const int64_t n = static_cast<TwF32Ctx*>(ctx)->n;
    std::vector<double> t;
    host_twiddle(n, n, 1, t);
    float* o = static_cast<float*>(host);
    for (size_t i = 0; i < t.size(); ++i) o[i] = (float)t[i];
}
struct WinF32Ctx { int w; double p; int64_t n; };
static void fill_win32(void* host, void* ctx) {
    WinF32Ctx* c = static_cast<WinF32Ctx*>(ctx);
    std::vector<double> w;
    host_window(c->w, c->p, (int)c->n, w);
    float* o = static_cast<float*>(host);
    for (size_t i = 0; i < w.size(); ++i) o[i] = (float)w[i];
}

template <int LR, int PP>
static int launch_stereo_pp(const StftArgs& a, hipStream_t st) {
    constexpr int FPW = stft_fpw<LR, PP>();
    constexpr int NW = (FftPlan<LR, PP>::TPF * FPW + 63) / 64;
    constexpr int lds = FPW * lds_fft_elems<float, LR>() * 8 + NW * 16 + 16 +
                        (LR == 13 && PP == 16 && FPW == 1 ? FftTail2_8192::TAB * 8 : 0);   // + the second pass' twiddle table
    auto kern = stft_stereo_kernel<LR, PP>;
    static thread_local uint64_t attr_devs = 0;            // bit d: attribute set for device d (it is per device)
    SA_TRY(allow_dynamic_lds(reinterpret_cast<const void*>(kern), (int)lds, &attr_devs));
    // a multiple of 8 workgroups (the kernel deals each residue class mod 8 its own contiguous range of frames)
    const int64_t cap = (grid_cap(64, "STFT") + 7) / 8 * 8, need = ((a.nframes + FPW - 1) / FPW + 7) / 8 * 8;
    hipLaunchKernelGGL(kern, dim3((unsigned)(need < cap ? need : cap)), dim3(FftPlan<LR, PP>::TPF * FPW), lds, st, a);
    SA_HIP(hipGetLastError());
    return SA_OK;
}

template <class M, int TEAMS>
static int launch_stft8192(const StftArgs& a, hipStream_t st) {
    static thread_local uint64_t attr_devs = 0;
    SA_TRY(allow_dynamic_lds(reinterpret_cast<const void*>(stft8192_kernel<M, TEAMS>), Stft32::LDS_BYTES * TEAMS, &attr_devs));
    const int64_t cap = (grid_cap(64 / TEAMS, "STFT") + 7) / 8 * 8, need = ((a.nframes + TEAMS - 1) / TEAMS + 7) / 8 * 8;
    hipLaunchKernelGGL((stft8192_kernel<M, TEAMS>), dim3((unsigned)(need < cap ? need : cap)), dim3(Stft32::T * TEAMS), Stft32::LDS_BYTES * TEAMS, st, a);
    SA_HIP(hipGetLastError());
    return SA_OK;
}
template <class M>
static int launch_stft8192q(const StftArgs& a, hipStream_t st) {
    static thread_local uint64_t attr_devs = 0;
    const int lds = Stft32Q::LDS_BYTES;
    SA_TRY(allow_dynamic_lds(reinterpret_cast<const void*>(stft8192q_kernel<M>), lds, &attr_devs));
    const int64_t cap = (grid_cap(64, "STFT") + 7) / 8 * 8, need = (a.nframes + 7) / 8 * 8;
    hipLaunchKernelGGL((stft8192q_kernel<M>), dim3((unsigned)(need < cap ? need : cap)), dim3(Stft32Q::T), lds, st, a);
    SA_HIP(hipGetLastError());
    return SA_OK;
}
// the window (or ones) times the 1/2 of the channel split and the 1/32768 of the PCM scale: exact (a power of two)
static void fill_win32q(void* host, void* ctx) {
    WinF32Ctx* c = static_cast<WinF32Ctx*>(ctx);
    std::vector<double> w;
    if (c->w == SA_WIN_NONE) w.assign((size_t)c->n, 1.0); else host_window(c->w, c->p, (int)c->n, w);
    float* o = static_cast<float*>(host);
    for (size_t i = 0; i < w.size(); ++i) o[i] = (float)w[i] * (0.5f / 32768.0f);
}
// SA_STFT_V (A/B aid) for nfft 8192: 0 = the 16-point-per-thread kernel with the folded radix-2 tail (round 2),
// 1 = 32 points per thread, plain fp32 arithmetic (default), 2 = the same with packed fp32 arithmetic, 3 = two lock-stepped
// teams per workgroup, 4 = the 32-point plan as four 35 KiB workgroups per CU (half exchanges, stft8192q.h)
static int stft_version() {
    static const int v = getenv("SA_STFT_V") ? atoi(getenv("SA_STFT_V")) : 1;
    return v;
}

// SA_STFT_PP (tuning aid): points per thread of the stereo kernel (8 or 16)
static int stft_pp_override() {
    static const int v = getenv("SA_STFT_PP") ? atoi(getenv("SA_STFT_PP")) : 0;
    return v;
}

template <int LR>
static int launch_stereo(const StftArgs& a, hipStream_t st) {
    const int pp = stft_pp_override() ? stft_pp_override() : STFT_PP(LR);
    if constexpr (LR >= 8) {
        if (pp == 16) return launch_stereo_pp<LR, 16>(a, st);
    }
    return launch_stereo_pp<LR, 8>(a, st);
}

int stft_pcm16_dev(const int16_t* pcm, int64_t n_samples, int32_t channels, int32_t nfft, int32_t hop, int32_t window,
                   double win_param, int32_t flags, float* mag, int32_t* peak_bin, float* peak_mag, float* scratch,
                   hipStream_t st) {
    SA_ENTER();
    int lr = 0;
    while ((1 << lr) < nfft) ++lr;
    if (nfft < 16 || (1 << lr) != nfft || nfft > (1 << 24)) { set_error("stft: nfft=%d must be a power of two in [16, 2^24]", nfft); return SA_EUNSUP; }
    if (hop < 1) { set_error("stft: hop must be >= 1"); return SA_EINVAL; }
    if (channels < 1) { set_error("stft: channels must be >= 1"); return SA_EINVAL; }
    if (window < SA_WIN_NONE || window > SA_WIN_LAST) { set_error("unknown window id %d", window); return SA_EINVAL; }
    if (flags & ~7) { set_error("stft: unknown flag bits"); return SA_EINVAL; }
    if (n_samples < nfft) return SA_OK;                 // no complete frame
    const int64_t nframes = (n_samples - nfft) / hop + 1;
    if (!pcm || !peak_bin || !peak_mag || ((flags & SA_F_WRITE_MAG) && !mag)) { set_error("stft: NULL buffer"); return SA_EINVAL; }
    if (channels == 2 && nfft <= 8192) {
        StftArgs a{};
        a.pcm = pcm; a.nframes = nframes; a.hop = hop; a.flags = flags; a.mag = mag; a.peak_bin = peak_bin; a.peak_mag = peak_mag;
        TwF32Ctx tc{nfft};
        a.twR = static_cast<const float2*>(get_cached_table("tw32:" + std::to_string(nfft), (size_t)nfft * 8, fill_tw32, &tc));
        if (!a.twR) return SA_EHIP;
        if (window != SA_WIN_NONE) {
            WinF32Ctx wc{window, win_param, nfft};
            char key[96];
            snprintf(key, sizeof(key), "win32:%d:%.17g:%d", window, window_has_param(window) && win_param > 0 ? win_param : 0.0, nfft);
            a.win = static_cast<const float*>(get_cached_table(key, (size_t)nfft * 4, fill_win32, &wc));
            if (!a.win) return SA_EHIP;
        }
        switch (lr) {
            case 4: return launch_stereo<4>(a, st);
            case 5: return launch_stereo<5>(a, st);
            case 6: return launch_stereo<6>(a, st);
            case 7: return launch_stereo<7>(a, st);
            case 8: return launch_stereo<8>(a, st);
            case 9: return launch_stereo<9>(a, st);
            case 10: return launch_stereo<10>(a, st);
            case 11: return launch_stereo<11>(a, st);
            case 12: return launch_stereo<12>(a, st);
            case 13:
                // SA_STFT_V=4: four small workgroups per CU (stft8192q.h).  Measured on C3's recording: 6.2 against 7.0 ms
                // without the spectrum stores, 8.9 against 7.9 ms with them — and its peak magnitudes differ from the default
                // kernel's in the last place (other FMA contractions), so it is not chosen by flags: opt-in only.
                if (stft_version() == 4) {
                    WinF32Ctx wc{window, win_param, nfft};
                    char key[96];
                    snprintf(key, sizeof(key), "win32q:%d:%.17g:%d", window, window_has_param(window) && win_param > 0 ? win_param : 0.0, nfft);
                    a.winq = static_cast<const float*>(get_cached_table(key, (size_t)nfft * 4, fill_win32q, &wc));
                    if (!a.winq) return SA_EHIP;
                    return launch_stft8192q<MathPlain>(a, st);
                }
                if (stft_version() == 3) return launch_stft8192<MathPlain, 2>(a, st);   // two lock-stepped teams per workgroup
                if (stft_version() == 2) return launch_stft8192<MathPk, 1>(a, st);
                if (stft_version() == 1) return launch_stft8192<MathPlain, 1>(a, st);
                return launch_stereo<13>(a, st);
        }
    }
    // generic channel count: de-interleave + convert, then strided batched rfft
    // (an odd hop starts every other frame on an odd sample: rfft_peak_strided_dev then loads element-wise)
    if (!scratch) { set_error("stft: this channel layout needs a scratch buffer of n_samples floats"); return SA_EINVAL; }
    const int64_t nb = nfft / 2 + 1;
    for (int ch = 0; ch < channels; ++ch) {
        hipLaunchKernelGGL(deinterleave_pcm16, dim3(num_cus() * 8), dim3(256), 0, st, pcm, scratch, n_samples, channels, ch);
        SA_HIP(hipGetLastError());
        SA_TRY(rfft_peak_strided_dev<float>(scratch, nframes, hop, nfft, window, win_param, flags,
                                            mag ? mag + (int64_t)ch * nframes * nb : nullptr, peak_bin + (int64_t)ch * nframes,
                                            peak_mag + (int64_t)ch * nframes, st));
    }
    return SA_OK;
}

}  // namespace sa

namespace sa {
// SURVEY §8(d) C3 synthetic recording (include/specengine.h: sa_fill_pcm16_stereo_dev).  Phases are reduced to a fraction of
// a cycle in fp64 before the sine (t reaches 2.4e4 s at 2^30 samples); the noise is counter-based, so any span can be
// regenerated independently.
__device__ __forceinline__ uint64_t pcm_mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
__global__ void __launch_bounds__(256) fill_pcm16_stereo_kernel(int16_t* __restrict__ pcm, int64_t n_samples, double fs, uint64_t seed) {
    constexpr int64_t SWEEP = 1 << 20;
    const double f0 = 100.0, f1 = 20000.0;
    for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < n_samples; n += (int64_t)gridDim.x * 256) {
        const double dn = (double)n, dm = (double)(n & (SWEEP - 1));
        const double c1 = 440.0 * dn / fs, c2 = 3000.0 * dn / fs;
        const double c3 = (f0 * dm + (f1 - f0) * dm * dm / (2.0 * (double)SWEEP)) / fs;
        const double l = 0.35 * sinpi(2.0 * (c1 - floor(c1))) + 0.25 * sinpi(2.0 * (c2 - floor(c2))) + 0.25 * sinpi(2.0 * (c3 - floor(c3)));
        double r = 0.0;
#pragma unroll
        for (int o = 0; o < 12; ++o) {
            const uint64_t h = pcm_mix64(seed ^ ((uint64_t)o << 56) ^ (uint64_t)(n >> o));
            r += (double)(h >> 11) * (2.0 / 9007199254740992.0) - 1.0;
        }
        r *= 0.08;
        const double ls = fmin(fmax(rint(l * 32767.0), -32768.0), 32767.0), rs = fmin(fmax(rint(r * 32767.0), -32768.0), 32767.0);
        const short2 v = make_short2((short)ls, (short)rs);
        reinterpret_cast<short2*>(pcm)[n] = v;
    }
}
}  // namespace sa

extern "C" {

int sa_fill_pcm16_stereo_dev(int16_t* interleaved, int64_t n_samples, double fs, uint64_t seed, void* stream) {
    SA_ENTER();
    if (!interleaved || n_samples < 0 || !(fs > 0) || ((uintptr_t)interleaved & 3)) { sa::set_error("sa_fill_pcm16_stereo_dev: bad arguments"); return SA_EINVAL; }
    if (n_samples == 0) return SA_OK;
    hipStream_t st = stream ? (hipStream_t)stream : sa::thread_stream();
    const int64_t nb = (n_samples + 255) / 256, cap = (int64_t)sa::num_cus() * 16;
    hipLaunchKernelGGL(sa::fill_pcm16_stereo_kernel, dim3((unsigned)(nb < cap ? nb : cap)), dim3(256), 0, st, interleaved, n_samples, fs, seed);
    SA_HIP(hipGetLastError());
    return SA_OK;
}

int64_t sa_stft_num_frames(int64_t n_samples, int32_t nfft, int32_t hop) {
    if (hop < 1 || nfft < 1 || n_samples < nfft) return 0;
    return (n_samples - nfft) / hop + 1;
}

int sa_stft_pcm16_dev(const int16_t* interleaved, int64_t n_samples, int32_t channels, int32_t nfft, int32_t hop,
                      int32_t window, double win_param, int32_t flags, float* mag, int32_t* peak_bin, float* peak_mag,
                      float* scratch, void* stream) {
    SA_ENTER();
    hipStream_t st = stream ? (hipStream_t)stream : sa::thread_stream();
    return sa::stft_pcm16_dev(interleaved, n_samples, channels, nfft, hop, window, win_param, flags, mag, peak_bin, peak_mag,
                              scratch, st);
}

}  // extern "C"

// Frames [f_lo, f_hi) of the recording on the calling thread's device; outputs land at their place among all
// `nframes` frames of each channel.  The samples a frame range needs (its hop-sized steps plus the nfft - hop halo the
// next range also reads) come from the host source — shards never exchange samples.
static int stft_host_range(const int16_t* interleaved, int32_t channels, int32_t nfft, int32_t hop, int32_t window, double win_param,
                           int32_t flags, float* mag, int32_t* peak_bin, float* peak_mag, int64_t nframes, int64_t f_lo, int64_t f_hi) {
    using namespace sa;
    hipStream_t st = thread_stream();
    const int64_t nb = nfft / 2 + 1;
    // process in segments of frames so the staging buffers stay bounded (<= ~1 GiB of spectra)
    int64_t seg = (int64_t)(1ull << 30) / (nb * 4 * channels);
    if (seg < 1) seg = 1;
    if (seg > f_hi - f_lo) seg = f_hi - f_lo;
    const int64_t seg_samples = (seg - 1) * hop + nfft;
    const bool wmag = flags & SA_F_WRITE_MAG;
    // staging from the thread's arena (small calls reuse its cached block; large ones fall back to one allocation each)
    Temp tmp;
    int16_t* dpcm = static_cast<int16_t*>(tmp.dev((size_t)seg_samples * channels * 2));
    float* dmag = wmag ? static_cast<float*>(tmp.dev((size_t)seg * nb * channels * 4)) : nullptr;
    float* dpm = static_cast<float*>(tmp.dev((size_t)seg * channels * 4));
    int32_t* dpb = static_cast<int32_t*>(tmp.dev((size_t)seg * channels * 4));
    float* dscr = !(channels == 2 && nfft <= 8192) ? static_cast<float*>(tmp.dev((size_t)seg_samples * 4)) : nullptr;
    int rc = SA_OK;
    hipError_t e = hipSuccess;
    if (!dpcm || (wmag && !dmag) || !dpm || !dpb || (!(channels == 2 && nfft <= 8192) && !dscr)) rc = SA_ENOMEM;
    for (int64_t f0 = f_lo; rc == SA_OK && f0 < f_hi; f0 += seg) {
        const int64_t nf = f_hi - f0 < seg ? f_hi - f0 : seg;
        const int64_t ns = (nf - 1) * hop + nfft;
        e = hipMemcpyAsync(dpcm, interleaved + f0 * hop * channels, (size_t)ns * channels * 2, hipMemcpyHostToDevice, st);
        if (e != hipSuccess) { set_error("stft: H2D failed: %s", hipGetErrorString(e)); rc = SA_EHIP; break; }
        rc = stft_pcm16_dev(dpcm, ns, channels, nfft, hop, window, win_param, flags, dmag, dpb, dpm, dscr, st);
        if (rc != SA_OK) break;
        for (int ch = 0; ch < channels && e == hipSuccess; ++ch) {
            if (wmag) e = hipMemcpyAsync(mag + ((int64_t)ch * nframes + f0) * nb, dmag + (int64_t)ch * nf * nb, (size_t)nf * nb * 4, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipMemcpyAsync(peak_bin + (int64_t)ch * nframes + f0, dpb + (int64_t)ch * nf, (size_t)nf * 4, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipMemcpyAsync(peak_mag + (int64_t)ch * nframes + f0, dpm + (int64_t)ch * nf, (size_t)nf * 4, hipMemcpyDeviceToHost, st);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) { set_error("stft: D2H failed: %s", hipGetErrorString(e)); rc = SA_EHIP; }
    }
    if (rc != SA_OK) (void)hipStreamSynchronize(st);      // nothing may still be using the arena when it is handed back
    return rc;            // every segment ended with a stream synchronise: the arena blocks are idle
}

extern "C" int sa_stft_pcm16(const int16_t* interleaved, int64_t n_samples, int32_t channels, int32_t nfft, int32_t hop, int32_t window,
                             double win_param, int32_t flags, float* mag, int32_t* peak_bin, float* peak_mag) {
    using namespace sa;
    SA_ENTER();
    if (channels < 1 || nfft < 1 || hop < 1) { set_error("stft: bad channels/nfft/hop"); return SA_EINVAL; }
    const int64_t nframes = sa_stft_num_frames(n_samples, nfft, hop);
    if (nframes == 0) return SA_OK;
    if (!interleaved) { set_error("stft: NULL input"); return SA_EINVAL; }
    if (!peak_bin || !peak_mag || ((flags & SA_F_WRITE_MAG) && !mag)) { set_error("stft: NULL output buffer"); return SA_EINVAL; }
    if (shard_count() >= 2 && nframes >= 2)               // frame ranges fan out over the shard devices (sa_shard_devices)
        return run_sharded(nframes, [&](int64_t lo, int64_t hi) {
            return stft_host_range(interleaved, channels, nfft, hop, window, win_param, flags, mag, peak_bin, peak_mag, nframes, lo, hi);
        });
    return stft_host_range(interleaved, channels, nfft, hop, window, win_param, flags, mag, peak_bin, peak_mag, nframes, 0, nframes);
}
