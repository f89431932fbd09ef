// stft8192.h — the STFT kernel of BASELINE config 3 (stereo PCM16, nfft = 8192), 32 points per thread (round 3).
//
// One 256-thread workgroup per frame, 8192 = 32 * 16 * 16 as three decimation-in-frequency passes in registers with
// TWO exchanges through LDS (the 16-point-per-thread kernel of stft.hip has three, plus the read of the folded tail):
//     n = 256 n_a + 16 n_b + n_c         k = k_a + 32 k_b + 512 k_c
//     pass 0  thread t = 16 n_b + n_c    32-point transform over n_a,  times w_8192^(t k_a)
//     pass 1  thread (p, n_c)            16-point transforms over n_b of the TWO sub-problems k_a = p and 32 - p
//                                        (p = 0: k_a = 0 and 16), times w_256^(n_c k_b)
//     pass 2  thread (s, p)              16-point transforms over n_c of final butterflies j_F = k_a + 32 k_b and
//                                        j_G = 512 - j_F: bins j_F + 512 k_c and j_G + 512 k_c
// The pairing k_a <-> 32 - k_a, k_b <-> 15 - k_b runs through both exchanges, so that
//   * every LDS access is a 16-byte ds_write_b128 / ds_read_b128 carrying one value of each partner (64 LDS instructions
//     per thread and frame, 4 x 64 KiB through the LDS pipe; before: 96 eight-byte accesses per 16 points, 6 x 64 KiB);
//   * after the last pass a thread holds Z[q] AND Z[8192 - q] for each of its bins: the channel split
//     L^[q] = (Z[q] + conj Z[N-q]) / 2, R^[q] = (Z[q] - conj Z[N-q]) / 2i needs no third exchange;
//   * 4 workgroup barriers per frame instead of 6.
// Layout of both exchanges (16-byte units): pair-row p at p * 257; exchange 1 unit t, exchange 2 unit 16 s + n_c.  The odd
// pitch puts the 16 lanes of a quarter-wave on distinct 16-byte bank groups in pass 2's gather (lanes = p); every other
// access is lane-contiguous.  Two workgroups share a CU (64.3 KiB of LDS each, <= 256 VGPRs at two waves per SIMD).
// Per-thread constants of a launch (window samples, twiddle bases, LDS addresses) stay in registers over the frame loop.
#pragma once
#include "fft_core.h"
#include "fft_pk.h"
#include "lds_fft.h"

// cache policy of the spectrum stores (aux operand of the buffer intrinsics; build-time A/B aid: 2 = nontemporal,
// 1 = sc0, 17 = sc0 sc1).  Measured on C3 (profiles/r03_c3_ab.txt): no policy beats the default.
#ifndef STFT_ST_AUX
#define STFT_ST_AUX 0
#endif

namespace sa {

struct Stft32 {
    static constexpr int N = 8192, T = 256, NB = N / 2 + 1;
    static constexpr int PITCH = 257;                                 // pair-row pitch, 16-byte units
    static constexpr int UNITS = 15 * PITCH + 256;                    // 4111 units = 65 776 bytes
    static constexpr int LDS_BYTES = UNITS * 16 + 64;                 // + per-wave peak records; per team
};

__device__ __forceinline__ constexpr int x32(int k) { return (k & 1) ? 16 + (k >> 1) : (k >> 1); }   // register of X[k]

template <class M, bool DB>
struct Stft32Emit {
    using C = typename M::C;
    float bl = -1.f, br = -1.f;
    int kl = 0x7fffffff, kr = 0x7fffffff;
    // one bin of both channels from Z[q] and Z[N - q]; the 1/2 of the split rides on the window
    __device__ __forceinline__ void emit(int q, int voff, int soff, C zq, C zm, bool count, const __amdgpu_buffer_rsrc_t& rl,
                                         const __amdgpu_buffer_rsrc_t& rr) {
        const C m2 = M::split_mag2(zq, zm);                                  // L = zq + conj zm, R = (zq - conj zm) / i
        const float mL = mag_sqrt(m2.x), mR = mag_sqrt(m2.y);
        const float oL = DB ? 20.0f * log10f(fmaxf(mL, SA_DB_FLOOR_F32)) : mL;
        const float oR = DB ? 20.0f * log10f(fmaxf(mR, SA_DB_FLOOR_F32)) : mR;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(oL), rl, voff, soff, STFT_ST_AUX);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(oR), rr, voff, soff, STFT_ST_AUX);
        const bool gl = count & (mL > bl), gr = count & (mR > br);          // bins arrive ascending: '>' keeps the first maximum
        bl = gl ? mL : bl; kl = gl ? q : kl;
        br = gr ? mR : br; kr = gr ? q : kr;
    }
    __device__ __forceinline__ void merge(const Stft32Emit& o) {            // lowest bin among equal maxima
        const bool tl_ = (o.bl > bl) | ((o.bl == bl) & (o.kl < kl)), tr_ = (o.br > br) | ((o.br == br) & (o.kr < kr));
        bl = tl_ ? o.bl : bl; kl = tl_ ? o.kl : kl;
        br = tr_ ? o.br : br; kr = tr_ ? o.kr : kr;
    }
};

// TEAMS = 2: two 256-thread teams in one workgroup, each with a frame (and an LDS buffer) of its own, running the same
// phases in lock step: the two waves of a SIMD are then in arithmetic at the same time (one workgroup per CU).
template <class M, int TEAMS>
__global__ void __launch_bounds__(256 * TEAMS, 2) stft8192_kernel(StftArgs a) {
    using K = Stft32;
    using C = typename M::C;
    constexpr int N = K::N, NB = K::NB;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int team = TEAMS == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
    float4* buf = reinterpret_cast<float4*>(smem_raw) + team * K::UNITS;
    float* red_m = reinterpret_cast<float*>(reinterpret_cast<float4*>(smem_raw) + TEAMS * K::UNITS) + 16 * team;   // [4 waves][2]
    int* red_k = reinterpret_cast<int*>(red_m + 8);
    const int tid = threadIdx.x & 255;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool wmag = (a.flags & SA_F_WRITE_MAG) && a.mag;
    const bool wdb = a.flags & SA_F_DB;
    const bool excl = a.flags & SA_F_EXCLUDE_DC_NYQ;

    // ---- per-thread constants of the launch ---------------------------------------------------------------------------
    float w[32];                                                          // window of samples 256 n_a + tid, x 1/2 x 1/32768
#pragma unroll
    for (int i = 0; i < 32; ++i) w[i] = (a.win ? a.win[i * 256 + tid] : 1.0f) * (0.5f / 32768.0f);
    // pass 0 twiddles w_8192^(tid k_a), k_a = 8 h + l: bases W_l (l = 1..7) and V_h (h = 1..3)
    C W[8], V[4];
#pragma unroll
    for (int l = 1; l < 8; ++l) { const float2 t = a.twR[tid * l]; W[l] = M::make(t.x, t.y); }
#pragma unroll
    for (int h = 1; h < 4; ++h) { const float2 t = a.twR[tid * 8 * h]; V[h] = M::make(t.x, t.y); }
    // pass 1 twiddles w_256^(n_c k_b), n_c = tid % 16
    C tw1[16];
#pragma unroll
    for (int k = 1; k < 16; ++k) { const float2 t = a.twR[32 * (tid & 15) * k]; tw1[k] = M::make(t.x, t.y); }
    // LDS addresses (16-byte units)
    const int p1 = tid >> 4, c1 = tid & 15;                               // pass 1: pair-row, n_c
    // pass 2: pair-row p2 and slot s2.  The slots are dealt so that a wave holds s2 = {2w, 15 - 2w, 2w + 1, 14 - 2w}: slot s2
    // owns the lower half of the 32-bin block 32 s2 (its F bins) and the upper half of block 480 - 32 s2 (its G bins), so
    // with the F / G roles of the slots s2 >= 8 SWAPPED (the writers of exchange 2 swap the two halves of their units,
    // a compile-time choice per slot) every store instruction of a wave covers 64 ADJACENT bins = 256 contiguous bytes
    // instead of four 64-byte pieces whose line partners arrive from another wave eight stores later.
    const int p2 = tid & 15, g2 = tid >> 4;
    const int s2 = (g2 & 1) ? 15 - (2 * (g2 >> 2) + ((g2 >> 1) & 1)) : 2 * (g2 >> 2) + ((g2 >> 1) & 1);
    const float4* rd1 = buf + p1 * K::PITCH + c1;                         // + 16 n_b
    float4* wr2 = buf + p1 * K::PITCH + c1;                               // + 16 s
    const float4* rd2 = buf + p2 * K::PITCH + 16 * s2;                    // + n_c
    // final butterflies of this thread: F bins jF + 512 k, G bins jG + 512 k
    int jF = p2 + 32 * s2, jG = 512 - jF;
    if (s2 >= 8) { jF = 512 - (p2 + 32 * s2); jG = p2 + 32 * s2; }      // swapped roles (pair-row 0 below keeps its own)
    if (p2 == 0) { jF = s2 <= 7 ? 32 * s2 : 496 - 32 * s2; jG = tid == 0 ? 256 : 512 - jF; }

    const uint32_t* pcm = reinterpret_cast<const uint32_t*>(a.pcm);
    uint32_t raw[32];
    const int64_t last = a.nframes - 1;
    auto load_frame = [&](int64_t f, int lane) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(pcm + f * a.hop), 0, N * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < 32; ++i) raw[i] = __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, i * 1024, 0);
    };
    // frame -> workgroup map as in stft_stereo_kernel: each residue class blockIdx % 8 (one XCD, one L2) walks its own
    // contiguous eighth of the frames, so the 75 % overlap of neighbouring frames is fetched into that L2 once
    const int64_t per_xcd = ((a.nframes + TEAMS - 1) / TEAMS + 7) / 8;    // in units of TEAMS consecutive frames
    const int64_t xcd_first = (int64_t)(blockIdx.x & 7) * per_xcd, slot_step = gridDim.x >> 3, slot0 = blockIdx.x >> 3;
    if (slot0 < per_xcd) { const int64_t f0 = (xcd_first + slot0) * TEAMS + team; load_frame(f0 <= last ? f0 : last, tid); }
    for (int64_t slot = slot0; slot < per_xcd; slot += slot_step) {
        const int64_t base = (xcd_first + slot) * TEAMS + team;
        const bool valid = base <= last;
        const int64_t f = valid ? base : last;
        int tq = tid;                                                     // opaque per frame: nothing is hoisted out of the loop
        asm volatile("" : "+v"(tq));
        C v[32];
#pragma unroll
        for (int i = 0; i < 32; ++i)
            v[i] = M::scale(M::make((float)(int16_t)(raw[i] & 0xFFFF), (float)(int16_t)(raw[i] >> 16)), w[i]);
        {
            const int64_t nxt = base + slot_step * TEAMS;                 // next round's frame (clamped; unused at the end)
            load_frame(nxt <= last ? nxt : last, tq);
        }
        // ---- pass 0: X[k_a] of this thread's column, times w^(tid k_a); pair (k_a, 32 - k_a) -> one 16-byte unit --------
        bfly32m<M>(v);
        {
            C Wl[8], Vh[4];
#pragma unroll
            for (int l = 1; l < 8; ++l) { Wl[l] = W[l]; M::opaque(Wl[l]); }
#pragma unroll
            for (int h = 1; h < 4; ++h) { Vh[h] = V[h]; M::opaque(Vh[h]); }
#pragma unroll
            for (int k = 1; k < 32; ++k) {
                const int l = k & 7, h = k >> 3;
                C t = v[x32(k)];
                if (l) t = M::mul(t, Wl[l]);
                if (h) t = M::mul(t, Vh[h]);
                v[x32(k)] = t;
            }
            float4* wr1 = buf + (tq & 255);
            wr1[0] = make_float4(v[x32(0)].x, v[x32(0)].y, v[x32(16)].x, v[x32(16)].y);
#pragma unroll
            for (int p = 1; p < 16; ++p)
                wr1[p * K::PITCH] = make_float4(v[x32(p)].x, v[x32(p)].y, v[x32(32 - p)].x, v[x32(32 - p)].y);
        }
        __syncthreads();
        // ---- pass 1: A = sub-problem k_a = p (or 0), B = k_a = 32 - p (or 16); 16 points over n_b each -------------------
        C* A = v;
        C* B = v + 16;
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            const float4 r = rd1[16 * n];
            A[n] = M::make(r.x, r.y); B[n] = M::make(r.z, r.w);
        }
        bfly16m<M>(A);
        bfly16m<M>(B);
#pragma unroll
        for (int k = 1; k < 16; ++k) {
            C t = tw1[k];
            M::opaque(t);
            A[k] = M::mul(A[k], t); B[k] = M::mul(B[k], t);
        }
        __syncthreads();                                                  // every thread has its pass-1 inputs: the rows may be rewritten
        // exchange 2: slot s carries (input of F, input of G) of the pass-2 thread (s, p)
        if (wave == 0) {
            // lanes 0..15 hold p = 0 (k_a = 0 and 16, each its own partner): slot 0 = (A0, A8), slots 1..7 = (A[s], A[16-s]),
            // slots 8..15 = (B[15-s], B[s]); the other lanes of the wave write the common pattern (A[s], B[15-s]) — halves
            // swapped for the slots s >= 8, whose readers run with the F / G roles exchanged
            const bool z = tid < 16;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const C f0 = s <= 7 ? A[s] : B[15 - s], g0 = s <= 7 ? B[15 - s] : A[s];
                const C f1 = s <= 7 ? A[s] : B[15 - s];
                const C g1 = s == 0 ? A[8] : s <= 7 ? A[16 - s] : B[s];
                wr2[16 * s] = make_float4(z ? f1.x : f0.x, z ? f1.y : f0.y, z ? g1.x : g0.x, z ? g1.y : g0.y);
            }
        } else {
#pragma unroll
            for (int s = 0; s < 8; ++s) wr2[16 * s] = make_float4(A[s].x, A[s].y, B[15 - s].x, B[15 - s].y);
#pragma unroll
            for (int s = 8; s < 16; ++s) wr2[16 * s] = make_float4(B[15 - s].x, B[15 - s].y, A[s].x, A[s].y);
        }
        __syncthreads();
        // ---- pass 2: F = Z[jF + 512 k], G = Z[jG + 512 k] ------------------------------------------------------------------
        C* F = v;
        C* G = v + 16;
#pragma unroll
        for (int n = 0; n < 16; ++n) {
            const float4 r = rd2[n];
            F[n] = M::make(r.x, r.y); G[n] = M::make(r.z, r.w);
        }
        bfly16m<M>(F);
        bfly16m<M>(G);
        // ---- channel split, |.|, stores, peaks -----------------------------------------------------------------------------
        const bool st_ok = wmag && valid;
        const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc(st_ok ? a.mag + f * NB : nullptr, 0, st_ok ? NB * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(st_ok ? a.mag + (a.nframes + f) * NB : nullptr, 0, st_ok ? NB * 4 : 0, 0x00020000);
        float bl, br;
        int kl, kr;
        auto epilogue = [&](auto DBT, auto W0T) {
            constexpr bool DB = decltype(DBT)::value, W0 = decltype(W0T)::value;   // W0: the wave that holds thread 0
            Stft32Emit<M, DB> e1, e2;
            const bool t0 = W0 && tq == 0;
            const bool c0 = !(excl && t0);                                  // bin 0 (and N/2 below) are thread 0's
            // bins jF + 512 k, k = 0..7 (ascending): Z[q] = F[k], Z[N - q] = G[15 - k]; thread 0 (jF = 0): Z[N - q] = F[16 - k]
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                C zm = G[15 - k];
                if constexpr (W0) { const C alt = F[(16 - k) & 15]; zm = M::make(t0 ? alt.x : zm.x, t0 ? alt.y : zm.y); }
                e1.emit(jF + 512 * k, jF * 4, 2048 * k, F[k], zm, k == 0 ? c0 : true, rl, rr);
            }
            // bins jG + 512 (15 - k), k = 15..8 (ascending): Z[q] = G[15 - k], Z[N - q] = F[k]; thread 0 (jG = 256): Z[N - q] = G[k]
#pragma unroll
            for (int k = 15; k >= 8; --k) {
                C zm = F[k];
                if constexpr (W0) { const C alt = G[k]; zm = M::make(t0 ? alt.x : zm.x, t0 ? alt.y : zm.y); }
                e2.emit(jG + 512 * (15 - k), jG * 4, 2048 * (15 - k), G[15 - k], zm, true, rl, rr);
            }
            if constexpr (W0) {
                if (tq == 0) e1.emit(N / 2, 0, (N / 2) * 4, F[8], F[8], !excl, rl, rr);   // Nyquist: Z[N/2] pairs with itself
            }
            e1.merge(e2);
            bl = e1.bl; br = e1.br; kl = e1.kl; kr = e1.kr;
        };
        if (wave == 0) { if (wdb) epilogue(std::true_type{}, std::true_type{}); else epilogue(std::false_type{}, std::true_type{}); }
        else           { if (wdb) epilogue(std::true_type{}, std::false_type{}); else epilogue(std::false_type{}, std::false_type{}); }
        {
            const float wl = wave_max_f32(bl), wr = wave_max_f32(br);
            kl = wave_min_u32(bl == wl ? (unsigned)kl : 0x7fffffffu);
            kr = wave_min_u32(br == wr ? (unsigned)kr : 0x7fffffffu);
            bl = wl; br = wr;
        }
        if ((tid & 63) == 0) { red_m[2 * wave] = bl; red_m[2 * wave + 1] = br; red_k[2 * wave] = kl; red_k[2 * wave + 1] = kr; }
        __syncthreads();                                                  // also: every thread has its pass-2 inputs (next frame may write)
        if (tid == 0 && valid) {
#pragma unroll
            for (int wv = 1; wv < 4; ++wv) {
                const float ol = red_m[2 * wv], orr = red_m[2 * wv + 1];
                const int okl = red_k[2 * wv], okr = red_k[2 * wv + 1];
                if (ol > bl || (ol == bl && okl < kl)) { bl = ol; kl = okl; }
                if (orr > br || (orr == br && okr < kr)) { br = orr; kr = okr; }
            }
            a.peak_bin[f] = kl; a.peak_mag[f] = bl;
            a.peak_bin[a.nframes + f] = kr; a.peak_mag[a.nframes + f] = br;
        }
    }
}

}  // namespace sa
