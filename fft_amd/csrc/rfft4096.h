// rfft4096.h — tuned fused  window -> rfft(4096) -> |Y| (-> dB) -> argmax  kernel, fp32, gfx950.
//
// Replaces per frame np.abs(fft(y*w)) + np.argmax (reference main.py:475, 479, 1029, 1627) for the
// headline configuration (BASELINE config 2: 2^20 frames x 4096 points).
//
// Design (wavefront-native, no workgroup barriers in the frame loop):
//   * ONE 64-lane wavefront owns ONE frame.  The 4096 reals are read with 16 coalesced
//     global_load_dwordx4 per lane (1 KiB per wave-instruction) straight into the registers that
//     the first radix-16 butterfly consumes: lane t, load j holds z[128 j + 2t], z[128 j + 2t + 1]
//     of the packed complex sequence z[c] = x[2c] + i x[2c+1] (M = 2048 points).
//   * M = 16 x 16 x 8 decimation-in-frequency passes run in registers (32 complex points per lane);
//     the two transposes between passes go through a PRIVATE 8.5 KiB LDS slab per wave, each done
//     in two halves so the slab is half a frame (in-order DS execution inside a wave replaces
//     barriers).  Exchange 1 is padded (row stride 136), exchange 2 XOR-swizzled: both are
//     bank-conflict free for ds_write_b128/ds_read_b64 resp. ds_write_b64/ds_read_b128.
//   * Twiddles are not table-looked-up per frame: each lane keeps 6 base twiddles in registers and
//     derives the powers with a depth-4 product tree; the window (pre-scaled by 1/2 for the
//     Hermitian pass) sits in a 16 KiB LDS table shared by the 16 waves of the workgroup.
//   * The last pass is assigned so that a lane owns both Z[k] and Z[2048-k]: the real-FFT
//     post-pass, magnitudes and the per-lane peak need no further exchange; stores are 256-byte
//     coalesced rows.  Lane 0 owns the two self-paired butterflies (k = 0, 128 mod 256) and DC/Nyquist.
//   * Workgroups are persistent (one 1024-thread workgroup per CU), waves stride over frames.
// HBM traffic per frame = 16 KiB read + 8196 B (+8 B) written: the algorithmic minimum.
#pragma once
#include "fft_core.h"

namespace sa {

constexpr int R4K_WAVES = 12;                       // waves per workgroup
constexpr int R4K_SLAB_BYTES = 8 * 136 * 8;         // 8704 B private exchange slab per wave
constexpr int R4K_LDS_BYTES = 4096 * 4 + R4K_WAVES * R4K_SLAB_BYTES;

__device__ __forceinline__ void wave_lds_fence() {
    // Orders this wave's LDS accesses for the compiler; the hardware executes one wave's DS
    // instructions in order, so no s_barrier is needed for a wave-private slab.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// p[k] = w^k for k = 1..15 (p[0] unused) with a depth-4 product tree.
__device__ __forceinline__ void twiddle_powers16(float2 w, float2* p) {
    p[1] = w;
    p[2] = cmul(w, w);
    p[3] = cmul(p[2], w);
    p[4] = cmul(p[2], p[2]);
    p[5] = cmul(p[4], w);
    p[6] = cmul(p[4], p[2]);
    p[7] = cmul(p[4], p[3]);
    p[8] = cmul(p[4], p[4]);
    p[9] = cmul(p[8], w);
    p[10] = cmul(p[8], p[2]);
    p[11] = cmul(p[8], p[3]);
    p[12] = cmul(p[8], p[4]);
    p[13] = cmul(p[8], p[5]);
    p[14] = cmul(p[8], p[6]);
    p[15] = cmul(p[8], p[7]);
}

// multiply by w16^i (i = 0..3), compile-time i
template <int I> __device__ __forceinline__ float2 mul_w16(float2 a) {
    constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
    if constexpr (I == 0) return a;
    else if constexpr (I == 1) return cmul(a, mk<float>(c1, -s1));
    else if constexpr (I == 2) return mk<float>((a.x + a.y) * h, (a.y - a.x) * h);
    else return cmul(a, mk<float>(s1, -c1));
}

__device__ __forceinline__ float fast_sqrt(float v) { return __builtin_amdgcn_sqrtf(v); }

// One Hermitian pair: zq = Z'[k], zm = Z'[2048-k] (Z' = Z/2), tw = w_4096^k.
// Returns |X[k]| in ma and |X[2048-k]| in mb.
__device__ __forceinline__ void herm_pair(float2 zq, float2 zm, float2 tw, float& ma, float& mb) {
    const float2 e = mk<float>(zq.x + zm.x, zq.y - zm.y);
    const float2 d = mk<float>(zq.x - zm.x, zq.y + zm.y);
    const float2 o = mk<float>(d.y, -d.x);
    const float2 wo = cmul(o, tw);
    const float2 xa = cadd(e, wo), xb = csub(e, wo);
    ma = fast_sqrt(xa.x * xa.x + xa.y * xa.y);
    mb = fast_sqrt(xb.x * xb.x + xb.y * xb.y);
}

// MODE: 0 = peak only, 1 = write |Y|, 2 = write dB
template <int MODE>
__global__ void __launch_bounds__(R4K_WAVES * 64)
rfft4096_wave_kernel(const float* __restrict__ x, int64_t batch, const float* __restrict__ win,
                     const float2* __restrict__ twM, const float2* __restrict__ twN, int excl_dc_nyq,
                     float* __restrict__ mag, int32_t* __restrict__ peak_bin, float* __restrict__ peak_mag,
                     float db_floor) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float4* wtab4 = reinterpret_cast<float4*>(smem_raw);                 // 1024 float4: 0.5 * window
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    char* slab = smem_raw + 4096 * 4 + wave * R4K_SLAB_BYTES;
    float4* slab4 = reinterpret_cast<float4*>(slab);
    float2* slab2 = reinterpret_cast<float2*>(slab);

    for (int i = threadIdx.x; i < 1024; i += R4K_WAVES * 64) {
        float4 w = win ? reinterpret_cast<const float4*>(win)[i] : make_float4(1.f, 1.f, 1.f, 1.f);
        w.x *= 0.5f; w.y *= 0.5f; w.z *= 0.5f; w.w *= 0.5f;
        wtab4[i] = w;
    }
    __syncthreads();

    // ---- per-lane constants ------------------------------------------------------------------
    const float2 w1a = twM[2 * lane], w1b = twM[2 * lane + 1];          // w_2048^{2t}, w_2048^{2t+1}
    const float2 w2b = twM[16 * (lane & 7)];                            // w_128^{n3}
    const bool special = (lane == 0);
    const int kA = special ? 128 : lane;                                 // slot-1 pair index base, i < 4
    const int kB = special ? -768 : lane;                                // slot-1 pair index base, i >= 4
    const int kC = 64 + lane;                                            // slot-2 pair index base
    const float2 wA = twN[kA];
    const float2 wB = special ? cconj(twN[768]) : wA;                    // w_4096^{-768}
    const float2 wC = twN[kC];
    // exchange-1 addresses
    const int x1_wr = lane;                                              // float4 index, + 68 * k1'
    const int x1_rd = (lane >> 3) * 136 + (lane & 7);                    // float2 index, + 8 * n2
    // exchange-2 write addresses (float2 index), + 128 * k2'
    const int n3 = lane & 7;
    const int k1lo = lane >> 3, k1hi = 8 + (lane >> 3);
    const int x2_wr0 = k1lo * 8 + ((((n3 >> 1) ^ ((k1lo >> 2) & 3)) << 1) | (n3 & 1));
    const int x2_wr1 = k1hi * 8 + ((((n3 >> 1) ^ ((k1hi >> 2) & 3)) << 1) | (n3 & 1));
    // exchange-2 read rows hl' (phase 0: t, t+64; phase 1: (128-t)&127, 64-t)
    const int hl_p1 = lane, hl_p2 = lane + 64, hl_q1 = (128 - lane) & 127, hl_q2 = 64 - lane;

    const int64_t nwaves = (int64_t)gridDim.x * R4K_WAVES;
    for (int64_t frame = (int64_t)blockIdx.x * R4K_WAVES + wave; frame < batch; frame += nwaves) {
        const float4* xf4 = reinterpret_cast<const float4*>(x + frame * 4096);
        float2 va[16], vb[16];
        {
            float4 r[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) r[j] = xf4[j * 64 + lane];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float4 w = wtab4[j * 64 + lane];
                va[j] = mk<float>(r[j].x * w.x, r[j].y * w.y);
                vb[j] = mk<float>(r[j].z * w.z, r[j].w * w.w);
            }
        }
        // ---- pass 1: radix-16 over n1 (stride 128), twiddle w_2048^{id k1} ----------------------
        bfly16<float>(va);
        bfly16<float>(vb);
        {
            float2 p[16];
            twiddle_powers16(w1a, p);
#pragma unroll
            for (int k = 1; k < 16; ++k) va[k] = cmul(va[k], p[k]);
            twiddle_powers16(w1b, p);
#pragma unroll
            for (int k = 1; k < 16; ++k) vb[k] = cmul(vb[k], p[k]);
        }
        // ---- exchange 1 + pass 2 (two halves: k1 < 8, k1 >= 8) ----------------------------------
        float2 b0[16], b1[16];
        float2 p2[16];
        twiddle_powers16(w2b, p2);
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            wave_lds_fence();
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float2 a = va[8 * ph + k], b = vb[8 * ph + k];
                slab4[x1_wr + 68 * k] = make_float4(a.x, a.y, b.x, b.y);
            }
            wave_lds_fence();
            float2* c = ph ? b1 : b0;
#pragma unroll
            for (int n2 = 0; n2 < 16; ++n2) c[n2] = slab2[x1_rd + 8 * n2];
            bfly16<float>(c);
#pragma unroll
            for (int k = 1; k < 16; ++k) c[k] = cmul(c[k], p2[k]);
        }
        // ---- exchange 2 + pass 3 (two halves: k2 < 8 -> hl < 128, k2 >= 8 -> hl >= 128) ---------
        float2 P1[8], P2[8], Q1[8], Q2[8];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            wave_lds_fence();
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                slab2[x2_wr0 + 128 * k] = b0[8 * q + k];
                slab2[x2_wr1 + 128 * k] = b1[8 * q + k];
            }
            wave_lds_fence();
            float2* d1 = q ? Q1 : P1;
            float2* d2 = q ? Q2 : P2;
            const int h1 = q ? hl_q1 : hl_p1, h2 = q ? hl_q2 : hl_p2;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 u = slab4[h1 * 4 + (c ^ ((h1 >> 2) & 3))];
                d1[2 * c] = mk<float>(u.x, u.y); d1[2 * c + 1] = mk<float>(u.z, u.w);
                const float4 v = slab4[h2 * 4 + (c ^ ((h2 >> 2) & 3))];
                d2[2 * c] = mk<float>(v.x, v.y); d2[2 * c + 1] = mk<float>(v.z, v.w);
            }
            bfly8<float>(d1);
            bfly8<float>(d2);
        }
        wave_lds_fence();
        // ---- Hermitian post-pass ---------------------------------------------------------------
        // slot 1: (U[i], V[7-i]); generic lanes U = P1, V = Q1; lane 0 owns butterflies 0 and 128.
        float m[34];
        int kk[34];
        float* mf = (MODE != 0) ? mag + frame * 2049 : nullptr;
        const float2 z0 = P1[0];
        {
            float2 U[8], V[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) U[i] = special ? Q1[i] : P1[i];
            U[4] = special ? P1[1] : P1[4]; U[5] = special ? P1[2] : P1[5];
            U[6] = special ? P1[3] : P1[6]; U[7] = special ? P1[4] : P1[7];
#pragma unroll
            for (int i = 4; i < 8; ++i) V[i] = Q1[i];
            V[3] = special ? P1[7] : Q1[3]; V[2] = special ? P1[6] : Q1[2];
            V[1] = special ? P1[5] : Q1[1]; V[0] = special ? P1[4] : Q1[0];
            const float2 tA0 = wA, tA1 = mul_w16<1>(wA), tA2 = mul_w16<2>(wA), tA3 = mul_w16<3>(wA);
            // i >= 4: w16^i = -i * w16^{i-4}
            const float2 tB0 = mul_mi(wB), tB1 = mul_mi(mul_w16<1>(wB)), tB2 = mul_mi(mul_w16<2>(wB)),
                         tB3 = mul_mi(mul_w16<3>(wB));
            const float2 tw[8] = {tA0, tA1, tA2, tA3, tB0, tB1, tB2, tB3};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                herm_pair(U[i], V[7 - i], tw[i], m[2 * i], m[2 * i + 1]);
                const int k = (i < 4 ? kA : kB) + 256 * i;
                kk[2 * i] = k; kk[2 * i + 1] = 2048 - k;
            }
        }
        {
            const float2 tC0 = wC, tC1 = mul_w16<1>(wC), tC2 = mul_w16<2>(wC), tC3 = mul_w16<3>(wC);
            const float2 tw[8] = {tC0, tC1, tC2, tC3, mul_mi(tC0), mul_mi(tC1), mul_mi(tC2), mul_mi(tC3)};
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                herm_pair(P2[i], Q2[7 - i], tw[i], m[16 + 2 * i], m[16 + 2 * i + 1]);
                const int k = kC + 256 * i;
                kk[16 + 2 * i] = k; kk[16 + 2 * i + 1] = 2048 - k;
            }
        }
        // DC / Nyquist (lane 0 only): X[0] = Re Z0 + Im Z0, X[2048] = Re Z0 - Im Z0, Z0 = 2 z0
        m[32] = fabsf(2.0f * (z0.x + z0.y));
        m[33] = fabsf(2.0f * (z0.x - z0.y));
        kk[32] = 0; kk[33] = 2048;
        if (MODE != 0) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                float v = m[i];
                if (MODE == 2) v = 20.0f * __log10f(fmaxf(v, db_floor));
                mf[kk[i]] = v;
            }
            if (special) {
                float v0 = m[32], v1 = m[33];
                if (MODE == 2) { v0 = 20.0f * __log10f(fmaxf(v0, db_floor)); v1 = 20.0f * __log10f(fmaxf(v1, db_floor)); }
                mf[0] = v0; mf[2048] = v1;
            }
        }
        // ---- dominant peak: max, then lowest bin holding the max (np.argmax tie rule) -------------
        float best = -1.0f;
#pragma unroll
        for (int i = 0; i < 32; ++i) best = fmaxf(best, m[i]);
        if (special && !excl_dc_nyq) best = fmaxf(best, fmaxf(m[32], m[33]));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) best = fmaxf(best, __shfl_xor(best, off));
        int bk = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < 32; ++i) bk = (m[i] == best) ? min(bk, kk[i]) : bk;
        if (special && !excl_dc_nyq) {
            if (m[33] == best) bk = min(bk, 2048);
            if (m[32] == best) bk = 0;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) bk = min(bk, __shfl_xor(bk, off));
        if (special) {
            peak_bin[frame] = bk;
            peak_mag[frame] = best;
        }
    }
}

}  // namespace sa
