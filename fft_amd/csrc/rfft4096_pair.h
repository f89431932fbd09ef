// rfft4096_pair.h — the 4096-point fp32 headline kernel with TWO waves per frame (round 3).
//
// Why: the one-wave-per-frame kernel (rfft4096.h) sits on the ceiling of its access shape.  A compute-free copy that
// moves the same bytes the same way — one wave issues the 16 one-KiB loads of a 16 KiB frame — streams at 5.5-5.8 TB/s,
// and it is the BYTES PER WAVE that set this figure, not the order in which addresses reach memory
// (tools/ubench/stream_pattern4.hip, profiles/r03_stream_pattern4.txt: the same loads issued by a cooperating workgroup
// frame after frame 5.6; in reverse / bit-reversed order 5.6; split in time 5.6-5.7; EIGHT loads per wave 5.9-6.0; a
// barrier between the two waves of a frame costs nothing).  So a frame is shared by two waves here, each loading half.
//
// The decomposition is rfft4096.h's (2048 packed complex points = 16 x 16 x 8, decimation in frequency):
//   * packed index p = 128 j + col; lane t of wave h loads rows j = 8h .. 8h+7 of columns 2t, 2t+1 (8 x 1 KiB);
//   * the first radix-16 butterfly over j is split between the waves: each writes its 8 windowed rows to LDS, reads the
//     partner's (one barrier), and wave 0 takes the even outputs k1 = 2 k1' (radix 8 over a_j + a_{j+8}), wave 1 the odd
//     ones (radix 8 over (a_j - a_{j+8}) w_16^j).  The area a wave read from the partner is its PRIVATE slab from then on
//     (in-order DS inside a wave, no barrier);
//   * every later pass works on one k1 at a time, so the waves are independent: exchange 1 + radix 16 over the columns,
//     exchange 2 + radix 8, each wave on its eight k1 — literally the two halves rfft4096.h runs one after the other;
//   * the bins of wave 0 are the even packed bins Z[2b + 256 k3], those of wave 1 the odd ones; Z[k] and Z[2048 - k] have
//     the same parity, so the Hermitian post-pass stays inside a wave (lane t owns final butterflies b = t and 128 - t
//     resp. 127 - t; lane 0 of wave 0 the two self-paired ones and DC / Nyquist, as in rfft4096.h);
//   * magnitudes wait in registers for the second barrier (both waves are done with their slabs), are staged as ONE
//     interleaved row in LDS, and after the third barrier each wave stores half of the row as 16-byte aligned pieces.
// One-shot workgroups of F frames (2 F waves), the symmetric-window half table of rfft4096.h, nontemporal accesses.
#pragma once
#include "rfft4096.h"

namespace sa {

constexpr int R4P_XAREA_BYTES = 2 * R4K_SLAB_BYTES;          // 17 408 B per frame: 16 rows of 68 float4 = two slabs
constexpr int r4p_lds_bytes(int frames) { return 2048 * 4 + frames * R4P_XAREA_BYTES + R4K_TW2_BYTES + frames * 32 + 64; }

// The body of one wave; h (which half of the frame it loads, which parity of bins it produces) is a compile-time
// constant, so that the two roles are straight-line code (as one function with run-time h the register arrays of the
// butterflies stayed in scratch memory: 9.5 ms instead of 4.6).
template <int MODE, int F, int NT, int h>
__device__ __forceinline__ void r4p_wave(char* smem_raw, int slot, const float* __restrict__ x, int64_t batch, int64_t frame_stride,
                                         const float* __restrict__ win, const float2* __restrict__ twM, const float2* __restrict__ twN,
                                         int excl_dc_nyq, float* __restrict__ mag, int32_t* __restrict__ peak_bin,
                                         float* __restrict__ peak_mag, float db_floor) {
    float4* wtab4 = reinterpret_cast<float4*>(smem_raw);                              // 512 float4: 0.5 * window, first half
    const int lane = threadIdx.x & 63;
    char* xarea = smem_raw + 2048 * 4 + slot * R4P_XAREA_BYTES;
    float2* tw2 = reinterpret_cast<float2*>(smem_raw + 2048 * 4 + F * R4P_XAREA_BYTES);   // [8][17]: w_128^{g k}
    unsigned long long* pk = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(tw2) + R4K_TW2_BYTES) + slot * 4;
    const int64_t frame = (int64_t)blockIdx.x * F + slot;
    const bool ok = frame < batch;

    // ---- the frame half: 8 coalesced 1 KiB loads (zero-length descriptor past the end: zeros, no memory touched) -------
    v4f r[8];
    {
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + (ok ? frame : 0) * frame_stride), 0,
                                                                             ok ? 4096 * 4 : 0, 0x00020000);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            r[j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(xrs, lane * 16, (8 * h + j) * 1024, NT & 0xff));
    }
    // per-lane twiddle bases straight from the plan's tables (in flight beside the frame): columns ca = 2 lane, cb = ca + 1
    const float2 b2a = twM[4 * lane], b2b = twM[4 * lane + 2];                        // w_2048^(2 col): powers over k1'
    const float2 b1a = twM[2 * lane], b1b = twM[2 * lane + 1];                        // w_2048^col: the odd outputs' extra factor
    // workgroup tables
    if (threadIdx.x < 128) tw2[(threadIdx.x >> 4) * 17 + (threadIdx.x & 15)] = twM[16 * (threadIdx.x >> 4) * (threadIdx.x & 15)];
    for (int i = threadIdx.x; i < 512; i += F * 128) {
        float4 w = win ? reinterpret_cast<const float4*>(win)[i] : make_float4(1.f, 1.f, 1.f, 1.f);
        w.x *= 0.5f; w.y *= 0.5f; w.z *= 0.5f; w.w *= 0.5f;
        wtab4[i] = w;
    }
    __syncthreads();

    // ---- window; own rows to LDS; partner's rows back ----------------------------------------------------------------------
    float4* xa4 = reinterpret_cast<float4*>(xarea);
    float2 va[8], vb[8];                                                              // columns 2 lane, 2 lane + 1
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if constexpr (h != 0) {                                                       // rows 8..15: mirrored half of the table
            const float4 w = wtab4[(7 - j) * 64 + (63 - lane)];
            va[j] = mk<float>(r[j][0] * w.w, r[j][1] * w.z);
            vb[j] = mk<float>(r[j][2] * w.y, r[j][3] * w.x);
        } else {
            const float4 w = wtab4[j * 64 + lane];
            va[j] = mk<float>(r[j][0] * w.x, r[j][1] * w.y);
            vb[j] = mk<float>(r[j][2] * w.z, r[j][3] * w.w);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) xa4[(8 * h + j) * 68 + lane] = make_float4(va[j].x, va[j].y, vb[j].x, vb[j].y);
    __syncthreads();
    // the partner's half of the area is this wave's private slab from here on
    char* slab = xarea + (1 - h) * R4K_SLAB_BYTES;
    float4* slab4 = reinterpret_cast<float4*>(slab);
    float2* slab2 = reinterpret_cast<float2*>(slab);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float4 o = slab4[j * 68 + lane];                                       // partner's row j (+ 8 (1 - h))
        const float2 oa = mk<float>(o.x, o.y), ob = mk<float>(o.z, o.w);
        if constexpr (h != 0) {                                                       // d_j = (a_j - a_{j+8}) w_16^j, a_j the partner's
            float2 da = csub(oa, va[j]), dbb = csub(ob, vb[j]);
            if (j == 1) { da = mul_w16<1>(da); dbb = mul_w16<1>(dbb); }
            else if (j == 2) { da = mul_w16<2>(da); dbb = mul_w16<2>(dbb); }
            else if (j == 3) { da = mul_w16<3>(da); dbb = mul_w16<3>(dbb); }
            else if (j == 4) { da = mul_mi(da); dbb = mul_mi(dbb); }
            else if (j == 5) { da = mul_mi(mul_w16<1>(da)); dbb = mul_mi(mul_w16<1>(dbb)); }
            else if (j == 6) { da = mul_mi(mul_w16<2>(da)); dbb = mul_mi(mul_w16<2>(dbb)); }
            else if (j == 7) { da = mul_mi(mul_w16<3>(da)); dbb = mul_mi(mul_w16<3>(dbb)); }
            va[j] = da; vb[j] = dbb;
        } else {
            va[j] = cadd(va[j], oa); vb[j] = cadd(vb[j], ob);
        }
    }
    // ---- this wave's half of pass 1: radix 8 over j -> k1 = 2 k1' + h, times w_2048^(col k1) ------------------------------
    bfly8<float>(va);
    bfly8<float>(vb);
    if constexpr (h != 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) { va[k] = cmul(va[k], b1a); vb[k] = cmul(vb[k], b1b); }
    }
    apply_twiddle_powers<float, 8>(va, b2a);
    apply_twiddle_powers<float, 8>(vb, b2b);
    // ---- exchange 1 + pass 2 (radix 16 over the columns of one k1'), rfft4096.h's layout ------------------------------------
    float2 c[16];
    {
        const int x1_rd = (lane >> 3) * 136 + (lane & 7);
        wave_lds_fence();
#pragma unroll
        for (int k = 0; k < 8; ++k) slab4[lane + 68 * k] = make_float4(va[k].x, va[k].y, vb[k].x, vb[k].y);
        wave_lds_fence();
#pragma unroll
        for (int n2 = 0; n2 < 16; ++n2) c[n2] = slab2[x1_rd + 8 * n2];
        bfly16<float>(c);
        apply_twiddles16_tab(c, tw2 + (lane & 7) * 17);
    }
    // ---- exchange 2 + pass 3: final butterfly b = k1' + 8 k2 is row b of the slab (8 values, XOR-swizzled quads) ------------
    float2 P[8], Q[8];
    const int bP = lane;
    const int bQ = h ? 127 - lane : (lane == 0 ? 64 : 128 - lane);
    {
        const int k1p = lane >> 3, c0 = lane & 7;
        const int sw_e = (k1p >> 2) & 1, sw_o = sw_e | 2;
        const int wr_e = k1p * 8 + ((((c0 >> 1) ^ sw_e) << 1) | (c0 & 1));
        const int wr_o = k1p * 8 + ((((c0 >> 1) ^ sw_o) << 1) | (c0 & 1));
        wave_lds_fence();
#pragma unroll
        for (int k2 = 0; k2 < 16; ++k2) slab2[((k2 & 1) ? wr_o : wr_e) + 64 * k2] = c[k2];
        wave_lds_fence();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 u = slab4[bP * 4 + (q ^ ((bP >> 2) & 3))];
            P[2 * q] = mk<float>(u.x, u.y); P[2 * q + 1] = mk<float>(u.z, u.w);
            const float4 v = slab4[bQ * 4 + (q ^ ((bQ >> 2) & 3))];
            Q[2 * q] = mk<float>(v.x, v.y); Q[2 * q + 1] = mk<float>(v.z, v.w);
        }
        bfly8<float>(P);
        bfly8<float>(Q);
    }
    // ---- Hermitian post-pass into registers: P[i] = Z'[hP + 256 i], Q[i] = Z'[256 - hP + 256 i] (Z' = Z / 2) ---------------
    float m[16], m_dc = 0.f, m_ny = 0.f;                                              // |X| of this lane's bins, in emit order
    unsigned long long key = 0;                                                       // (|X| bits << 32) | (4095 - k)
    auto emit = [&](float v, int k, auto IDX) {                                       // IDX: compile-time slot of m[]
        const unsigned long long cnd = ((unsigned long long)__float_as_uint(v) << 32) | (unsigned)(4095 - k);
        key = cnd > key ? cnd : key;
        m[decltype(IDX)::value] = MODE == 2 ? 20.0f * __log10f(fmaxf(v, db_floor)) : v;
    };
    const bool special = h == 0 && lane == 0;
    int kA, kAm, kB, kBm;                                                             // bin bases of the four emit families
    if constexpr (h == 0) {
        // rfft4096.h's slot 1 with its lane index replaced by 2 lane (lane 0: butterflies 0 and 128, DC / Nyquist)
        const float2 z0 = P[0];
        kA = special ? 128 : 2 * lane;
        kB = special ? 256 : 2 * lane + 1024;
        kAm = 256 - kA;
        kBm = special ? 1024 : 256 - 2 * lane;
        const float2 wA = twN[special ? 128 : 2 * lane];
        const float2 wB = special ? cconj(twN[768]) : wA;
        float2 U[8], V[8];
        // (by value: a conditional expression over two array elements is an lvalue, the compiler then selects between
        // the ADDRESSES and both arrays stay in scratch memory)
        auto sel = [&](float2 a, float2 b) { return mk<float>(special ? a.x : b.x, special ? a.y : b.y); };
#pragma unroll
        for (int i = 0; i < 4; ++i) U[i] = sel(Q[i], P[i]);
        U[4] = sel(P[1], P[4]); U[5] = sel(P[2], P[5]); U[6] = sel(P[3], P[6]); U[7] = sel(P[4], P[7]);
#pragma unroll
        for (int i = 4; i < 8; ++i) V[i] = Q[i];
        V[3] = sel(P[7], Q[3]); V[2] = sel(P[6], Q[2]); V[1] = sel(P[5], Q[1]); V[0] = sel(P[4], Q[0]);
        auto slot1 = [&](auto I) {
            constexpr int i = decltype(I)::value;
            float ma, mb;
            herm_pair<false>(U[i], V[7 - i], mul_w16<i>(wA), ma, mb);
            emit(ma, kA + 256 * i, std::integral_constant<int, 4 * i>{});
            emit(mb, kAm + 256 * (7 - i), std::integral_constant<int, 4 * i + 1>{});
            herm_pair<false>(U[i + 4], V[3 - i], mul_mi(mul_w16<i>(wB)), ma, mb);
            if constexpr (i == 3) ma = special ? mb : ma;      // lane 0: the pair (P4, P4), bin 1024 twice — one value (rfft4096.h)
            emit(ma, kB + 256 * i, std::integral_constant<int, 4 * i + 2>{});
            emit(mb, kBm + 256 * (3 - i), std::integral_constant<int, 4 * i + 3>{});
        };
        slot1(std::integral_constant<int, 0>{}); slot1(std::integral_constant<int, 1>{});
        slot1(std::integral_constant<int, 2>{}); slot1(std::integral_constant<int, 3>{});
        // DC / Nyquist (lane 0): X[0] = Re Z0 + Im Z0, X[2048] = Re Z0 - Im Z0, Z0 = 2 z0
        const float v0 = fabsf(2.0f * (z0.x + z0.y)), v1 = fabsf(2.0f * (z0.x - z0.y));
        if (special && !excl_dc_nyq) {
            unsigned kh = (unsigned)(key >> 32), kl = (unsigned)key;
            const unsigned e0 = __float_as_uint(v0), e1 = __float_as_uint(v1);
            if (e0 >= kh) { kh = e0; kl = 4095u; }                                    // bin 0 wins every tie
            if (e1 > kh) { kh = e1; kl = 4095u - 2048u; }                             // bin 2048 loses every tie
            key = ((unsigned long long)kh << 32) | kl;
        }
        m_dc = MODE == 2 ? 20.0f * __log10f(fmaxf(v0, db_floor)) : v0;                // bins 0 and 2048 (lane 0 stages them)
        m_ny = MODE == 2 ? 20.0f * __log10f(fmaxf(v1, db_floor)) : v1;
        // (lane 0 emits bin 1024 twice, from the pair (P4, P4): made the same value above, to the same place)
    } else {
        kA = 2 * lane + 1;                                                            // slot-2 form of rfft4096.h: no special lane
        kAm = 255 - 2 * lane;
        kB = kA; kBm = kAm;
        const float2 wC = twN[2 * lane + 1];
        auto slot2 = [&](auto I) {
            constexpr int i = decltype(I)::value;
            const float2 t = mul_w16<i>(wC);
            float ma, mb;
            herm_pair<false>(P[i], Q[7 - i], t, ma, mb);
            emit(ma, kA + 256 * i, std::integral_constant<int, 4 * i>{});
            emit(mb, kAm + 256 * (7 - i), std::integral_constant<int, 4 * i + 1>{});
            herm_pair<false>(P[i + 4], Q[3 - i], mul_mi(t), ma, mb);
            emit(ma, kA + 256 * (i + 4), std::integral_constant<int, 4 * i + 2>{});
            emit(mb, kAm + 256 * (3 - i), std::integral_constant<int, 4 * i + 3>{});
        };
        slot2(std::integral_constant<int, 0>{}); slot2(std::integral_constant<int, 1>{});
        slot2(std::integral_constant<int, 2>{}); slot2(std::integral_constant<int, 3>{});
    }
    // ---- peak of this wave; both waves are done with their slabs after the barrier ----------------------------------------------
    {
        const float lane_best = __uint_as_float((unsigned)(key >> 32));
        const unsigned lane_bin = 4095u - (unsigned)key;
        const float best = wave_max_f32(lane_best);
        const int bk = wave_min_u32(lane_best == best ? lane_bin : 0x7fffffffu);
        if (lane == 0) pk[h] = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)(4095 - bk);
    }
    __syncthreads();
    // ---- stage the row: float k + o of the area is bin k (o = frame & 3 re-aligns the 2049-float rows to 16 bytes) -----------
    const int o = (int)(frame & 3);
    float* rowf = reinterpret_cast<float*>(xarea);
    if constexpr (h == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rowf[o + kA + 256 * i] = m[4 * i];
            rowf[o + kAm + 256 * (7 - i)] = m[4 * i + 1];
            rowf[o + kB + 256 * i] = m[4 * i + 2];
            rowf[o + kBm + 256 * (3 - i)] = m[4 * i + 3];
        }
        if (special) { rowf[o] = m_dc; rowf[o + 2048] = m_ny; }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rowf[o + kA + 256 * i] = m[4 * i];
            rowf[o + kAm + 256 * (7 - i)] = m[4 * i + 1];
            rowf[o + kA + 256 * (i + 4)] = m[4 * i + 2];
            rowf[o + kAm + 256 * (3 - i)] = m[4 * i + 3];
        }
    }
    __syncthreads();
    // ---- each wave stores half of the row as 16-byte aligned pieces; wave 0 the 4 - o head and o + 1 tail floats ---------------
    const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(ok ? mag + frame * 2049 : nullptr, 0, ok ? 2049 * 4 : 0, 0x00020000);
    const v4u* rowq = reinterpret_cast<const v4u*>(xarea);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int q = lane + 64 * (4 * h + j);
        const v4u d = rowq[q];
        if (q > 0) __builtin_amdgcn_raw_buffer_store_b128(d, mrs, q * 16 - o * 4, 0, NT >> 8);   // quad 0 holds the previous row's tail
    }
    if (h == 0 && lane < 5) {
        const int ke = lane < 4 - o ? lane : 2044 + lane;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(rowf[ke + o]), mrs, ke * 4, 0, NT >> 8);
    }
    if (special && ok) {
        const unsigned long long a0 = pk[0], a1 = pk[1];                              // larger key = larger |X|, then lower bin
        const unsigned long long best = a0 > a1 ? a0 : a1;
        peak_bin[frame] = 4095 - (int)(unsigned)best;
        peak_mag[frame] = __uint_as_float((unsigned)(best >> 32));
    }
}

// MODE: 1 = write |Y|, 2 = write dB.  F: frames per workgroup.
template <int MODE, int F, int NT>
__global__ void __launch_bounds__(F * 128) rfft4096_pair_kernel(const float* __restrict__ x, int64_t batch, int64_t frame_stride,
                                                                const float* __restrict__ win, const float2* __restrict__ twM,
                                                                const float2* __restrict__ twN, int excl_dc_nyq, float* __restrict__ mag,
                                                                int32_t* __restrict__ peak_bin, float* __restrict__ peak_mag, float db_floor) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave & 1) r4p_wave<MODE, F, NT, 1>(smem_raw, wave >> 1, x, batch, frame_stride, win, twM, twN, excl_dc_nyq, mag, peak_bin, peak_mag, db_floor);
    else r4p_wave<MODE, F, NT, 0>(smem_raw, wave >> 1, x, batch, frame_stride, win, twM, twN, excl_dc_nyq, mag, peak_bin, peak_mag, db_floor);
}

}  // namespace sa
