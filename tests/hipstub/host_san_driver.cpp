// host_san_driver.cpp — exercises libspecengine's HOST code (built with -fsanitize=..., linked against hipstub) through
// the C ABI: everything around the kernels, from several threads at once.  Exit code 0 = every call behaved and the
// sanitizer stayed silent.  Test infrastructure (tests/test_host_sanitizers.py runs it).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <thread>
#include <vector>

#include "../../include/specengine.h"

extern "C" long hipstub_live_allocs(void);
extern "C" long hipstub_launches(void);

static std::atomic<int> g_fail{0};
#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) {                                                               \
            fprintf(stderr, "FAIL %s:%d: %s  [%s]\n", __FILE__, __LINE__, #cond, sa_last_error()); \
            ++g_fail;                                                                \
        }                                                                            \
    } while (0)

static void pipeline_call(int seed, int64_t batch, int n) {
    // > 64 MiB of frames: the 3-deep upload / compute / download pipeline with its helper thread
    std::vector<float> x((size_t)batch * n, (float)seed), mag((size_t)batch * (n / 2 + 1), -1.f), pm(batch, -1.f);
    std::vector<int32_t> pb(batch, -1);
    CHECK(sa_rfft_peak_f32(x.data(), batch, n, SA_WIN_HANN, 0.0, SA_F_WRITE_MAG, mag.data(), pb.data(), pm.data()) == SA_OK);
    CHECK(sa_rfft_peak_f32(x.data(), batch, n, SA_WIN_HANN, 0.0, 0, nullptr, pb.data(), pm.data()) == SA_OK);
}

int main() {
    CHECK(sa_init(-1) == SA_OK);
    CHECK(sa_device_count() >= 1);
    // 1. argument errors never reach the device layer
    {
        float x[16];
        int32_t pb;
        float pm;
        CHECK(sa_rfft_peak_f32(x, 1, 1000, 0, 0, 0, nullptr, &pb, &pm) == SA_EUNSUP);
        CHECK(sa_rfft_peak_f32(x, 1, 8, 0, 0, 0, nullptr, &pb, &pm) == SA_EINVAL);
        CHECK(sa_rfft_peak_f32(nullptr, 1, 16, 0, 0, 0, nullptr, &pb, &pm) == SA_EINVAL);
        CHECK(sa_rfft_peak_f32(x, 1, 16, 99, 0, 0, nullptr, &pb, &pm) == SA_EINVAL);
        CHECK(strlen(sa_last_error()) > 0);
    }
    // 2. plan cache + host table builders (windows, exact twiddles) for many sizes, from several threads at once
    {
        std::vector<std::thread> th;
        for (int t = 0; t < 6; ++t)
            th.emplace_back([t] {
                for (int l = 4; l <= 15; ++l) {
                    const int n = 1 << l;
                    std::vector<double> x(n, 1.0), mag(n / 2 + 1);
                    int32_t pb;
                    double pm;
                    for (int w = 0; w <= SA_WIN_LAST; ++w)
                        if (l <= 14) CHECK(sa_rfft_peak_f64(x.data(), 1, n, w, 0.0, SA_F_WRITE_MAG, mag.data(), &pb, &pm) == SA_OK);
                    std::vector<float> xf(n, 1.f), mf(n / 2 + 1);
                    float pmf;
                    CHECK(sa_rfft_peak_f32(xf.data(), 1, n, (t + l) % (SA_WIN_LAST + 1), 0.0, SA_F_WRITE_MAG, mf.data(), &pb, &pmf) == SA_OK);
                }
            });
        for (auto& t : th) t.join();
    }
    // 3. the pipelined host path from two threads at once (the GUI's pool worker + Tk thread, main.py:448, 745)
    {
        std::thread a(pipeline_call, 1, 3 * 4096 + 17, 4096), b(pipeline_call, 2, 2 * 4096 + 1, 4096);
        a.join(); b.join();
    }
    // 4. short-lived threads hand their stream + arena to the next one (runtime.hip's pool)
    {
        const int before = sa_thread_state_count();
        for (int r = 0; r < 20; ++r) {
            std::thread([r] {
                std::vector<double> x(1500 + r, 0.5), mag((1500 + r) / 2 + 1);
                int32_t pb;
                double pm;
                CHECK(sa_fft_mag_f64(x.data(), 1, 1500 + r, SA_WIN_NONE, 0.0, SA_F_WRITE_MAG, mag.data(), &pb, &pm) == SA_OK);   // Bluestein tables
            }).join();
        }
        CHECK(sa_thread_state_count() <= before + 1);
    }
    // 5. every other host entry point once: c2c (1/2/3 passes), long real frames, STFT, synthesis, fused pipeline,
    //    metrics, find_peaks, spline, analyze
    {
        for (int l : {8, 14, 21}) {
            std::vector<double> in((size_t)2 << l, 0.25), out((size_t)2 << l);
            CHECK(sa_fft_c2c_f64(in.data(), out.data(), (int64_t)1 << l, 1, 0) == SA_OK);
            CHECK(sa_fft_c2c_f64(in.data(), out.data(), (int64_t)1 << l, 1, SA_FFT_INVERSE) == SA_OK);
        }
        {
            const int n = 1 << 17;
            std::vector<float> x(n, 1.f), mag(n / 2 + 1);
            int32_t pb;
            float pm;
            CHECK(sa_rfft_peak_f32(x.data(), 1, n, SA_WIN_BLACKMAN, 0.0, SA_F_WRITE_MAG | SA_F_DB, mag.data(), &pb, &pm) == SA_OK);
        }
        {
            const int64_t ns = 8192 + 2048 * 9 + 5;
            std::vector<int16_t> pcm(ns * 2, 100);
            const int64_t nf = sa_stft_num_frames(ns, 8192, 2048);
            CHECK(nf == 10);
            std::vector<float> mag(2 * nf * 4097), pm(2 * nf);
            std::vector<int32_t> pb(2 * nf);
            CHECK(sa_stft_pcm16(pcm.data(), ns, 2, 8192, 2048, SA_WIN_BLACKMAN, 0.0, SA_F_WRITE_MAG, mag.data(), pb.data(), pm.data()) == SA_OK);
            std::vector<int16_t> mono(ns, 7);
            const int64_t nfm = sa_stft_num_frames(ns, 1024, 333);
            std::vector<float> pmm(nfm);
            std::vector<int32_t> pbm(nfm);
            CHECK(sa_stft_pcm16(mono.data(), ns, 1, 1024, 333, SA_WIN_HANN, 0.0, 0, nullptr, pbm.data(), pmm.data()) == SA_OK);
        }
        {
            sa_gen_params p{};
            p.waveform = SA_WF_SENO; p.N = 2000; p.duration = 0.1; p.Fc = 1000; p.Fs = 20000; p.Fm = 100; p.vpp_half = 1; p.apply_vpp = 1;
            std::vector<double> t(2000), y(2000), mag(1001);
            int32_t pb;
            double pm;
            CHECK(sa_synth_f64(&p, t.data(), y.data()) == SA_OK);
            CHECK(sa_compute_f64(&p, SA_WIN_NONE, 0.0, SA_F_WRITE_MAG, t.data(), y.data(), mag.data(), &pb, &pm) == SA_OK);
            std::vector<sa_gen_params> ps(5, p);
            std::vector<float> m32(5 * 8193), y32(5 * 16384), pm32(5);
            std::vector<int32_t> pb32(5);
            CHECK(sa_synth_analyze(ps.data(), 5, 2048, 8, SA_WIN_GAUSSIAN, 0.0, SA_F_WRITE_MAG, m32.data(), pb32.data(), pm32.data(), y32.data()) == SA_OK);
        }
        {
            std::vector<double> y(3 * 777, 1.0);
            std::vector<sa_time_stats> ts(3);
            CHECK(sa_time_stats_f64(y.data(), 3, 777, ts.data()) == SA_OK);
            std::vector<int32_t> idx(3 * 400), cnt(3);
            // no kernel runs under the stub, so the per-row counts the call reads back are whatever the arena held:
            // either outcome of its overflow check is a clean return
            const int rc_fp = sa_find_peaks_f64(y.data(), 3, 777, 0.0, 0.0, 10, 400, idx.data(), cnt.data());
            CHECK(rc_fp == SA_OK || rc_fp == SA_EINVAL);
            std::vector<double> half(3 * 389, 2.0);
            std::vector<sa_spec_metrics> sm(3);
            CHECK(sa_spectral_metrics_f64(half.data(), 3, 776, 1.0, 10.0, sm.data()) == SA_OK);
            std::vector<float> x(7 * 4096, 1.f), pmf(7);
            std::vector<sa_time_stats> ts2(7);
            std::vector<sa_spec_metrics> sm2(7);
            std::vector<int32_t> pb(7);
            CHECK(sa_analyze_frames_f32(x.data(), 7, 4096, SA_WIN_HANN, 0.0, 1e5, 10.0, ts2.data(), sm2.data(), pb.data(), pmf.data()) == SA_OK);
            CHECK(sa_analyze_frames_f32(x.data(), 7, 4096, SA_WIN_HANN, 0.0, 1e5, 10.0, nullptr, sm2.data(), nullptr, nullptr) == SA_OK);
            std::vector<double> tk(40), yk(40, 1.0), to(500), yo(500);
            for (int i = 0; i < 40; ++i) tk[i] = i * 0.5;
            CHECK(sa_cubic_upsample_f64(tk.data(), yk.data(), 1, 40, 500, to.data(), yo.data()) == SA_OK);
            tk[7] = tk[6];
            CHECK(sa_cubic_upsample_f64(tk.data(), yk.data(), 1, 40, 500, to.data(), yo.data()) == SA_EINVAL);
        }
    }
    // 6. single-process fan-out: configure, run from two threads, reconfigure under load, switch off
    {
        CHECK(sa_shard_devices(nullptr, 5) == SA_OK && sa_shard_count() == 5);
        std::thread a(pipeline_call, 3, 4096 + 3, 4096), b(pipeline_call, 4, 7, 4096);
        std::thread c([] { for (int g : {2, 8, 3}) CHECK(sa_shard_devices(nullptr, g) == SA_OK); });
        a.join(); b.join(); c.join();
        const int32_t bad[2] = {0, 99};
        CHECK(sa_shard_devices(bad, 2) == SA_EINVAL);
        float x[64];
        int32_t pb[4];
        float pm[4];
        CHECK(sa_rfft_peak_f32(x, 4, 1000, 0, 0, 0, nullptr, pb, pm) == SA_EUNSUP);     // error text travels back from a shard
        CHECK(strstr(sa_last_error(), "not a power of two") != nullptr);
        CHECK(sa_shard_devices(nullptr, 0) == SA_OK && sa_shard_count() == 0);
    }
    // 7. the distributed plan's bookkeeping (virtual ranks), every exchange mode, natural order, host entry point
    {
        for (int mode = SA_DIST_XCHG_AUTO; mode <= SA_DIST_XCHG_FUSED; ++mode)
            for (int P : {1, 2, 8}) {
                sa_dist_plan* pl = nullptr;
                CHECK(sa_dist_plan_create(&pl, 1 << 16, P, nullptr, SA_DIST_XCHG(mode) | SA_DIST_CHUNKS(3) | SA_DIST_NATURAL_IN | SA_DIST_NATURAL_OUT) == SA_OK);
                if (!pl) continue;
                sa_dist_info inf;
                CHECK(sa_dist_plan_info(pl, P - 1, &inf) == SA_OK && inf.slab_elems == (1 << 16) / P);
                CHECK(sa_dist_fill(pl, SA_DIST_KAT_TONE, 77, 0) == SA_OK);
                sa_dist_stats st;
                CHECK(sa_dist_execute(pl, &st) == SA_OK && st.exchanges == 3 && st.n_ranks == P);
                double err = 0;
                CHECK(sa_dist_check(pl, SA_DIST_KAT_TONE, 77, &err) == SA_OK);
                CHECK(sa_dist_plan_destroy(pl) == SA_OK);
            }
        std::vector<double> in(2 << 12, 1.0), out(2 << 12);
        CHECK(sa_fft_c2c_f64_dist(in.data(), out.data(), 1 << 12, 4, 0, nullptr) == SA_OK);
        sa_dist_plan* pl = nullptr;
        CHECK(sa_dist_plan_create(&pl, 1 << 12, 3, nullptr, 0) == SA_EINVAL && pl == nullptr);
    }
    // 8. shutdown while other threads are still calling: it waits for calls in flight, later calls rebuild their plans
    {
        std::atomic<bool> stop{false};
        std::vector<std::thread> th;
        for (int t = 0; t < 4; ++t)
            th.emplace_back([&stop, t] {
                std::vector<float> x(4 * 2048, 1.f), pm(4);
                std::vector<int32_t> pb(4);
                while (!stop.load()) CHECK(sa_rfft_peak_f32(x.data(), 4, 2048, 1 + t % 3, 0.0, 0, nullptr, pb.data(), pm.data()) == SA_OK);
            });
        // ... and a thread looping over a distributed plan: sa_dist_execute fetches cached twiddle tables that sa_shutdown
        // frees, so it must sit behind the same gate; a sharded caller whose workers sa_shutdown stops
        th.emplace_back([&stop] {
            sa_dist_plan* pl = nullptr;
            CHECK(sa_dist_plan_create(&pl, 1 << 14, 4, nullptr, SA_DIST_XCHG(SA_DIST_XCHG_DMA) | SA_DIST_CHUNKS(2)) == SA_OK);
            if (!pl) return;
            sa_dist_stats st;
            double err = 0;
            while (!stop.load()) {
                CHECK(sa_dist_fill(pl, SA_DIST_KAT_IMPULSE, 5, 0) == SA_OK);
                CHECK(sa_dist_execute(pl, &st) == SA_OK && st.pairs_timed == 1);
                CHECK(sa_dist_check(pl, SA_DIST_KAT_IMPULSE, 5, &err) == SA_OK);
            }
            CHECK(sa_dist_plan_destroy(pl) == SA_OK);
        });
        CHECK(sa_shard_devices(nullptr, 3) == SA_OK);
        th.emplace_back([&stop] {
            std::vector<float> x(16 * 1024, 1.f), pm(16);
            std::vector<int32_t> pb(16);
            while (!stop.load()) CHECK(sa_rfft_peak_f32(x.data(), 16, 1024, 1, 0.0, 0, nullptr, pb.data(), pm.data()) == SA_OK);
        });
        for (int r = 0; r < 50; ++r) sa_shutdown();
        stop = true;
        for (auto& t : th) t.join();
        CHECK(sa_shard_count() == 3);                     // the configured fan-out survives sa_shutdown
        CHECK(sa_shard_devices(nullptr, 0) == SA_OK && sa_shard_count() == 0);
    }
    sa_shutdown();
    CHECK(hipstub_launches() > 100);
    printf("host_san_driver: %d failure(s), %ld kernel launches intercepted, %ld stub allocations live after shutdown\n",
           g_fail.load(), hipstub_launches(), hipstub_live_allocs());
    return g_fail.load() ? 1 : 0;
}
