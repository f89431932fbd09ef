// spline.hip — a10: the zoomed time plot's smoothing.
//
//   cs = CubicSpline(t_visible, y_visible); t_interp = np.linspace(t_visible[0], t_visible[-1], 500); y = cs(t_interp)
//   (reference main.py:1385-1392, constants INTERP_THRESHOLD / INTERP_SAMPLES main.py:26-27)
//
// scipy.interpolate.CubicSpline with its default bc_type='not-a-knot' (scipy/interpolate/_cubic.py): a tridiagonal
// system for the knot derivatives s_i, then per interval the power-basis coefficients PPoly evaluates.  One
// wavefront per row: the knots sit in LDS, lane 0 runs the (serial, <= 2048-row) Thomas elimination, all lanes
// evaluate the outputs.  fp64 throughout; the knots may be non-uniform.  The fused generator pipeline
// (synth_pipeline.hip) carries its own fp32, uniform-knot version of the same spline.
#include <math.h>
#include <string.h>

#include "runtime.h"

#pragma clang fp contract(off)      // np.linspace is i * step + start with two roundings; keep it that way

namespace sa {

constexpr int SPLINE_MAX_KNOTS = 2048;

// t, y: [batch][m] knots; t_out, y_out: [batch][n_out].  LDS: 5 m doubles.
__global__ void __launch_bounds__(64) cubic_upsample_kernel(const double* __restrict__ t, const double* __restrict__ y,
                                                            int64_t batch, int m, int n_out, double* __restrict__ t_out,
                                                            double* __restrict__ y_out) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* xk = reinterpret_cast<double*>(smem_raw);    // knots
    double* yk = xk + m;
    double* s = yk + m;                                  // derivatives at the knots
    double* cp = s + m;                                  // Thomas work: modified super-diagonal
    double* dp = cp + m;                                 //              modified right-hand side
    const int lane = threadIdx.x;
    for (int64_t row = blockIdx.x; row < batch; row += gridDim.x) {
        __syncthreads();                                 // the previous row's readers are done
        for (int i = lane; i < m; i += 64) { xk[i] = t[row * m + i]; yk[i] = y[row * m + i]; }
        __syncthreads();
        if (lane == 0) {
            auto dx = [&](int i) { return xk[i + 1] - xk[i]; };
            auto slope = [&](int i) { return (yk[i + 1] - yk[i]) / (xk[i + 1] - xk[i]); };
            if (m == 2) {
                s[0] = s[1] = slope(0);
            } else if (m == 3) {
                // scipy: the parabola through the three points.  Rows: [1 1 0], [dx1 2(dx0+dx1) dx0], [0 1 1]
                const double d0 = dx(0), d1 = dx(1), s0 = slope(0), s1 = slope(1);
                const double b0 = 2 * s0, b1 = 3 * (d0 * s1 + d1 * s0), b2 = 2 * s1;
                // eliminate: s0' = b0 - s1', s2' = b2 - s1'  ->  s1' (2(d0+d1) - d1 - d0) = b1 - d1 b0 - d0 b2
                const double mid = (b1 - d1 * b0 - d0 * b2) / (d0 + d1);
                s[1] = mid; s[0] = b0 - mid; s[2] = b2 - mid;
            } else {
                // row i: lo_i s_{i-1} + di_i s_i + up_i s_{i+1} = b_i
                // row 0 (not-a-knot): dx1 s0 + (x2 - x0) s1 = ((dx0 + 2 d) dx1 slope0 + dx0^2 slope1) / d,  d = x2 - x0
                double d = xk[2] - xk[0];
                double di = dx(1), up = d;
                double b = ((dx(0) + 2 * d) * dx(1) * slope(0) + dx(0) * dx(0) * slope(1)) / d;
                cp[0] = up / di;
                dp[0] = b / di;
                for (int i = 1; i < m - 1; ++i) {
                    const double lo = dx(i);
                    di = 2 * (dx(i - 1) + dx(i));
                    up = dx(i - 1);
                    b = 3 * (dx(i) * slope(i - 1) + dx(i - 1) * slope(i));
                    const double den = di - lo * cp[i - 1];
                    cp[i] = up / den;
                    dp[i] = (b - lo * dp[i - 1]) / den;
                }
                // last row: (x_{m-1} - x_{m-3}) s_{m-2} + dx_{m-3} s_{m-1} = (dx_{m-2}^2 slope_{m-3} + (2 d + dx_{m-2}) dx_{m-3} slope_{m-2}) / d
                d = xk[m - 1] - xk[m - 3];
                const double lo = d;
                di = dx(m - 3);
                b = (dx(m - 2) * dx(m - 2) * slope(m - 3) + (2 * d + dx(m - 2)) * dx(m - 3) * slope(m - 2)) / d;
                const double den = di - lo * cp[m - 2];
                s[m - 1] = (b - lo * dp[m - 2]) / den;
                for (int i = m - 2; i >= 0; --i) s[i] = dp[i] - cp[i] * s[i + 1];
            }
        }
        __syncthreads();
        // np.linspace(x0, x1, n_out): step = (x1 - x0) / (n_out - 1); y = arange * step + x0; last point = x1 exactly
        const double x0 = xk[0], x1 = xk[m - 1];
        const double step = n_out > 1 ? (x1 - x0) / (double)(n_out - 1) : 0.0;
        for (int q = lane; q < n_out; q += 64) {
            double xq = (double)q * step + x0;
            if (q == n_out - 1 && n_out > 1) xq = x1;
            // interval: largest i <= m - 2 with xk[i] <= xq (PPoly's find_interval; xq is inside [x0, x1] up to rounding)
            int lo = 0, hi = m - 1;
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (xk[mid] <= xq) lo = mid; else hi = mid;
            }
            const int i = lo;
            const double h = xk[i + 1] - xk[i];
            const double sl = (yk[i + 1] - yk[i]) / h;
            const double tt = (s[i] + s[i + 1] - 2 * sl) / h;
            const double c3 = tt / h, c2 = (sl - s[i]) / h - tt, c1 = s[i], c0 = yk[i];
            const double u = xq - xk[i];
            t_out[row * n_out + q] = xq;
            y_out[row * n_out + q] = ((c3 * u + c2) * u + c1) * u + c0;
        }
    }
}

int cubic_upsample_dev(const double* t, const double* y, int64_t batch, int32_t m, int32_t n_out, double* t_out, double* y_out,
                       hipStream_t st) {
    SA_ENTER();
    if (batch < 0 || m < 2 || n_out < 1) { set_error("cubic_upsample: need batch >= 0, m >= 2 knots, n_out >= 1"); return SA_EINVAL; }
    if (m > SPLINE_MAX_KNOTS) { set_error("cubic_upsample: m=%d knots exceed the supported %d", m, SPLINE_MAX_KNOTS); return SA_EUNSUP; }
    if (batch == 0) return SA_OK;
    if (!t || !y || !t_out || !y_out) { set_error("cubic_upsample: NULL buffer"); return SA_EINVAL; }
    const int lds = 5 * m * (int)sizeof(double);
    static thread_local uint64_t attr_devs = 0;
    SA_TRY(allow_dynamic_lds(reinterpret_cast<const void*>(cubic_upsample_kernel), 5 * SPLINE_MAX_KNOTS * (int)sizeof(double), &attr_devs));
    const int64_t cap = (int64_t)num_cus() * 16;
    hipLaunchKernelGGL(cubic_upsample_kernel, dim3((unsigned)(batch < cap ? batch : cap)), dim3(64), lds, st, t, y, batch, m, n_out,
                       t_out, y_out);
    SA_HIP(hipGetLastError());
    return SA_OK;
}

}  // namespace sa

extern "C" {

int sa_cubic_upsample_f64_dev(const double* t, const double* y, int64_t batch, int32_t m, int32_t n_out, double* t_out,
                              double* y_out, void* stream) {
    SA_ENTER();
    return sa::cubic_upsample_dev(t, y, batch, m, n_out, t_out, y_out, stream ? (hipStream_t)stream : sa::thread_stream());
}

int sa_cubic_upsample_f64(const double* t, const double* y, int64_t batch, int32_t m, int32_t n_out, double* t_out, double* y_out) {
    using namespace sa;
    SA_ENTER();
    if (batch < 0 || m < 2 || n_out < 1) { set_error("cubic_upsample: need batch >= 0, m >= 2 knots, n_out >= 1"); return SA_EINVAL; }
    if (m > SPLINE_MAX_KNOTS) { set_error("cubic_upsample: m=%d knots exceed the supported %d", m, SPLINE_MAX_KNOTS); return SA_EUNSUP; }
    if (batch == 0) return SA_OK;
    if (!t || !y || !t_out || !y_out) { set_error("cubic_upsample: NULL buffer"); return SA_EINVAL; }
    // knots must be finite and strictly increasing (scipy raises ValueError for anything else)
    for (int64_t b = 0; b < batch; ++b)
        for (int32_t i = 0; i < m; ++i) {
            const double v = t[b * m + i];
            if (!(v - v == 0.0) || !(y[b * m + i] - y[b * m + i] == 0.0)) { set_error("cubic_upsample: non-finite knot in row %lld", (long long)b); return SA_EINVAL; }
            if (i > 0 && !(v > t[b * m + i - 1])) { set_error("cubic_upsample: `x` must be strictly increasing sequence (row %lld)", (long long)b); return SA_EINVAL; }
        }
    hipStream_t st = thread_stream();
    const size_t kb = (size_t)batch * m * 8, ob = (size_t)batch * n_out * 8;
    Temp tmp;
    char* d = static_cast<char*>(tmp.dev(2 * kb + 2 * ob));
    char* h = static_cast<char*>(tmp.pinned(2 * kb + 2 * ob));
    if (!d || !h) return SA_ENOMEM;
    memcpy(h, t, kb);
    memcpy(h + kb, y, kb);
    SA_HIP(hipMemcpyAsync(d, h, 2 * kb, hipMemcpyHostToDevice, st));
    SA_TRY(cubic_upsample_dev(reinterpret_cast<const double*>(d), reinterpret_cast<const double*>(d + kb), batch, m, n_out,
                              reinterpret_cast<double*>(d + 2 * kb), reinterpret_cast<double*>(d + 2 * kb + ob), st));
    SA_HIP(hipMemcpyAsync(h + 2 * kb, d + 2 * kb, 2 * ob, hipMemcpyDeviceToHost, st));
    SA_HIP(hipStreamSynchronize(st));
    memcpy(t_out, h + 2 * kb, ob);
    memcpy(y_out, h + 2 * kb + ob, ob);
    return SA_OK;
}

}  // extern "C"
