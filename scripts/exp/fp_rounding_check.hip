// Is the device's f32 square root (as the kernels of this library spell it: __fsqrt_rn, sqrtf) the correctly rounded one the host's sqrtss is?
// Exhaustive over every positive normal float in [2^-20, 2^20) would be 335 M values; this checks a stride through all of them plus every float in [1, 4).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
__global__ void k(const float *x, float *a, float *b, size_t n)
{
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) { a[i] = __fsqrt_rn(x[i]); b[i] = sqrtf(x[i]); }
}
__global__ void kdiv(const float *x, const float *y, float *q, const double *xd, const double *yd, double *qd, size_t n)
{
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) { q[i] = x[i] / y[i]; qd[i] = xd[i] / yd[i]; }
}
__global__ void kd(const double *x, double *a, size_t n)
{
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i < n) a[i] = sqrt(x[i]);
}
int main()
{
    std::vector<float> h;
    for (unsigned bits = 0x3f800000u; bits < 0x40800000u; ++bits) { float f; std::memcpy(&f, &bits, 4); h.push_back(f); }            // [1, 4): every mantissa, both exponent parities
    for (unsigned bits = 0x35800000u; bits < 0x49800000u; bits += 37) { float f; std::memcpy(&f, &bits, 4); h.push_back(f); }      // 2^-20 .. 2^20, stride 37
    for (unsigned bits = 1; bits < 0x00800000u; bits += 101) { float f; std::memcpy(&f, &bits, 4); h.push_back(f); }               // denormals
    const size_t n = h.size();
    float *dx, *da, *db;
    hipMalloc(&dx, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4);
    hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
    k<<<unsigned((n + 255) / 256), 256>>>(dx, da, db, n);
    std::vector<float> a(n), b(n);
    hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost);
    size_t bad_a = 0, bad_b = 0;
    for (size_t i = 0; i < n; ++i) {
        const float want = std::sqrt(h[i]);
        if (std::memcmp(&want, &a[i], 4)) { if (bad_a++ < 3) std::printf("__fsqrt_rn(%a) = %a, host %a\n", h[i], a[i], want); }
        if (std::memcmp(&want, &b[i], 4)) { if (bad_b++ < 3) std::printf("sqrtf(%a) = %a, host %a\n", h[i], b[i], want); }
    }
    std::printf("%zu values: __fsqrt_rn differs from the host's sqrt on %zu, sqrtf on %zu\n", n, bad_a, bad_b);
    {   // f64: sqrt(double) as the solver kernels use it, on 2^24 arguments spread over [2^-40, 2^40) with random low mantissa bits
        const size_t m = size_t(1) << 24;
        std::vector<double> hd(m), ad(m);
        unsigned long long st = 88172645463325252ull;
        for (size_t i = 0; i < m; ++i) {
            st ^= st << 13; st ^= st >> 7; st ^= st << 17;
            const unsigned long long bits = ((1023ull - 40 + (st >> 57) % 80) << 52) | (st & 0xfffffffffffffull);
            std::memcpy(&hd[i], &bits, 8);
        }
        double *ddx, *dda;
        hipMalloc(&ddx, m * 8); hipMalloc(&dda, m * 8);
        hipMemcpy(ddx, hd.data(), m * 8, hipMemcpyHostToDevice);
        kd<<<unsigned((m + 255) / 256), 256>>>(ddx, dda, m);
        hipMemcpy(ad.data(), dda, m * 8, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t i = 0; i < m; ++i) { const double want = std::sqrt(hd[i]); if (std::memcmp(&want, &ad[i], 8)) { if (bad++ < 3) std::printf("sqrt(%a) = %a, host %a\n", hd[i], ad[i], want); } }
        std::printf("%zu f64 values: sqrt differs from the host's on %zu\n", m, bad);
    }
    {   // divisions, f32 and f64, on 2^24 random operand pairs
        const size_t m = size_t(1) << 24;
        std::vector<float> x(m), y(m), q(m);
        std::vector<double> xd(m), yd(m), qd(m);
        unsigned long long st = 0x9e3779b97f4a7c15ull;
        auto next = [&] { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
        for (size_t i = 0; i < m; ++i) {
            unsigned a = unsigned(((127u - 20 + unsigned(next() >> 58) % 40) << 23) | unsigned(next() & 0x7fffff)), b = unsigned(((127u - 20 + unsigned(next() >> 58) % 40) << 23) | unsigned(next() & 0x7fffff));
            std::memcpy(&x[i], &a, 4); std::memcpy(&y[i], &b, 4);
            unsigned long long c = ((1023ull - 30 + (next() >> 57) % 60) << 52) | (next() & 0xfffffffffffffull), d = ((1023ull - 30 + (next() >> 57) % 60) << 52) | (next() & 0xfffffffffffffull);
            std::memcpy(&xd[i], &c, 8); std::memcpy(&yd[i], &d, 8);
        }
        float *dx2, *dy2, *dq2; double *dxd, *dyd, *dqd;
        hipMalloc(&dx2, m * 4); hipMalloc(&dy2, m * 4); hipMalloc(&dq2, m * 4); hipMalloc(&dxd, m * 8); hipMalloc(&dyd, m * 8); hipMalloc(&dqd, m * 8);
        hipMemcpy(dx2, x.data(), m * 4, hipMemcpyHostToDevice); hipMemcpy(dy2, y.data(), m * 4, hipMemcpyHostToDevice);
        hipMemcpy(dxd, xd.data(), m * 8, hipMemcpyHostToDevice); hipMemcpy(dyd, yd.data(), m * 8, hipMemcpyHostToDevice);
        kdiv<<<unsigned((m + 255) / 256), 256>>>(dx2, dy2, dq2, dxd, dyd, dqd, m);
        hipMemcpy(q.data(), dq2, m * 4, hipMemcpyDeviceToHost); hipMemcpy(qd.data(), dqd, m * 8, hipMemcpyDeviceToHost);
        size_t bad32 = 0, bad64 = 0;
        for (size_t i = 0; i < m; ++i) {
            const volatile float w32 = x[i] / y[i]; const volatile double w64 = xd[i] / yd[i];
            const float w32n = w32; const double w64n = w64;
            if (std::memcmp(&w32n, &q[i], 4)) ++bad32;
            if (std::memcmp(&w64n, &qd[i], 8)) ++bad64;
        }
        std::printf("%zu operand pairs: f32 division differs from the host's on %zu, f64 division on %zu\n", m, bad32, bad64);
    }
    return 0;
}
