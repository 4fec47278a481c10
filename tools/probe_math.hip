// Probe: accuracy of v_rcp_f64 / v_rsq_f64 / v_sqrt_f64 on gfx950 (max relative error over random inputs),
// raw and after 1 / 2 Newton steps.  Build: hipcc --offload-arch=gfx950 -O3 tools/probe_math.hip -o /tmp/probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double *x, double *o, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = x[i];
    double r0 = __builtin_amdgcn_rcp(v);
    double r1 = fma(fma(-v, r0, 1.0), r0, r0);
    double r2 = fma(fma(-v, r1, 1.0), r1, r1);
    double y0 = __builtin_amdgcn_rsq(v);
    double e0 = fma(-v * y0, y0, 1.0);
    double y1 = fma(0.5 * y0, e0, y0);
    double e1 = fma(-v * y1, y1, 1.0);
    double y2 = fma(0.5 * y1, e1, y1);
    double s0 = __builtin_amdgcn_sqrt(v);
    o[i * 7 + 0] = r0; o[i * 7 + 1] = r1; o[i * 7 + 2] = r2;
    o[i * 7 + 3] = y0; o[i * 7 + 4] = y1; o[i * 7 + 5] = y2; o[i * 7 + 6] = s0;
}
int main()
{
    const int n = 1 << 20;
    std::vector<double> h(n), o(n * 7);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        double u = (double)(s >> 11) / 9007199254740992.0;
        h[i] = std::ldexp(1.0 + u, (int)(s % 120) - 60);
    }
    double *dx, *dout;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, n * 56);
    hipMemcpy(dx, h.data(), n * 8, hipMemcpyHostToDevice);
    k<<<n / 256, 256>>>(dx, dout, n);
    hipMemcpy(o.data(), dout, n * 56, hipMemcpyDeviceToHost);
    double m[7] = {0};
    for (int i = 0; i < n; ++i) {
        long double v = h[i];
        long double ref[7] = {1 / v, 1 / v, 1 / v, 1 / sqrtl(v), 1 / sqrtl(v), 1 / sqrtl(v), sqrtl(v)};
        for (int j = 0; j < 7; ++j) {
            double e = (double)fabsl((o[i * 7 + j] - ref[j]) / ref[j]);
            if (e > m[j]) m[j] = e;
        }
    }
    printf("max rel err: rcp raw %.3e  +1NR %.3e  +2NR %.3e | rsq raw %.3e  +1NR %.3e  +2NR %.3e | sqrt raw %.3e\n",
           m[0], m[1], m[2], m[3], m[4], m[5], m[6]);
    return 0;
}
