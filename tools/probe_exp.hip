// How often does the device's fp64 exp() differ from the host libm's on the builder's weight arguments (-r^2 / 1.2, r in [0.1, 6] m)?
// hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probe_exp.hip -o tools/probe_exp && tools/probe_exp
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
__global__ void k(const double* x, double* y, long n) {
    long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (i < n) y[i] = exp(x[i]);
}
int main() {
    const long n = 1 << 22;
    std::vector<double> x(n), y(n);
    srand(1);
    for (long i = 0; i < n; ++i) {
        double r = 0.1 + 5.9 * (rand() / (double)RAND_MAX);
        x[i] = -(r * r) / 1.2 * (1.0 + 1e-9 * (rand() / (double)RAND_MAX));
    }
    double *dx, *dy;
    hipMalloc(&dx, n * 8); hipMalloc(&dy, n * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    k<<<(n + 255) / 256, 256>>>(dx, dy, n);
    hipMemcpy(y.data(), dy, n * 8, hipMemcpyDeviceToHost);
    long diff = 0, diff2 = 0, ldiff = 0;
    for (long i = 0; i < n; ++i) {
        double h = exp(x[i]);
        long long a, b;
        memcpy(&a, &h, 8); memcpy(&b, &y[i], 8);
        long long d = a > b ? a - b : b - a;
        if (d) ++diff;
        if (d > 1) ++diff2;
        // against long double (x87: 64-bit mantissa) rounded to double = correctly rounded except at 2^-11 of the ties
        double cr = (double)expl((long double)x[i]);
        if (cr != h) ++ldiff;
    }
    printf("n %ld: device != host libm in %ld (%.4f %%), by more than 1 ulp in %ld; host libm != round(expl) in %ld (%.4f %%)\n", n, diff, 100.0 * diff / n,
           diff2, ldiff, 100.0 * ldiff / n);
    return 0;
}
