// Probe: effective shader clock = clock64() ticks / wall_clock64() (100 MHz) under (a) one busy wave, (b) all CUs busy.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void spin(long long iters, long long* out) {
    long long c0 = clock64(), w0 = wall_clock64();
    float x = threadIdx.x;
    for (long long i = 0; i < iters; ++i) x = x * 1.0000001f + 0.5f;
    long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long)x; }
}
int main() {
    long long* d; hipMalloc(&d, 64); long long h[3];
    int rate = 0; hipDeviceGetAttribute(&rate, hipDeviceAttributeWallClockRate, 0);
    for (int rep = 0; rep < 3; ++rep)
    for (int blocks : {1, 256, 2048}) {
        hipLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, 0, 2000000LL, d);
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("blocks %4d: %lld shader ticks / %lld wall ticks (wall rate %d kHz) -> %.0f MHz, %.2f ms\n", blocks, h[0], h[1], rate,
               (double)h[0] / ((double)h[1] / rate) / 1e3, (double)h[1] / rate);
    }
    return 0;
}
