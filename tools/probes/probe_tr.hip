// Probe: semantics of ds_read_tr16_b64 (ds_read_b64_tr_b16) and 4-byte-aligned float4 global loads on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
struct __attribute__((packed, aligned(4))) F4 { float v[4]; };

__global__ void k_tr(const unsigned short* in, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 4];
    const int l = threadIdx.x;
    for (int i = l; i < 256; i += 64) lds[i] = in[i];
    __syncthreads();
    // each lane supplies the address of 4 consecutive u16: lane-linear image
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(lds + l * 4));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
__global__ void k_unal(const float* in, float* out) {
    const int l = threadIdx.x;
    const F4 f = *reinterpret_cast<const F4*>(in + 1 + 4 * l);   // 4-byte aligned only
    out[l] = f.v[0] + f.v[1] + f.v[2] + f.v[3];
}
int main() {
    unsigned short h[256], o[256];
    for (int i = 0; i < 256; ++i) h[i] = i;
    unsigned short *di, *dout;
    hipMalloc(&di, 512); hipMalloc(&dout, 512);
    hipMemcpy(di, h, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_tr, dim3(1), dim3(64), 0, 0, di, dout);
    hipMemcpy(o, dout, 512, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l) {
        for (int j = 0; j < 4; ++j) {
            const int expect = (l & 15) + j * 16 + (l >> 4) * 64;
            if (o[l * 4 + j] != expect) ok = 0;
        }
    }
    printf("tr16 formula lds[(l&15)+j*16+(l>>4)*64]: %s\n", ok ? "MATCH" : "MISMATCH");
    if (!ok) for (int l = 0; l < 64; ++l) printf("lane %d: %d %d %d %d\n", l, o[l*4], o[l*4+1], o[l*4+2], o[l*4+3]);
    float hf[512], of[64]; for (int i = 0; i < 512; ++i) hf[i] = (float)i;
    float *fi, *fo; hipMalloc(&fi, 2048); hipMalloc(&fo, 256);
    hipMemcpy(fi, hf, 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_unal, dim3(1), dim3(64), 0, 0, fi, fo);
    hipError_t e = hipMemcpy(of, fo, 256, hipMemcpyDeviceToHost);
    int ok2 = e == hipSuccess;
    for (int l = 0; l < 64 && ok2; ++l) { float ex = 4.0f * (1 + 4 * l) + 6.0f; if (of[l] != ex) ok2 = 0; }
    printf("unaligned float4 global load: %s (%s)\n", ok2 ? "OK" : "BAD", hipGetErrorString(e));
    return 0;
}
