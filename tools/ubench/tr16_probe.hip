// Semantics probe of ds_read_b64_tr_b16 (gfx950): lane l reads the 4 consecutive 16-bit elements at element index 4*l of an LDS array
// holding s[i] = i; the printed value 4*L + e says "came from lane L's element e".   hipcc --offload-arch=gfx950 -O2 tr16_probe.hip -o /tmp/tr16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4i16 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) v4i16 lds_v4i16;
__global__ void k(short* out) {
    __shared__ __attribute__((aligned(16))) short s[256];
    for (int i = threadIdx.x; i < 256; i += 64) s[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x;
    v4i16 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4i16*)(s + l * 4));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
    short* d;
    short h[256];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) printf("  L%2d.e%d", h[l * 4 + j] / 4, h[l * 4 + j] % 4);
        printf("\n");
    }
    return 0;
}
