// Probe: which lane does v_mov_b32_dpp wave_shl:1 / wave_shr:1 read from on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *p) {
    const int v = threadIdx.x;
    p[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false);        // wave_shl:1
    p[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);   // wave_shr:1
}
int main() {
    int *d, h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("wave_shl:1 lanes 0,1,15,16,31,32,62,63 <- %d %d %d %d %d %d %d %d\n", h[0], h[1], h[15], h[16], h[31], h[32], h[62], h[63]);
    printf("wave_shr:1 lanes 0,1,15,16,31,32,62,63 <- %d %d %d %d %d %d %d %d\n", h[64], h[65], h[79], h[80], h[95], h[96], h[126], h[127]);
    return 0;
}
