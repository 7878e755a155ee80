#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out) {
    int v = threadIdx.x + 100;
    out[threadIdx.x]       = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false); // wave_shl:1
    out[64 + threadIdx.x]  = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false); // wave_shr:1
    out[128 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x134, 0xf, 0xf, false); // wave_rol:1
    out[192 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x13c, 0xf, 0xf, false); // wave_ror:1
    out[256 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x102, 0xf, 0xf, false); // row_shl:2
    out[320 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x112, 0xf, 0xf, false); // row_shr:2
}
int main() {
    int* d; (void)hipMalloc(&d, 384 * 4); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d); int h[384]; (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char* names[] = {"wave_shl:1", "wave_shr:1", "wave_rol:1", "wave_ror:1", "row_shl:2", "row_shr:2"};
    for (int r = 0; r < 6; ++r) { printf("%-11s:", names[r]); for (int l : {0, 1, 2, 14, 15, 16, 17, 31, 32, 47, 48, 62, 63}) printf(" [%d]=%d", l, h[r * 64 + l]); printf("\n"); }
    return 0;
}
