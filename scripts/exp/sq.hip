// Debug aid: are the _rn intrinsics of this ROCm correctly rounded?  __fsqrt_rn against sqrtf (-fhip-fp32-correctly-rounded-divide-sqrt)
// and against the double-precision root narrowed (exact for float), __fdiv_rn / __ddiv_rn against the plain operators, on random
// operands; prints mismatch counts.   ./sq.bin
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long *bad, unsigned seed)
{
    unsigned long long s = 88172645463325252ull + (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761ull + seed;
    unsigned b0 = 0, b1 = 0, b2 = 0, b3 = 0;
    for (int i = 0; i < 4096; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const float a = __uint_as_float(0x3e000000u + (unsigned)(s & 0x03ffffffu));          // [0.125, 32)
        const float b = __uint_as_float(0x3f000000u + (unsigned)((s >> 26) & 0x01ffffffu));   // [0.5, 8)
        const float r0 = __fsqrt_rn(a), r1 = sqrtf(a), r2 = (float)sqrt((double)a);
        b0 += r0 != r2;
        b1 += r1 != r2;
        b2 += __fdiv_rn(a, b) != a / b;
        const double da = (double)a * 1.000000123, db = (double)b * 0.999999871;
        b3 += __ddiv_rn(da, db) != da / db;
    }
    atomicAdd(&bad[0], b0); atomicAdd(&bad[1], b1); atomicAdd(&bad[2], b2); atomicAdd(&bad[3], b3);
}
int main()
{
    unsigned long long *d, h[4] = {0, 0, 0, 0};
    hipMalloc(&d, 32); hipMemcpy(d, h, 32, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(256), dim3(256), 0, 0, d, 1u);
    hipMemcpy(h, d, 32, hipMemcpyDeviceToHost);
    printf("of %d operands: __fsqrt_rn != exact %llu, sqrtf != exact %llu, __fdiv_rn != '/' %llu, __ddiv_rn != '/' %llu\n", 256 * 256 * 4096, h[0], h[1], h[2], h[3]);
    return 0;
}
