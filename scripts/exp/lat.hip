// lat.hip - round 5: what does a FIRST TOUCH cost?  The fused kernel's workgroups march down the rows of 16-byte-per-pixel planes
// (row stride 30 KB at 1080p): every row is a new set of lines and - if the TLB works in 4 KB pages - a new page per plane.
// One wave per workgroup reads 1 KB (64 lanes x 16 B) at base + i * stride, waits for it, and goes on; time per dependent step for
// strides from 1 KB to 4 MB on a cold buffer, with the rest of the chip idle ("alone": 1 workgroup per XCD) and loaded (every CU busy
// with the same walk).  Build: hipcc --offload-arch=gfx950 -O3 -w scripts/exp/lat.hip -o scripts/exp/lat.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(64) void k_walk(const char *base, size_t stride, int steps, size_t per_wg, unsigned long long *out, float *sink)
{
    const char *p = base + (size_t)blockIdx.x * per_wg + threadIdx.x * 16;
    float acc = 0.f;
    const unsigned long long t0 = wall_clock64();
    for (int i = 0; i < steps; ++i) {
        const f4 v = *(const f4 *)(p + (size_t)i * stride);
        acc += v.x;                               // (dependent use: the next load is issued only after this one returned)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (acc == 123.456f) sink[0] = acc;
}
int main()
{
    const size_t bytes = (size_t)8 << 30;
    char *buf; unsigned long long *out; float *sink;
    hipMalloc(&buf, bytes); hipMalloc(&out, 4096 * 8); hipMalloc(&sink, 4);
    hipMemset(buf, 0, bytes);
    hipDeviceSynchronize();
    const int steps = 256;
    for (int wgs : {8, 256, 2048}) {
        for (size_t stride : {(size_t)1024, (size_t)4096, (size_t)8192, (size_t)30720, (size_t)65536, (size_t)262144, (size_t)2097152}) {
            size_t per_wg = stride * steps;                     // every workgroup its own region (cold lines, cold pages)
            if (per_wg * wgs > bytes) per_wg = bytes / wgs / 1024 * 1024;
            int st = steps;
            if ((size_t)st * stride > per_wg) st = (int)(per_wg / stride);
            if (st < 8) continue;
            // flush caches between runs: stream over another part of the buffer is too slow; rely on regions >> L2 + new offsets
            static size_t shift = 0; shift = (shift + 4096 * 3) % 65536;
            hipLaunchKernelGGL(k_walk, dim3(wgs), dim3(64), 0, 0, buf + shift * 0, stride, st, per_wg, out, sink);
            hipDeviceSynchronize();
            std::vector<unsigned long long> h(wgs);
            hipMemcpy(h.data(), out, wgs * 8, hipMemcpyDeviceToHost);
            double s = 0; for (auto v : h) s += (double)v;
            printf("wgs %5d stride %8zu steps %4d: %.1f ns per dependent step (100 MHz clock)\n", wgs, stride, st, s / wgs / st * 10.0);
            // second pass over the same addresses: lines now in L2 / MALL, pages in the TLBs
            hipLaunchKernelGGL(k_walk, dim3(wgs), dim3(64), 0, 0, buf, stride, st, per_wg, out, sink);
            hipDeviceSynchronize();
            hipMemcpy(h.data(), out, wgs * 8, hipMemcpyDeviceToHost);
            s = 0; for (auto v : h) s += (double)v;
            printf("                                   second pass: %.1f ns\n", s / wgs / st * 10.0);
        }
    }
    return 0;
}
