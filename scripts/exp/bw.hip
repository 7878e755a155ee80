// Scratch microbenchmark: HBM write-only / read-only / copy ceilings.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
#define IDX size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; size_t s = (size_t)gridDim.x * blockDim.x
__global__ void k_write(f4 *p, size_t n) { IDX; for (; i < n; i += s) { f4 v = {1.f, 2.f, 3.f, (float)i}; p[i] = v; } }
__global__ void k_write_nt(f4 *p, size_t n) { IDX; for (; i < n; i += s) { f4 v = {1.f, 2.f, 3.f, (float)i}; __builtin_nontemporal_store(v, p + i); } }
__global__ void k_write1(float *p, size_t n) { IDX; for (; i < n; i += s) p[i] = (float)i; }
__global__ void k_write1_nt(float *p, size_t n) { IDX; for (; i < n; i += s) __builtin_nontemporal_store((float)i, p + i); }
__global__ void k_read(const f4 *p, size_t n, float *out) { IDX; float a = 0; for (; i < n; i += s) { f4 v = p[i]; a += v.x + v.y + v.z + v.w; } if (a == 123.456f) out[0] = a; }
__global__ void k_read_nt(const f4 *p, size_t n, float *out) { IDX; float a = 0; for (; i < n; i += s) { f4 v = __builtin_nontemporal_load(p + i); a += v.x + v.y + v.z + v.w; } if (a == 123.456f) out[0] = a; }
__global__ void k_copy(const f4 *p, f4 *q, size_t n) { IDX; for (; i < n; i += s) q[i] = p[i]; }
template <typename F> float timeit(F f) { hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b); float best = 1e9; for (int i = 0; i < 5; ++i) { (void)hipEventRecord(a); f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b); float ms; (void)hipEventElapsedTime(&ms, a, b); if (i && ms < best) best = ms; } return best; }
int main() {
    size_t n = (size_t)1920 * 1080 * 256;  // f4 elements -> 8.49 GB
    f4 *a, *b; float *o; (void)hipMalloc(&a, n * 16); (void)hipMalloc(&b, n * 16); (void)hipMalloc(&o, 4); (void)hipMemset(a, 0, n * 16); (void)hipMemset(b, 0, n * 16);
    for (int g : {2048, 16384}) {
        float t;
        t = timeit([&] { hipLaunchKernelGGL(k_write, dim3(g), dim3(256), 0, 0, a, n); }); printf("grid %6d write16   %.3f ms  %.0f GB/s\n", g, t, n * 16 / t / 1e6);
        t = timeit([&] { hipLaunchKernelGGL(k_write_nt, dim3(g), dim3(256), 0, 0, a, n); }); printf("grid %6d write16nt %.3f ms  %.0f GB/s\n", g, t, n * 16 / t / 1e6);
        t = timeit([&] { hipLaunchKernelGGL(k_write1, dim3(g), dim3(256), 0, 0, (float *)a, n); }); printf("grid %6d write4    %.3f ms  %.0f GB/s\n", g, t, n * 4 / t / 1e6);
        t = timeit([&] { hipLaunchKernelGGL(k_write1_nt, dim3(g), dim3(256), 0, 0, (float *)a, n); }); printf("grid %6d write4nt  %.3f ms  %.0f GB/s\n", g, t, n * 4 / t / 1e6);
        t = timeit([&] { hipLaunchKernelGGL(k_read, dim3(g), dim3(256), 0, 0, a, n, o); }); printf("grid %6d read16    %.3f ms  %.0f GB/s\n", g, t, n * 16 / t / 1e6);
        t = timeit([&] { hipLaunchKernelGGL(k_read_nt, dim3(g), dim3(256), 0, 0, a, n, o); }); printf("grid %6d read16nt  %.3f ms  %.0f GB/s\n", g, t, n * 16 / t / 1e6);
        t = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, 0, a, b, n / 2); }); printf("grid %6d copy16    %.3f ms  %.0f GB/s (R+W)\n", g, t, n * 16 / t / 1e6);
    }
    return 0;
}
