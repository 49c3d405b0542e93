// experiment (round 3): does hipExtAnyOrderLaunch let a kernel start while its predecessor in the SAME stream still runs, and is a consumer
// that spins on a flag its predecessor sets safe (in-order dispatch: every producer workgroup is placed before any consumer workgroup)?
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ inline unsigned long long wall() { return __builtin_readcyclecounter(); }
__device__ inline unsigned long long wclk() { unsigned long long t; asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }   // 100 MHz

__global__ void producer(unsigned *counter, unsigned long long *stamps, int work_iters)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) stamps[0] = wclk();
    float x = threadIdx.x;
    for (int i = 0; i < work_iters; ++i) x = x * 1.0001f + 0.5f;
    if (x == 12345.f) stamps[7] = 1;
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned t = atomicAdd(counter, 1u);
        if (t == gridDim.x - 1) stamps[1] = wclk();
    }
}

__global__ void consumer(unsigned *counter, unsigned expect, unsigned long long *stamps, unsigned *timeouts, int max_spins)
{
    __shared__ int ok;
    if (threadIdx.x == 0) {
        if (blockIdx.x == 0) stamps[2] = wclk();
        int spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < expect && spins < max_spins) { __builtin_amdgcn_s_sleep(8); ++spins; }
        ok = spins < max_spins;
        if (!ok) atomicAdd(timeouts, 1u);
        if (blockIdx.x == 0) stamps[3] = wclk();
    }
    __syncthreads();
    if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1) stamps[4] = wclk();
}

int main()
{
    unsigned *counter, *timeouts;
    unsigned long long *stamps;
    CK(hipMalloc(&counter, 4)); CK(hipMalloc(&timeouts, 4)); CK(hipMalloc(&stamps, 64));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int mode = 0; mode < 2; ++mode) {
        for (int trial = 0; trial < 3; ++trial) {
            const int gp[3] = {88, 1015, 4000}, gc[3] = {1015, 88, 8000};
            CK(hipMemsetAsync(counter, 0, 4, st)); CK(hipMemsetAsync(timeouts, 0, 4, st)); CK(hipMemsetAsync(stamps, 0, 64, st));
            CK(hipStreamSynchronize(st));
            hipExtLaunchKernelGGL(producer, dim3(gp[trial]), dim3(256), 0, st, nullptr, nullptr, 0, counter, stamps, 20000);
            hipExtLaunchKernelGGL(consumer, dim3(gc[trial]), dim3(256), 0, st, nullptr, nullptr, mode ? hipExtAnyOrderLaunch : 0, counter, unsigned(gp[trial]), stamps, timeouts, 1 << 20);
            CK(hipStreamSynchronize(st));
            unsigned long long h[8]; unsigned to;
            CK(hipMemcpy(h, stamps, 64, hipMemcpyDeviceToHost)); CK(hipMemcpy(&to, timeouts, 4, hipMemcpyDeviceToHost));
            printf("mode %s producer %d wgs consumer %d wgs: producer start 0, producer last-done %.2f us, consumer first-start %.2f us, consumer wg0 released %.2f us, consumer last wg end %.2f us, timeouts %u\n",
                   mode ? "ANYORDER" : "inorder ", gp[trial], gc[trial], (h[1] - h[0]) / 100.0, (double(h[2]) - double(h[0])) / 100.0, (double(h[3]) - double(h[0])) / 100.0, (double(h[4]) - double(h[0])) / 100.0, to);
        }
    }
    // back-to-back chain latency: 10 tiny dependent kernels, in-order vs any-order + flag
    return 0;
}
