// How many dependent kernel launches per second ONE process gets out of this GPU from K host threads, each on its own stream (K = 1, 2, 4, 8): what bounds K
// independent frame pipelines whose frames are ~80 small launches each (profiles/r06_pipelines.txt). hipcc --offload-arch=gfx950 -O2 -pthread launch_rate.hip
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void tiny(int *p, int spin) { if (threadIdx.x == 0) { int v = 0; for (int i = 0; i < spin; ++i) v += __builtin_amdgcn_s_memtime() & 1; if (v < 0) *p = v; } }
int main(int argc, char **argv)
{
    const int N = 20000;
    int *d = nullptr;
    hipMalloc(&d, 4);
    const bool quick = argc > 1;
    for (int wgs : {1, 256}) for (int spin : {0, 200}) for (int K : {1, 2, 3, 4, 6, 8, 12, 16}) {
        if (quick && (wgs != 1 || spin != 200)) continue;
        std::vector<hipStream_t> st(K);
        for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
        std::atomic<bool> go{false};
        std::atomic<int> ready{0};
        std::vector<std::thread> th;
        for (int k = 0; k < K; ++k) th.emplace_back([&, k] {
            for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(tiny, dim3(wgs), dim3(256), 0, st[k], d, spin);
            hipStreamSynchronize(st[k]);
            ready.fetch_add(1);
            while (!go.load()) std::this_thread::yield();
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(wgs), dim3(256), 0, st[k], d, spin);
            hipStreamSynchronize(st[k]);
        });
        while (ready.load() < K) std::this_thread::yield();
        const auto t0 = std::chrono::steady_clock::now();
        go.store(true);
        for (auto &t : th) t.join();
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("workgroups %3d, ~%d cycles of work, K = %d streams / threads: %.0f launches per second in total (%.2f us per launch per stream)\n", wgs, spin * 40, K, double(K) * N / s, 1e6 * s / N);
        for (auto &s2 : st) hipStreamDestroy(s2);
    }
    return 0;
}
