// Dependent chain of small kernels: launched one by one into a stream vs replayed as a captured hipGraph (the question: does a graph shorten the kernel-to-kernel
// boundary of a chain the host is already far ahead of -- a mapper frame is ~70 such launches). hipcc --offload-arch=gfx950 -O2 graph_rate.hip -o graph_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void tiny(int *p, int spin) { if (threadIdx.x == 0) { int v = 0; for (int i = 0; i < spin; ++i) v += __builtin_amdgcn_s_memtime() & 1; if (v < 0) *p = v; } }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    int *d = nullptr;
    (void)hipMalloc(&d, 4);
    hipStream_t st;
    (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    const int CH = 70, REP = 300;
    for (int wgs : {1, 88, 1024}) for (int spin : {0, 400, 2000}) {
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(tiny, dim3(wgs), dim3(256), 0, st, d, spin);
        (void)hipStreamSynchronize(st);
        double t0 = now();
        for (int r = 0; r < REP; ++r) for (int i = 0; i < CH; ++i) hipLaunchKernelGGL(tiny, dim3(wgs), dim3(256), 0, st, d, spin);
        (void)hipStreamSynchronize(st);
        const double t_stream = (now() - t0) / (REP * CH);
        hipGraph_t g; hipGraphExec_t ge;
        (void)hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < CH; ++i) hipLaunchKernelGGL(tiny, dim3(wgs), dim3(256), 0, st, d, spin);
        (void)hipStreamEndCapture(st, &g);
        if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { std::printf("instantiate failed\n"); return 1; }
        for (int r = 0; r < 5; ++r) (void)hipGraphLaunch(ge, st);
        (void)hipStreamSynchronize(st);
        t0 = now();
        for (int r = 0; r < REP; ++r) (void)hipGraphLaunch(ge, st);
        (void)hipStreamSynchronize(st);
        const double t_graph = (now() - t0) / (REP * CH);
        // one chain alone, host waits for it (the latency a frame sees)
        t0 = now();
        for (int r = 0; r < 50; ++r) { for (int i = 0; i < CH; ++i) hipLaunchKernelGGL(tiny, dim3(wgs), dim3(256), 0, st, d, spin); (void)hipStreamSynchronize(st); }
        const double l_stream = (now() - t0) / 50;
        t0 = now();
        for (int r = 0; r < 50; ++r) { (void)hipGraphLaunch(ge, st); (void)hipStreamSynchronize(st); }
        const double l_graph = (now() - t0) / 50;
        std::printf("workgroups %4d, spin %4d: per kernel of a %d-kernel chain: stream %.2f us, graph %.2f us | one chain + wait: stream %.1f us, graph %.1f us\n", wgs, spin, CH,
                    1e6 * t_stream, 1e6 * t_graph, 1e6 * l_stream, 1e6 * l_graph);
        (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    }
    return 0;
}
