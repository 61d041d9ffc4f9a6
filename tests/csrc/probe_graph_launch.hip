// What would a hipGraph buy the forward's launch sequence?  Host time to queue N small kernels (a) one hipLaunchKernelGGL at a time, (b) as ONE hipGraphLaunch
// of a graph captured from the same sequence, and (c) the same graph after its kernel nodes' parameters were updated (what a forward whose pointers or
// camera changed would have to do: hipGraphExecKernelNodeSetParams per node).  GPU idle before each measurement; the kernels do next to nothing.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/pgl tests/csrc/probe_graph_launch.hip && /tmp/pgl
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void k_small(float* p, int i) { if (threadIdx.x == 0 && blockIdx.x == 0) p[i] += 1.0f; }

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    float* buf; hipMalloc(&buf, 4096); hipMemset(buf, 0, 4096);
    hipStream_t s; hipStreamCreate(&s);
    for (int N : {8, 15, 30}) {
        // (a) plain launches
        double best_a = 1e9;
        for (int rep = 0; rep < 50; rep++) {
            hipStreamSynchronize(s);
            const double t0 = now_us();
            for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s, buf, i);
            const double t1 = now_us();
            if (t1 - t0 < best_a) best_a = t1 - t0;
        }
        // (b) graph
        hipGraph_t g; hipGraphExec_t ge;
        hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
        for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s, buf, i);
        hipStreamEndCapture(s, &g);
        if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) { printf("instantiate failed\n"); return 1; }
        double best_b = 1e9, sum_b = 0;
        for (int rep = 0; rep < 50; rep++) {
            hipStreamSynchronize(s);
            const double t0 = now_us();
            hipGraphLaunch(ge, s);
            const double t1 = now_us();
            if (t1 - t0 < best_b) best_b = t1 - t0;
            sum_b += t1 - t0;
        }
        // GPU-side: time from launch to completion of the N kernels, both ways
        double gpu_a = 1e9, gpu_b = 1e9;
        for (int rep = 0; rep < 20; rep++) {
            hipStreamSynchronize(s); double t0 = now_us();
            for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_small, dim3(64), dim3(256), 0, s, buf, i);
            hipStreamSynchronize(s); double t1 = now_us(); if (t1 - t0 < gpu_a) gpu_a = t1 - t0;
            t0 = now_us(); hipGraphLaunch(ge, s); hipStreamSynchronize(s); t1 = now_us(); if (t1 - t0 < gpu_b) gpu_b = t1 - t0;
        }
        // (c) update every node's parameters, then launch
        size_t nn = 0; hipGraphGetNodes(g, nullptr, &nn);
        std::vector<hipGraphNode_t> nodes(nn); hipGraphGetNodes(g, nodes.data(), &nn);
        double best_c = 1e9;
        for (int rep = 0; rep < 50; rep++) {
            hipStreamSynchronize(s);
            const double t0 = now_us();
            for (size_t k = 0; k < nn; k++) {
                hipKernelNodeParams kp;
                if (hipGraphKernelNodeGetParams(nodes[k], &kp) != hipSuccess) continue;
                int idx = (int)k; void* args[2] = {&buf, &idx};
                kp.kernelParams = args;
                hipGraphExecKernelNodeSetParams(ge, nodes[k], &kp);
            }
            hipGraphLaunch(ge, s);
            const double t1 = now_us();
            if (t1 - t0 < best_c) best_c = t1 - t0;
        }
        printf("N = %2d kernels: host time to queue -- plain launches %.1f us (%.2f each), one graph launch %.1f us (mean %.1f), graph after updating all node parameters %.1f us; "
               "launch-to-done on an idle GPU: plain %.1f us, graph %.1f us\n", N, best_a, best_a / N, best_b, sum_b / 50, best_c, gpu_a, gpu_b);
        hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
    return 0;
}
