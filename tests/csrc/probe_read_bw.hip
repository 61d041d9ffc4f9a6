// What does a SHORT read-mostly kernel get out of HBM on this chip?  (The 6.3 TB/s "copy rate" of the guide is a long read+write
// stream.)  Reads N bytes once with 16-byte-per-lane loads, in three shapes, and reports useful GB/s per size:
//   stream  grid-stride over the whole array, one float4 per lane per iteration (fully coalesced, many blocks)
//   block12 one wave per contiguous 12 KB block, twelve 1 KB loads in flight, then a reduction (the sh_color_kernel shape)
//   block12h the same with 40 % of the 192-byte rows masked out (their pieces are not loaded)
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/prb tests/csrc/probe_read_bw.hip && /tmp/prb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void __launch_bounds__(256) k_stream(const float4* __restrict__ src, size_t n4, float* out)
{
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { const float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 123.456f) out[0] = acc;
}

template <bool HOLES>
__global__ void __launch_bounds__(256) k_block12(const float4* __restrict__ src, size_t nblocks, float* out)
{
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wave >= nblocks) return;
    const int lane = threadIdx.x & 63;
    const float4* s = src + wave * 768;
    // a fixed pseudo-random 60 % of the rows is alive
    uint64_t alive = 0x9E3779B97F4A7C15ull * (wave + 1); alive ^= alive >> 29; alive |= (alive << 7) & 0x5555555555555555ull;
    float4 v[12];
#pragma unroll
    for (int it = 0; it < 12; it++) {
        const int q = it * 64 + lane, row = q / 12;
        const bool ok = !HOLES || ((alive >> row) & 1ull);
        v[it] = ok ? s[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float acc = 0.f;
#pragma unroll
    for (int it = 0; it < 12; it++) acc += v[it].x + v[it].y + v[it].z + v[it].w;
    if (acc == 123.456f) out[0] = acc;
}

int main(int argc, char** argv)
{
    float* out; hipMalloc(&out, 4);
    // "cold" mode (any argument): 1.5 GB of other memory is written between the repetitions, as the kernels around a real call do --
    // nothing of the array is left in L2 / the 256 MB infinity cache, and the caches are full of dirty lines when the reads start
    const bool cold = argc > 1;
    char* junk = nullptr; const size_t junk_bytes = (size_t)1536 << 20;
    if (cold) hipMalloc(&junk, junk_bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (size_t mb : {48, 192, 768, 3072}) {
        const size_t bytes = mb << 20, n4 = bytes / 16, nblocks = bytes / 12288;
        float4* src; hipMalloc(&src, bytes); hipMemset(src, 0, bytes);
        for (int kind = 0; kind < 3; kind++) {
            float best = 1e9f;
            for (int rep = 0; rep < 8; rep++) {
                if (cold) hipMemsetAsync(junk, rep, junk_bytes, 0);
                hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(k_stream, dim3(256 * 16), dim3(256), 0, 0, src, n4, out);
                else if (kind == 1) hipLaunchKernelGGL(k_block12<false>, dim3((unsigned)((nblocks + 3) / 4)), dim3(256), 0, 0, src, nblocks, out);
                else hipLaunchKernelGGL(k_block12<true>, dim3((unsigned)((nblocks + 3) / 4)), dim3(256), 0, 0, src, nblocks, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep >= 2 && ms < best) best = ms;
            }
            printf("%5zu MB  %-9s %8.1f us  %7.0f GB/s (of the full array)\n", mb, kind == 0 ? "stream" : (kind == 1 ? "block12" : "block12h"), best * 1e3f, bytes / (best * 1e-3) / 1e9);
        }
        hipFree(src);
    }
    return 0;
}
