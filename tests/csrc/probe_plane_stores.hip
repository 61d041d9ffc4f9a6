// What do the geo forward's per-pixel output stores cost, and which lane -> pixel shape makes them cheapest?  The kernel writes 35 + planes of
// a 1080p frame (cam_feat 20, warped 15: 290 MB) from waves that each own a half tile (16 x 8 pixels), in planar CHW layout.  Shapes:
//   q8x8    lane l = pixel (l % 8, l / 8) of an 8 x 8 quadrant, two store instructions per plane: eight 32-byte row pieces each (today's epilogue)
//   r16x4   lane l = pixel (l % 16, l / 16): two instructions per plane, four 64-byte row pieces each
//   v4      lane l < 32 owns four horizontally adjacent pixels (float4): ONE instruction per plane, eight 64-byte pieces, half the lanes idle
//   v4x2    the same with lanes 32..63 writing the NEXT plane: one instruction per two planes
//   linear  64 consecutive floats per instruction (not a tile shape: the ceiling for dword stores)
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/pps tests/csrc/probe_plane_stores.hip && /tmp/pps
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int W = 1920, H = 1080, GX = 120, GY = 68, NPL = 35;

__device__ __forceinline__ void tile_of(int b, int& tx0, int& ty0)
{   // 8 x 8-tile blocks dealt round-robin to the XCDs (the library's map for this kernel), two waves per tile
    const int xcd = b & 7, idx = b >> 3;
    const int per = 128, blk = (idx / per) * 8 + xcd, within = idx % per;
    const int nbx = (GX + 7) / 8, t = within >> 1, sub = within & 1;
    const int tx = (blk % nbx) * 8 + t % 8, ty = (blk / nbx) * 8 + t / 8;
    tx0 = (tx < GX && ty < GY) ? tx * 16 : -1; ty0 = ty * 16 + sub * 8;
}

template <int SHAPE>
__global__ void __launch_bounds__(64) k_store(float* __restrict__ out, int npl, float seed)
{
    int tx0, ty0; tile_of(blockIdx.x, tx0, ty0);
    if (tx0 < 0) return;
    const int lane = threadIdx.x;
    const size_t HW = (size_t)W * H;
    if (SHAPE == 0) {
#pragma unroll 1
        for (int p = 0; p < npl; p++)
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int x = tx0 + q * 8 + (lane & 7), y = ty0 + (lane >> 3);
                if (y < H) out[(size_t)p * HW + (size_t)y * W + x] = seed + (float)(p + lane);
            }
    } else if (SHAPE == 1) {
#pragma unroll 1
        for (int p = 0; p < npl; p++)
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const int x = tx0 + (lane & 15), y = ty0 + q * 4 + (lane >> 4);
                if (y < H) out[(size_t)p * HW + (size_t)y * W + x] = seed + (float)(p + lane);
            }
    } else if (SHAPE == 2) {
#pragma unroll 1
        for (int p = 0; p < npl; p++) {
            const int x = tx0 + (lane & 3) * 4, y = ty0 + ((lane & 31) >> 2);
            if (lane < 32 && y < H) *reinterpret_cast<float4*>(out + (size_t)p * HW + (size_t)y * W + x) = make_float4(seed, seed + p, seed + lane, seed);
        }
    } else if (SHAPE == 3) {
#pragma unroll 1
        for (int p = 0; p < npl; p += 2) {
            const int pp = p + (lane >> 5);
            const int x = tx0 + (lane & 3) * 4, y = ty0 + ((lane & 31) >> 2);
            if (pp < npl && y < H) *reinterpret_cast<float4*>(out + (size_t)pp * HW + (size_t)y * W + x) = make_float4(seed, seed + p, seed + lane, seed);
        }
    } else {
#pragma unroll 1
        for (int p = 0; p < npl; p++)
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const size_t i = ((size_t)blockIdx.x * 2 + q) * 64 + lane;
                if (i < HW) out[(size_t)p * HW + i] = seed + (float)(p + lane);
            }
    }
}

int main()
{
    const size_t HW = (size_t)W * H;
    float* out; hipMalloc(&out, HW * NPL * sizeof(float));
    char* junk; const size_t junk_bytes = (size_t)1024 << 20; hipMalloc(&junk, junk_bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[5] = {"q8x8", "r16x4", "v4", "v4x2", "linear"};
    const int grid = ((GX + 7) / 8) * ((GY + 7) / 8);
    const int nblocks = ((grid + 7) / 8) * 8 * 128;
    for (int npl : {35, 28, 14}) {
        for (int shape = 0; shape < 5; shape++) {
            float best = 1e9f;
            for (int rep = 0; rep < 6; rep++) {
                hipMemsetAsync(junk, rep, junk_bytes, 0);          // dirty caches, as inside a step
                hipEventRecord(e0);
                const dim3 g(shape == 4 ? (unsigned)((HW + 127) / 128) : (unsigned)nblocks);
                if (shape == 0) hipLaunchKernelGGL(k_store<0>, g, dim3(64), 0, 0, out, npl, (float)rep);
                if (shape == 1) hipLaunchKernelGGL(k_store<1>, g, dim3(64), 0, 0, out, npl, (float)rep);
                if (shape == 2) hipLaunchKernelGGL(k_store<2>, g, dim3(64), 0, 0, out, npl, (float)rep);
                if (shape == 3) hipLaunchKernelGGL(k_store<3>, g, dim3(64), 0, 0, out, npl, (float)rep);
                if (shape == 4) hipLaunchKernelGGL(k_store<4>, g, dim3(64), 0, 0, out, npl, (float)rep);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("%2d planes  %-7s %8.1f us  %7.0f GB/s\n", npl, names[shape], best * 1e3, (double)HW * npl * 4 / (best * 1e-3) / 1e9);
        }
    }
    return 0;
}
