// Every workgroup -> tile layout of ibgs_amd/csrc/common.h (tile_map_item / tile_map_grid) must hand out every (tile, wave of the tile)
// exactly once, whatever the grid: round-robin, runs of N items per XCD, X x Y-tile blocks per XCD, with 1, 2 or 4 waves per tile.
// Build + run: hipcc --offload-arch=gfx950 -O2 -I ibgs_amd/csrc -o /tmp/ttm tests/csrc/test_tile_map.hip && /tmp/ttm   (prints "tile map ok")
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "common.h"

using namespace ibgs;
namespace ibgs { void set_error(const char*, ...) {} }

__global__ void visit(TileMap m, int gx, int gy, int ipt, unsigned* seen)
{
    int tile, sub;
    if (threadIdx.x == 0 && tile_map_item(m, blockIdx.x, gx, gy, ipt, tile, sub)) atomicAdd(&seen[tile * ipt + sub], 1u);
}

int main()
{
    const int grids[][2] = {{1, 1}, {3, 2}, {7, 5}, {16, 16}, {25, 25}, {50, 50}, {120, 68}, {513, 129}, {9, 1}, {1, 33}};
    const TileMap maps[] = {{TMAP_RR, 1, 1, 1}, {TMAP_GROUP, 2, 1, 1}, {TMAP_GROUP, 16, 1, 1}, {TMAP_GROUP, 1024, 1, 1}, {TMAP_BLOCK, 1, 2, 2},
                            {TMAP_BLOCK, 1, 4, 4}, {TMAP_BLOCK, 1, 8, 4}, {TMAP_BLOCK, 1, 8, 8}, {TMAP_BLOCK, 1, 3, 5}};
    unsigned* d; hipMalloc(&d, sizeof(unsigned) * 513 * 129 * 4);
    int bad = 0, cases = 0;
    for (auto& g : grids) for (auto& m : maps) for (int ipt : {1, 2, 4}) {
        const int gx = g[0], gy = g[1], n = gx * gy * ipt;
        hipMemset(d, 0, sizeof(unsigned) * n);
        const int grid = tile_map_grid(m, gx, gy, ipt);
        hipLaunchKernelGGL(visit, dim3(grid), dim3(64), 0, 0, m, gx, gy, ipt, d);
        std::vector<unsigned> h(n);
        hipMemcpy(h.data(), d, sizeof(unsigned) * n, hipMemcpyDeviceToHost);
        int wrong = 0;
        for (int i = 0; i < n; i++) wrong += h[i] != 1u;
        if (wrong) { printf("grid %dx%d ipt %d mode %d g %d b %dx%d: %d of %d items not visited exactly once\n", gx, gy, ipt, m.mode, m.g, m.bx, m.by, wrong, n); bad++; }
        cases++;
    }
    printf(bad ? "tile map WRONG in %d of %d cases\n" : "tile map ok (%d of %d cases wrong)\n", bad, cases);
    return bad ? 1 : 0;
}
