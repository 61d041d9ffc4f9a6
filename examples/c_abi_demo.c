/* The C ABI without Python or torch: renders 3 Gaussians into a 64x48 image with ibgs_forward, back-propagates a
 * constant dL/dC with ibgs_backward and prints a few numbers.  Only the HIP runtime (for device memory) and
 * libibgs_rast.so are linked -- the same calls a cgo / JNI / ctypes binding of the reference would make.
 *
 *   gcc -D__HIP_PLATFORM_AMD__ -I /opt/rocm/include -I include examples/c_abi_demo.c -L ibgs_amd -libgs_rast \
 *       -L /opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/ibgs_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/c_abi_demo && /tmp/c_abi_demo
 *   (plain C compiler; -D__HIP_PLATFORM_AMD__ is what hip_runtime_api.h itself asks for)
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ibgs_rast.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

static void* dev(const void* host, size_t bytes)
{
    void* p = NULL;
    CHECK(hipMalloc(&p, bytes ? bytes : 4));
    if (host) CHECK(hipMemcpy(p, host, bytes, hipMemcpyHostToDevice)); else CHECK(hipMemset(p, 0, bytes ? bytes : 4));
    return p;
}

static char* g_binning = NULL;
static char* alloc_binning(size_t bytes, void* user) { (void)user; if (g_binning) (void)hipFree(g_binning); CHECK(hipMalloc((void**)&g_binning, bytes)); return g_binning; }

int main(void)
{
    enum { P = 3, W = 64, H = 48 };
    /* camera at the origin looking down +z: world == view; pinhole with tan(fov/2) = 0.5 / 0.375 */
    const float tanx = 0.5f, tany = 0.375f, zn = 0.01f, zf = 100.0f;
    float vm[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};           /* world_view_transform, row-major as the reference stores it */
    float pm[16] = {0};                                                      /* full_proj_transform = view @ projection (getProjectionMatrix, transposed) */
    pm[0] = 1.0f / tanx; pm[5] = 1.0f / tany; pm[10] = zf / (zf - zn); pm[11] = 1.0f; pm[14] = -(zf * zn) / (zf - zn);
    float campos[3] = {0, 0, 0}, bg[3] = {0.1f, 0.2f, 0.3f};
    float means[P * 3] = {0.0f, 0.0f, 3.0f, 0.6f, 0.2f, 4.0f, -0.5f, -0.3f, 5.0f};
    float scales[P * 3] = {0.3f, 0.2f, 0.1f, 0.25f, 0.25f, 0.25f, 0.5f, 0.1f, 0.3f};
    float rots[P * 4] = {1, 0, 0, 0, 0.9238795f, 0, 0.3826834f, 0, 1, 0, 0, 0};
    float opac[P] = {0.8f, 0.6f, 0.9f};
    float colors[P * 3] = {1, 0, 0, 0, 1, 0, 0, 0, 1};

    ibgs_forward_args f; memset(&f, 0, sizeof f);
    if (ibgs_sizeof_forward_args() != sizeof f) { fprintf(stderr, "header / library mismatch\n"); return 2; }
    f.P = P; f.W = W; f.H = H;
    f.means3D = dev(means, sizeof means); f.colors_precomp = dev(colors, sizeof colors); f.opacities = dev(opac, sizeof opac);
    f.scales = dev(scales, sizeof scales); f.rotations = dev(rots, sizeof rots); f.scale_modifier = 1.0f;
    f.bg = dev(bg, sizeof bg); f.viewmatrix = dev(vm, sizeof vm); f.projmatrix = dev(pm, sizeof pm); f.campos = dev(campos, sizeof campos);
    f.tanfovx = tanx; f.tanfovy = tany; f.n_src = 1; f.buffer_length = 4;
    f.geom_bytes = ibgs_required_geom(P); f.geom = dev(NULL, f.geom_bytes);
    f.img_bytes = ibgs_required_img(W, H); f.img = dev(NULL, f.img_bytes);
    f.binning_alloc = alloc_binning;
    float* d_color = dev(NULL, sizeof(float) * 3 * W * H); int32_t* d_radii = dev(NULL, sizeof(int32_t) * P);
    f.out_color = d_color; f.radii = d_radii;
    const int64_t R = ibgs_forward(&f);
    if (R < 0) { fprintf(stderr, "ibgs_forward: %s\n", ibgs_last_error()); return 1; }

    static float color[3 * W * H]; int32_t radii[P];
    CHECK(hipMemcpy(color, d_color, sizeof color, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(radii, d_radii, sizeof radii, hipMemcpyDeviceToHost));
    double sum = 0; for (int i = 0; i < 3 * W * H; i++) sum += color[i];
    printf("%s: R = %lld tile entries, radii = %d %d %d, centre pixel = (%.4f %.4f %.4f), image sum = %.4f\n", ibgs_version(), (long long)R,
           radii[0], radii[1], radii[2], color[(H / 2) * W + W / 2], color[W * H + (H / 2) * W + W / 2], color[2 * W * H + (H / 2) * W + W / 2], sum);

    ibgs_backward_args b; memset(&b, 0, sizeof b);
    b.P = P; b.W = W; b.H = H; b.R = R;
    b.means3D = f.means3D; b.colors_precomp = f.colors_precomp; b.scales = f.scales; b.rotations = f.rotations; b.scale_modifier = 1.0f;
    b.bg = f.bg; b.viewmatrix = f.viewmatrix; b.projmatrix = f.projmatrix; b.campos = f.campos; b.tanfovx = tanx; b.tanfovy = tany; b.n_src = 1;
    b.radii = d_radii; b.geom = f.geom; b.binning = g_binning; b.img = f.img;
    static float ones[3 * W * H]; for (int i = 0; i < 3 * W * H; i++) ones[i] = 1.0f;
    b.dL_dcolor = dev(ones, sizeof ones);
    b.grad_acc = dev(NULL, sizeof(float) * 16 * P);
    float *g2 = dev(NULL, 12 * P), *g2a = dev(NULL, 12 * P), *gop = dev(NULL, 4 * P), *gcol = dev(NULL, 12 * P), *gm = dev(NULL, 12 * P),
          *gcov = dev(NULL, 24 * P), *gs = dev(NULL, 12 * P), *gr = dev(NULL, 16 * P);
    b.dL_dmean2D = g2; b.dL_dmean2D_abs = g2a; b.dL_dopacity = gop; b.dL_dcolors = gcol; b.dL_dmean3D = gm; b.dL_dcov3D = gcov; b.dL_dscale = gs; b.dL_drot = gr;
    if (ibgs_backward(&b) < 0) { fprintf(stderr, "ibgs_backward: %s\n", ibgs_last_error()); return 1; }
    float hop[P], hcol[3 * P];
    CHECK(hipMemcpy(hop, gop, sizeof hop, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(hcol, gcol, sizeof hcol, hipMemcpyDeviceToHost));
    printf("dL/dopacity = %.4f %.4f %.4f   dL/dcolour[0] = %.4f %.4f %.4f (= its blend weight summed over the image, three times)\n",
           hop[0], hop[1], hop[2], hcol[0], hcol[1], hcol[2]);
    const int ok = R > 0 && radii[0] > 0 && isfinite(sum) && hcol[0] > 0 && fabsf(hcol[0] - hcol[1]) < 1e-4f * hcol[0];
    printf("%s\n", ok ? "c_abi_demo OK" : "c_abi_demo FAILED");
    return ok ? 0 : 1;
}
