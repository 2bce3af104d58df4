// cabi_demo.cpp - libswnerf_hip.so driven from plain C++ (no Python, no torch): the C ABI of include/swnerf.h is the
// drop-in boundary, PyTorch is only one possible owner of the device memory.  Renders a small lego-like frame with seeded
// weights: get_rays -> ray batch -> coarse pass (+ hierarchical resampling) -> fine pass, and dumps inputs and outputs so
// that tests/test_00_bench_launcher.py can replay the same render through the Python mirrors and compare bit for bit.
//
//   hipcc --offload-arch=gfx950 -O2 -I include examples/cabi_demo.cpp -L sw-nerf_amd/swnerf -lswnerf_hip \
//         -Wl,-rpath,$PWD/sw-nerf_amd/swnerf -o cabi_demo && ./cabi_demo dump.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "swnerf.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define SW_OK(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #x, r_, swnerf_last_error()); return 3; } } while (0)

static unsigned long long g_state = 0x9E3779B97F4A7C15ull;
static float uniform01() {                       // splitmix64 -> [0,1)
    unsigned long long z = (g_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    return (float)((z >> 40) * (1.0 / 16777216.0));
}
static float normal01() { float u1 = uniform01() + 1e-7f, u2 = uniform01(); return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2); }

int main(int argc, char** argv) {
    const int H = 16, W = 16, N = H * W, S = 64, NI = 128, LP = 10, LD = 4;
    // the 24 tensors of one vallina_NeRF in state_dict order (include/swnerf.h), He-normal weights, small biases
    const int shapes[12][2] = {{256, 63}, {256, 256}, {256, 256}, {256, 256}, {256, 256}, {256, 319}, {256, 256}, {256, 256},
                               {128, 283}, {256, 256}, {1, 256}, {3, 128}};
    std::vector<std::vector<float>> host[2];
    std::vector<float*> dev_params[2];
    float* packed[2];
    for (int net = 0; net < 2; ++net) {
        for (int t = 0; t < 12; ++t) {
            const int o = shapes[t][0], i = shapes[t][1];
            std::vector<float> w((size_t)o * i), b(o);
            const float sd = sqrtf(2.f / i);
            for (auto& v : w) v = normal01() * sd;
            for (auto& v : b) v = normal01() * 0.05f;
            if (t == 10) {                                                // alpha_linear: positive weights on the (non-negative) features
                for (auto& v : w) v = fabsf(v) * 0.1f;                    // and a negative bias -> a fog of varying density,
                b[0] = net ? -0.3f : -0.25f;                              // neither empty nor opaque (sigma ~ 0.1 .. 1 over a path of 4)
            }
            host[net].push_back(w); host[net].push_back(b);
        }
        for (auto& h : host[net]) {
            float* d; HIP_OK(hipMalloc(&d, h.size() * sizeof(float)));
            HIP_OK(hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
            dev_params[net].push_back(d);
        }
        HIP_OK(hipMalloc(&packed[net], swnerf_packed_floats(SWNERF_NET_CANON) * sizeof(float)));
        SW_OK(swnerf_pack_net(SWNERF_NET_CANON, dev_params[net].data(), LP, LD, 0, packed[net], nullptr));
    }
    // camera: 4 units from the origin on +z looking down -z (c2w = [I | (0,0,4)]), blender-lego field of view
    const float c2w[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 4.f};
    const double focal = 0.5 * W / tan(0.5 * 0.6911112070083618);
    float *rays_o, *rays_d, *rb, *z_fine, *z_std, *rgb0, *rgb, *disp, *acc;
    HIP_OK(hipMalloc(&rays_o, N * 3 * sizeof(float))); HIP_OK(hipMalloc(&rays_d, N * 3 * sizeof(float)));
    HIP_OK(hipMalloc(&rb, N * 11 * sizeof(float))); HIP_OK(hipMalloc(&z_fine, (size_t)N * (S + NI) * sizeof(float)));
    HIP_OK(hipMalloc(&z_std, N * sizeof(float))); HIP_OK(hipMalloc(&rgb0, N * 3 * sizeof(float)));
    HIP_OK(hipMalloc(&rgb, N * 3 * sizeof(float))); HIP_OK(hipMalloc(&disp, N * sizeof(float))); HIP_OK(hipMalloc(&acc, N * sizeof(float)));
    SW_OK(swnerf_get_rays(H, W, focal, focal, 0.5 * W, 0.5 * H, 0, c2w, 0, N, rays_o, rays_d, nullptr));
    SW_OK(swnerf_pack_ray_batch(rays_o, rays_d, N, 2.0, 6.0, 0, 0.0, 0, H, W, focal, rb, nullptr));
    swnerf_pass_args a = {};
    a.ray_batch = rb; a.n_rays = N; a.cols = 11; a.kind = SWNERF_NET_CANON; a.packed = packed[0];
    a.L_pos = LP; a.L_dir = LD; a.n_samples = S; a.white_bkgd = 1; a.rgb_map = rgb0;
    a.n_importance = NI; a.z_fine = z_fine; a.z_std = z_std;
    SW_OK(swnerf_render_pass(&a, nullptr));                               // coarse pass + resampling
    swnerf_pass_args f = {};
    f.ray_batch = rb; f.n_rays = N; f.cols = 11; f.kind = SWNERF_NET_CANON; f.packed = packed[1];
    f.L_pos = LP; f.L_dir = LD; f.n_samples = S + NI; f.z_vals = z_fine; f.white_bkgd = 1;
    f.rgb_map = rgb; f.disp_map = disp; f.acc_map = acc;
    SW_OK(swnerf_render_pass(&f, nullptr));                               // fine pass on the sorted union
    HIP_OK(hipDeviceSynchronize());
    std::vector<float> h_rgb(N * 3), h_rgb0(N * 3), h_acc(N);
    HIP_OK(hipMemcpy(h_rgb.data(), rgb, N * 3 * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(h_rgb0.data(), rgb0, N * 3 * sizeof(float), hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(h_acc.data(), acc, N * sizeof(float), hipMemcpyDeviceToHost));
    double sum = 0, amin = 1e9, amax = -1e9;
    for (float v : h_rgb) { if (!(v >= -1e-6f && v <= 1.f + 1e-5f)) { fprintf(stderr, "rgb out of range: %g\n", v); return 4; } sum += v; }
    for (float v : h_acc) { amin = v < amin ? v : amin; amax = v > amax ? v : amax; }
    printf("cabi_demo: %d rays, 64+128 samples, version %d: mean rgb %.6f, acc in [%.3f, %.3f]\n", N, swnerf_version(), sum / (N * 3), amin, amax);
    if (argc > 1) {                                                       // weights (2 x 24 tensors), then rgb0, rgb, acc
        FILE* fp = fopen(argv[1], "wb");
        if (!fp) return 5;
        for (int net = 0; net < 2; ++net) for (auto& h : host[net]) fwrite(h.data(), sizeof(float), h.size(), fp);
        fwrite(h_rgb0.data(), sizeof(float), h_rgb0.size(), fp); fwrite(h_rgb.data(), sizeof(float), h_rgb.size(), fp);
        fwrite(h_acc.data(), sizeof(float), h_acc.size(), fp);
        fclose(fp);
    }
    return 0;
}
