// rays.hip -- a1-a3: pixel grid -> camera directions -> world rays -> unit-sphere far bound.
// Reference util/ray.py:8-12 (grid), :25-31 (directions, no half-pixel offset, +z forward),
// :46-54 (rotate, normalise, origin), :81-99 (sphere intersection); record layout
// dataset/many_object_scenes.py:191-199.
#include "clift_dev.h"

struct CamP {
    float fx, fy, cx, cy;
    float R[9];
    float t[3];
    float nearp;
};

__global__ __launch_bounds__(256) void k_gen_rays(CamP c, int H, int W, float* __restrict__ rays, int* __restrict__ bad) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= H * W) return;
    const int j = p / W, i = p - j * W;
    const float dx = ((float)i - c.cx) / c.fx, dy = ((float)j - c.cy) / c.fy, dz = 1.f;
    float d[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) d[a] = dx * c.R[a * 3 + 0] + dy * c.R[a * 3 + 1] + dz * c.R[a * 3 + 2];
    const float n = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
#pragma unroll
    for (int a = 0; a < 3; ++a) d[a] = d[a] / n;
    const float od = c.t[0] * d[0] + c.t[1] * d[1] + c.t[2] * d[2];
    const float dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    const float oo = c.t[0] * c.t[0] + c.t[1] * c.t[1] + c.t[2] * c.t[2];
    const float disc = od * od + (1.f - oo) * dd;
    if (!(disc >= 0.f) && bad) atomicAdd(bad, 1);
    const float farv = (sqrtf(disc) - od) / dd;
    float4* o = reinterpret_cast<float4*>(rays + (size_t)p * 8);
    o[0] = make_float4(c.t[0], c.t[1], c.t[2], d[0]);
    o[1] = make_float4(d[1], d[2], c.nearp, farv);
}

extern "C" int clift_gen_rays(int H, int W, const float* h_K9, const float* h_c2w16, float near_plane, float* rays,
                              int* bad_count, clift_stream_t s) {
    CLIFT_REQUIRE(H > 0 && W > 0, "clift_gen_rays: bad image size %dx%d", H, W);
    CamP c;
    c.fx = h_K9[0]; c.fy = h_K9[4]; c.cx = h_K9[2]; c.cy = h_K9[5];
    for (int a = 0; a < 3; ++a) {
        for (int b = 0; b < 3; ++b) c.R[a * 3 + b] = h_c2w16[a * 4 + b];
        c.t[a] = h_c2w16[a * 4 + 3];
    }
    c.nearp = near_plane;
    k_gen_rays<<<cdiv((long)H * W, 256), 256, 0, as_stream(s)>>>(c, H, W, rays, bad_count);
    return clift_check_launch("clift_gen_rays");
}
