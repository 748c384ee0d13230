// Grid maintenance kernels: what prune_grid / accelerate call into.
//   k_dilate            misc_kernel.cu:24-58    3x3x3 binary dilation, edges clamped
//   k_accel_dist_set    misc_kernel.cu:113-144  marks, level by level, the 2^k-cells that hold a vertex
//   k_accel_dist_prop   misc_kernel.cu:147-180  empty links become -(1 + number of empty pyramid levels)
//   k_weight_render     misc_kernel.cu:186-284, 286-400, 888-937  per-vertex max of the compositing weight (dense
//                       density volume) / of the transmittance in front of the sample (sparse grid) over a camera's rays
// One voxel (or pixel) per thread.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace asurf {

__global__ void __launch_bounds__(256)
k_dilate(const uint8_t* __restrict__ grid, uint8_t* __restrict__ out, int sx, int sy, int sz) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (long long)sx * sy * sz) return;
    const int z = (int)(tid % sz);
    const int xy = (int)(tid / sz);
    const int y = xy % sy, x = xy / sy;
    const int xs[3] = {max(x - 1, 0), x, min(x + 1, sx - 1)};
    const int ys[3] = {max(y - 1, 0), y, min(y + 1, sy - 1)};
    const int zs[3] = {max(z - 1, 0), z, min(z + 1, sz - 1)};
    uint8_t any = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int k = 0; k < 3; ++k) any |= grid[((size_t)xs[i] * sy + ys[j]) * sz + zs[k]];
    out[tid] = any ? 1 : 0;
}

__host__ __device__ inline int div2_ceil(int x) { return ((x - 1) >> 1) + 1; }  // cuda_util.cuh int_div2_ceil

template <bool SET>
__global__ void __launch_bounds__(256)
k_accel_dist(int32_t* __restrict__ links, uint8_t* __restrict__ tmp, int sx0, int sy0, int sz0) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= (long long)sx0 * sy0 * sz0) return;
    int sx = sx0, sy = sy0, sz = sz0;
    int z = (int)(tid % sz);
    const int xy = (int)(tid / sz);
    int y = xy % sy, x = xy / sy;
    uint8_t* base = tmp;
    const int val = links[tid];
    if (SET ? (val >= 0) : (val < 0)) {
        int result = -1;
        while (sx > 1 && sy > 1 && sz > 1) {
            x >>= 1;
            y >>= 1;
            z >>= 1;
            sx = div2_ceil(sx);
            sy = div2_ceil(sy);
            sz = div2_ceil(sz);
            const size_t idx = ((size_t)x * sy + y) * sz + z;
            if (SET) {
                base[idx] = 1;
            } else {
                if (base[idx]) break;
                result -= 1;
            }
            base += (size_t)sx * sy * sz;
        }
        if (!SET) links[tid] = result;
    }
}

// ---- trilinear sampling of an [N,C] plane at arbitrary points (svox2_kernel.cu:11-246) ----
// One thread per (point, channel): the 8 corner reads of a point are C contiguous floats each.
// A point is mapped by p * scaling + offset, clamped to [0, size-1]; an empty corner reads as `dflt`
// (0 for density / SH, empty_raw for sample_grid_raw_alpha, default_surf for the surface plane).
// ---- weight renders (resample's camera-weight criterion, the weight-masked surface initialisation) ----
struct CamK {
    float c2w[12];  // rows of the 3x4 camera-to-world matrix
    float fx, fy, cx, cy;
    int width, height;
};

// max for non-negative floats (misc weights and transmittances): integer order == float order there.
// A plain read first: after the first few cameras almost every sample is below what the vertex holds.
__device__ __forceinline__ void max_nonneg(float* p, float v) {
    if (v > *p) atomicMax(reinterpret_cast<int*>(p), __builtin_bit_cast(int, v));
}

// SPARSE = false: `data` is a dense [X,Y,Z] density volume and the vertex keeps the largest compositing
// weight (grid_trace_ray, misc_kernel.cu:186-284).  SPARSE = true: densities come through `links`
// (missing = 0) and the vertex keeps the largest transmittance in front of a sample (:286-400).
template <bool SPARSE>
__global__ void __launch_bounds__(256)
k_weight_render(const float* __restrict__ data, const int32_t* __restrict__ links, int sx, int sy, int sz, CamK cam,
                float step_size, float stop_thresh, int last_sample_opaque, float off0, float off1, float off2,
                float sc0, float sc1, float sc2, float* __restrict__ out) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= cam.width * cam.height) return;
    const int iy = tid / cam.width, ix = tid % cam.width;
    // cam2world_ray, render_util.cuh:599-617 (OpenCV convention, pixel centres)
    float x = ((float)ix + 0.5f - cam.cx) / cam.fx;
    float y = ((float)iy + 0.5f - cam.cy) / cam.fy;
    float z = sqrtf((float)((double)(x * x + y * y) + 1.0));
    x /= z;
    y /= z;
    z = 1.0f / z;
    float d[3], o[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        d[i] = cam.c2w[4 * i] * x + cam.c2w[4 * i + 1] * y + cam.c2w[4 * i + 2] * z;
        o[i] = cam.c2w[4 * i + 3];
    }
    o[0] = fmaf(o[0], sc0, off0);
    o[1] = fmaf(o[1], sc1, off1);
    o[2] = fmaf(o[2], sc2, off2);
    d[0] *= sc0;
    d[1] *= sc1;
    d[2] *= sc2;
    const float rn = 1.f / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    d[0] *= rn;
    d[1] *= rn;
    d[2] *= rn;
    const float world_step = rn * step_size;
    const int size[3] = {sx, sy, sz};
    float t = 0.f, tmax = 2e3f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float invdir = (float)(1.0 / (double)d[i]);
        const float t1 = (-0.5f - o[i]) * invdir;
        const float t2 = ((float)size[i] - 0.5f - o[i]) * invdir;
        if (d[i] != 0.f) {
            t = fmaxf(t, fminf(t1, t2));
            tmax = fminf(tmax, fmaxf(t1, t2));
        }
    }
    if (t > tmax) return;
    const int st0 = sy * sz, st1 = sz;
    float lli = 0.f;  // log light intensity
    // t <= tmax <= 2e3 and step_size > 0 (checked by the caller): the loop ends
    while (t <= tmax) {
        float pos[3];
        int l[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            pos[j] = o[j] + t * d[j];
            pos[j] = fminf(fmaxf(pos[j], 0.f), (float)size[j] - 1.f);
            l[j] = min((int)pos[j], size[j] - 2);
            pos[j] -= (float)l[j];
        }
        const int idx = l[0] * st0 + l[1] * st1 + l[2];
        float v[8];
        if constexpr (SPARSE) {
            const int32_t* p = links + idx;
            const int32_t lk[8] = {p[0], p[1], p[st1], p[st1 + 1], p[st0], p[st0 + 1], p[st0 + st1], p[st0 + st1 + 1]};
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = lk[i] >= 0 ? data[lk[i]] : 0.f;
        } else {
            const float* p = data + idx;
            v[0] = p[0];
            v[1] = p[1];
            v[2] = p[st1];
            v[3] = p[st1 + 1];
            v[4] = p[st0];
            v[5] = p[st0 + 1];
            v[6] = p[st0 + st1];
            v[7] = p[st0 + st1 + 1];
        }
        const float i00 = fmaf(pos[2], v[1] - v[0], v[0]), i01 = fmaf(pos[2], v[3] - v[2], v[2]);
        const float i10 = fmaf(pos[2], v[5] - v[4], v[4]), i11 = fmaf(pos[2], v[7] - v[6], v[6]);
        const float i0 = fmaf(pos[1], i01 - i00, i00), i1 = fmaf(pos[1], i11 - i10, i10);
        float sigma = fmaf(pos[0], i1 - i0, i0);
        if (!SPARSE && last_sample_opaque && t + step_size > tmax) {
            sigma += 1e9f;
            lli = 0.f;
        }
        if (sigma > 1e-8f) {
            const float log_att = -world_step * sigma;
            const float tr = expf(lli);
            const float val = SPARSE ? tr : tr * (1.f - expf(log_att));
            lli += log_att;
            float* q = out + idx;
            max_nonneg(q, val);
            max_nonneg(q + 1, val);
            max_nonneg(q + st1, val);
            max_nonneg(q + st1 + 1, val);
            max_nonneg(q + st0, val);
            max_nonneg(q + st0 + 1, val);
            max_nonneg(q + st0 + st1, val);
            max_nonneg(q + st0 + st1 + 1, val);
            if (expf(lli) < stop_thresh) break;
        }
        t += step_size;
    }
}

struct SampleXf {
    float offset[3], scaling[3];
    int size[3];
};

__device__ __forceinline__ const int32_t* sample_cell(const int32_t* links, const SampleXf& X,
                                                      const float* __restrict__ points, long long pid,
                                                      float* w) {
    int l[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float p = fmaf(points[3 * pid + i], X.scaling[i], X.offset[i]);  // transform_coord, cuda_util.cuh:63-69
        p = fminf(fmaxf(p, 0.f), X.size[i] - 1.f);
        l[i] = min((int)p, X.size[i] - 2);
        w[i] = p - l[i];
    }
    return links + ((size_t)l[0] * X.size[1] * X.size[2] + (size_t)l[1] * X.size[2] + l[2]);
}

__global__ void __launch_bounds__(256)
k_sample_plane(const int32_t* __restrict__ links, SampleXf X, const float* __restrict__ plane, int C, float dflt,
               const float* __restrict__ points, long long P, float* __restrict__ out) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= P * C) return;
    const int idx = (int)(tid % C);
    const long long pid = tid / C;
    float w[3];
    const int32_t* lp = sample_cell(links, X, points, pid, w);
    const int offy = X.size[2], offx = X.size[1] * X.size[2];
    auto rd = [&](int u) { const int l = lp[u]; return l >= 0 ? plane[(size_t)l * C + idx] : dflt; };
    auto lerp = [](float a, float b, float t) { return fmaf(t, b - a, a); };  // cuda_util.cuh:75-77
    const float ix0y0 = lerp(rd(0), rd(1), w[2]);
    const float ix0y1 = lerp(rd(offy), rd(offy + 1), w[2]);
    const float ix0 = lerp(ix0y0, ix0y1, w[1]);
    const float ix1y0 = lerp(rd(offx), rd(offx + 1), w[2]);
    const float ix1y1 = lerp(rd(offy + offx), rd(offy + offx + 1), w[2]);
    const float ix1 = lerp(ix1y0, ix1y1, w[1]);
    out[tid] = lerp(ix0, ix1, w[0]);
}

__global__ void __launch_bounds__(256)
k_sample_plane_backward(const int32_t* __restrict__ links, SampleXf X, int C, const float* __restrict__ points,
                        long long P, const float* __restrict__ grad_out, float* __restrict__ grad_plane) {
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= P * C) return;
    const int idx = (int)(tid % C);
    const long long pid = tid / C;
    float w[3];
    const int32_t* lp = sample_cell(links, X, points, pid, w);
    const int offy = X.size[2], offx = X.size[1] * X.size[2];
    const float go = grad_out[tid];
    const float xb = w[0], yb = w[1], zb = w[2], xa = 1.f - w[0], ya = 1.f - w[1], za = 1.f - w[2];
    auto add = [&](int u, float v) {
        const int l = lp[u];
        if (l >= 0) atomicAdd(&grad_plane[(size_t)l * C + idx], v);
    };
    const float xago = xa * go, xbgo = xb * go;
    float t = ya * xago;
    add(0, t * za);
    add(1, t * zb);
    t = yb * xago;
    add(offy, t * za);
    add(offy + 1, t * zb);
    t = ya * xbgo;
    add(offx, t * za);
    add(offx + 1, t * zb);
    t = yb * xbgo;
    add(offx + offy, t * za);
    add(offx + offy + 1, t * zb);
}

// cubic_extract_iso_pts_kernel (svox2_kernel.cu:248-372): iso-surface points of the listed cells along
// n x n axis-parallel lines per direction.  One thread per (cell, i, j) -- the reference runs one per cell --
// so a 100 k-cell batch fills the chip; the three directions of a line pair share the cell's 16 loads.
__global__ void __launch_bounds__(256)
k_iso_pts(const int32_t* __restrict__ links, int sx, int sy, int sz, const float* __restrict__ level_data,
          const float* __restrict__ mask_data, const int32_t* __restrict__ cell_ids, long long Q, int n_sample,
          float density_thresh, float* __restrict__ out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int nn = n_sample * n_sample;
    if (t >= Q * nn) return;
    const long long tid = t / nn;
    const int ij = (int)(t - tid * nn), i = ij / n_sample, j = ij - i * n_sample;
    const int xyz = cell_ids[tid];
    const int z = xyz % sz, xy = xyz / sz, y = xy % sy, x = xy / sy;
    if (x >= sx - 1 || y >= sy - 1 || z >= sz - 1) return;
    float surf[8], mask[8];
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int lk = links[((size_t)(x + ((c >> 2) & 1)) * sy + (y + ((c >> 1) & 1))) * sz + (z + (c & 1))];
        ok = ok && lk >= 0;
        surf[c] = lk >= 0 ? level_data[lk] : 0.f;
        mask[c] = lk >= 0 ? mask_data[lk] : 0.f;
    }
    if (!ok) return;
    const float step = 1.f / (float)(n_sample - 1);
    const float pos1 = i * step, pos2 = j * step;
#pragma unroll
    for (int dir_id = 0; dir_id < 3; ++dir_id) {
        double o[3] = {0., 0., 0.};
        float d[3] = {0.f, 0.f, 0.f};
        d[dir_id] = 1.f;
        o[dir_id == 0 ? 1 : 0] = pos1;
        o[dir_id == 2 ? 1 : 2] = pos2;
        double fs[4], st[3] = {-1, -1, -1};
        surface_cubic(surf, o, d, fs);
        solve_cubic_vieta(fs[0], fs[1], fs[2], fs[3], st);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            if (!(st[k] >= 0. && st[k] <= 1.)) continue;
            const float pt[3] = {(float)(o[0] + (double)d[0] * st[k]), (float)(o[1] + (double)d[1] * st[k]),
                                 (float)(o[2] + (double)d[2] * st[k])};
            if (trilerp8(mask, pt) >= density_thresh) {
                float* dst = out + ((size_t)tid * 3 * nn + (size_t)(i * n_sample * 3 + j * 3 + dir_id)) * 3;
                dst[0] = pt[0] + x;
                dst[1] = pt[1] + y;
                dst[2] = pt[2] + z;
                break;
            }
        }
    }
}

}  // namespace asurf
