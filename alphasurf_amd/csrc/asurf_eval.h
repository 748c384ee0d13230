// Evaluation renders: the traversal + cubic core of the hot path
// without colour, density gate, truncation window or fake samples.  surf_trav.cu:564-1705, kernels
// :3457-3580.  One ray per lane; the walk is the hierarchical exact DDA of the training path over
// the per-call occupancy pyramid (links present + a level set inside the voxel's value range).
//
// Two forms.  k_eval_comp (the default) runs behind the training path's k_sweep + k_geom: the candidates of every ray
// are in the pool, in order, and a lane per ray folds them into the requested quantity.  k_eval is the one-kernel
// form -- the stepping walk with the cubic solved in the loop -- for the rays the pipeline hands back, for
// stop_thresh > 1 (the render then ends at the first work voxel, sample or not) and for ASURF_EVAL=walk.
#pragma once
#include "asurf_pipeline.h"

namespace asurf {

enum EvalMode {
    EVAL_EXPECTED = 0,      // expected termination depth                      out0[Q]
    EVAL_MODE_DEPTH = 1,    // depth of the max-weight sample, 0 unless sum w > param
    EVAL_FIRST_DEPTH = 2,   // depth of the first sample with alpha > param
    EVAL_FIRST_ALPHA = 3,   // alpha of the first sample with alpha > param
    EVAL_NORMAL = 4,        // surface-field gradient at the first sample with alpha > 0   out0[Q,3]
    EVAL_EXTRACT = 5,       // up to max_sample (depth, alpha) pairs with alpha > param    out0,out1[Q,max_sample]
};

template <int MODE, bool CLDS>
__global__ void __launch_bounds__(kBlock)
k_eval(KGrid g, KOpt o, const float* __restrict__ origins, const float* __restrict__ dirs, long long Q,
       float param, int max_sample, float* __restrict__ out0, float* __restrict__ out1,
       const int* __restrict__ ray_list, const int* __restrict__ n_list) {
    extern __shared__ unsigned int s_coarse[];
    const int tid = threadIdx.x;
    // ray_list: only the rays the pipeline handed back (usually none: leave before anything is staged)
    if (ray_list != nullptr && (long long)blockIdx.x * kBlock >= (long long)*n_list) return;
    if constexpr (CLDS) {
        for (int i = tid; i < g.coarse_words; i += kBlock) s_coarse[i] = g.coarse[i];
        __syncthreads();
    }
    long long q = (long long)blockIdx.x * kBlock + tid;
    if (ray_list != nullptr) {
        if (q >= (long long)*n_list) return;
        q = ray_list[q];
    }
    if (q >= Q) return;
    Ray r;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        r.o[k] = origins[3 * q + k];
        r.d[k] = dirs[3 * q + k];
    }
    // world_step / step_size = 1 / |dir * scaling| (render_util.cuh:536-547)
    const float ds0 = r.d[0] * g.scaling[0], ds1 = r.d[1] * g.scaling[1], ds2 = r.d[2] * g.scaling[2];
    const float world_step = (1.f / sqrtf(ds0 * ds0 + ds1 * ds1 + ds2 * ds2)) * o.step_size;
    setup_ray(g, o, r);
    constexpr int NOUT = (MODE == EVAL_NORMAL) ? 3 : 1;
    float* my0 = out0 + (size_t)q * (MODE == EVAL_EXTRACT ? max_sample : NOUT);
    float* my1 = (MODE == EVAL_EXTRACT) ? out1 + (size_t)q * max_sample : nullptr;
    if (r.tmin > r.tmax) {
        if constexpr (MODE == EVAL_EXTRACT) {
            my0[0] = 0.f;  // the rows arrive zeroed (surf_trav.cu:4066-4067)
        } else {
#pragma unroll
            for (int i = 0; i < NOUT; ++i) my0[i] = 0.f;
        }
        return;
    }
    float outv = 0.f, weight_acc = 0.f, max_weight = -1.f, log_transmit = 0.f;
    float res[3] = {0.f, 0.f, 0.f};
    int sample_id = 0;
    bool done = false;
    Walk w;
    walk_init(g, r, w);
    bool active = true;
    int guard = 4 * (g.size[0] + g.size[1] + g.size[2]) + 64;  // see k_walk
    while (active && !done && --guard > 0) {
        const int kind = walk_hier<CLDS>(g, s_coarse, w, o.step_size, r.tmax, true);
        active = (w.t <= r.tmax);
        if (kind != 2) continue;
        Vox v;
        load_voxel(g, r, pack_voxel(w.vx, w.vy, w.vz), v);
        for (int li = 0; li < g.level_set_num && !done; ++li) {
            const float lvf = g.level_set[li];
            if (lvf < v.smin || lvf > v.smax) continue;
            double st[3] = {-1, -1, -1};
            solve_cubic_vieta(v.fs[0] - (double)lvf, v.fs[1], v.fs[2], v.fs[3], st);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (done || !(st[j] > 0)) continue;
                float pos[3];
                if (!root_position(r, v, st[j], pos)) continue;
                const float alpha = alpha_act(trilerp8(v.dens, pos), o.alpha_activation_type);
                const double depth = (st[j] + (double)v.t_close) / (double)o.step_size * (double)world_step;
                if constexpr (MODE == EVAL_EXPECTED || MODE == EVAL_MODE_DEPTH) {
                    const float pcnt = -1 * logf(1 - alpha);
                    const float weight = expf(log_transmit) * (1.f - expf(-pcnt));
                    log_transmit -= pcnt;
                    if constexpr (MODE == EVAL_EXPECTED) {
                        outv = (float)((double)outv + (double)weight * depth);
                    } else {
                        weight_acc += weight;
                        if (weight > max_weight) {
                            max_weight = weight;
                            outv = (float)depth;
                        }
                    }
                } else if constexpr (MODE == EVAL_FIRST_DEPTH) {
                    if (alpha > param) {
                        res[0] = (float)depth;
                        done = true;
                    }
                } else if constexpr (MODE == EVAL_FIRST_ALPHA) {
                    if (alpha > param) {
                        res[0] = alpha;
                        done = true;
                    }
                } else if constexpr (MODE == EVAL_NORMAL) {
                    if (alpha > 0) {
                        field_grad(v.surf, pos, res);
                        done = true;
                    }
                } else {
                    if (alpha > param) {
                        my0[sample_id] = (float)depth;
                        my1[sample_id] = alpha;
                        if (++sample_id >= max_sample) done = true;
                    }
                }
            }
        }
        if constexpr (MODE == EVAL_EXPECTED || MODE == EVAL_MODE_DEPTH) {
            if (expf(log_transmit) < o.stop_thresh) break;  // once per visited voxel, :786-789
        }
    }
    if constexpr (MODE == EVAL_EXPECTED) my0[0] = outv;
    else if constexpr (MODE == EVAL_MODE_DEPTH) my0[0] = (weight_acc > param) ? outv : 0.f;
    else if constexpr (MODE == EVAL_FIRST_DEPTH || MODE == EVAL_FIRST_ALPHA) my0[0] = res[0];
    else if constexpr (MODE == EVAL_NORMAL) {
        my0[0] = res[0];
        my0[1] = res[1];
        my0[2] = res[2];
    }
}

// ------------------------------------------------------------------------------------
// The same quantities from the candidates k_geom left (surf_trav.cu:564-1705): one ray per lane, its candidates in
// order (k_comp's way through the slots: cand_mask).  `ts` of a candidate is t_close + st rounded to float; the
// reference forms the depth from the double sum (a relative 1e-7 apart).
template <int MODE>
__global__ void __launch_bounds__(kBlock)
k_eval_comp(KGrid g, KOpt o, const float* __restrict__ dirs, long long Q, PipeBuf P, float param, int max_sample,
            float* __restrict__ out0, float* __restrict__ out1) {
    const long long q = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (q >= Q) return;
    const int n = P.n_items[q];
    if (n < 0) return;  // handed back: k_eval renders this ray
    const float ds0 = dirs[3 * q] * g.scaling[0], ds1 = dirs[3 * q + 1] * g.scaling[1], ds2 = dirs[3 * q + 2] * g.scaling[2];
    const float world_step = (1.f / sqrtf(ds0 * ds0 + ds1 * ds1 + ds2 * ds2)) * o.step_size;
    constexpr int NOUT = (MODE == EVAL_NORMAL) ? 3 : 1;
    float* my0 = out0 + (size_t)q * (MODE == EVAL_EXTRACT ? max_sample : NOUT);
    float* my1 = (MODE == EVAL_EXTRACT) ? out1 + (size_t)q * max_sample : nullptr;
    float outv = 0.f, weight_acc = 0.f, max_weight = -1.f, log_transmit = 0.f;
    float res[3] = {0.f, 0.f, 0.f};
    int sample_id = 0;
    bool done = false;
    const unsigned long long cmask = P.cand_mask[q];
    const bool scan = (cmask >> 63) != 0ull;
    unsigned long long todo = cmask & ~(1ull << 63);
    const int spb = (n > P.C1) ? P.sp_base[q] : 0;
    int s_scan = 63;  // (below slot 63 the cells hold a word only where cand_mask says so)
    while (!done) {
        int s;
        if (todo != 0ull) {
            s = __ffsll((long long)todo) - 1;
            todo &= todo - 1ull;
        } else if (scan && s_scan < n) {
            s = s_scan++;
        } else {
            break;
        }
        const int ci = (s < P.C1) ? P.citem[(size_t)s * Q + q] : P.sp_citem[spb + s - P.C1];
        const int nc = ci & 3, base = ci >> 2;
        for (int c = 0; c < nc && !done; ++c) {
            const CandRec cr = P.pool[base + c];
            const float alpha = alpha_act(cr.raw, o.alpha_activation_type);
            const double depth = (double)cr.ts / (double)o.step_size * (double)world_step;
            if constexpr (MODE == EVAL_EXPECTED || MODE == EVAL_MODE_DEPTH) {
                const float pcnt = -1 * logf(1 - alpha);
                const float weight = expf(log_transmit) * (1.f - expf(-pcnt));
                log_transmit -= pcnt;
                if constexpr (MODE == EVAL_EXPECTED) {
                    outv = (float)((double)outv + (double)weight * depth);
                } else {
                    weight_acc += weight;
                    if (weight > max_weight) {
                        max_weight = weight;
                        outv = (float)depth;
                    }
                }
            } else if constexpr (MODE == EVAL_FIRST_DEPTH) {
                if (alpha > param) {
                    res[0] = (float)depth;
                    done = true;
                }
            } else if constexpr (MODE == EVAL_FIRST_ALPHA) {
                if (alpha > param) {
                    res[0] = alpha;
                    done = true;
                }
            } else if constexpr (MODE == EVAL_NORMAL) {
                if (alpha > 0) {
                    const int pv = (s < P.C1) ? P.items[(size_t)s * Q + q] : P.sp_items[spb + s - P.C1];
                    const int32_t* p = g.links + ((size_t)(pv >> 20) * g.offx + (size_t)((pv >> 10) & 1023) * g.offy +
                                                  (pv & 1023));
                    const int lk[8] = {p[0], p[1], p[g.offy], p[g.offy + 1], p[g.offx], p[g.offx + 1],
                                       p[g.offx + g.offy], p[g.offx + g.offy + 1]};
                    float surf[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) surf[i] = g.surface[lk[i]];
                    const float pos[3] = {cr.px, cr.py, cr.pz};
                    field_grad(surf, pos, res);
                    done = true;
                }
            } else {
                if (alpha > param) {
                    my0[sample_id] = (float)depth;
                    my1[sample_id] = alpha;
                    if (++sample_id >= max_sample) done = true;
                }
            }
        }
        if constexpr (MODE == EVAL_EXPECTED || MODE == EVAL_MODE_DEPTH) {
            if (nc > 0 && expf(log_transmit) < o.stop_thresh) break;  // once per visited voxel, :786-789
        }
    }
    if constexpr (MODE == EVAL_EXPECTED) my0[0] = outv;
    else if constexpr (MODE == EVAL_MODE_DEPTH) my0[0] = (weight_acc > param) ? outv : 0.f;
    else if constexpr (MODE == EVAL_FIRST_DEPTH || MODE == EVAL_FIRST_ALPHA) my0[0] = res[0];
    else if constexpr (MODE == EVAL_NORMAL) {
        my0[0] = res[0];
        my0[1] = res[1];
        my0[2] = res[2];
    }
}

}  // namespace asurf
