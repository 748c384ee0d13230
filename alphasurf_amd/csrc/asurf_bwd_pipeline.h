// asurf_bwd_pipeline.h -- the fused backward pass as three dense kernels (gfx950, wave64).
//
// The forward pass leaves one record per composited sample (voxel, level set, root index,
// intersection ordinal, raw opacity, re-weighted alpha, pre-clamp colour, position).  With the
// colours on record the only part of trace_ray_surf_trav_backward (surf_trav.cu:1710-2911) that
// is sequential along a ray is scalar arithmetic, so the pass splits into
//
//   k_bwd_seq : one RAY per lane -- the running quantities (transmittance, `accum`, the sample
//               index of the fused losses, stop_thresh) and every term that needs the ray's
//               sample cache; emits per sample: weight, dL/dcolour per channel, the first part
//               of dL/d(rwalpha) and of dL/dt; appends the visited (record, ray) cells to a dense
//               list, record-major within a wave (neighbouring entries = neighbouring rays; for an
//               image batch the wave's 64 rays are an 8 x 8 pixel tile: round 4).
//   k_bwd_sh  : one SAMPLE per wave step over that list, lanes = (corner, SH coefficient) --
//               a chunk of 64 entries (two chunks for a long list) is sorted by voxel first
//               (wave_sort64 / wave_sort128); contributions of consecutive samples in the same
//               voxel are summed in registers, the rows of a face shared with the next voxel are
//               carried over, the rest is flushed with row-contiguous atomics into grad_sh; d/dpos
//               of the colour from the voxel's SH rows held in registers.
//   k_bwd_geo : one SAMPLE per lane over the list (chunks sorted likewise) -- reloads the voxel,
//               classifies the cubic (the forward stored the position), finishes dL/d(raw alpha)
//               and dL/dt, root gradient; a segmented shuffle reduction merges the density /
//               surface contributions of neighbouring lanes in the same voxel before the scatter.
//
// The last two are launched with a fixed grid of kListBlocks blocks that stride over the list.
// What an atomic costs there: one request at the L2 per 64-byte line an instruction touches, 21 G
// requests/s on the whole chip (variants/micro/atomics.hip) -- fewer, longer runs are the lever.
#pragma once
#include "asurf_kernels.h"
#include "asurf_pipeline.h"
#include "asurf_sweep.h"   // wave_incl_scan

namespace asurf {

struct SampleRec {  // planes [R][Q]
    // written by the forward pass
    int* v;      // packed voxel
    int* m;      // bit0 fake, bits1-2 root, bits3-7 level set, bits8.. intersection ordinal + 1
    float* raw;  // raw opacity
    float* a;    // re-weighted alpha
    float* c0;   // colour before the +0.5 / clamp, per channel
    float* c1;
    float* c2;
    float* px;   // position inside the voxel
    float* py;
    float* pz;
    // written by k_bwd_seq
    float* w;    // compositing weight (backward's clamped form)
    float* g0;   // weight * [colour in range] * gout, per channel
    float* g1;
    float* g2;
    float* cgr;  // dL/d(rwalpha): compositing + cache-dependent loss terms
    float* gst;  // cache-dependent part of dL/dt
    // written by k_bwd_sh
    float* gx;   // dL/dpos through the SH coefficients
    float* gy;
    float* gz;
    int* n;      // [Q] records of the ray (< 0: overflowed, ray belongs to the walking kernel)
    unsigned* list;  // dense list of visited (record, ray) cells: record << 27 | ray (k_bwd_seq); kRecShards segments
    int* n_list;     // their cursors, kCtrShardStride ints apart
    int list_cap;    // cells per segment
    int* n_eff;  // [Q] records the backward actually visits (stop_thresh)
    float* sws;  // [Q] clamped sum of the cached weights (sparsity term)
    int R;
};

// The list as the consumers see it: the segments' running sums turn a position in their concatenation into a cell.
struct RecList {
    int seg_end[kRecShards];
    int n;
    __device__ __forceinline__ void init(const SampleRec& S) {
        n = 0;
#pragma unroll
        for (int sh = 0; sh < kRecShards; ++sh) {
            n += min(S.n_list[sh * kCtrShardStride], S.list_cap);
            seg_end[sh] = n;
        }
    }
    __device__ __forceinline__ unsigned cell(const SampleRec& S, int i) const {  // i < n
        int sh = 0;
#pragma unroll
        for (int k = 0; k < kRecShards - 1; ++k) sh += (i >= seg_end[k]) ? 1 : 0;
        int seg0 = 0;
#pragma unroll
        for (int k = 0; k < kRecShards - 1; ++k) seg0 = (sh == k + 1) ? seg_end[k] : seg0;
        return S.list[(size_t)sh * S.list_cap + (i - seg0)];
    }
};

// ------------------------------------------------------------------------------------
// one ray of k_bwd_seq; returns the number of records the backward visits
__device__ inline int bwd_seq_ray(const KOpt& o, const KLoss& L, float trunc_a, long long Q, long long q,
                                  const float* __restrict__ rgb_gt, const float* __restrict__ rgb_out, int M,
                                  const float* __restrict__ cache_a, const float* __restrict__ cache_w,
                                  const float* __restrict__ cache_t, const int* __restrict__ cache_n,
                                  const SampleRec& S) {
    const int n_rec = S.n[q];
    if (n_rec <= 0) {
        S.n_eff[q] = 0;
        return 0;
    }
    float gout[3];
    float accum;
    {
        float cc[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {  // surf_trav.cu:3306-3316
            cc[i] = rgb_out[3 * q + i];
            const float resid = cc[i] - rgb_gt[3 * q + i];
            gout[i] = resid * L.norm_l2 * L.lambda_l2;
            gout[i] += (resid > 0.f) ? (L.norm_l1 * L.lambda_l1) : (-L.norm_l1 * L.lambda_l1);
        }
        accum = fmaf(cc[0], gout[0], fmaf(cc[1], gout[1], cc[2] * gout[2]));  // :1800-1802
    }
    // prologue over the sample cache (:1756-1793); entries >= n are implicit zeros
    const int n = (M > 0) ? min(cache_n[q], M) : 0;
    auto AL = [&](int i) -> float { return (i < n) ? cache_a[(size_t)i * Q + q] : 0.f; };
    auto WE = [&](int i) -> float { return (i < n) ? cache_w[(size_t)i * Q + q] : 0.f; };
    auto TS = [&](int i) -> float { return (i < n) ? cache_t[(size_t)i * Q + q] : 0.f; };
    float sample_alpha_sum = 0.f, sample_weight_sum = 0.f;
    for (int i = 0; i < n; ++i) {
        sample_alpha_sum += AL(i);
        sample_weight_sum += WE(i);
    }
    sample_alpha_sum = clamp_lo(sample_alpha_sum);
    sample_weight_sum = clamp_lo(sample_weight_sum);
    float Den_Dasum = 0.f, Den_Dwsum = 0.f, sample_t_mean = 0.f;
    int valid_sample_n = 0;
    float shared_Dmeant_sign = 0.f;
    int max_sample_id = 0;
    {
        float max_weight = 0.f;
        for (int i = 0; i < n; ++i) {
            const float a = AL(i), ww = WE(i);
            Den_Dasum += a * (log_ratio(a, sample_alpha_sum) + 1.f) / sqrf(sample_alpha_sum);
            Den_Dwsum += ww * (log_ratio(ww, sample_weight_sum) + 1.f) / sqrf(sample_weight_sum);
            sample_t_mean += ww / sample_weight_sum * TS(i);
            if (ww > max_weight) {
                max_weight = ww;
                max_sample_id = i;
            }
        }
        for (int i = 0; i < n; ++i) {
            const float ts = TS(i);
            if (ts > 0.f) {
                valid_sample_n++;
                shared_Dmeant_sign += (sample_t_mean > ts) ? 1.f : ((sample_t_mean < ts) ? -1.f : 0.f);
            }
        }
    }
    S.sws[q] = sample_weight_sum;

    float log_transmit = 0.f;
    int sample_i = 0;
    int r = 0;
    for (; r < n_rec; ++r) {
        const size_t idx = (size_t)r * Q + q;
        const int meta = S.m[idx];
        const bool fake = meta & 1;
        const float rwalpha = S.a[idx];
        const float pcnt = -logf(clamp_lo(1.f - rwalpha));  // :2099
        const float weight = expf(log_transmit) * (1.f - expf(-pcnt));
        // colour terms (:2104-2122)
        const float s0 = S.c0[idx] + 0.5f, s1 = S.c1[idx] + 0.5f, s2 = S.c2[idx] + 0.5f;
        const float t0 = fmaxf(s0, 0.f), t1 = fmaxf(s1, 0.f), t2 = fmaxf(s2, 0.f);
        const float total_color = (t0 * gout[0] + t2 * gout[2]) + t1 * gout[1];
        S.w[idx] = weight;
        S.g0[idx] = weight * (t0 == s0 ? 1.f : 0.f) * gout[0];
        S.g1[idx] = weight * (t1 == s1 ? 1.f : 0.f) * gout[1];
        S.g2[idx] = weight * (t2 == s2 ? 1.f : 0.f) * gout[2];
        accum -= weight * total_color;
        float cgr = accum / fminf(rwalpha - 1.f, -1e-8f) + total_color * expf(log_transmit);
        // l_dist / l_entropy (:2141-2210, 2602-2666)
        if (!fake || o.fake_sample_l_dist) {
            const float denom = fminf(rwalpha - 1.f, -1e-8f);
            if (L.lambda_l_dist > 0.f) {
                float Dl = 0.f;
                float log_Tk = log_transmit;
                for (int k = sample_i; k < valid_sample_n; ++k) {
                    float Dwk = 0.f;
                    const float tk = TS(k);
                    for (int j = 0; j < n; ++j) Dwk += WE(j) * fabsf(tk - TS(j));
                    if (k == sample_i) {
                        Dl += Dwk * expf(log_transmit);
                    } else {
                        log_Tk += logf(clamp_lo(1.f - AL(k - 1)));
                        Dl += Dwk * expf(log_Tk) * AL(k) / denom;
                    }
                }
                cgr += L.lambda_l_dist * Dl;
            }
            if (L.lambda_l_entropy > 0.f) {
                auto Den_Dw = [&](int k) -> float {
                    if (L.no_norm_weight_l_entropy) return -(logf(clamp_lo(WE(k))) + 1.f);
                    return -(log_ratio(WE(k), sample_weight_sum) + 1.f) / sample_weight_sum + Den_Dwsum;
                };
                float Den_Dai = Den_Dw(sample_i) * expf(log_transmit);
                float log_Tk = log_transmit;
                for (int k = sample_i + 1; k < valid_sample_n; ++k) {
                    const float Dwk = Den_Dw(k);
                    log_Tk += logf(clamp_lo(1.f - AL(k - 1)));
                    Den_Dai += Dwk * expf(log_Tk) * AL(k) / denom;
                }
                cgr += L.lambda_l_entropy * Den_Dai;
            }
        }
        log_transmit -= pcnt;
        float gst = 0.f;
        if (!fake || o.fake_sample_l_dist) {
            const float ts_i = TS(sample_i);
            if (L.lambda_l_dist_a > 0.f) {
                float ga = 0.f;
                for (int j = 0; j < n; ++j) ga += AL(j) * fabsf(ts_i - TS(j));
                cgr += L.lambda_l_dist_a * ga;
            }
            if (L.lambda_l_entropy_a > 0.f) {
                const float Den_Dai = -(log_ratio(AL(sample_i), sample_alpha_sum) + 1.f) / sample_alpha_sum;
                cgr += L.lambda_l_entropy_a * (Den_Dai + Den_Dasum);
            }
            if (!fake) {  // dL/dt terms exist for real intersections only (:2365-2400)
                if (L.lambda_l_dist > 0.f || L.lambda_l_dist_a > 0.f) {
                    float gw = 0.f, ga = 0.f;
                    const float wi = WE(sample_i), ai = AL(sample_i);
                    for (int j = 0; j < n; ++j) {
                        const float tj = TS(j);
                        const float sgn = (ts_i > tj) ? 1.f : (ts_i < tj) ? -1.f : 0.f;
                        gw += sgn * wi * WE(j);
                        ga += sgn * ai * AL(j);
                    }
                    gst += L.lambda_l_dist * gw + L.lambda_l_dist_a * ga;
                }
                if (L.lambda_l_samp_dist > 0.f) {
                    const float sgn = (sample_t_mean > ts_i) ? 1.f : (sample_t_mean < ts_i) ? -1.f : 0.f;
                    gst += L.lambda_l_samp_dist *
                           (shared_Dmeant_sign * AL(sample_i) / sample_weight_sum + sgn * (-1.f));
                }
                const float trunc_reweight =
                    o.truncated_vol_render ? trunc_rw((float)((meta >> 8) - 1), trunc_a, o.trunc_vol_weight_min)
                                           : 1.f;
                if (L.lambda_conv_mode_samp > 0.f && trunc_reweight > o.trunc_vol_weight_min) {
                    const float tm = TS(max_sample_id);
                    const float gcm = (ts_i > tm) ? 1.f : (ts_i < tm) ? -1.f : 0.f;
                    gst += L.lambda_conv_mode_samp * gcm;
                }
            }
            if (sample_i < M - 1) sample_i += 1;
        }
        S.cgr[idx] = cgr;
        S.gst[idx] = gst;
        // the reference tests stop_thresh once per voxel, after all of its samples (:2893-2895)
        const bool last_of_voxel = (r + 1 >= n_rec) || (S.v[(size_t)(r + 1) * Q + q] != S.v[idx]);
        if (last_of_voxel && expf(log_transmit) < o.stop_thresh) {
            ++r;
            break;
        }
    }
    S.n_eff[q] = min(r, n_rec);
    return min(r, n_rec);
}

__global__ void __launch_bounds__(kBlock)
k_bwd_seq(KOpt o, KLoss L, float trunc_a, long long Q, const float* __restrict__ rgb_gt,
          const float* __restrict__ rgb_out, int M, const float* __restrict__ cache_a,
          const float* __restrict__ cache_w, const float* __restrict__ cache_t,
          const int* __restrict__ cache_n, SampleRec S, int rpw, int tile_w, int tile_tw) {
    // (rpw rays a wave, as k_comp).  tile_w > 0: the batch is an image of that width in row order and a wave takes a
    // tile of tile_tw x (64 / tile_tw) pixels of it instead of 64 pixels of a row (rpw = 64; the host has checked that
    // the tiles partition the image): the list below is record-major within the wave, so a chunk of k_bwd_sh /
    // k_bwd_geo is then one depth of a 2-D patch of rays, which falls into fewer voxels than a row of 64 does
    long long q = ((long long)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * rpw + (threadIdx.x & 63);
    if (tile_w > 0) {
        const int wg = (int)blockIdx.x * (kBlock / 64) + (int)(threadIdx.x >> 6);
        const int tiles_x = tile_w / tile_tw, lane = (int)(threadIdx.x & 63);
        const int ty = wg / tiles_x, tx = wg - ty * tiles_x;
        const int ly = lane / tile_tw, lx = lane - ly * tile_tw;
        q = (long long)(ty * (64 / tile_tw) + ly) * tile_w + tx * tile_tw + lx;
    }
    int nt = 0;
    if ((int)(threadIdx.x & 63) < rpw && q < Q) nt = bwd_seq_ray(o, L, trunc_a, Q, q, rgb_gt, rgb_out, M, cache_a, cache_w, cache_t, cache_n, S);
    // dense list of the (record, ray) cells the backward visits, record-major within the wave
    int wave_total;
    wave_excl_scan(nt, &wave_total);
    // one allocation per block (a single counter serialises ten thousand per-wave atomics)
    __shared__ int s_tot[kBlock / 64 + 1];
    if ((threadIdx.x & 63) == 0) s_tot[threadIdx.x >> 6] = wave_total;
    __syncthreads();
    if (threadIdx.x == 0) {
        int sum = 0;
        for (int w = 0; w < kBlock / 64; ++w) {
            const int t = s_tot[w];
            s_tot[w] = sum;
            sum += t;
        }
        const int shard = (int)(blockIdx.x % kRecShards);
        s_tot[kBlock / 64] = shard * S.list_cap + (sum > 0 ? atomicAdd(&S.n_list[shard * kCtrShardStride], sum) : 0);
    }
    __syncthreads();
    if (wave_total > 0) {
        int base0 = s_tot[kBlock / 64] + s_tot[threadIdx.x >> 6];
        const unsigned long long below = (1ull << (threadIdx.x & 63)) - 1ull;
        for (int j = 0; __any(nt > j); ++j) {
            const unsigned long long m = __ballot(nt > j);
            if (nt > j) S.list[base0 + __popcll(m & below)] = ((unsigned)j << kRecRayBits) | (unsigned)q;
            base0 += __popcll(m);
        }
    }
}

// ------------------------------------------------------------------------------------
// The 64 keys of a wave in ascending order (bitonic network: 21 compare-exchange steps, two ds_bpermute each).
// k_bwd_sh and k_bwd_geo sort the samples of a chunk by voxel before they merge runs of one voxel: the chunk is one
// depth of 64 neighbouring rays, its samples fall into ~29 voxels (bench scene) that come back again and again along
// the list -- A B A B C B ... -- and every change of voxel writes rows out.  Sorted, a voxel is ONE run, and the
// voxel after it in the order (z fastest in the packed number) mostly shares a face with it.
__device__ __forceinline__ unsigned long long wave_sort64(unsigned long long key) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const unsigned lo = __shfl_xor((unsigned)key, j), hi = __shfl_xor((unsigned)(key >> 32), j);
            const unsigned long long other = ((unsigned long long)hi << 32) | lo;
            const bool up = (lane & k) == 0;          // this block of k lanes sorts ascending (k = 64: all of them)
            const bool lower = (lane & j) == 0;       // this lane keeps the smaller key of the pair when ascending
            const bool take_min = (up == lower);
            const bool other_less = other < key;
            key = (take_min == other_less) ? other : key;
        }
    }
    return key;
}

// 128 keys, two a lane (element = lane + 64 * register), ascending: the long lists are sorted two chunks at a time --
// on the bench scene the 128 samples of two neighbouring pixel tiles fall into 1.7 times the voxels of one tile's 64.
__device__ __forceinline__ void wave_sort128(unsigned long long& e0, unsigned long long& e1) {
    const int lane = threadIdx.x & 63;
    auto exchange = [&](unsigned long long& key, int j, bool up) {
        const unsigned lo = __shfl_xor((unsigned)key, j), hi = __shfl_xor((unsigned)(key >> 32), j);
        const unsigned long long other = ((unsigned long long)hi << 32) | lo;
        const bool take_min = (up == ((lane & j) == 0));
        key = (take_min == (other < key)) ? other : key;
    };
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            // element index = lane (+ 64): bit k of it decides the direction; for k = 64 that bit is the register
            exchange(e0, j, k == 64 ? true : (lane & k) == 0);
            exchange(e1, j, k == 64 ? false : (lane & k) == 0);
        }
    }
    {   // k = 128, j = 64: within the lane
        const unsigned long long lo = e0 < e1 ? e0 : e1, hi = e0 < e1 ? e1 : e0;
        e0 = lo;
        e1 = hi;
    }
#pragma unroll
    for (int j = 32; j > 0; j >>= 1) {
        exchange(e0, j, true);
        exchange(e1, j, true);
    }
}

// ------------------------------------------------------------------------------------
template <int B>
__global__ void __launch_bounds__(kBlock)
k_bwd_sh(KGrid g, KGrads G, int want_dpos, const float* __restrict__ dirs, long long Q, SampleRec S, int sort_chunk) {
    const TlScope tl(6);
    using C = Coop<B>;
    constexpr int SP = B | 1;  // odd LDS row stride of the per-sample SH basis
    __shared__ float s_sph[kBlock * SP];
    __shared__ LaneRec s_rec[kBlock];
    __shared__ float4 s_smp[kBlock];  // position in the voxel | real sample?   (in), d/dpos (out)
    __shared__ float4 s_gch[kBlock];  // d(loss)/d(colour channel) | packed voxel
    __shared__ float s_carry[(kBlock / 64) * 8 * C::D];  // per wave: the run's sums while they change corner slots
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    RecList RL;
    RL.init(S);
    const int n_list = RL.n;
    const int wave_id = blockIdx.x * (kBlock / 64) + (tid >> 6);
    const int wbase = tid & ~63;
    const float* sph_wave = s_sph + wbase * SP;
    const LaneRec* rec_wave = s_rec + wbase;
    // phase A layout: two samples at a time, lane = (half, channel, corner)
    const int half = lane >> 5, chA = (lane >> 3) & 3, cA = lane & 7;
    // phase B layout: lanes = (corner slot, SH coefficient): row-contiguous atomics
    const int cs = lane / C::D;
    const int e = lane - cs * C::D;
    const int ch = e / B;
    const int k = e - ch * B;
    const bool on = cs < C::CPR;
  // one dense chunk of list entries per wave and turn (see k_bwd_geo).  A wave works its chunk off two samples at a
  // time, a dependent round trip each: a short list is cut into shorter chunks (8 entries at the training script's
  // batch size: 0.049 -> 0.02 ms), a long one into chunks of 64, whose runs of one voxel merge more atomics.
  // (a long list: two chunks are sorted together and worked off one after the other, the run going on across them;
  // sort_chunk 2 -- the tests -- does so with a list of any length)
  const bool wide = (sort_chunk && n_list >= (1 << 18)) || sort_chunk == 2;
  const int chunk = (wide || n_list >= (1 << 18)) ? 64 : n_list >= (1 << 17) ? 32 : n_list >= (1 << 16) ? 16 : 8;
  const int take = wide ? 128 : chunk;
  for (int base = wave_id * take; base < n_list; base += (int)gridDim.x * (kBlock / 64) * take) {
    // the chunk's entries, sorted by voxel (then by record and ray: the order of the sums does not depend on timing)
    auto entry = [&](int i) -> unsigned long long {
        if (i >= n_list) return ~0ull;
        const unsigned c0 = RL.cell(S, i);
        const int pv0 = S.v[(size_t)(c0 >> kRecRayBits) * Q + (c0 & ((1u << kRecRayBits) - 1u))];
        return ((unsigned long long)(unsigned)pv0 << 32) | c0;
    };
    unsigned long long ent0 = lane < chunk ? entry(base + lane) : ~0ull;
    unsigned long long ent1 = wide ? entry(base + 64 + lane) : ~0ull;
    if (wide) wave_sort128(ent0, ent1);
    else if (sort_chunk && n_list >= (1 << 16)) ent0 = wave_sort64(ent0);   // (a short list is not coherent: nothing to merge)
    float run[C::ROUNDS];
    int run_link[C::ROUNDS];   // the rows the run's sums belong to (this lane's corners)
#pragma unroll
    for (int rd = 0; rd < C::ROUNDS; ++rd) {
        run[rd] = 0.f;
        run_link[rd] = 0;
    }
    int run_pv = -1;
    int held = -1;  // voxel whose SH rows this lane holds (phase A, per half)
    float val[B];
#pragma unroll
    for (int kk = 0; kk < B; ++kk) val[kk] = 0.f;
    auto flush = [&]() {
        if (run_pv < 0 || G.grad_sh == nullptr) return;
#pragma unroll
        for (int rd = 0; rd < C::ROUNDS; ++rd) {
            const int c = rd * C::CPR + cs;
            if (on && c < 8) atomicAdd(&G.grad_sh[(size_t)run_link[rd] * C::D + e], run[rd]);
            run[rd] = 0.f;
        }
    };
   for (int part = 0; part < (wide ? 2 : 1); ++part) {
    const unsigned long long ent = part ? ent1 : ent0;
    const bool mine = ent != ~0ull;
    const unsigned cell = mine ? (unsigned)ent : 0u;
    const long long q = (long long)(cell & ((1u << kRecRayBits) - 1u));
    const int r = (int)(cell >> kRecRayBits);
    const unsigned long long need0 = __ballot(mine);
    const size_t idx = (size_t)r * Q + q;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the previous turn's readers are done
    if (mine) {
        const int pv = (int)(ent >> 32);
        const int vx = pv >> 20, vy = (pv >> 10) & 1023, vz = pv & 1023;
        const int32_t* p = g.links + ((size_t)vx * g.offx + (size_t)vy * g.offy + vz);
        s_rec[tid].link[0] = p[0];
        s_rec[tid].link[1] = p[1];
        s_rec[tid].link[2] = p[g.offy];
        s_rec[tid].link[3] = p[g.offy + 1];
        s_rec[tid].link[4] = p[g.offx];
        s_rec[tid].link[5] = p[g.offx + 1];
        s_rec[tid].link[6] = p[g.offx + g.offy];
        s_rec[tid].link[7] = p[g.offx + g.offy + 1];
        float d[3] = {dirs[3 * q], dirs[3 * q + 1], dirs[3 * q + 2]};
        float sph[B];
        eval_sh<B>(d, sph);
#pragma unroll
        for (int kk = 0; kk < B; ++kk) s_sph[tid * SP + kk] = sph[kk];
        // fake samples carry no d/dpos (surf_trav.cu:2722)
        s_smp[tid] = make_float4(S.px[idx], S.py[idx], S.pz[idx], (S.m[idx] & 1) ? 0.f : 1.f);
        s_gch[tid] = make_float4(S.g0[idx], S.g1[idx], S.g2[idx], __builtin_bit_cast(float, pv));
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    unsigned long long need = need0;
    // Neighbouring rays mostly sit in the same voxel: their contributions to the 8 SH rows are
    // summed in registers and flushed with one atomic per (row, coefficient) when the voxel changes
    // (wave-uniform test, so the lanes never diverge on it).
    while (need) {
        int r0, r1;
        bool two;
        next_pair(need, &r0, &r1, &two);
        // ---- phase A: per lane (channel, corner): A = w_c * gout[ch]; d/dpos through the colour ----
        const int rr = wbase + (half ? r1 : r0);
        const bool mineA = chA < 3 && (half == 0 || two);
        const float4 sp = s_smp[rr];
        const float4 sg = s_gch[rr];
        const float gch = chA == 0 ? sg.x : chA == 1 ? sg.y : sg.z;
        const float wx = (cA & 4) ? sp.x : 1.f - sp.x;
        const float wy = (cA & 2) ? sp.y : 1.f - sp.y;
        const float wz = (cA & 1) ? sp.z : 1.f - sp.z;
        const float A = mineA ? (wx * wy * wz) * gch : 0.f;
        if (want_dpos) {  // :2249-2264
            const int pvA = __builtin_bit_cast(int, sg.w);
            if (mineA && pvA != held) {
                held = pvA;
                const float* row = g.sh + ((size_t)s_rec[rr].link[cA] * C::D + chA * B);
#pragma unroll
                for (int kk = 0; kk < B; ++kk) val[kk] = row[kk];
            }
            float dsh = 0.f;
#pragma unroll
            for (int kk = 0; kk < B; ++kk) dsh = fmaf(s_sph[rr * SP + kk], val[kk], dsh);
            const float cgd = mineA ? gch * dsh * sp.w : 0.f;
            const float gx = dpp_sum32(cgd * (((cA & 4) ? wy : -wy) * wz));
            const float gy = dpp_sum32(cgd * (((cA & 2) ? wx : -wx) * wz));
            const float gz = dpp_sum32(cgd * (((cA & 1) ? wx : -wx) * wy));
            if ((lane & 31) == 31 && (half == 0 || two)) s_smp[rr] = make_float4(gx, gy, gz, 0.f);
        }
        // ---- phase B: grad_sh[c][e] += A[ch(e)][c] * sph[k(e)], merged over runs of one voxel ----
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 0 || two) {
                const int sl = h ? r1 : r0;
                const int pv = __builtin_bit_cast(int, s_gch[wbase + sl].w);  // wave-uniform
                if (pv != run_pv) {
                    // Neighbouring rays mostly step to a voxel that shares a FACE with the run's: the four rows of
                    // that face keep accumulating (they are corners c ^ b of the new voxel) and only the four rows
                    // of the far face are written out.  The sums change corner slot through LDS.
                    // (the same holds for an edge or a corner: two rows or one)
                    const int dx = (pv >> 20) - (run_pv >> 20), dy = ((pv >> 10) & 1023) - ((run_pv >> 10) & 1023),
                              dz = (pv & 1023) - (run_pv & 1023);
                    const bool adjacent = run_pv >= 0 && dx >= -1 && dx <= 1 && dy >= -1 && dy <= 1 && dz >= -1 && dz <= 1;
                    if (adjacent && G.grad_sh != nullptr) {
                        const int b = (dx ? 4 : 0) | (dy ? 2 : 0) | (dz ? 1 : 0);            // axes moved along
                        const int hi = (dx > 0 ? 4 : 0) | (dy > 0 ? 2 : 0) | (dz > 0 ? 1 : 0);  // ... upwards
                        float* carry = s_carry + (tid >> 6) * (8 * C::D);
#pragma unroll
                        for (int rd = 0; rd < C::ROUNDS; ++rd) {
                            const int c = rd * C::CPR + cs;
                            if (on && c < 8) {
                                const bool shared = (c & b) == hi;  // on the old voxel's side towards the new one
                                if (!shared) atomicAdd(&G.grad_sh[(size_t)run_link[rd] * C::D + e], run[rd]);
                                carry[c * C::D + e] = shared ? run[rd] : 0.f;
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
                        for (int rd = 0; rd < C::ROUNDS; ++rd) {
                            const int c = rd * C::CPR + cs;
                            run[rd] = (on && c < 8) ? carry[(c ^ b) * C::D + e] : 0.f;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    } else {
                        flush();
                    }
                    run_pv = pv;
#pragma unroll
                    for (int rd = 0; rd < C::ROUNDS; ++rd) {
                        const int c = rd * C::CPR + cs;
                        run_link[rd] = (on && c < 8) ? rec_wave[sl].link[c] : 0;
                    }
                }
                const float sphk = on ? sph_wave[sl * SP + k] : 0.f;  // :2120-2122
#pragma unroll
                for (int rd = 0; rd < C::ROUNDS; ++rd) {
                    const int c = rd * C::CPR + cs;
                    const float Av = __builtin_bit_cast(
                        float, __builtin_amdgcn_ds_bpermute((h * 32 + ch * 8 + (c & 7)) << 2, __builtin_bit_cast(int, A)));
                    if (on && c < 8) run[rd] = fmaf(Av, sphk, run[rd]);
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (mine) {
        const float4 go = want_dpos ? s_smp[tid] : make_float4(0.f, 0.f, 0.f, 0.f);
        S.gx[idx] = go.x;
        S.gy[idx] = go.y;
        S.gz[idx] = go.z;
    }
   }  // part
    flush();
  }
  tl.end();
}

// ------------------------------------------------------------------------------------
// k_bwd_sh with the SH gradient rows of a whole chunk merged before they leave (round 4; ASURF_BWD_SH=tab -- measured
// slower than k_bwd_sh at its present occupancy and left off, see the end of this comment).
//
// k_bwd_sh merges along the list only: the sums of a run of samples in ONE voxel stay in registers, a face shared with
// the next run's voxel is carried over, everything else is written out -- 3.0 rows of 27 floats a sample on the bench
// scene (WRITE_SIZE 148 MB a launch).  But the 64 samples of a chunk are neighbouring rays of an image, they fall into a
// patch of a few dozen voxels, and the vertices of that patch are shared in every direction: 2.1 distinct rows a sample
// in image order, 1.07 when the rays come as 8 x 8 pixel tiles (asurf_fused_params::image_width; counted on the bench
// scene).  So a wave first finds the DISTINCT rows of its chunk -- a hash table of row numbers in LDS (slot = row mod
// kTabRows, linear probing) --, sorts the chunk's 512 (sample, corner) terms by row with a counting sort, and then walks
// the rows, 21 at a time, lane = (row, channel): the row's terms are summed in B registers, the row's SH values are
// loaded ONCE for the colour's d/dpos (k_bwd_sh gathers the 8 rows of every sample, pair after pair), and a row leaves
// once.  A sample one of whose corners finds no place within kTabProbes probes (an incoherent batch: 512 different
// rows) adds to memory directly and takes k_bwd_sh's d/dpos.
// Measured (MI355X, config 2, wave timelines of variants/timeline.py): a chunk takes a wave 50 us here against 87-150 us
// in k_bwd_sh, with 0.36 of the atomics -- but the tables are 14.5 KB of LDS a wave, 8 waves share a CU where k_bwd_sh
// has 20, and the launch is four rounds of waves: 0.25 ms in tile order (0.33 in image order, where the table of 256
// rows overflows) against k_bwd_sh's 0.17.  Forms on the way here: the row sums themselves in LDS (ds_add_f32 into a
// 13.8 KB table, 6 waves a CU: 0.49 ms); lane = (row, coefficient), two rows a turn, every lane recomputing the term's
// weight (188 VALU instructions a sample: 0.30 ms).  What it needs to pay: half the LDS (links reloaded for the rare
// direct sample, 128 rows, counters aliased with the staging buffer) and k_bwd_geo merged the same way.
template <int B>
struct BwdShTab {
    static constexpr int kWaves = 4;
    static constexpr int kThreads = 64 * kWaves;
    static constexpr int kTabRows = 256;   // power of two (136 distinct rows a chunk of 64 in image order, 68 in tile order)
    static constexpr int kTabProbes = 8;
};

template <int B>
__global__ void __launch_bounds__(BwdShTab<B>::kThreads)
k_bwd_sh_tab(KGrid g, KGrads G, int want_dpos, const float* __restrict__ dirs, long long Q, SampleRec S) {
    const TlScope tl(6);
    using C = Coop<B>;
    using T = BwdShTab<B>;
    constexpr int NT = T::kThreads, H = T::kTabRows;
    constexpr int SP = (B + 3) & ~3;  // LDS row stride of the per-sample SH basis: rows are read whole (16-byte pieces)
    __shared__ float s_sph[NT * SP];
    __shared__ LaneRec s_rec[NT];
    __shared__ float4 s_smp[NT];  // position in the voxel | real sample?
    __shared__ float4 s_gch[NT];  // d(loss)/d(colour channel) | packed voxel
    __shared__ float4 s_dpos[NT]; // d/dpos (out)
    __shared__ int s_key[T::kWaves][H];             // row number of the slot (-1: free)
    __shared__ int s_cnt32[T::kWaves][H];           // terms of the slot still to be placed (0 between chunks)
    __shared__ unsigned short s_off[T::kWaves][H];  // first term of the slot in s_list
    __shared__ unsigned char s_len[T::kWaves][H];   // terms of the slot (<= 64: a sample's 8 rows are different)
    __shared__ unsigned short s_used[T::kWaves][H];
    __shared__ unsigned short s_list[T::kWaves][512];   // sample << 3 | corner, sorted by slot
    __shared__ float s_stage[T::kWaves][21 * C::D];     // a turn's row sums on their way to the (row, coefficient) layout
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    RecList RL;
    RL.init(S);
    const int n_list = RL.n;
    const int wave_id = blockIdx.x * T::kWaves + (tid >> 6);
    const int wbase = tid & ~63;
    const LaneRec* rec_wave = s_rec + wbase;
    int* const key = s_key[tid >> 6];
    unsigned short* const off = s_off[tid >> 6];
    unsigned char* const len = s_len[tid >> 6];
    unsigned short* const used = s_used[tid >> 6];
    unsigned short* const list = s_list[tid >> 6];
    int* const cnt32 = s_cnt32[tid >> 6];
    for (int i = lane; i < H; i += 64) {
        key[i] = -1;
        cnt32[i] = 0;
    }
    const int half = lane >> 5, chA = (lane >> 3) & 3, cA = lane & 7;
    const int cs = lane / C::D;
    const int e = lane - cs * C::D;
    const int ch = e / B;
    const int k = e - ch * B;
    const bool on = cs < C::CPR;
  const int chunk = n_list >= (1 << 18) ? 64 : n_list >= (1 << 17) ? 32 : n_list >= (1 << 16) ? 16 : 8;
  for (int base = wave_id * chunk; base < n_list; base += (int)gridDim.x * T::kWaves * chunk) {
    const bool mine = lane < chunk && base + lane < n_list;
    const unsigned cell = mine ? RL.cell(S, base + lane) : 0u;
    const long long q = (long long)(cell & ((1u << kRecRayBits) - 1u));
    const int r = (int)(cell >> kRecRayBits);
    const size_t idx = (size_t)r * Q + q;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the previous turn's readers are done
    constexpr int kNone = 0xffff;
    int sl8[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) sl8[c] = kNone;
    if (mine) {
        const int pv = S.v[idx];
        const int vx = pv >> 20, vy = (pv >> 10) & 1023, vz = pv & 1023;
        const int32_t* p = g.links + ((size_t)vx * g.offx + (size_t)vy * g.offy + vz);
        int lk[8];
        lk[0] = p[0];
        lk[1] = p[1];
        lk[2] = p[g.offy];
        lk[3] = p[g.offy + 1];
        lk[4] = p[g.offx];
        lk[5] = p[g.offx + 1];
        lk[6] = p[g.offx + g.offy];
        lk[7] = p[g.offx + g.offy + 1];
        float d[3] = {dirs[3 * q], dirs[3 * q + 1], dirs[3 * q + 2]};
        float sph[B];
        eval_sh<B>(d, sph);
#pragma unroll
        for (int kk = 0; kk < B; ++kk) s_sph[tid * SP + kk] = sph[kk];
        // fake samples carry no d/dpos (surf_trav.cu:2722)
        s_smp[tid] = make_float4(S.px[idx], S.py[idx], S.pz[idx], (S.m[idx] & 1) ? 0.f : 1.f);
        s_gch[tid] = make_float4(S.g0[idx], S.g1[idx], S.g2[idx], __builtin_bit_cast(float, pv));
        s_dpos[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
        // the sample's 8 rows take their places in the table
        bool all_in = G.grad_sh != nullptr;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            s_rec[tid].link[c] = lk[c];
            int h = lk[c] & (H - 1), got = -1;
            for (int pr = 0; pr < T::kTabProbes && lk[c] >= 0; ++pr) {
                const int old = atomicCAS(&key[h], -1, lk[c]);
                if (old == -1 || old == lk[c]) {
                    got = h;
                    break;
                }
                h = (h + 1) & (H - 1);
            }
            sl8[c] = got < 0 ? kNone : got;
            all_in = all_in && got >= 0;
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            sl8[c] = all_in ? sl8[c] : kNone;
            if (all_in) atomicAdd(&cnt32[sl8[c]], 1);
        }
    }
    const unsigned long long direct = __ballot(mine && sl8[0] == kNone);   // samples that add to memory themselves
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // ---- counting sort of the (sample, corner) terms by slot ----
    int n_used = 0, n_terms = 0;   // (wave-uniform)
#pragma unroll
    for (int h0 = 0; h0 < H; h0 += 64) {
        const int c = cnt32[h0 + lane];
        const int incl = wave_incl_scan(c);
        off[h0 + lane] = (unsigned short)(n_terms + incl - c);
        len[h0 + lane] = (unsigned char)c;
        const unsigned long long mu = __ballot(c > 0);
        if (c > 0) used[n_used + __popcll(mu & ((1ull << lane) - 1ull))] = (unsigned short)(h0 + lane);
        n_used += __popcll(mu);
        n_terms += bcast_i(incl, 63);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (mine && sl8[0] != kNone) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int at = atomicSub(&cnt32[sl8[c]], 1) - 1;   // (the counters are 0 again when every term is placed)
            list[off[sl8[c]] + at] = (unsigned short)((lane << 3) | c);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // ---- the chunk's rows, 21 at a time: lane (row, channel) ----
    //   gradient:  row[ch*B + k] += (w_c * gout[ch]) * sph[k] over the row's terms (:2120-2122), w_c the corner's
    //              trilinear weight, summed in B registers; the sums go through LDS to the (row, coefficient) layout
    //              and leave row-contiguous, CPR rows an instruction.  (A first form had lane = (row, coefficient), two
    //              rows a turn: every lane of a row recomputed the term's weight -- 188 VALU instructions a sample.)
    //   d/dpos of the colour (:2249-2264): a term's share is gout[ch] * <sph, row[ch]> * d(w_c)/dpos -- the row's SH
    //              values are loaded ONCE a row, all rows of a turn together, and the shares are added to the sample's
    //              three sums in LDS.  k_bwd_sh gathers the 8 rows of every sample, pair after pair: 32 dependent
    //              round trips a chunk, which is what that kernel lasts (63 % of its wave cycles wait).
    if (G.grad_sh != nullptr || want_dpos) {
        constexpr int RPT = 21;   // rows a turn: 21 x 3 channels = 63 lanes
        float* const stage = s_stage[tid >> 6];
        const int rg = lane / 3, chl = lane - rg * 3;
        for (int i0 = 0; i0 < n_used; i0 += RPT) {
            const bool act = rg < RPT && i0 + rg < n_used;
            const int slot = act ? used[i0 + rg] : 0;
            const int lo = off[slot], n = act ? len[slot] : 0;
            float acc[B], shr[B];
#pragma unroll
            for (int kk = 0; kk < B; ++kk) {
                acc[kk] = 0.f;
                shr[kk] = 0.f;
            }
            if (want_dpos && act) {
                const float* row = g.sh + ((size_t)key[slot] * C::D + chl * B);
#pragma unroll
                for (int kk = 0; kk < B; ++kk) shr[kk] = row[kk];
            }
            for (int j = 0; __any(j < n); ++j) {
                if (j < n) {
                    const int ent = list[lo + j];
                    const int sm = wbase + (ent >> 3), c = ent & 7;
                    const float4 sp = s_smp[sm];
                    const float4 sg = s_gch[sm];
                    const float wx = (c & 4) ? sp.x : 1.f - sp.x;
                    const float wy = (c & 2) ? sp.y : 1.f - sp.y;
                    const float wz = (c & 1) ? sp.z : 1.f - sp.z;
                    const float gch = chl == 0 ? sg.x : chl == 1 ? sg.y : sg.z;
                    const float A = (wx * wy * wz) * gch;
                    float dsh = 0.f;
#pragma unroll
                    for (int kk = 0; kk < B; ++kk) {
                        const float sk = s_sph[sm * SP + kk];
                        acc[kk] = fmaf(A, sk, acc[kk]);
                        dsh = fmaf(sk, shr[kk], dsh);
                    }
                    if (want_dpos) {
                        const float t = gch * dsh * sp.w;
                        const float t1 = __shfl_down(t, 1), t2 = __shfl_down(t, 2);   // (the row's other two channels)
                        const float ts = (t + t1) + t2;
                        if (chl == 0 && ts != 0.f) {
                            atomicAdd(&s_dpos[sm].x, ts * (((c & 4) ? wy : -wy) * wz));
                            atomicAdd(&s_dpos[sm].y, ts * (((c & 2) ? wx : -wx) * wz));
                            atomicAdd(&s_dpos[sm].z, ts * (((c & 1) ? wx : -wx) * wy));
                        }
                    }
                }
            }
            if (G.grad_sh != nullptr) {
                if (rg < RPT) {
#pragma unroll
                    for (int kk = 0; kk < B; ++kk) stage[rg * C::D + chl * B + kk] = acc[kk];
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                const int n_rows = min(RPT, n_used - i0);
                for (int r0 = 0; r0 < n_rows; r0 += C::CPR) {
                    if (on && r0 + cs < n_rows) {
                        const float v = stage[(r0 + cs) * C::D + e];
                        if (v != 0.f) atomicAdd(&G.grad_sh[(size_t)key[used[i0 + r0 + cs]] * C::D + e], v);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
        }
    }
    if (G.grad_sh != nullptr) {
        // samples without a place in the table: their 8 rows one after the other
        unsigned long long dm = direct;
        while (dm) {
            const int sl = __ffsll((long long)dm) - 1;
            dm &= dm - 1ull;
            const float4 sp = s_smp[wbase + sl];
            const float4 sg = s_gch[wbase + sl];
            const float gch = ch == 0 ? sg.x : ch == 1 ? sg.y : sg.z;
            const float sphk = on ? s_sph[(wbase + sl) * SP + k] : 0.f;
#pragma unroll
            for (int rd = 0; rd < C::ROUNDS; ++rd) {
                const int c = rd * C::CPR + cs;
                if (on && c < 8) {
                    const float wx = (c & 4) ? sp.x : 1.f - sp.x;
                    const float wy = (c & 2) ? sp.y : 1.f - sp.y;
                    const float wz = (c & 1) ? sp.z : 1.f - sp.z;
                    const float v = ((wx * wy * wz) * gch) * sphk;
                    const int link = rec_wave[sl].link[c];
                    if (v != 0.f && link >= 0) atomicAdd(&G.grad_sh[(size_t)link * C::D + e], v);
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (int i = lane; i < n_used; i += 64) key[used[i]] = -1;
    // (slots claimed by a sample that then went direct hold a key and no terms: they are freed here)
    if (direct != 0ull) {
#pragma unroll
        for (int h0 = 0; h0 < H; h0 += 64) key[h0 + lane] = -1;
    }
    // ---- phase A: d/dpos of the colour through the SH rows (k_bwd_sh's), two samples at a time ----
    if (want_dpos && direct != 0ull) {   // (the samples without a place in the table only)
        unsigned long long need = direct;
        int held = -1;  // voxel whose SH rows this lane holds (per half)
        float val[B];
#pragma unroll
        for (int kk = 0; kk < B; ++kk) val[kk] = 0.f;
        while (need) {
            int r0, r1;
            bool two;
            next_pair(need, &r0, &r1, &two);
            const int rr = wbase + (half ? r1 : r0);
            const bool mineA = chA < 3 && (half == 0 || two);
            const float4 sp = s_smp[rr];
            const float4 sg = s_gch[rr];
            const float gch = chA == 0 ? sg.x : chA == 1 ? sg.y : sg.z;
            const float wx = (cA & 4) ? sp.x : 1.f - sp.x;
            const float wy = (cA & 2) ? sp.y : 1.f - sp.y;
            const float wz = (cA & 1) ? sp.z : 1.f - sp.z;
            const int pvA = __builtin_bit_cast(int, sg.w);
            if (mineA && pvA != held) {   // :2249-2264
                held = pvA;
                const float* row = g.sh + ((size_t)s_rec[rr].link[cA] * C::D + chA * B);
#pragma unroll
                for (int kk = 0; kk < B; ++kk) val[kk] = row[kk];
            }
            float dsh = 0.f;
#pragma unroll
            for (int kk = 0; kk < B; ++kk) dsh = fmaf(s_sph[rr * SP + kk], val[kk], dsh);
            const float cgd = mineA ? gch * dsh * sp.w : 0.f;
            const float gx = dpp_sum32(cgd * (((cA & 4) ? wy : -wy) * wz));
            const float gy = dpp_sum32(cgd * (((cA & 2) ? wx : -wx) * wz));
            const float gz = dpp_sum32(cgd * (((cA & 1) ? wx : -wx) * wy));
            if ((lane & 31) == 31 && (half == 0 || two)) s_dpos[rr] = make_float4(gx, gy, gz, 0.f);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (mine) {
        const float4 go = want_dpos ? s_dpos[tid] : make_float4(0.f, 0.f, 0.f, 0.f);
        S.gx[idx] = go.x;
        S.gy[idx] = go.y;
        S.gz[idx] = go.z;
    }
  }
  tl.end();
}

__device__ __forceinline__ void acc8(float* dst, const float* w) {
#pragma unroll
    for (int c = 0; c < 8; ++c) dst[c] += w[c];
}

// ------------------------------------------------------------------------------------
// (2 / 3 / 4 waves a SIMD -- 188 registers / 168 and 12 bytes of scratch / 128 and 172 bytes: 0.1105 / 0.1077 / 0.1197 ms)
#ifndef ASURF_BGEO_WAVES
#define ASURF_BGEO_WAVES 3
#endif
// (RELAXED: as k_geom's -- 199 registers instead of 168 + 44 bytes of scratch; 5 184 rays: 0.0276 -> 0.0252 ms)
template <bool RELAXED>
__global__ void __launch_bounds__(kBlock, RELAXED ? 1 : ASURF_BGEO_WAVES)
k_bwd_geo(KGrid g, KOpt o, KLoss L, KGrads G, const float* __restrict__ origins,
          const float* __restrict__ dirs, long long Q, SampleRec S, int sort_chunk) {
    const TlScope tl(7);
  __shared__ float s_head_d[kBlock / 64][64][9], s_head_s[kBlock / 64][64][9];  // per wave: the heads' rows (see the scatter)
  __shared__ int s_head_l[kBlock / 64][64][9];
  RecList RL;
  RL.init(S);
  const int n_list = RL.n;
  const int lane = threadIdx.x & 63;
  const int wave_id = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  // (a long list: two chunks are sorted together -- wave_sort128 -- and then taken one after the other)
  // (measured: at 454 k entries -- the bench scene -- 0.081 ms without, 0.086 with; at 2.7 M and 7 M entries -- the heavy
  // scenes -- 0.36 -> 0.32 and 1.17 -> 0.92 ms)
  const bool wide = (sort_chunk && n_list >= (1 << 20)) || sort_chunk == 2;
  const int take = wide ? 128 : 64;
  for (int base = wave_id * take; base < n_list; base += (int)gridDim.x * (kBlock / 64) * take) {
    // the chunk's entries, sorted by voxel (wave_sort64: a voxel is one run, and most runs follow a face neighbour)
    auto entry = [&](int i) -> unsigned long long {
        if (i >= n_list) return ~0ull;
        const unsigned c0 = RL.cell(S, i);
        const int pv0 = S.v[(size_t)(c0 >> kRecRayBits) * Q + (c0 & ((1u << kRecRayBits) - 1u))];
        return ((unsigned long long)(unsigned)pv0 << 32) | c0;
    };
    unsigned long long ent0 = entry(base + lane);
    unsigned long long ent1 = wide ? entry(base + 64 + lane) : ~0ull;
    if (wide) wave_sort128(ent0, ent1);
    else if (sort_chunk && n_list >= (1 << 16)) ent0 = wave_sort64(ent0);
   for (int part = 0; part < (wide ? 2 : 1); ++part) {
    const unsigned long long ent = part ? ent1 : ent0;
    const bool mine = ent != ~0ull;
    // per-sample contributions to the 8 corner rows of the density and the surface gradient
    float gd[8] = {0, 0, 0, 0, 0, 0, 0, 0}, gs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int lk8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int key = -1 - lane;  // lanes without a sample never join a run
    float gstd_acc = 0.f;
    if (mine) {
    const unsigned cell = (unsigned)ent;
    const long long q = (long long)(cell & ((1u << kRecRayBits) - 1u));
    const int r_i = (int)(cell >> kRecRayBits);
    const size_t idx = (size_t)r_i * Q + q;
    const int meta = S.m[idx];
    const bool fake = meta & 1;
    const int st_id = (meta >> 1) & 3;
    const int li = (meta >> 3) & 31;
    const int ord = (meta >> 8) - 1;  // intersect_i at the sample
    Ray r;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        r.o[k] = origins[3 * q + k];
        r.d[k] = dirs[3 * q + k];
    }
    setup_ray(g, o, r);
    Vox v;
    key = (int)(ent >> 32);
    load_voxel(g, r, key, v);
#pragma unroll
    for (int c = 0; c < 8; ++c) lk8[c] = v.lk[c];
    const float raw_alpha = S.raw[idx];
    const float weight = S.w[idx];
    const float sws = S.sws[q];
    float cgr = S.cgr[idx];
    const float alpha = alpha_act(raw_alpha, o.alpha_activation_type);
    const float trunc_reweight =
        o.truncated_vol_render ? trunc_rw((float)ord, g.trunc_a, o.trunc_vol_weight_min) : 1.f;
    auto sparsity_term = [&](float curr, float a) -> float {  // :2286-2300
        const float _1_a = clamp_lo(1.f - a);
        const double add = -(double)L.sparsity_loss * (1.0 / fmin((double)(_1_a * logf(_1_a)), -1e-8)) *
                           (double)(1.f - weight / sws);
        return (float)((double)curr + add);
    };
    auto inwards = [&](const float* pos, float& curr) {  // :2302-2323
        float sg[3];
        field_grad(v.surf, pos, sg);
        const float nn = clamp_lo(sqrtf(sqrf(sg[0]) + sqrf(sg[1]) + sqrf(sg[2])));
        const float nd = (-sg[0] / nn) * r.d[0] + (-sg[1] / nn) * r.d[1] + (-sg[2] / nn) * r.d[2];
        if (nd > 0.f) curr += L.lambda_inwards_norm_loss * sqrf(nd);
    };

    if (!fake) {
        const float lvf = g.level_set[li];
        double fs[4] = {v.fs[0] - (double)lvf, v.fs[1], v.fs[2], v.fs[3]};
        // the forward stored the sample position; only the branch of the root gradient is needed
        const int root_type = classify_cubic_vieta(fs[0], fs[1], fs[2], fs[3]);
        const float pos[3] = {S.px[idx], S.py[idx], S.pz[idx]};
        if (L.sparsity_loss > 0.f && raw_alpha > 0.f) cgr = sparsity_term(cgr, alpha);
        if (L.lambda_inwards_norm_loss > 0.f) inwards(pos, cgr);
        const float curr_grad_alpha = cgr * trunc_reweight;
        float curr_grad_raw_alpha = curr_grad_alpha * alpha_act_grad(alpha, o.alpha_activation_type);
        {
            float cw[8];
            corner_weights(pos, curr_grad_raw_alpha, cw);
            acc8(gd, cw);
        }
        if (L.lambda_l_di > 0.f && alpha < L.l_di_alpha_thresh)
            curr_grad_raw_alpha += L.lambda_l_di * (-1.f) * alpha_act_grad(alpha, o.alpha_activation_type);
        float grad_xyz[3] = {S.gx[idx], S.gy[idx], S.gz[idx]};
        trilerp_dpos(v.dens, pos, curr_grad_raw_alpha, grad_xyz);
        float grad_st = grad_xyz[0] * r.d[0] + grad_xyz[1] * r.d[1] + grad_xyz[2] * r.d[2];
        grad_st += S.gst[idx];
        float grad_fs[4] = {grad_st, grad_st, grad_st, grad_st};
        cubic_root_grad_vieta(root_type, st_id, fs, grad_fs);
        float gsurf[8];
        const float nof[3] = {(float)((double)v.new_origin[0] - (double)v.vx),
                              (float)((double)v.new_origin[1] - (double)v.vy),
                              (float)((double)v.new_origin[2] - (double)v.vz)};
        surface_cubic_grad(nof, r.d, grad_fs, gsurf);
        acc8(gs, gsurf);
        if (alpha < L.surf_sparse_alpha_thresh) {  // :2436-2445
            float cw[8];
            corner_weights(pos, L.lambda_inplace_surf_sparse, cw);
            acc8(gs, cw);
        }
    } else {  // fake sample (:2460-2866)
        float pos[3];
        fake_position(r, v, pos);
        double miu, sd;
        const float dist = fake_distance(g, o, v, pos, &miu, &sd);
        const float reweight = expf((float)(-.5 * (double)sqrf(dist / g.fake_sample_std)));
        const float rw_alpha = alpha * reweight * trunc_reweight;
        if (L.sparsity_loss > 0.f && raw_alpha > 0.f) cgr = sparsity_term(cgr, rw_alpha);
        if (L.lambda_inwards_norm_loss > 0.f) inwards(pos, cgr);
        const float curr_grad_alpha = cgr * reweight * trunc_reweight;
        const float curr_grad_raw_alpha = curr_grad_alpha * alpha_act_grad(alpha, o.alpha_activation_type);
        {
            float cw[8];
            corner_weights(pos, curr_grad_raw_alpha, cw);
            acc8(gd, cw);
        }
        const float grad_fake_dist =
            cgr * (-alpha * trunc_reweight * dist * reweight / sqrf(g.fake_sample_std));
        float grad_ns[8];
        corner_weights(pos, grad_fake_dist, grad_ns);
        float gsurf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (!o.fake_sample_normalize_surf) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) gsurf[ks] = grad_ns[ks];
        } else {
            const double std3 = cubd(sd);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
                for (int kn = 0; kn < 8; ++kn) {
                    double jac = (double)v.surf[kn] * (miu - (double)v.surf[ks]) / 8.f / std3;
                    if (ks == kn) jac += 1.f / sd;
                    gsurf[ks] = (float)((double)gsurf[ks] + (double)grad_ns[kn] * jac);
                }
            }
        }
        acc8(gs, gsurf);
        if (G.grad_fake_sample_std != nullptr) {
            const float gstd = cgr * alpha * sqrf(dist) * reweight * trunc_reweight /
                               (g.fake_sample_std * g.fake_sample_std * g.fake_sample_std);
            gstd_acc += gstd;
        }
    }
    }  // mine
    // Neighbouring list entries are neighbouring rays, mostly in the same voxel: runs of equal
    // voxels are summed across lanes (segmented doubling over contiguous lanes) and only the
    // first lane of a run touches memory -- the [N,1] planes cost one L2 transaction per
    // scattered atomic, so fewer atomics is what counts.
    const int key_up = __shfl_up(key, 1);
    const bool head = (lane == 0) || (key_up != key);
    // run number = heads at or below this lane: monotone, so equal numbers at distance d mean
    // every lane in between belongs to the same run (a voxel may come back in a later run)
    const int run = __popcll(__ballot(head) & ((2ull << lane) - 1ull));
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int k2 = __shfl_down(run, d);
        const bool take = (lane + d < 64) && (k2 == run);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float a = __shfl_down(gd[c], d), b2 = __shfl_down(gs[c], d);
            gd[c] += take ? a : 0.f;
            gs[c] += take ? b2 : 0.f;
        }
    }
    // Successive runs mostly sit in voxels that share a face: the four rows of that face are handed to the next
    // run's head lane (they are its corners c ^ b) and leave from there with that run's own terms -- one atomic per
    // plane instead of two.  A row a head has itself received is not handed on.
    {
        const bool is_head = mine && head;
        const unsigned long long H = __ballot(is_head);
        const unsigned long long above = (lane < 63) ? (H & ~((2ull << lane) - 1ull)) : 0ull;
        const unsigned long long below = H & ((1ull << lane) - 1ull);
        const int nh = above ? __ffsll((long long)above) - 1 : -1;
        const int ph = below ? 63 - __clzll((long long)below) : -1;
        const int key_n = __shfl(key, nh >= 0 ? nh : lane);
        const int key_p = __shfl(key, ph >= 0 ? ph : lane);
        // step from voxel a to voxel b: bits of the axes moved along (0: not a neighbour) | << 3 those moved upwards
        auto step_code = [](int a, int b2) {
            const int dx = (b2 >> 20) - (a >> 20), dy = ((b2 >> 10) & 1023) - ((a >> 10) & 1023), dz = (b2 & 1023) - (a & 1023);
            const bool near = dx >= -1 && dx <= 1 && dy >= -1 && dy <= 1 && dz >= -1 && dz <= 1;
            const int bb = (dx ? 4 : 0) | (dy ? 2 : 0) | (dz ? 1 : 0);
            const int hh = (dx > 0 ? 4 : 0) | (dy > 0 ? 2 : 0) | (dz > 0 ? 1 : 0);
            return near ? (bb | (hh << 3)) : 0;
        };
        const int cn = (is_head && nh >= 0) ? step_code(key, key_n) : 0;
        const int cp = (is_head && ph >= 0) ? step_code(key_p, key) : 0;
        const int bn = cn & 7, hin = cn >> 3, bp = cp & 7, hip = cp >> 3;
        // what the previous head hands over depends on ITS step from its predecessor
        const int cpp = __shfl(cp, ph >= 0 ? ph : lane);
        const int bpp = cpp & 7, hipp = cpp >> 3;
        float pd[8], ps[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            pd[c] = __shfl(gd[c], ph >= 0 ? ph : lane);
            ps[c] = __shfl(gs[c], ph >= 0 ? ph : lane);
        }
        if (is_head) {
            float od[8], os[8];
            bool give[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                // received into c: the previous head's corner c ^ bp, if that lies on its face towards this voxel and
                // not on its face towards ITS predecessor
                const int cq = c ^ bp;  // the previous voxel's corner at this vertex
                const bool prev_shared = bp != 0 && (cq & bp) == hip;
                const bool prev_received = bpp != 0 && ((cq ^ bpp) & bpp) == hipp;
                const bool recv = prev_shared && !prev_received;
                float rd_ = 0.f, rs_ = 0.f;
#pragma unroll
                for (int m = 1; m < 8; ++m) {
                    rd_ = bp == m ? pd[c ^ m] : rd_;
                    rs_ = bp == m ? ps[c ^ m] : rs_;
                }
                od[c] = gd[c] + (recv ? rd_ : 0.f);
                os[c] = gs[c] + (recv ? rs_ : 0.f);
                // handed to the next head: on my face towards it, unless it also lies on my face towards the previous
                // head (the same geometric rule the receiver applies to me: prev_received above)
                const bool next_shared = bn != 0 && (c & bn) == hin;
                const bool on_prev_face = bp != 0 && ((c ^ bp) & bp) == hip;
                give[c] = next_shared && !on_prev_face;
            }
            // The scatter.  An atomic costs one request per 64-byte line an INSTRUCTION touches, whatever the number of
            // lanes in the line (variants/micro/atomics.hip: 21 G lines/s on the whole chip, scattered floats or rows
            // alike), and the rows of a voxel's two corners along z are neighbours in memory wherever the links were
            // numbered in grid order.  So the heads do not issue their eight corners one after the other: they
            // leave them in LDS, and the wave writes them out eight heads at a time, lane = (head, corner).
            const int hr = __popcll(H & ((1ull << lane) - 1ull));
            float* hd = s_head_d[threadIdx.x >> 6][hr];
            float* hs = s_head_s[threadIdx.x >> 6][hr];
            int* hl = s_head_l[threadIdx.x >> 6][hr];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                hd[c] = od[c];
                hs[c] = os[c];
                hl[c] = give[c] ? -1 : lk8[c];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const int n_heads = __popcll(H);
        for (int h0 = 0; h0 < n_heads; h0 += 8) {
            const int hh = h0 + (lane >> 3), c = lane & 7;
            if (hh < n_heads) {
                const int lk = s_head_l[threadIdx.x >> 6][hh][c];
                if (lk >= 0) {
                    if (G.grad_density != nullptr) atomicAdd(&G.grad_density[lk], s_head_d[threadIdx.x >> 6][hh][c]);
                    if (G.grad_surface != nullptr) atomicAdd(&G.grad_surface[lk], s_head_s[threadIdx.x >> 6][hh][c]);
                    if (G.mask != nullptr) G.mask[lk] = 1;
                    if (G.mask2 != nullptr) G.mask2[lk] = 1;
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // (the next turn writes the table again)
    }
    if (G.grad_fake_sample_std != nullptr) {
        const float tot = wave_sum(gstd_acc);
        if (lane == 0 && tot != 0.f) atomicAdd(G.grad_fake_sample_std, tot);
    }
   }  // part
  }
  tl.end();
}

}  // namespace asurf
