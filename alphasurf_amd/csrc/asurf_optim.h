// asurf_optim.h -- per-element RMSProp / SGD steps.
//
// Replaces rmsprop_step / sgd_step of the reference (svox2/csrc/optim_kernel.cu:14-265): dense,
// bool-row-mask and int64-row-index variants; the gradient element is zeroed after use; the last
// channel may use its own learning rate.  Pure streaming: 12 B read + 12 B written per element.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace asurf {

// mode 0: every row; 1: rows with mask[row] != 0; 2: rows listed in indices[0..n_idx).
// Lane -> (row, column) without a division: a row owns a group of CP = 2^k >= n_cols consecutive lanes (CP <= 64,
// so a group never straddles a wave), lanes >= n_cols of a group idle (27 columns: 5 of 32).  A masked-out row
// costs its group one byte read; the rows that are updated stream 12 B in and 12 B out per element, row-contiguous.
// Wider rows (n_cols > 64) take the generic kernel below.
template <bool RMS>
__device__ __forceinline__ void optim_once(float* __restrict__ data, float* __restrict__ rms, float* __restrict__ grad,
                                           long long off, bool last, float beta, float lr, float epsilon, float minval,
                                           float lr_last) {
    const float g = grad[off];
    const float step = last ? lr_last : lr;
    if (RMS) {  // rmsprop_once, optim_kernel.cu:15-25
        float r = rms[off];
        r = (r == 0.f) ? g * g : fmaf(beta, r - g * g, g * g);  // lerp(g^2, rms, beta)
        rms[off] = r;
        data[off] = fmaxf(data[off] - step * g / (sqrtf(r) + epsilon), minval);
    } else {  // sgd_once, :96-102
        data[off] -= step * g;
    }
    grad[off] = 0.f;
}

// Row mask -> list of the rows it names, count left ON THE DEVICE (the steps below read it there): the reference
// decides between mask and index list on the host (`torch.count_nonzero(...).item()`, svox2.py:6326-6332), a device ->
// host round trip per optimizer step that stalls the launch queue of a whole training iteration.  Block = 4096 rows:
// per-thread counts, one scan, one atomic per block; the order of the list is irrelevant to its readers.
constexpr int kCompactRowsPerBlock = 4096;
__global__ void __launch_bounds__(256)
k_compact_rows(const uint8_t* __restrict__ mask, long long n_rows, long long* __restrict__ indices,
               int* __restrict__ count) {
    __shared__ int s_wave[4], s_base;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const long long row0 = (long long)blockIdx.x * kCompactRowsPerBlock;
    unsigned flags = 0;
#pragma unroll
    for (int j = 0; j < kCompactRowsPerBlock / 256; ++j) {
        const long long row = row0 + t + 256 * j;
        if (row < n_rows && mask[row]) flags |= 1u << j;
    }
    const int c = __popc(flags);
    int incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d);
        if (lane >= d) incl += v;
    }
    if (lane == 63) s_wave[w] = incl;
    __syncthreads();
    if (t == 0) {
        const int total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        s_base = total > 0 ? atomicAdd(count, total) : 0;
    }
    __syncthreads();
    int at = s_base + incl - c;
    for (int k = 0; k < w; ++k) at += s_wave[k];
#pragma unroll
    for (int j = 0; j < kCompactRowsPerBlock / 256; ++j)
        if (flags & (1u << j)) indices[at++] = row0 + t + 256 * j;
}

// n_idx_dev != nullptr: the length of `indices` is read from the device (k_compact_rows' count)
template <bool RMS>
__global__ void __launch_bounds__(256)
k_optim_rows(float* __restrict__ data, float* __restrict__ rms, float* __restrict__ grad, long long n_rows, int n_cols,
             int cp_shift, int mode, const uint8_t* __restrict__ mask, const long long* __restrict__ indices,
             long long n_idx, float beta, float lr, float epsilon, float minval, float lr_last,
             const int* __restrict__ n_idx_dev) {
    if (n_idx_dev != nullptr) {  // the list is k_compact_rows': walk it if it is short, else the mask it came from
        n_idx = *n_idx_dev;
        mode = (mask != nullptr && 4 * n_idx >= n_rows) ? 1 : 2;
    }
    const long long groups = mode == 2 ? n_idx : n_rows;
    const long long stride = ((long long)gridDim.x * blockDim.x) >> cp_shift;
    const int chnl = (int)(threadIdx.x & ((1u << cp_shift) - 1u));
    if (chnl >= n_cols) return;
    for (long long grp = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> cp_shift; grp < groups; grp += stride) {
        long long row = grp;
        if (mode == 1) {
            if (!mask[row]) continue;
        } else if (mode == 2) {
            row = indices[grp];
        }
        optim_once<RMS>(data, rms, grad, row * n_cols + chnl, chnl == n_cols - 1, beta, lr, epsilon, minval, lr_last);
    }
}

// ---- the masked steps of a training iteration (round 4) --------------------------------------------------------------
// k_compact_rows allocates a block's rows in ONE list with an atomic that returns its place: 320 blocks on one address,
// 15 us for a 1.3 M-row mask -- as long as the step itself -- and a memset in front of it.  Here the list is cut into
// segments of kSegRows rows of the mask: a block lists the rows of its four segments with ballots, writes each segment's
// count, and nothing is shared between blocks; the step's blocks take segments in turn, a lane group a listed row.
// A full mask costs the step what the mask walk cost it (a count a segment instead of a byte a row), a sparse one its
// rows plus a count per segment -- no decision between list and mask, no read-back.
constexpr int kSegRows = 256;
constexpr int kSegBlockRows = 1024;   // rows a block of k_compact_segments lists (four segments: 1 280 blocks at 1.3 M rows)
__global__ void __launch_bounds__(256)
k_compact_segments(const uint8_t* __restrict__ mask, long long n_rows, int* __restrict__ seg_rows, int* __restrict__ seg_count) {
    __shared__ int s_cnt[kSegBlockRows / kSegRows][4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const long long row0 = (long long)blockIdx.x * kSegBlockRows;
    unsigned flags = 0;
    int before[kSegBlockRows / kSegRows];   // set rows of this wave in front of this lane, per segment
#pragma unroll
    for (int j = 0; j < kSegBlockRows / kSegRows; ++j) {
        const long long row = row0 + t + kSegRows * j;
        const bool set = row < n_rows && mask[row] != 0;
        const unsigned long long b = __ballot(set);
        flags |= set ? 1u << j : 0u;
        before[j] = __popcll(b & ((1ull << lane) - 1ull));
        if (lane == 0) s_cnt[j][w] = __popcll(b);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kSegBlockRows / kSegRows; ++j) {
        const long long seg = (long long)blockIdx.x * (kSegBlockRows / kSegRows) + j;
        if (seg * kSegRows >= n_rows) break;
        int at = before[j];
        for (int k = 0; k < w; ++k) at += s_cnt[j][k];
        if (flags & (1u << j)) seg_rows[seg * kSegRows + at] = (int)(row0 + t + kSegRows * j);
        if (t == 0) seg_count[seg] = s_cnt[j][0] + s_cnt[j][1] + s_cnt[j][2] + s_cnt[j][3];
    }
}

// The same in ONE kernel (what the training iteration runs): a block reads the 256 mask bytes of a segment, lists the
// rows that are set in LDS with ballots, and its lane groups take the listed rows -- no list in memory, no second
// launch (list + step: 5.7 + 7.8 us a step at 1.3 M rows, three steps an iteration; this: one launch).
template <bool RMS>
__global__ void __launch_bounds__(256)
k_optim_masked(float* __restrict__ data, float* __restrict__ rms, float* __restrict__ grad, long long n_rows, int n_cols,
               int cp_shift, const uint8_t* __restrict__ mask, int n_seg, float beta, float lr, float epsilon,
               float minval, float lr_last) {
    __shared__ int s_rows[kSegRows];
    __shared__ int s_cnt[4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int chnl = (int)(t & ((1u << cp_shift) - 1u));
    const int gpb = (int)blockDim.x >> cp_shift;
    for (int seg = blockIdx.x; seg < n_seg; seg += gridDim.x) {
        const long long row = (long long)seg * kSegRows + t;
        const bool set = row < n_rows && mask[row] != 0;
        if (cp_shift == 0) {   // a column: the thread that read the byte takes the row
            if (set) optim_once<RMS>(data, rms, grad, row, true, beta, lr, epsilon, minval, lr_last);
            continue;
        }
        const unsigned long long b = __ballot(set);
        if (lane == 0) s_cnt[w] = __popcll(b);
        __syncthreads();
        int at = __popcll(b & ((1ull << lane) - 1ull));
        for (int k = 0; k < w; ++k) at += s_cnt[k];
        if (set) s_rows[at] = (int)(row - (long long)seg * kSegRows);
        const int n = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        __syncthreads();
        if (chnl < n_cols) {
            for (int i = t >> cp_shift; i < n; i += gpb) {
                const long long r = (long long)seg * kSegRows + s_rows[i];
                optim_once<RMS>(data, rms, grad, r * n_cols + chnl, chnl == n_cols - 1, beta, lr, epsilon, minval, lr_last);
            }
        }
        __syncthreads();   // (the next segment writes the tables again)
    }
}

template <bool RMS>
__global__ void __launch_bounds__(256)
k_optim_segments(float* __restrict__ data, float* __restrict__ rms, float* __restrict__ grad, int n_cols, int cp_shift,
                 const int* __restrict__ seg_rows, const int* __restrict__ seg_count, int n_seg, float beta, float lr,
                 float epsilon, float minval, float lr_last) {
    const int chnl = (int)(threadIdx.x & ((1u << cp_shift) - 1u));
    if (chnl >= n_cols) return;
    const int gpb = (int)blockDim.x >> cp_shift;   // lane groups (rows at a time) a block
    for (int seg = blockIdx.x; seg < n_seg; seg += gridDim.x) {
        const int n = seg_count[seg];
        for (int i = (int)threadIdx.x >> cp_shift; i < n; i += gpb) {
            const long long row = seg_rows[(size_t)seg * kSegRows + i];
            optim_once<RMS>(data, rms, grad, row * n_cols + chnl, chnl == n_cols - 1, beta, lr, epsilon, minval, lr_last);
        }
    }
}

template <bool RMS>
__global__ void __launch_bounds__(256)
k_optim_step(float* __restrict__ data, float* __restrict__ rms, float* __restrict__ grad,
             long long n_rows, int n_cols, int mode, const uint8_t* __restrict__ mask,
             const long long* __restrict__ indices, long long n_idx, float beta, float lr,
             float epsilon, float minval, float lr_last, const int* __restrict__ n_idx_dev) {
    if (n_idx_dev != nullptr) {
        n_idx = *n_idx_dev;
        mode = (mask != nullptr && 4 * n_idx >= n_rows) ? 1 : 2;
    }
    const long long total = (mode == 2 ? n_idx : n_rows) * n_cols;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x; tid < total; tid += stride) {
        long long row = tid / n_cols;
        const int chnl = (int)(tid - row * n_cols);
        if (mode == 1) {
            if (!mask[row]) continue;
        } else if (mode == 2) {
            row = indices[row];
        }
        optim_once<RMS>(data, rms, grad, row * n_cols + chnl, chnl == n_cols - 1, beta, lr, epsilon, minval, lr_last);
    }
}

}  // namespace asurf
