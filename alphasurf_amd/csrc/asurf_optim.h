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

template <bool RMS>
__global__ void __launch_bounds__(256)
k_optim_rows(float* __restrict__ data, float* __restrict__ rms, float* __restrict__ grad, long long n_rows, int n_cols,
             int cp_shift, int mode, const uint8_t* __restrict__ mask, const long long* __restrict__ indices,
             long long n_idx, float beta, float lr, float epsilon, float minval, float lr_last) {
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

template <bool RMS>
__global__ void __launch_bounds__(256)
k_optim_step(float* __restrict__ data, float* __restrict__ rms, float* __restrict__ grad,
             long long n_rows, int n_cols, int mode, const uint8_t* __restrict__ mask,
             const long long* __restrict__ indices, long long n_idx, float beta, float lr,
             float epsilon, float minval, float lr_last) {
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
