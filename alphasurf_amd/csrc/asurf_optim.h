// asurf_optim.h -- per-element RMSProp / SGD steps.
//
// Replaces rmsprop_step / sgd_step of the reference (svox2/csrc/optim_kernel.cu:14-265): dense,
// bool-row-mask and int64-row-index variants; the gradient element is zeroed after use; the last
// channel may use its own learning rate.  Pure streaming: 12 B read + 12 B written per element.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace asurf {

// mode 0: every row; 1: rows with mask[row] != 0; 2: rows listed in indices[0..n_idx)
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
        const long long off = row * n_cols + chnl;
        const float g = grad[off];
        const float step = (chnl == n_cols - 1) ? lr_last : lr;
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
}

}  // namespace asurf
