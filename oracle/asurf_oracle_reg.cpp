/*
 * asurf_oracle_reg.cpp -- CPU ORACLE (test infrastructure, not product code) for the
 * per-iteration regularisers.  Sequential restatement of
 * loss_kernel.cu's sparse kernels and of the eikonal loss the reference evaluates with
 * autograd.  Only tests/ may load it.
 *
 * Parity pinning: surface-normal loss and eikonal loss against gradients produced by the
 * reference's own PyTorch implementations (svox2.py:5165-5345, 5433-5472; the procedure of
 * test/test_cuda_utils.py:305-363) -- tests/golden/gen_golden_reg.py.  The two TV kernels and
 * the sparsify kernel have no PyTorch counterpart in the reference: "parity unpinned" for those
 * three; they are checked by finite differences of the loss they are the gradient of.
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstddef>

extern "C" {

static inline void decode(const int32_t* size, int xyz, int& x, int& y, int& z) {
    z = xyz % size[2];
    const int xy = xyz / size[2];
    y = xy % size[1];
    x = xy / size[1];
}

static inline void add_set(float* grad, uint8_t* mask, int lnk, int C, int idx, float val, float idelta) {
    if (lnk >= 0 && val != 0.f) {
        grad[(size_t)lnk * C + idx] += val * idelta;
        if (mask) mask[lnk] = 1;
    }
}

/* loss_kernel.cu:739-808 (surf = 0) and :810-895 (surf = 1); launchers :1327-1427 */
int oracle_tv_grad_sparse(const int32_t* links, const int32_t* size, const float* data, int32_t C,
                          const float* density, int32_t density_C, const int32_t* rand_cells,
                          int64_t n_cells, uint8_t* mask_out, int32_t start_dim, int32_t end_dim, float scale,
                          int32_t ignore_edge, float edge_value, int32_t ignore_last_z, int32_t surf,
                          int32_t alpha_dependency, float* grad) {
    if (n_cells <= 0) return 0;
    const int offx = size[1] * size[2], offy = size[2];
    const float sc = scale / (float)(int)n_cells;
    const float scaling[3] = {size[0] * (1.f / 256.f), size[1] * (1.f / 256.f), size[2] * (1.f / 256.f)};
    for (int64_t cell = 0; cell < n_cells; ++cell) {
        int x, y, z;
        decode(size, rand_cells[cell], x, y, z);
        const int32_t* lp = links + ((size_t)x * offx + (size_t)y * offy + z);
        if (ignore_edge && *lp == 0) continue;
        const int l000 = lp[0];
        const int l001 = ((z + 1 < size[2]) && (!ignore_last_z || z != size[2] - 2)) ? lp[1] : 0;
        const int l010 = (y + 1 < size[1]) ? lp[offy] : 0;
        const int l100 = (x + 1 < size[0]) ? lp[offx] : 0;
        if (ignore_last_z && z == size[2] - 2) continue;
        for (int idx = start_dim; idx < end_dim; ++idx) {
            const float dflt = surf ? edge_value : 0.f;
            const float v000 = l000 >= 0 ? data[(size_t)l000 * C + idx] : dflt;
            const float nullv = ignore_edge ? v000 : dflt;
            const float v001 = l001 >= 0 ? data[(size_t)l001 * C + idx] : nullv;
            const float v010 = l010 >= 0 ? data[(size_t)l010 * C + idx] : nullv;
            const float v100 = l100 >= 0 ? data[(size_t)l100 * C + idx] : nullv;
            float dx = v100 - v000, dy = v010 - v000, dz = v001 - v000;
            float idelta = sc * (1.f / sqrtf(1e-9f + dx * dx + dy * dy + dz * dz));
            if (surf && alpha_dependency) {
                const float a0 = l000 >= 0 ? density[(size_t)l000 * density_C + idx] : 0.f;
                const float a1 = l001 >= 0 ? density[(size_t)l001 * density_C + idx] : 0.f;
                const float a2 = l010 >= 0 ? density[(size_t)l010 * density_C + idx] : 0.f;
                const float a3 = l100 >= 0 ? density[(size_t)l100 * density_C + idx] : 0.f;
                const float mx = std::max(a0, std::max(a1, std::max(a2, a3)));
                if (mx < 0.1f) idelta /= std::max(mx * 10, 1e-1f);
            }
            dx *= scaling[0];
            dy *= scaling[1];
            dz *= scaling[2];
            add_set(grad, mask_out, l000, C, idx, -(dx + dy + dz), idelta);
            add_set(grad, mask_out, l001, C, idx, dz, idelta);
            add_set(grad, mask_out, l010, C, idx, dy, idelta);
            add_set(grad, mask_out, l100, C, idx, dx, idelta);
        }
    }
    return 0;
}

namespace {
struct Cell {
    int lk[8];
    float s[8];
    bool exist;
};

void load_cell(const int32_t* links, const int32_t* size, const float* surface, int x, int y, int z, Cell& c) {
    const int offx = size[1] * size[2], offy = size[2];
    c.exist = (x < size[0] - 1) && (y < size[1] - 1) && (z < size[2] - 1);
    if (!c.exist) return;
    const int32_t* p = links + ((size_t)x * offx + (size_t)y * offy + z);
    const int off[8] = {0, 1, offy, offy + 1, offx, offx + 1, offx + offy, offx + offy + 1};
    for (int i = 0; i < 8; ++i) {
        c.lk[i] = p[off[i]];
        if (c.lk[i] < 0) c.exist = false;
    }
    if (!c.exist) return;
    for (int i = 0; i < 8; ++i) c.s[i] = surface[c.lk[i]];
}

void normal_of(const Cell& c, float* n) {  // __CALC_DX / DY / DZ, render_util.cuh:1905-1913
    n[0] = ((c.s[4] + c.s[5] + c.s[6] + c.s[7]) - (c.s[0] + c.s[1] + c.s[2] + c.s[3])) / 4;
    n[1] = ((c.s[2] + c.s[3] + c.s[6] + c.s[7]) - (c.s[0] + c.s[1] + c.s[4] + c.s[5])) / 4;
    n[2] = ((c.s[1] + c.s[3] + c.s[5] + c.s[7]) - (c.s[0] + c.s[2] + c.s[4] + c.s[6])) / 4;
}

bool one_side(const float* s, int n, float lv) {
    bool le = true, ge = true;
    for (int i = 0; i < n; ++i) {
        le = le && (s[i] <= lv);
        ge = ge && (s[i] >= lv);
    }
    return le || ge;
}

void split_add(const Cell& c, const float* gn, float scale, uint8_t* mask, float* grad) {
    for (int i = 0; i < 8; ++i) {
        const float g = ((i & 4) ? 0.25f : -0.25f) * gn[0] + ((i & 2) ? 0.25f : -0.25f) * gn[1] +
                        ((i & 1) ? 0.25f : -0.25f) * gn[2];
        const float val = scale * g;
        if (val != 0.f) {
            grad[c.lk[i]] += val;
            if (mask) mask[c.lk[i]] = 1;
        }
    }
}
}  // namespace

/* loss_kernel.cu:398-436, 1572-1622; add_surface_normal_grad, render_util.cuh:1870-2130 */
int oracle_surface_normal_grad_sparse(const int32_t* links, const int32_t* size, const float* surface,
                                      const int32_t* rand_cells, int64_t n_cells, uint8_t* mask_out,
                                      float lv_set, float scale, int32_t con_check, int32_t ignore_empty,
                                      int32_t use_l1, float* grad) {
    if (n_cells <= 0) return 0;
    const float sc_all = scale / (float)(int)n_cells;
    for (int64_t cell = 0; cell < n_cells; ++cell) {
        int x, y, z;
        decode(size, rand_cells[cell], x, y, z);
        Cell c0;
        load_cell(links, size, surface, x, y, z, c0);
        if (!c0.exist) continue;
        const bool empty0 = ignore_empty ? one_side(c0.s, 8, lv_set) : false;
        float n0[3];
        normal_of(c0, n0);
        Cell cn[3];
        bool use[3];
        int count = 0;
        for (int i = 0; i < 3; ++i) {
            load_cell(links, size, surface, x + (i == 0), y + (i == 1), z + (i == 2), cn[i]);
            bool ok = cn[i].exist;
            if (ok && con_check) {
                const int bit = 4 >> i;
                float f[4];
                int k = 0;
                for (int j = 0; j < 8; ++j)
                    if (!(j & bit)) f[k++] = cn[i].s[j];
                ok = !one_side(f, 4, lv_set);
            }
            if (ok && ignore_empty) ok = !empty0 || !one_side(cn[i].s, 8, lv_set);
            use[i] = ok;
            count += ok;
        }
        if (!count) continue;
        const float N0 = sqrtf(1e-9f + n0[0] * n0[0] + n0[1] * n0[1] + n0[2] * n0[2]);
        const float sc = sc_all * 1.f / count;
        for (int i = 0; i < 3; ++i) {
            if (!use[i]) continue;
            float n1[3];
            normal_of(cn[i], n1);
            const float N1 = sqrtf(1e-9f + n1[0] * n1[0] + n1[1] * n1[1] + n1[2] * n1[2]);
            float w[3];
            for (int b = 0; b < 3; ++b) {
                const float L = n0[b] / N0 - n1[b] / N1;
                w[b] = use_l1 ? ((L > 0.f) ? 1.f : (L == 0.f ? 0.f : -1.f)) : 2.f * L;
            }
            float d0[3], d1[3];
            const float i0 = 1.f / (N0 * N0 * N0), i1 = 1.f / (N1 * N1 * N1);
            for (int a = 0; a < 3; ++a) {
                float s0 = 0.f, s1 = 0.f;
                for (int b = 0; b < 3; ++b) {
                    s0 += w[b] * (((a == b) ? 1.f / N0 : 0.f) - n0[a] * n0[b] * i0);
                    s1 += w[b] * (((a == b) ? -1.f / N1 : 0.f) + n1[a] * n1[b] * i1);
                }
                d0[a] = s0;
                d1[a] = s1;
            }
            split_add(c0, d0, sc, mask_out, grad);
            split_add(cn[i], d1, sc, mask_out, grad);
        }
    }
    return 0;
}

/* loss_kernel.cu:665-738, 1512-1570 */
int oracle_alpha_surf_sparsify_grad_sparse(const int32_t* links, const int32_t* size, const float* alpha,
                                           const float* surf, const int32_t* rand_cells, int64_t n_cells,
                                           uint8_t* mask_out, float scale_alpha, float scale_surf,
                                           int32_t surf_decrease, float surf_thresh, float alpha_bound,
                                           float surf_bound, float* grad_alpha, float* grad_surf) {
    const int offx = size[1] * size[2], offy = size[2];
    for (int64_t cell = 0; cell < n_cells; ++cell) {
        int x, y, z;
        decode(size, rand_cells[cell], x, y, z);
        const int l = links[(size_t)x * offx + (size_t)y * offy + z];
        if (l < 0) continue;
        if (mask_out) mask_out[l] = 1;
        const float a = alpha[l];
        const float safe = 1.f / std::max(a, 1e-8f);
        if (a > alpha_bound) grad_alpha[l] += scale_alpha * safe;
        const bool reg = surf_decrease ? (surf[l] > surf_bound) : (surf[l] < surf_bound);
        if (reg && a < surf_thresh) grad_surf[l] += surf_decrease ? (scale_surf * safe) : (-scale_surf * safe);
    }
    return 0;
}

/* svox2.py:5433-5472 (_surface_eikonal_loss_grad_check): returns the loss in *loss_out */
int oracle_surface_eikonal_grad_sparse(const int32_t* links, const int32_t* size, const float* surface,
                                       const int32_t* rand_cells, int64_t n_cells, uint8_t* mask_out,
                                       float scale, float h, float* grad, double* loss_out) {
    const int offx = size[1] * size[2], offy = size[2];
    double loss = 0;
    const float sc = scale / (float)(int)n_cells;
    const float inv_h2 = 1.f / (h * h);
    for (int64_t cell = 0; cell < n_cells; ++cell) {
        int x, y, z;
        decode(size, rand_cells[cell], x, y, z);
        const int32_t* lp = links + ((size_t)x * offx + (size_t)y * offy + z);
        const int l000 = lp[0];
        if (l000 < 0) continue;
        const int l100 = (x + 1 < size[0]) ? lp[offx] : -1;
        const int l010 = (y + 1 < size[1]) ? lp[offy] : -1;
        const int l001 = (z + 1 < size[2]) ? lp[1] : -1;
        const float s0 = surface[l000];
        const float dx = (l100 >= 0 ? surface[l100] : s0) - s0;
        const float dy = (l010 >= 0 ? surface[l010] : s0) - s0;
        const float dz = (l001 >= 0 ? surface[l001] : s0) - s0;
        const float g2 = (dx * dx + dy * dy + dz * dz) * inv_h2;
        loss += (double)(sc * (1.f - g2) * (1.f - g2));
        const float k = -4.f * sc * (1.f - g2) * inv_h2;
        const float gx = k * dx, gy = k * dy, gz = k * dz;
        if (gx != 0.f) grad[l100] += gx;
        if (gy != 0.f) grad[l010] += gy;
        if (gz != 0.f) grad[l001] += gz;
        const float gc = -(gx + gy + gz);
        if (gc != 0.f) grad[l000] += gc;
        if (mask_out) {
            mask_out[l000] = 1;
            if (gx != 0.f) mask_out[l100] = 1;
            if (gy != 0.f) mask_out[l010] = 1;
            if (gz != 0.f) mask_out[l001] = 1;
        }
    }
    if (loss_out) *loss_out = loss;
    return 0;
}

}  // extern "C"
