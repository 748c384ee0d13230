"""Known-answer scenes for the gates that only the reference's CUDA path has (difference 3:
`only_outward_intersect` surf_trav.cu:350-367, the density gates :230-239 / :380, `stop_thresh` :544-547), worked out
WITHOUT the oracle: closed forms in float64 numpy.

The scene is a slab: a dense R^3 grid whose surface field is s(x) = w - |x_a - c| along one axis a, with c on a
vertex plane.  s is piecewise linear with its kink on a vertex plane, so the trilinear interpolant IS s: the level
set 0 consists of the two planes x_a = c -+ w exactly, every ray meets each plane once, and where it does is
elementary.  The raw opacity is linear in another coordinate (sigma = s0 + s1 * x_b: trilinear interpolation
reproduces it exactly too) and the colour is constant per channel (SH band 0 only), so a sample's alpha depends on
where the crossing is and nothing else does.  Per ray, with samples i = 1, 2 in travel order that pass the gates:

    alpha_i = 1 - exp(-sigma(p_i))                          (exp activation; sigma(p_i) > sigma_thresh, > 0)
    rgb_k   = sum_i T_i alpha_i c_k + T_end * bg,           T_1 = 1, T_{i+1} = T_i (1 - alpha_i)
              ... the background term is dropped once T < stop_thresh after a sample, and so are later samples
    c_k     = max(0.5 + Y_00 * sh_k, 0)

and the gradients of L = lambda_l2 * sum |rgb - gt|^2 / (3 Q) follow by the chain rule: a sample's dL/dsigma and
dL/dc_k are spread over the 8 corners of its voxel with the trilinear weights W_v(p_i).  The surface gradient is
predicted only where it must vanish (constant sigma: nothing depends on where the crossing is): along a ray the
slab's field is exactly linear, and on that branch the reference's root gradient drops the terms of the vanishing
cubic coefficients (render_util.cuh:1340-1415; test/test_cuda_utils.py asserts the zeros) -- it is not the derivative
of the crossing there, and restating that formula here would test nothing.

A second scene, a sphere (s = r - |x - c|, positive inside as in the reference's initialisation), puts the true
cubic solver on the path: with constant sigma and constant colour a ray's RGB depends only on HOW MANY crossings
pass the gates -- two, or the entry crossing alone with only_outward_intersect -- and the sums of grad_density and of
grad_sh over all rows are closed form (trilinear weights add up to 1).
Grid coordinates throughout (the scenes use offset 0, scaling 1)."""
import numpy as np

Y00 = 0.28209479177387814


def slab_scene(reso=32, axis=0, center=16, half_width=5.3, sigma0=0.4, sigma1=0.05, sigma_axis=1,
               sh=(0.6, -0.3, 1.1), basis_dim=1):
    R = reso
    ax = np.arange(R, dtype=np.float64)
    coords = np.meshgrid(ax, ax, ax, indexing="ij")
    n = R ** 3
    links = np.arange(n, dtype=np.int32).reshape(R, R, R)
    surface = (half_width - np.abs(coords[axis] - center)).reshape(n, 1).astype(np.float32)
    density = (sigma0 + sigma1 * coords[sigma_axis]).reshape(n, 1).astype(np.float32)
    shd = np.zeros((n, 3 * basis_dim), np.float32)
    for k in range(3):
        shd[:, k * basis_dim] = sh[k]
    sc = dict(links=links, density=density, surface=surface, sh=shd, level_set=np.zeros(1, np.float32),
              offset=np.zeros(3, np.float32), scaling=np.ones(3, np.float32), basis_dim=basis_dim,
              fake_sample_std=1.0, truncated_vol_render_a=1.0, surface_type=102)
    model = dict(reso=R, axis=axis, center=float(center), half_width=float(half_width), sigma0=float(sigma0),
                 sigma1=float(sigma1), sigma_axis=sigma_axis, sh=np.asarray(sh, np.float64))
    return sc, model


def slab_rays(model, n=400, seed=0):
    """Rays that start outside the grid and cross both planes well inside the vertex lattice, from either side, plus
    a few that run parallel to the slab outside it (background only)."""
    rs = np.random.RandomState(seed)
    R, a = model["reso"], model["axis"]
    lo, hi = model["center"] - model["half_width"], model["center"] + model["half_width"]
    o = np.zeros((n, 3))
    d = np.zeros((n, 3))
    for q in range(n):
        side = rs.rand() < 0.5
        p_in = rs.uniform(4.0, R - 5.0, size=3)       # where the ray meets the first plane
        p_out = rs.uniform(4.0, R - 5.0, size=3)      # ... and the second one
        p_in[a], p_out[a] = (lo, hi) if side else (hi, lo)
        dirv = p_out - p_in
        dirv /= np.linalg.norm(dirv)
        o[q] = p_in - dirv * rs.uniform(30.0, 40.0)
        d[q] = dirv
    # rays that miss the slab: parallel to it, at x_a = 2.5
    for q in range(0, n, 25):
        dirv = rs.randn(3)
        dirv[a] = 0.0
        dirv /= np.linalg.norm(dirv)
        p = rs.uniform(6.0, R - 7.0, size=3)
        p[a] = 2.5
        o[q], d[q] = p - 40.0 * dirv, dirv
    return o.astype(np.float32), d.astype(np.float32)


def expected(model, origins, dirs, rgb_gt, only_outward=False, sigma_thresh=-1e4, stop_thresh=-1e4, bg=1.0,
             lambda_l2=1.0):
    """float64 closed forms: rgb [Q,3], grad_density [N,1], grad_sh [N,3], grad_surface [N,1], samples per ray, and the
    per ray the
    smallest margin by which one of its gate decisions was taken (a test ray should not sit on a gate)."""
    R, a, b = model["reso"], model["axis"], model["sigma_axis"]
    c, w, s0, s1 = model["center"], model["half_width"], model["sigma0"], model["sigma1"]
    col = np.maximum(0.5 + Y00 * model["sh"], 0.0)
    Q = origins.shape[0]
    o = origins.astype(np.float64)
    d = dirs.astype(np.float64)
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    N = R ** 3
    rgb = np.zeros((Q, 3))
    g_den, g_sh, g_surf = np.zeros(N), np.zeros((N, 3)), np.zeros(N)
    n_samples = np.zeros(Q, np.int64)
    margin = np.full(Q, np.inf)
    recs = []
    for q in range(Q):
        samples = []
        if abs(d[q, a]) > 1e-12:
            ts = sorted([(c - w - o[q, a]) / d[q, a], (c + w - o[q, a]) / d[q, a]])
            for t in ts:
                p = o[q] + t * d[q]
                assert t > 0 and np.all(p > 1.0) and np.all(p < R - 2.0), "the test rays cross inside the lattice"
                grad_s_dot_d = -np.sign(p[a] - c) * d[q, a]          # s = w - |x_a - c|
                if only_outward and not grad_s_dot_d > 0.0:          # surf_trav.cu:350-367: kept iff grad s . d > 0
                    continue
                sigma = s0 + s1 * p[b]
                margin[q] = min(margin[q], abs(sigma - sigma_thresh))
                if not sigma > sigma_thresh:                         # :380 (the voxel gate :230-239 agrees: sigma is
                    continue                                         # monotone, a voxel below the threshold holds no sample)
                alpha = 1.0 - np.exp(-sigma) if sigma >= 0 else 0.0
                samples.append((p, alpha, grad_s_dot_d))
        # compositing with the stop rule
        T, used, stopped = 1.0, [], False
        for p, alpha, gsd in samples:
            used.append((p, alpha, gsd, T))
            T *= 1.0 - alpha
            margin[q] = min(margin[q], abs(T - stop_thresh))
            if T < stop_thresh:                                      # :544-547: the ray ends, the background is dropped
                stopped = True
                break
        n_samples[q] = len(used)
        for p, alpha, gsd, Ti in used:
            rgb[q] += Ti * alpha * col
        if not stopped:
            rgb[q] += T * bg
        recs.append((used, stopped, T))
    for q in range(Q):
        used, stopped, T_end = recs[q]
        gk = lambda_l2 * 2.0 * (rgb[q] - rgb_gt[q].astype(np.float64)) / (3.0 * Q)
        for i, (p, alpha, gsd, Ti) in enumerate(used):
            # d rgb_k / d alpha_i: own term, minus what the later samples and the background lose
            later = np.zeros(3)
            Tj = Ti * (1.0 - alpha)
            for (p2, a2, _, _) in used[i + 1:]:
                later += Tj * a2 * col
                Tj *= 1.0 - a2
            if not stopped:
                later += Tj * bg
            drgb_dalpha = Ti * col - later / (1.0 - alpha)
            dL_dalpha = float(gk @ drgb_dalpha)
            dL_dsigma = dL_dalpha * (1.0 - alpha)                     # exp activation, alpha > 0
            dL_dcol = gk * Ti * alpha
            del gsd
            v = np.floor(p).astype(np.int64)
            f = p - v
            for cx in range(2):
                for cy in range(2):
                    for cz in range(2):
                        wgt = (f[0] if cx else 1 - f[0]) * (f[1] if cy else 1 - f[1]) * (f[2] if cz else 1 - f[2])
                        row = ((v[0] + cx) * R + (v[1] + cy)) * R + (v[2] + cz)
                        g_den[row] += wgt * dL_dsigma
                        g_sh[row] += wgt * dL_dcol * Y00 * (col > 0)
    return dict(rgb=rgb, grad_density=g_den.reshape(N, 1), grad_sh=g_sh,
                grad_surface=(g_surf.reshape(N, 1) if s1 == 0.0 else None), n_samples=n_samples, margin=margin)


def sphere_scene(reso=32, radius=9.37, center=(15.3, 16.1, 15.8), sigma=0.9, sh=(0.6, -0.3, 1.1)):
    R = reso
    ax = np.arange(R, dtype=np.float64)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    n = R ** 3
    dist = np.sqrt((X - center[0]) ** 2 + (Y - center[1]) ** 2 + (Z - center[2]) ** 2)
    sc = dict(links=np.arange(n, dtype=np.int32).reshape(R, R, R), density=np.full((n, 1), sigma, np.float32),
              surface=(radius - dist).reshape(n, 1).astype(np.float32), sh=np.tile(np.asarray(sh, np.float32), (n, 1)),
              level_set=np.zeros(1, np.float32), offset=np.zeros(3, np.float32), scaling=np.ones(3, np.float32),
              basis_dim=1, fake_sample_std=1.0, truncated_vol_render_a=1.0, surface_type=102)
    return sc, dict(reso=R, radius=radius, center=np.asarray(center, np.float64), sigma=sigma,
                    sh=np.asarray(sh, np.float64))


def sphere_rays(model, n=300, seed=0):
    """Rays from outside the grid whose distance from the centre is < 0.7 r (two clean crossings, far from grazing),
    and every tenth one > 1.3 r (no crossing)."""
    rs = np.random.RandomState(seed)
    c, r = model["center"], model["radius"]
    o, d, hits = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros(n, bool)
    for q in range(n):
        dirv = rs.randn(3)
        dirv /= np.linalg.norm(dirv)
        perp = np.cross(dirv, rs.randn(3))
        perp /= np.linalg.norm(perp)
        hits[q] = (q % 10) != 0
        b = rs.uniform(0.0, 0.7 * r) if hits[q] else rs.uniform(1.3 * r, 1.4 * r)
        o[q], d[q] = c + b * perp - 45.0 * dirv, dirv
    return o.astype(np.float32), d.astype(np.float32), hits


def sphere_expected(model, hits, rgb_gt, only_outward=False, sigma_thresh=-1e4, stop_thresh=-1e4, bg=1.0):
    """rgb [Q,3], the sums over all rows of grad_density and of grad_sh (per channel), samples per ray."""
    col = np.maximum(0.5 + Y00 * model["sh"], 0.0)
    sigma = model["sigma"]
    Q = hits.shape[0]
    alpha = 1.0 - np.exp(-sigma)
    n_pass = (1 if only_outward else 2) if sigma > sigma_thresh else 0
    rgb = np.zeros((Q, 3))
    n_samples = np.zeros(Q, np.int64)
    info = []
    for q in range(Q):
        T, k, stopped = 1.0, 0, False
        if hits[q]:
            for _ in range(n_pass):
                rgb[q] += T * alpha * col
                T *= 1.0 - alpha
                k += 1
                if T < stop_thresh:
                    stopped = True
                    break
        if not stopped:
            rgb[q] += T * bg
        n_samples[q] = k
        info.append((k, stopped))
    sum_den, sum_sh = 0.0, np.zeros(3)
    for q in range(Q):
        k, stopped = info[q]
        gk = 2.0 * (rgb[q] - rgb_gt[q].astype(np.float64)) / (3.0 * Q)
        # rgb = col (1 - (1-a)^k) + [not stopped] bg (1-a)^k, every sample sharing the one sigma
        drgb_dalpha = (col - (0.0 if stopped else bg)) * k * (1.0 - alpha) ** max(k - 1, 0) if k else np.zeros(3)
        sum_den += float(gk @ drgb_dalpha) * (1.0 - alpha)
        sum_sh += gk * (1.0 - (1.0 - alpha) ** k) * Y00 * (col > 0)
    return dict(rgb=rgb, sum_grad_density=sum_den, sum_grad_sh=sum_sh, n_samples=n_samples)
