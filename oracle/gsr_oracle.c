/*
 * oracle/gsr_oracle.c -- CPU restatement of the differentiable tile rasterizer.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path may import, link or
 * call this file.  Allowed users: tests/, __graft_entry__.smoke(), and the
 * cpu_baseline leg of bench.py (as the thing timed *beside* the GPU path).
 *
 * PARITY UNPINNED.  The arithmetic of this path lives in the third-party,
 * UN-VENDORED submodule `submodules/diff-gaussian-rasterization`
 * (/root/reference/.gitmodules:4-6 -> github.com/ashawkey/diff-gaussian-rasterization,
 * installed by /root/reference/requirements.txt:11; no gitlink SHA => version
 * unpinned, directory empty in the reference checkout).  The reference has no
 * tests or golden vectors for it.  This file therefore restates the module's
 * *published algorithm* (Kerbl et al. 2023, "3D Gaussian Splatting", sec. 4-6
 * and appendix; plus the fork's depth / alpha outputs) in our own words, and is
 * anchored to the reference only where reference code exists:
 *   - calling convention      scene/gaussian_model_ht.py:806-894
 *   - matrix layout           scene/cameras.py:76-98 (transposed => column-major when read linearly)
 *   - SH basis / constants    utils/sh_utils.py:24-112, gaussian_model_ht.py:859-862 (+0.5, clamp_min 0)
 *   - covariance from S,R     utils/general_utils.py:62-108, gaussian_model_ht.py:50-55
 * Those four ARE pinned by fixtures under tests/golden (made by tools/make_golden.py
 * importing the reference's Python in the authoring container).
 *
 * Precision: arithmetic in `real` (double unless -DREAL=float).  The view-space
 * depth used for culling, sorting and the depth feature is always computed in
 * binary32 with one fixed fmaf sequence so that ordering is bit-identical to the
 * HIP path (see depth_key()).  Discrete decisions (alpha cut, transmittance
 * stop, radius ceil, tile rect) are additionally reported with an "ambiguous"
 * margin mask so parity tests can separate rounding-induced branch flips from
 * real errors.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

#ifndef REAL
#define REAL double
#endif
#ifndef R_EXP
#define R_EXP exp
#define R_SQRT sqrt
#endif
typedef REAL real;

#define TILE 16
#define ALPHA_MIN ((real)(1.0 / 255.0))
#define ALPHA_MAX ((real)0.99)
#define T_STOP ((real)1e-4)
#define NEAR_Z 0.2f
#define LOWPASS ((real)0.3)

/* SH constants: utils/sh_utils.py:24-48 */
static const double SH_C0 = 0.28209479177387814;
static const double SH_C1 = 0.4886025119029199;
static const double SH_C2[5] = {1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
                               -1.0925484305920792, 0.5462742152960396};
static const double SH_C3[7] = {-0.5900435899266435, 2.890611442640554, -0.4570457994644658,
                                0.3731763325901154, -0.4570457994644658, 1.445305721320277,
                                -0.5900435899266435};

typedef struct {
    int32_t N;          /* number of Gaussians */
    int32_t M;          /* SH coefficients stored per Gaussian (stride), e.g. 16 */
    int32_t D;          /* active SH degree 0..3 */
    int32_t W, H;
    int32_t prefiltered;
    float scale_modifier, tanfovx, tanfovy;
    const float *means3D;       /* [N,3] */
    const float *scales;        /* [N,3] or NULL */
    const float *rotations;     /* [N,4] (w,x,y,z) or NULL */
    const float *cov3D_precomp; /* [N,6] or NULL */
    const float *opacities;     /* [N] */
    const float *shs;           /* [N,M,3] or NULL */
    const float *colors_precomp;/* [N,3] or NULL */
    const float *viewmatrix;    /* 16, column-major when read linearly */
    const float *projmatrix;    /* 16, same */
    const float *campos;        /* 3 */
    const float *bg;            /* 3 */
} GsrOracleIn;

typedef struct {
    GsrOracleIn in;
    int32_t tiles_x, tiles_y;
    /* per Gaussian */
    int32_t *radius;
    int32_t *rect;        /* [N,4] xmin,ymin,xmax,ymax in tiles */
    int32_t *rect_outer, *rect_inner; /* [N,4] only meaningful where g_ambig */
    float *depth;         /* binary32 view z (sort key and depth feature) */
    real *xy;             /* [N,2] pixel coords */
    real *conic;          /* [N,3] */
    real *rgb;            /* [N,3] */
    real *cov3d;          /* [N,6] */
    real *cov2d;          /* [N,3] a,b,c incl. low-pass */
    uint8_t *clamped;     /* [N,3] SH colour clamped at 0 */
    uint8_t *g_ambig;     /* [N]  */
    /* binning */
    int64_t R;
    int64_t *tile_start;  /* [T+1] */
    uint32_t *list;       /* [R] Gaussian ids, per tile in blend order */
    /* per pixel */
    real *final_T;
    uint32_t *n_contrib;
    real *acc;            /* [P,5] C0,C1,C2,D,A without background */
    uint8_t *px_ambig;    /* bit 0: a blend decision of this pixel sits on a rounding edge; bit 1: a tile-rect edge */
    uint32_t *px_flips;   /* [P] or NULL: which of those decisions this pixel takes the OTHER way (gsr_oracle_resolve_branches) */
    int64_t pairs_evaluated;
} Ctx;

/* ---- binary32 view depth: ONE fixed rounding sequence shared with the HIP path ---- */
static inline float depth_key(const float *vm, const float *p)
{
    return fmaf(vm[10], p[2], fmaf(vm[6], p[1], fmaf(vm[2], p[0], vm[14])));
}

static inline void rect_of(real px, real py, int rad, int tx, int ty, int *r)
{
    real b[4] = {(px - rad) / TILE, (py - rad) / TILE, (px + rad + TILE - 1) / TILE, (py + rad + TILE - 1) / TILE};
    for (int k = 0; k < 4; k++) {
        int v = b[k] < 0 ? 0 : (b[k] > 1e6 ? 1000000 : (int)b[k]);
        int hi = (k & 1) ? ty : tx;
        r[k] = v > hi ? hi : v;
    }
}

static inline real frac_dist(real x) { real r = x - floor(x); return r < 1 - r ? r : 1 - r; }

/* covariance from scale and quaternion: Sigma = R S^2 R^T.
 * Twin: utils/general_utils.py:76-108 (build_rotation without the normalisation, which the
 * in-kernel route does not apply -- gaussian_model_ht.py:839 passes the already-normalised
 * get_rotation) + gaussian_model_ht.py:50-55 (L L^T, strip_symmetric order xx,xy,xz,yy,yz,zz). */
static void cov3d_from_scale_rot(const float *s, float mod, const float *q, real *cov)
{
    real r = q[0], x = q[1], y = q[2], z = q[3];
    real R[3][3] = {
        {1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)},
        {2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)},
        {2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)}};
    real sv[3] = {(real)mod * s[0], (real)mod * s[1], (real)mod * s[2]};
    real L[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) L[i][j] = R[i][j] * sv[j];
    real S[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) {
            real a = 0;
            for (int k = 0; k < 3; k++) a += L[i][k] * L[j][k];
            S[i][j] = a;
        }
    cov[0] = S[0][0]; cov[1] = S[0][1]; cov[2] = S[0][2];
    cov[3] = S[1][1]; cov[4] = S[1][2]; cov[5] = S[2][2];
}

/* SH -> RGB.  Twin: utils/sh_utils.py:57-100; sh layout [M,3] coefficient-major
 * (gaussian_model_ht.py:176-179 concatenates dc and rest along dim 1). */
static void sh_to_rgb(int deg, const float *sh, const real dir[3], real out[3])
{
    real x = dir[0], y = dir[1], z = dir[2];
    for (int c = 0; c < 3; c++) {
#define S_(k) ((real)sh[(k) * 3 + c])
        real res = SH_C0 * S_(0);
        if (deg > 0) {
            res = res - SH_C1 * y * S_(1) + SH_C1 * z * S_(2) - SH_C1 * x * S_(3);
            if (deg > 1) {
                real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                res = res + SH_C2[0] * xy * S_(4) + SH_C2[1] * yz * S_(5) +
                      SH_C2[2] * (2 * zz - xx - yy) * S_(6) + SH_C2[3] * xz * S_(7) +
                      SH_C2[4] * (xx - yy) * S_(8);
                if (deg > 2) {
                    res = res + SH_C3[0] * y * (3 * xx - yy) * S_(9) + SH_C3[1] * xy * z * S_(10) +
                          SH_C3[2] * y * (4 * zz - xx - yy) * S_(11) +
                          SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * S_(12) +
                          SH_C3[4] * x * (4 * zz - xx - yy) * S_(13) +
                          SH_C3[5] * z * (xx - yy) * S_(14) + SH_C3[6] * x * (xx - 3 * yy) * S_(15);
                }
            }
        }
#undef S_
        out[c] = res;
    }
}

/* rounding-edge margins of the blend decisions (absolute); calibrated in tests/test_oracle_cpu.py */
static double g_margin_alpha = 2e-8, g_margin_T = 1e-9, g_margin_power = 1e-7;
void gsr_oracle_set_margins(double a, double t, double p) { g_margin_alpha = a; g_margin_T = t; g_margin_power = p; }

static int cmp_u64(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}

void gsr_oracle_free(Ctx *c)
{
    if (!c) return;
    free(c->radius); free(c->rect); free(c->rect_outer); free(c->rect_inner); free(c->depth); free(c->xy); free(c->conic); free(c->rgb);
    free(c->cov3d); free(c->cov2d); free(c->clamped); free(c->g_ambig); free(c->tile_start);
    free(c->list); free(c->final_T); free(c->n_contrib); free(c->acc); free(c->px_ambig); free(c->px_flips);
    free(c);
}

/* ------------------------------------------------------------------------------------------
 * Stage 1: per-Gaussian projection ("preprocess").  Published algorithm: 3DGS paper sec. 4
 * (EWA projection Sigma' = J W Sigma W^T J^T), sec. 6 (tile assignment).  Constants recalled
 * from the public module: near cull z<=0.2, frustum clamp 1.3*tanfov, low-pass +0.3,
 * radius ceil(3 sqrt(lambda_max)) with lambda = mid +- sqrt(max(0.1, mid^2-det)),
 * p_w = 1/(w+1e-7), pixel centre ((ndc+1)*S-1)/2.
 * ------------------------------------------------------------------------------------------ */
static void preprocess_all(Ctx *c)
{
    const GsrOracleIn *I = &c->in;
    const float *vm = I->viewmatrix, *pm = I->projmatrix;
    const real fx = I->W / (2 * (real)I->tanfovx), fy = I->H / (2 * (real)I->tanfovy);
    const int tx = c->tiles_x, ty = c->tiles_y;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < I->N; i++) {
        c->radius[i] = 0;
        c->g_ambig[i] = 0;
        const float *p = I->means3D + 3 * (size_t)i;
        float zk = depth_key(vm, p);
        c->depth[i] = zk;
        if (!(zk > NEAR_Z)) continue; /* near-plane cull; NaN culled too */
        real X = p[0], Y = p[1], Z = p[2];
        real hx = pm[0] * X + pm[4] * Y + pm[8] * Z + pm[12];
        real hy = pm[1] * X + pm[5] * Y + pm[9] * Z + pm[13];
        real hw = pm[3] * X + pm[7] * Y + pm[11] * Z + pm[15];
        real pw = 1 / (hw + (real)1e-7);
        real ndcx = hx * pw, ndcy = hy * pw;

        real *cov = c->cov3d + 6 * (size_t)i;
        if (I->cov3D_precomp) {
            for (int k = 0; k < 6; k++) cov[k] = I->cov3D_precomp[6 * (size_t)i + k];
        } else {
            cov3d_from_scale_rot(I->scales + 3 * (size_t)i, I->scale_modifier,
                                 I->rotations + 4 * (size_t)i, cov);
        }
        /* view-space position (rows of the true view matrix = strided reads of vm) */
        real t0 = vm[0] * X + vm[4] * Y + vm[8] * Z + vm[12];
        real t1 = vm[1] * X + vm[5] * Y + vm[9] * Z + vm[13];
        real t2 = vm[2] * X + vm[6] * Y + vm[10] * Z + vm[14];
        real limx = (real)1.3 * I->tanfovx, limy = (real)1.3 * I->tanfovy;
        real txtz = t0 / t2, tytz = t1 / t2;
        real cx = txtz < -limx ? -limx : (txtz > limx ? limx : txtz);
        real cy = tytz < -limy ? -limy : (tytz > limy ? limy : tytz);
        t0 = cx * t2; t1 = cy * t2;
        real J00 = fx / t2, J02 = -fx * t0 / (t2 * t2), J11 = fy / t2, J12 = -fy * t1 / (t2 * t2);
        /* M = J * Wr, Wr(r,k) = vm[k*4+r] */
        real m0[3], m1[3];
        for (int k = 0; k < 3; k++) {
            m0[k] = J00 * vm[k * 4 + 0] + J02 * vm[k * 4 + 2];
            m1[k] = J11 * vm[k * 4 + 1] + J12 * vm[k * 4 + 2];
        }
        real S[3][3] = {{cov[0], cov[1], cov[2]}, {cov[1], cov[3], cov[4]}, {cov[2], cov[4], cov[5]}};
        real Sm0[3], Sm1[3];
        for (int r = 0; r < 3; r++) {
            Sm0[r] = S[r][0] * m0[0] + S[r][1] * m0[1] + S[r][2] * m0[2];
            Sm1[r] = S[r][0] * m1[0] + S[r][1] * m1[1] + S[r][2] * m1[2];
        }
        real a = m0[0] * Sm0[0] + m0[1] * Sm0[1] + m0[2] * Sm0[2] + LOWPASS;
        real b = m0[0] * Sm1[0] + m0[1] * Sm1[1] + m0[2] * Sm1[2];
        real cc = m1[0] * Sm1[0] + m1[1] * Sm1[1] + m1[2] * Sm1[2] + LOWPASS;
        real det = a * cc - b * b;
        if (det == 0) continue;
        real dinv = 1 / det;
        real mid = (real)0.5 * (a + cc);
        real disc = mid * mid - det;
        if (disc < (real)0.1) disc = (real)0.1;
        real l1 = mid + R_SQRT(disc), l2 = mid - R_SQRT(disc);
        real rad_f = 3 * R_SQRT(l1 > l2 ? l1 : l2);
        int rad = (int)ceil(rad_f);
        real px = ((ndcx + 1) * I->W - 1) * (real)0.5, py = ((ndcy + 1) * I->H - 1) * (real)0.5;
        /* tile rectangle: (int) truncation then clamp to the grid */
        int rc[4];
        rect_of(px, py, rad, tx, ty, rc);
        int x0 = rc[0], y0 = rc[1], x1 = rc[2], y1 = rc[3];
        /* rounding-edge analysis: outer/inner rect over radius and sub-pixel perturbations */
        {
            int rad_alt = rad;
            real fr = rad_f - floor(rad_f);
            real tol = (real)4e-6 * (rad_f > 1 ? rad_f : 1); /* ~8x the binary32 error of 3 sqrt(lambda) */
            if (fr < tol) rad_alt = rad - 1;          /* just above an integer: a float path may round down */
            else if (1 - fr < tol) rad_alt = rad + 1; /* just below: may round up */
            int o[4] = {x0, y0, x1, y1}, n[4] = {x0, y0, x1, y1};
            const real e = (real)4e-7 * (I->W > I->H ? I->W : I->H); /* pixels: ~8x the binary32 error of (ndc+1)*S/2 */
            for (int ra = 0; ra < 2; ra++)
                for (int sx = -1; sx <= 1; sx += 2)
                    for (int sy = -1; sy <= 1; sy += 2) {
                        int q[4];
                        rect_of(px + sx * e, py + sy * e, ra ? rad_alt : rad, tx, ty, q);
                        if (q[0] < o[0]) o[0] = q[0]; if (q[1] < o[1]) o[1] = q[1];
                        if (q[2] > o[2]) o[2] = q[2]; if (q[3] > o[3]) o[3] = q[3];
                        if (q[0] > n[0]) n[0] = q[0]; if (q[1] > n[1]) n[1] = q[1];
                        if (q[2] < n[2]) n[2] = q[2]; if (q[3] < n[3]) n[3] = q[3];
                    }
            if (rad_alt != rad || o[0] != n[0] || o[1] != n[1] || o[2] != n[2] || o[3] != n[3]) {
                c->g_ambig[i] = 1;
                for (int k = 0; k < 4; k++) { c->rect_outer[4 * (size_t)i + k] = o[k]; c->rect_inner[4 * (size_t)i + k] = n[k]; }
            }
        }
        c->rect[4 * (size_t)i + 0] = x0; c->rect[4 * (size_t)i + 1] = y0;
        c->rect[4 * (size_t)i + 2] = x1; c->rect[4 * (size_t)i + 3] = y1;
        c->xy[2 * (size_t)i] = px; c->xy[2 * (size_t)i + 1] = py;
        c->cov2d[3 * (size_t)i] = a; c->cov2d[3 * (size_t)i + 1] = b; c->cov2d[3 * (size_t)i + 2] = cc;
        c->conic[3 * (size_t)i] = cc * dinv; c->conic[3 * (size_t)i + 1] = -b * dinv;
        c->conic[3 * (size_t)i + 2] = a * dinv;
        if ((x1 - x0) * (y1 - y0) == 0) {
            /* off-screen: invisible, but an ambiguous one may pop in on the other side */
            continue;
        }
        /* colour */
        real *rgb = c->rgb + 3 * (size_t)i;
        if (I->colors_precomp) {
            for (int k = 0; k < 3; k++) { rgb[k] = I->colors_precomp[3 * (size_t)i + k]; c->clamped[3 * (size_t)i + k] = 0; }
        } else {
            real d[3] = {X - I->campos[0], Y - I->campos[1], Z - I->campos[2]};
            real n = R_SQRT(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            d[0] /= n; d[1] /= n; d[2] /= n;
            real col[3];
            sh_to_rgb(I->D, I->shs + (size_t)i * I->M * 3, d, col);
            for (int k = 0; k < 3; k++) {
                real v = col[k] + (real)0.5;
                c->clamped[3 * (size_t)i + k] = v < 0;
                rgb[k] = v < 0 ? 0 : v;
            }
        }
        c->radius[i] = rad;
    }
}

/* Stage 2: duplicate per touched tile, order by (tile, depth bits, emission order).
 * Published algorithm: 3DGS paper sec. 6 ("instantiate each Gaussian per tile, key = tile|depth,
 * one radix sort").  A stable sort on the 64-bit key is equivalent to the per-tile stable
 * sort on depth bits done here. */
static void bin_all(Ctx *c)
{
    const int N = c->in.N, T = c->tiles_x * c->tiles_y, tx = c->tiles_x;
    c->tile_start = (int64_t *)calloc((size_t)T + 1, sizeof(int64_t));
    for (int i = 0; i < N; i++) {
        if (c->radius[i] <= 0) continue;
        const int32_t *r = c->rect + 4 * (size_t)i;
        for (int y = r[1]; y < r[3]; y++)
            for (int x = r[0]; x < r[2]; x++) c->tile_start[y * tx + x + 1]++;
    }
    for (int t = 0; t < T; t++) c->tile_start[t + 1] += c->tile_start[t];
    c->R = c->tile_start[T];
    c->list = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)(c->R ? c->R : 1));
    int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * (size_t)T);
    memcpy(cur, c->tile_start, sizeof(int64_t) * (size_t)T);
    for (int i = 0; i < N; i++) {
        if (c->radius[i] <= 0) continue;
        const int32_t *r = c->rect + 4 * (size_t)i;
        for (int y = r[1]; y < r[3]; y++)
            for (int x = r[0]; x < r[2]; x++) c->list[cur[y * tx + x]++] = (uint32_t)i;
    }
    free(cur);
#pragma omp parallel
    {
        uint64_t *keys = NULL;
        size_t cap = 0;
#pragma omp for schedule(dynamic, 8)
        for (int t = 0; t < T; t++) {
            int64_t s = c->tile_start[t], e = c->tile_start[t + 1];
            size_t n = (size_t)(e - s);
            if (n < 2) continue;
            if (n > cap) { free(keys); cap = n * 2; keys = (uint64_t *)malloc(cap * sizeof(uint64_t)); }
            /* within a tile, emission order == increasing Gaussian id, so (depth bits, id) is the
             * stable order */
            for (size_t k = 0; k < n; k++) {
                uint32_t g = c->list[s + k], db;
                memcpy(&db, &c->depth[g], 4);
                keys[k] = ((uint64_t)db << 32) | g;
            }
            qsort(keys, n, sizeof(uint64_t), cmp_u64);
            for (size_t k = 0; k < n; k++) c->list[s + k] = (uint32_t)(keys[k] & 0xffffffffu);
        }
        free(keys);
    }
}

/* Stage 3: per-pixel front-to-back compositing (3DGS paper eq. 3; fork adds depth = sum z a T and
 * alpha = sum a T).  Constants recalled from the public module: alpha = min(0.99, o*exp(power)),
 * skip alpha<1/255, skip power>0, stop before T would drop below 1e-4.
 *
 * walk_pixel is the ONE statement of these decisions: the forward, the branch resolution and the backward all
 * go through it.  Three of the decisions are discrete (power > 0, alpha < 1/255, T(1-alpha) < 1e-4); an
 * implementation in binary32 can take one the other way when the quantity sits within rounding of its
 * threshold.  Each such rounding-edge decision met on the walk is an EVENT, numbered in order; bit j of `flips`
 * takes event j the other way.  flips = 0 is the exact-arithmetic walk. */
typedef struct {
    real T, C[3], D, A;
    uint32_t last;      /* 1-based list position of the last blended instance (n_contrib) */
    int events;         /* rounding-edge decisions met */
    int64_t pairs;
} PixOut;

typedef struct { uint32_t g; real alpha, G, T; } Kept;   /* one blended instance: id, clamped alpha, exp(power), T before it */

static void walk_pixel(const Ctx *c, int64_t s, int64_t e, int px, int py, uint32_t flips, PixOut *out, Kept *kept, int *n_kept)
{
    const GsrOracleIn *I = &c->in;
    const double coord_ulp = 6e-8 * (I->W > I->H ? I->W : I->H);
    real T = 1, C[3] = {0, 0, 0}, Dp = 0, A = 0;
    uint32_t contributor = 0, last = 0;
    int ev = 0, nk = 0;
    int64_t pairs = 0;
    double relT = 0;
    for (int64_t k = s; k < e; k++) {
        contributor++;
        pairs++;
        uint32_t g = c->list[k];
        real dx = c->xy[2 * (size_t)g] - px, dy = c->xy[2 * (size_t)g + 1] - py;
        const real *co = c->conic + 3 * (size_t)g;
        real o = I->opacities[g];
        real power = (real)-0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
        /* binary32 error model of the HIP path: the pixel-space mean carries ~ulp(max(W,H)),
         * so power carries |grad power| * ulp plus its own rounding; alpha = o*exp(power)
         * inherits alpha * dpower.  The running T collects the relative errors of (1-alpha). */
        double dpow = 4.0 * ((fabs((double)(co[0] * dx + co[1] * dy)) + fabs((double)(co[2] * dy + co[1] * dx))) * coord_ulp +
                             6e-8 * (fabs((double)(co[0] * dx * dx)) + fabs((double)(co[2] * dy * dy)) + fabs((double)(co[1] * dx * dy)) + 1.0));
        int skip = power > 0;
        if (power > (real)-(g_margin_power + dpow) && o >= ALPHA_MIN) {
            if (ev < 32 && ((flips >> ev) & 1u)) skip = !skip;
            ev++;
        }
        if (skip) continue;
        real G = R_EXP(power);
        real alpha = o * G;
        if (alpha > ALPHA_MAX) alpha = ALPHA_MAX;
        double dalpha = (double)alpha * dpow + g_margin_alpha;
        skip = alpha < ALPHA_MIN;
        if (fabs((double)(alpha - ALPHA_MIN)) < dalpha) {
            if (ev < 32 && ((flips >> ev) & 1u)) skip = !skip;
            ev++;
        }
        if (skip) continue;
        real test_T = T * (1 - alpha);
        relT += dalpha / (1.0 - (double)alpha) + 1.2e-7;
        int stop = test_T < T_STOP;
        if (fabs((double)(test_T - T_STOP)) < (double)test_T * relT + g_margin_T) {
            if (ev < 32 && ((flips >> ev) & 1u)) stop = !stop;
            ev++;
        }
        if (stop) break;
        real w = alpha * T;
        const real *rgb = c->rgb + 3 * (size_t)g;
        C[0] += rgb[0] * w; C[1] += rgb[1] * w; C[2] += rgb[2] * w;
        Dp += (real)c->depth[g] * w;
        A += w;
        if (kept) { kept[nk].g = g; kept[nk].alpha = alpha; kept[nk].G = G; kept[nk].T = T; }
        nk++;
        T = test_T;
        last = contributor;
    }
    out->T = T; out->C[0] = C[0]; out->C[1] = C[1]; out->C[2] = C[2]; out->D = Dp; out->A = A;
    out->last = last; out->events = ev; out->pairs = pairs;
    if (n_kept) *n_kept = nk;
}

static void store_pixel(Ctx *c, size_t pid, const PixOut *o)
{
    c->final_T[pid] = o->T;
    c->n_contrib[pid] = o->last;
    real *acc = c->acc + 5 * pid;
    acc[0] = o->C[0]; acc[1] = o->C[1]; acc[2] = o->C[2]; acc[3] = o->D; acc[4] = o->A;
}

static void blend_all(Ctx *c)
{
    const GsrOracleIn *I = &c->in;
    const int W = I->W, H = I->H, tx = c->tiles_x;
    int64_t pairs = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : pairs)
    for (int t = 0; t < c->tiles_x * c->tiles_y; t++) {
        int ty0 = (t / tx) * TILE, tx0 = (t % tx) * TILE;
        int64_t s = c->tile_start[t], e = c->tile_start[t + 1];
        for (int py = ty0; py < ty0 + TILE && py < H; py++)
            for (int px = tx0; px < tx0 + TILE && px < W; px++) {
                PixOut o;
                walk_pixel(c, s, e, px, py, 0u, &o, NULL, NULL);
                pairs += o.pairs;
                size_t pid = (size_t)py * W + px;
                store_pixel(c, pid, &o);
                c->px_ambig[pid] = o.events ? 1 : 0;
            }
    }
    c->pairs_evaluated = pairs;
    /* Gaussians whose radius / rect sits on a rounding edge taint the tiles a float path might add or
     * drop: outer rect minus inner rect */
    for (int i = 0; i < I->N; i++) {
        if (!c->g_ambig[i]) continue;
        const int32_t *o = c->rect_outer + 4 * (size_t)i, *n = c->rect_inner + 4 * (size_t)i;
        for (int tyy = o[1]; tyy < o[3]; tyy++)
            for (int txx = o[0]; txx < o[2]; txx++) {
                if (txx >= n[0] && txx < n[2] && tyy >= n[1] && tyy < n[3]) continue;
                /* a contested tile only matters where this Gaussian would actually blend (alpha >= 1/255) */
                const real *co = c->conic + 3 * (size_t)i;
                const real o_ = I->opacities[i];
                for (int y = tyy * TILE; y < (tyy + 1) * TILE && y < H; y++)
                    for (int x = txx * TILE; x < (txx + 1) * TILE && x < W; x++) {
                        real dx = c->xy[2 * (size_t)i] - x, dy = c->xy[2 * (size_t)i + 1] - y;
                        real power = (real)-0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                        if (power > (real)1e-3) continue;
                        real alpha = o_ * R_EXP(power > 0 ? 0 : power);
                        if (alpha >= ALPHA_MIN * (real)0.999) c->px_ambig[(size_t)y * W + x] |= 2;
                    }
            }
    }
}

Ctx *gsr_oracle_prepare(const GsrOracleIn *in, int do_bin)
{
    Ctx *c = (Ctx *)calloc(1, sizeof(Ctx));
    c->in = *in;
    const size_t N = (size_t)(in->N > 0 ? in->N : 1);
    c->tiles_x = (in->W + TILE - 1) / TILE;
    c->tiles_y = (in->H + TILE - 1) / TILE;
    c->radius = (int32_t *)calloc(N, sizeof(int32_t));
    c->rect = (int32_t *)calloc(N * 4, sizeof(int32_t));
    c->rect_outer = (int32_t *)calloc(N * 4, sizeof(int32_t));
    c->rect_inner = (int32_t *)calloc(N * 4, sizeof(int32_t));
    c->depth = (float *)calloc(N, sizeof(float));
    c->xy = (real *)calloc(N * 2, sizeof(real));
    c->conic = (real *)calloc(N * 3, sizeof(real));
    c->rgb = (real *)calloc(N * 3, sizeof(real));
    c->cov3d = (real *)calloc(N * 6, sizeof(real));
    c->cov2d = (real *)calloc(N * 3, sizeof(real));
    c->clamped = (uint8_t *)calloc(N * 3, 1);
    c->g_ambig = (uint8_t *)calloc(N, 1);
    preprocess_all(c);
    if (do_bin) bin_all(c);
    return c;
}

/* Full forward.  Outputs may be NULL.  Returns an opaque context for gsr_oracle_backward. */
Ctx *gsr_oracle_forward(const GsrOracleIn *in, float *out_color, float *out_depth, float *out_alpha,
                        int32_t *out_radii, uint8_t *out_px_ambig, uint8_t *out_g_ambig)
{
    Ctx *c = gsr_oracle_prepare(in, 1);
    const size_t P = (size_t)in->W * in->H;
    c->final_T = (real *)calloc(P ? P : 1, sizeof(real));
    c->n_contrib = (uint32_t *)calloc(P ? P : 1, sizeof(uint32_t));
    c->acc = (real *)calloc((P ? P : 1) * 5, sizeof(real));
    c->px_ambig = (uint8_t *)calloc(P ? P : 1, 1);
    blend_all(c);
    for (size_t p = 0; p < P; p++) {
        const real *acc = c->acc + 5 * p;
        if (out_color)
            for (int k = 0; k < 3; k++) out_color[k * P + p] = (float)(acc[k] + c->final_T[p] * in->bg[k]);
        if (out_depth) out_depth[p] = (float)acc[3];
        if (out_alpha) out_alpha[p] = (float)acc[4];
        if (out_px_ambig) out_px_ambig[p] = c->px_ambig[p];
    }
    for (int i = 0; i < in->N; i++) {
        if (out_radii) out_radii[i] = c->radius[i];
        if (out_g_ambig) out_g_ambig[i] = c->g_ambig[i];
    }
    return c;
}

/* Branch resolution against an implementation under test.
 * A pixel whose walk met m >= 1 rounding-edge decisions has up to 2^m legitimate outcomes (each such decision may go
 * either way under binary32 rounding).  For every such pixel try the combinations of its first `max_events` events
 * and ADOPT the branch whose (colour, alpha) is closest to `got`: the pixel's stored state and px_flips are replaced,
 * so the outputs returned here AND a following gsr_oracle_backward belong to that same branch.  The caller then
 * holds the implementation to the ordinary tolerance on these pixels too; a pixel that matches none of its branches
 * shows up with a large out_err.  out_events[p] = m of the exact walk (saturating at 255).  Returns the number of
 * pixels whose adopted branch is not the exact-arithmetic one. */
int64_t gsr_oracle_resolve_branches(Ctx *c, const float *got_color, const float *got_alpha, int max_events,
                                    uint8_t *out_events, float *out_err, float *out_color, float *out_depth, float *out_alpha)
{
    const GsrOracleIn *I = &c->in;
    const int W = I->W, H = I->H, tx = c->tiles_x;
    const size_t P = (size_t)W * H;
    if (max_events > 12) max_events = 12;
    if (!c->px_flips) c->px_flips = (uint32_t *)calloc(P ? P : 1, sizeof(uint32_t));
    int64_t changed = 0;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : changed)
    for (int t = 0; t < c->tiles_x * c->tiles_y; t++) {
        int ty0 = (t / tx) * TILE, tx0 = (t % tx) * TILE;
        int64_t s = c->tile_start[t], e = c->tile_start[t + 1];
        for (int py = ty0; py < ty0 + TILE && py < H; py++)
            for (int px = tx0; px < tx0 + TILE && px < W; px++) {
                size_t pid = (size_t)py * W + px;
                PixOut best, o;
                walk_pixel(c, s, e, px, py, 0u, &best, NULL, NULL);
                const int m0 = best.events;
                uint32_t best_mask = 0;
                double best_err = 0;
                {
                    double err = 0;
                    for (int k = 0; k < 3; k++) {
                        double d = fabs((double)(best.C[k] + best.T * I->bg[k]) - (double)got_color[k * P + pid]);
                        if (d > err) err = d;
                    }
                    double d = fabs((double)best.A - (double)got_alpha[pid]);
                    best_err = d > err ? d : err;
                }
                if (m0 > 0) {
                    int E = m0 < max_events ? m0 : max_events;
                    for (uint32_t mask = 1; mask < (1u << E); mask++) {
                        walk_pixel(c, s, e, px, py, mask, &o, NULL, NULL);
                        if (o.events > E && o.events <= max_events) E = o.events;   /* a flipped stop reveals later events */
                        double err = 0;
                        for (int k = 0; k < 3; k++) {
                            double d = fabs((double)(o.C[k] + o.T * I->bg[k]) - (double)got_color[k * P + pid]);
                            if (d > err) err = d;
                        }
                        double d = fabs((double)o.A - (double)got_alpha[pid]);
                        if (d > err) err = d;
                        if (err < best_err) { best_err = err; best = o; best_mask = mask; }
                    }
                }
                if (best_mask) { changed++; store_pixel(c, pid, &best); }
                c->px_flips[pid] = best_mask;
                if (out_events) out_events[pid] = (uint8_t)(m0 > 255 ? 255 : m0);
                if (out_err) out_err[pid] = (float)best_err;
                if (out_color) for (int k = 0; k < 3; k++) out_color[k * P + pid] = (float)(best.C[k] + best.T * I->bg[k]);
                if (out_depth) out_depth[pid] = (float)best.D;
                if (out_alpha) out_alpha[pid] = (float)best.A;
            }
    }
    return changed;
}

int64_t gsr_oracle_num_rendered(const Ctx *c) { return c->R; }
int64_t gsr_oracle_pairs(const Ctx *c) { return c->pairs_evaluated; }

/* copy out the binning result (per-tile ranges + blend-ordered Gaussian ids) */
void gsr_oracle_get_binning(const Ctx *c, int64_t *tile_start, uint32_t *list)
{
    int T = c->tiles_x * c->tiles_y;
    if (tile_start) memcpy(tile_start, c->tile_start, sizeof(int64_t) * ((size_t)T + 1));
    if (list && c->R) memcpy(list, c->list, sizeof(uint32_t) * (size_t)c->R);
}

/* copy out per-Gaussian projected state (as float) for stage-level parity tests */
void gsr_oracle_get_geom(const Ctx *c, float *xy, float *depth, float *conic, float *rgb, int32_t *rect)
{
    size_t N = (size_t)c->in.N;
    for (size_t i = 0; i < N; i++) {
        if (xy) { xy[2 * i] = (float)c->xy[2 * i]; xy[2 * i + 1] = (float)c->xy[2 * i + 1]; }
        if (depth) depth[i] = c->depth[i];
        if (conic) for (int k = 0; k < 3; k++) conic[3 * i + k] = (float)c->conic[3 * i + k];
        if (rgb) for (int k = 0; k < 3; k++) rgb[3 * i + k] = (float)c->rgb[3 * i + k];
        if (rect) for (int k = 0; k < 4; k++) rect[4 * i + k] = c->rect[4 * i + k];
    }
}

/* ------------------------------------------------------------------------------------------
 * Backward.  Derivatives of the forward above, derived directly (not transcribed):
 *   out = sum_i c_i a_i T_i + T_f bg ; T_i = prod_{j<i}(1-a_j)
 *   d out / d a_i = c_i T_i - (sum_{k>i} c_k a_k T_k + T_f bg) / (1 - a_i)
 * The clamp at 0.99 passes the gradient straight through and the frustum clamp treats the
 * clamped coordinate as constant in d/dz -- both as in the public module (recalled).
 * Gradients are accumulated in double.  d_means2D is returned in the module's convention:
 * d/d(ndc) = d/d(pixel) * (W/2, H/2) (consumed by gaussian_model_ht.py:718-721).
 * ------------------------------------------------------------------------------------------ */
void gsr_oracle_backward(const Ctx *c, const float *g_color, const float *g_depth, const float *g_alpha,
                         double *d_means3D, double *d_means2D, double *d_opacity, double *d_colors,
                         double *d_shs, double *d_scales, double *d_rotations, double *d_cov3D,
                         double *d_camera /* 35 = viewmatrix[16], projmatrix[16], campos[3] (linear order) or NULL */)
{
    const GsrOracleIn *I = &c->in;
    const int N = I->N, W = I->W, H = I->H, tx = c->tiles_x;
    const size_t P = (size_t)W * H;
    double *g_xy = (double *)calloc((size_t)N * 2 + 1, sizeof(double));    /* d/d pixel coords */
    double *g_conic = (double *)calloc((size_t)N * 3 + 1, sizeof(double)); /* true partials A,B,C */
    double *g_rgb = (double *)calloc((size_t)N * 3 + 1, sizeof(double));
    double *g_z = (double *)calloc((size_t)N + 1, sizeof(double));
    double *g_op = (double *)calloc((size_t)N + 1, sizeof(double));

    int64_t longest = 1;
    for (int t = 0; t < c->tiles_x * c->tiles_y; t++)
        if (c->tile_start[t + 1] - c->tile_start[t] > longest) longest = c->tile_start[t + 1] - c->tile_start[t];
#pragma omp parallel
    {
    Kept *kept = (Kept *)malloc(sizeof(Kept) * (size_t)longest);
#pragma omp for schedule(dynamic, 4)
    for (int t = 0; t < c->tiles_x * c->tiles_y; t++) {
        int ty0 = (t / tx) * TILE, tx0 = (t % tx) * TILE;
        int64_t s = c->tile_start[t], e = c->tile_start[t + 1];
        for (int py = ty0; py < ty0 + TILE && py < H; py++)
            for (int px = tx0; px < tx0 + TILE && px < W; px++) {
                size_t pid = (size_t)py * W + px;
                double gC[3] = {g_color ? g_color[pid] : 0, g_color ? g_color[P + pid] : 0,
                                g_color ? g_color[2 * P + pid] : 0};
                double gD = g_depth ? g_depth[pid] : 0, gA = g_alpha ? g_alpha[pid] : 0;
                if (gC[0] == 0 && gC[1] == 0 && gC[2] == 0 && gD == 0 && gA == 0) continue;
                /* the same walk as the forward took for this pixel (same branch of every rounding-edge decision) */
                PixOut po;
                int nk = 0;
                walk_pixel(c, s, e, px, py, c->px_flips ? c->px_flips[pid] : 0u, &po, kept, &nk);
                double Tf = po.T;
                double bgdot = Tf * (I->bg[0] * gC[0] + I->bg[1] * gC[1] + I->bg[2] * gC[2]);
                /* suffix sums start as the totals and shrink as we walk front to back */
                double sufC[3] = {po.C[0], po.C[1], po.C[2]}, sufD = po.D, sufA = po.A;
                for (int q = 0; q < nk; q++) {
                    uint32_t g = kept[q].g;
                    double dx = (double)c->xy[2 * (size_t)g] - px, dy = (double)c->xy[2 * (size_t)g + 1] - py;
                    const real *co = c->conic + 3 * (size_t)g;
                    double o = I->opacities[g];
                    double G = kept[q].G, alpha = kept[q].alpha, T = kept[q].T;
                    double w = alpha * T;
                    const real *rgb = c->rgb + 3 * (size_t)g;
                    double z = c->depth[g];
                    sufC[0] -= rgb[0] * w; sufC[1] -= rgb[1] * w; sufC[2] -= rgb[2] * w;
                    sufD -= z * w; sufA -= w;
                    double inv1a = 1.0 / (1.0 - alpha);
                    double dLda = 0;
                    for (int ch = 0; ch < 3; ch++) dLda += gC[ch] * (rgb[ch] * T - sufC[ch] * inv1a);
                    dLda += gD * (z * T - sufD * inv1a);
                    dLda += gA * (T - sufA * inv1a);
                    dLda -= bgdot * inv1a;
                    double dLdG = o * dLda;
                    double dLdpow = G * dLdG;
                    double gx = dLdpow * (-co[0] * dx - co[1] * dy);
                    double gy = dLdpow * (-co[2] * dy - co[1] * dx);
#pragma omp atomic
                    g_xy[2 * (size_t)g] += gx;
#pragma omp atomic
                    g_xy[2 * (size_t)g + 1] += gy;
#pragma omp atomic
                    g_conic[3 * (size_t)g] += -0.5 * dx * dx * dLdpow;
#pragma omp atomic
                    g_conic[3 * (size_t)g + 1] += -dx * dy * dLdpow;
#pragma omp atomic
                    g_conic[3 * (size_t)g + 2] += -0.5 * dy * dy * dLdpow;
#pragma omp atomic
                    g_op[g] += G * dLda;
                    for (int ch = 0; ch < 3; ch++) {
#pragma omp atomic
                        g_rgb[3 * (size_t)g + ch] += w * gC[ch];
                    }
#pragma omp atomic
                    g_z[g] += w * gD;
                }
            }
    }
    free(kept);
    }

    const float *vm = I->viewmatrix, *pm = I->projmatrix;
    const double fx = W / (2 * (double)I->tanfovx), fy = H / (2 * (double)I->tanfovy);
    if (d_camera) for (int k = 0; k < 35; k++) d_camera[k] = 0;
#pragma omp parallel
    {
    double camloc[35];
    for (int k = 0; k < 35; k++) camloc[k] = 0;
#pragma omp for schedule(static)
    for (int i = 0; i < N; i++) {
        double dm[3] = {0, 0, 0};
        if (d_means2D) { d_means2D[3 * (size_t)i] = d_means2D[3 * (size_t)i + 1] = d_means2D[3 * (size_t)i + 2] = 0; }
        if (d_opacity) d_opacity[i] = 0;
        if (d_colors) for (int k = 0; k < 3; k++) d_colors[3 * (size_t)i + k] = 0;
        if (d_shs) for (int k = 0; k < I->M * 3; k++) d_shs[(size_t)i * I->M * 3 + k] = 0;
        if (d_scales) for (int k = 0; k < 3; k++) d_scales[3 * (size_t)i + k] = 0;
        if (d_rotations) for (int k = 0; k < 4; k++) d_rotations[4 * (size_t)i + k] = 0;
        if (d_cov3D) for (int k = 0; k < 6; k++) d_cov3D[6 * (size_t)i + k] = 0;
        if (d_means3D) for (int k = 0; k < 3; k++) d_means3D[3 * (size_t)i + k] = 0;
        if (c->radius[i] <= 0) continue;
        const float *p = I->means3D + 3 * (size_t)i;
        double X = p[0], Y = p[1], Z = p[2];
        if (d_opacity) d_opacity[i] = g_op[i];

        /* ---- colour ---- */
        double gr[3] = {g_rgb[3 * (size_t)i], g_rgb[3 * (size_t)i + 1], g_rgb[3 * (size_t)i + 2]};
        if (I->colors_precomp) {
            if (d_colors) for (int k = 0; k < 3; k++) d_colors[3 * (size_t)i + k] = gr[k];
        } else {
            for (int k = 0; k < 3; k++) if (c->clamped[3 * (size_t)i + k]) gr[k] = 0;
            double d0[3] = {X - I->campos[0], Y - I->campos[1], Z - I->campos[2]};
            double n2 = d0[0] * d0[0] + d0[1] * d0[1] + d0[2] * d0[2], n = sqrt(n2);
            double x = d0[0] / n, y = d0[1] / n, z = d0[2] / n;
            const float *sh = I->shs + (size_t)i * I->M * 3;
            double *dsh = d_shs ? d_shs + (size_t)i * I->M * 3 : NULL;
            double dRdx[3] = {0, 0, 0}, dRdy[3] = {0, 0, 0}, dRdz[3] = {0, 0, 0};
            double basis[16], bx[16], by[16], bz[16];
            memset(basis, 0, sizeof basis); memset(bx, 0, sizeof bx); memset(by, 0, sizeof by); memset(bz, 0, sizeof bz);
            basis[0] = SH_C0;
            if (I->D > 0) {
                basis[1] = -SH_C1 * y; by[1] = -SH_C1;
                basis[2] = SH_C1 * z;  bz[2] = SH_C1;
                basis[3] = -SH_C1 * x; bx[3] = -SH_C1;
                if (I->D > 1) {
                    double xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                    basis[4] = SH_C2[0] * xy; bx[4] = SH_C2[0] * y; by[4] = SH_C2[0] * x;
                    basis[5] = SH_C2[1] * yz; by[5] = SH_C2[1] * z; bz[5] = SH_C2[1] * y;
                    basis[6] = SH_C2[2] * (2 * zz - xx - yy);
                    bx[6] = SH_C2[2] * -2 * x; by[6] = SH_C2[2] * -2 * y; bz[6] = SH_C2[2] * 4 * z;
                    basis[7] = SH_C2[3] * xz; bx[7] = SH_C2[3] * z; bz[7] = SH_C2[3] * x;
                    basis[8] = SH_C2[4] * (xx - yy); bx[8] = SH_C2[4] * 2 * x; by[8] = SH_C2[4] * -2 * y;
                    if (I->D > 2) {
                        basis[9] = SH_C3[0] * y * (3 * xx - yy);
                        bx[9] = SH_C3[0] * 6 * xy; by[9] = SH_C3[0] * (3 * xx - 3 * yy);
                        basis[10] = SH_C3[1] * xy * z;
                        bx[10] = SH_C3[1] * yz; by[10] = SH_C3[1] * xz; bz[10] = SH_C3[1] * xy;
                        basis[11] = SH_C3[2] * y * (4 * zz - xx - yy);
                        bx[11] = SH_C3[2] * -2 * xy; by[11] = SH_C3[2] * (4 * zz - xx - 3 * yy); bz[11] = SH_C3[2] * 8 * yz;
                        basis[12] = SH_C3[3] * z * (2 * zz - 3 * xx - 3 * yy);
                        bx[12] = SH_C3[3] * -6 * xz; by[12] = SH_C3[3] * -6 * yz; bz[12] = SH_C3[3] * (6 * zz - 3 * xx - 3 * yy);
                        basis[13] = SH_C3[4] * x * (4 * zz - xx - yy);
                        bx[13] = SH_C3[4] * (4 * zz - 3 * xx - yy); by[13] = SH_C3[4] * -2 * xy; bz[13] = SH_C3[4] * 8 * xz;
                        basis[14] = SH_C3[5] * z * (xx - yy);
                        bx[14] = SH_C3[5] * 2 * xz; by[14] = SH_C3[5] * -2 * yz; bz[14] = SH_C3[5] * (xx - yy);
                        basis[15] = SH_C3[6] * x * (xx - 3 * yy);
                        bx[15] = SH_C3[6] * (3 * xx - 3 * yy); by[15] = SH_C3[6] * -6 * xy;
                    }
                }
            }
            int nc = (I->D + 1) * (I->D + 1);
            for (int k = 0; k < nc; k++)
                for (int ch = 0; ch < 3; ch++) {
                    if (dsh) dsh[k * 3 + ch] = basis[k] * gr[ch];
                    dRdx[ch] += bx[k] * sh[k * 3 + ch];
                    dRdy[ch] += by[k] * sh[k * 3 + ch];
                    dRdz[ch] += bz[k] * sh[k * 3 + ch];
                }
            double gdir[3] = {0, 0, 0};
            for (int ch = 0; ch < 3; ch++) {
                gdir[0] += dRdx[ch] * gr[ch]; gdir[1] += dRdy[ch] * gr[ch]; gdir[2] += dRdz[ch] * gr[ch];
            }
            /* through v/|v| : (I - d d^T)/|v| */
            double dot = gdir[0] * x + gdir[1] * y + gdir[2] * z;
            dm[0] += (gdir[0] - x * dot) / n; dm[1] += (gdir[1] - y * dot) / n; dm[2] += (gdir[2] - z * dot) / n;
            /* direction = (p - campos)/|.|  =>  d/dcampos = -d/dp of this term */
            camloc[32] -= dm[0]; camloc[33] -= dm[1]; camloc[34] -= dm[2];
        }

        /* ---- conic -> cov2D ---- */
        double a = c->cov2d[3 * (size_t)i], b = c->cov2d[3 * (size_t)i + 1], cc = c->cov2d[3 * (size_t)i + 2];
        double det = a * cc - b * b;
        double d2i = 1.0 / (det * det + 1e-7);
        double gA = g_conic[3 * (size_t)i], gB = g_conic[3 * (size_t)i + 1], gCc = g_conic[3 * (size_t)i + 2];
        double ga = d2i * (-cc * cc * gA + b * cc * gB - b * b * gCc);
        double gb = d2i * (2 * b * cc * gA - (det + 2 * b * b) * gB + 2 * a * b * gCc);
        double gc = d2i * (-b * b * gA + a * b * gB - a * a * gCc);

        /* ---- cov2D -> cov3D, view-space mean ---- */
        double t0 = vm[0] * X + vm[4] * Y + vm[8] * Z + vm[12];
        double t1 = vm[1] * X + vm[5] * Y + vm[9] * Z + vm[13];
        double t2 = vm[2] * X + vm[6] * Y + vm[10] * Z + vm[14];
        double limx = 1.3 * I->tanfovx, limy = 1.3 * I->tanfovy;
        double txtz = t0 / t2, tytz = t1 / t2;
        double xmul = (txtz < -limx || txtz > limx) ? 0 : 1, ymul = (tytz < -limy || tytz > limy) ? 0 : 1;
        double cx = txtz < -limx ? -limx : (txtz > limx ? limx : txtz);
        double cy = tytz < -limy ? -limy : (tytz > limy ? limy : tytz);
        t0 = cx * t2; t1 = cy * t2;
        double J00 = fx / t2, J02 = -fx * t0 / (t2 * t2), J11 = fy / t2, J12 = -fy * t1 / (t2 * t2);
        double m0[3], m1[3];
        for (int k = 0; k < 3; k++) {
            m0[k] = J00 * vm[k * 4 + 0] + J02 * vm[k * 4 + 2];
            m1[k] = J11 * vm[k * 4 + 1] + J12 * vm[k * 4 + 2];
        }
        const real *cv = c->cov3d + 6 * (size_t)i;
        double S[3][3] = {{cv[0], cv[1], cv[2]}, {cv[1], cv[3], cv[4]}, {cv[2], cv[4], cv[5]}};
        double gS[6]; /* wrt the 6 unique entries */
        gS[0] = ga * m0[0] * m0[0] + gb * m0[0] * m1[0] + gc * m1[0] * m1[0];
        gS[3] = ga * m0[1] * m0[1] + gb * m0[1] * m1[1] + gc * m1[1] * m1[1];
        gS[5] = ga * m0[2] * m0[2] + gb * m0[2] * m1[2] + gc * m1[2] * m1[2];
        gS[1] = 2 * ga * m0[0] * m0[1] + gb * (m0[0] * m1[1] + m0[1] * m1[0]) + 2 * gc * m1[0] * m1[1];
        gS[2] = 2 * ga * m0[0] * m0[2] + gb * (m0[0] * m1[2] + m0[2] * m1[0]) + 2 * gc * m1[0] * m1[2];
        gS[4] = 2 * ga * m0[1] * m0[2] + gb * (m0[1] * m1[2] + m0[2] * m1[1]) + 2 * gc * m1[1] * m1[2];
        double Sm0[3], Sm1[3], gm0[3], gm1[3];
        for (int r = 0; r < 3; r++) {
            Sm0[r] = S[r][0] * m0[0] + S[r][1] * m0[1] + S[r][2] * m0[2];
            Sm1[r] = S[r][0] * m1[0] + S[r][1] * m1[1] + S[r][2] * m1[2];
        }
        for (int r = 0; r < 3; r++) { gm0[r] = 2 * ga * Sm0[r] + gb * Sm1[r]; gm1[r] = 2 * gc * Sm1[r] + gb * Sm0[r]; }
        double gJ00 = 0, gJ02 = 0, gJ11 = 0, gJ12 = 0;
        for (int k = 0; k < 3; k++) {
            gJ00 += gm0[k] * vm[k * 4 + 0]; gJ02 += gm0[k] * vm[k * 4 + 2];
            gJ11 += gm1[k] * vm[k * 4 + 1]; gJ12 += gm1[k] * vm[k * 4 + 2];
        }
        double tz2 = 1 / (t2 * t2), tz3 = tz2 / t2;
        double gt0 = xmul * -fx * tz2 * gJ02;
        double gt1 = ymul * -fy * tz2 * gJ12;
        double gt2 = -fx * tz2 * gJ00 - fy * tz2 * gJ11 + 2 * fx * t0 * tz3 * gJ02 + 2 * fy * t1 * tz3 * gJ12;
        for (int k = 0; k < 3; k++)
            dm[k] += vm[k * 4 + 0] * gt0 + vm[k * 4 + 1] * gt1 + vm[k * 4 + 2] * gt2;
        {   /* camera: t = V p_h (rows 0..2, depth feature folded into row 2), rotation block through M = J Wr */
            double ph[4] = {X, Y, Z, 1.0};
            double g2 = gt2 + g_z[i];
            for (int k = 0; k < 4; k++) {
                camloc[k * 4 + 0] += gt0 * ph[k]; camloc[k * 4 + 1] += gt1 * ph[k]; camloc[k * 4 + 2] += g2 * ph[k];
            }
            for (int k = 0; k < 3; k++) {
                camloc[k * 4 + 0] += gm0[k] * J00; camloc[k * 4 + 1] += gm1[k] * J11;
                camloc[k * 4 + 2] += gm0[k] * J02 + gm1[k] * J12;
            }
        }

        /* ---- screen position -> mean ---- */
        double hx = pm[0] * X + pm[4] * Y + pm[8] * Z + pm[12];
        double hy = pm[1] * X + pm[5] * Y + pm[9] * Z + pm[13];
        double hw = pm[3] * X + pm[7] * Y + pm[11] * Z + pm[15];
        double mw = 1 / (hw + 1e-7);
        double gnx = g_xy[2 * (size_t)i] * 0.5 * W, gny = g_xy[2 * (size_t)i + 1] * 0.5 * H;
        if (d_means2D) { d_means2D[3 * (size_t)i] = gnx; d_means2D[3 * (size_t)i + 1] = gny; }
        for (int k = 0; k < 3; k++) {
            dm[k] += (pm[k * 4 + 0] * mw - pm[k * 4 + 3] * hx * mw * mw) * gnx +
                     (pm[k * 4 + 1] * mw - pm[k * 4 + 3] * hy * mw * mw) * gny;
        }
        {   /* projection matrix rows 0, 1, 3 (row 2 = clip z is unused) */
            double ph[4] = {X, Y, Z, 1.0};
            for (int k = 0; k < 4; k++) {
                camloc[16 + k * 4 + 0] += gnx * mw * ph[k]; camloc[16 + k * 4 + 1] += gny * mw * ph[k];
                camloc[16 + k * 4 + 3] += -(gnx * hx + gny * hy) * mw * mw * ph[k];
            }
        }
        /* ---- depth feature = view z ---- */
        for (int k = 0; k < 3; k++) dm[k] += vm[k * 4 + 2] * g_z[i];
        if (d_means3D) for (int k = 0; k < 3; k++) d_means3D[3 * (size_t)i + k] = dm[k];

        /* ---- cov3D -> scale, rotation ---- */
        if (I->cov3D_precomp) {
            if (d_cov3D) for (int k = 0; k < 6; k++) d_cov3D[6 * (size_t)i + k] = gS[k];
        } else {
            const float *q = I->rotations + 4 * (size_t)i, *sc = I->scales + 3 * (size_t)i;
            double r = q[0], x = q[1], y = q[2], z = q[3];
            double Rm[3][3] = {
                {1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)},
                {2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)},
                {2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)}};
            double sv[3] = {I->scale_modifier * (double)sc[0], I->scale_modifier * (double)sc[1], I->scale_modifier * (double)sc[2]};
            /* Sigma = L L^T, L = R diag(s).  full symmetric gradient */
            double Gf[3][3] = {{gS[0], 0.5 * gS[1], 0.5 * gS[2]}, {0.5 * gS[1], gS[3], 0.5 * gS[4]}, {0.5 * gS[2], 0.5 * gS[4], gS[5]}};
            double gL[3][3]; /* 2 Gf L */
            for (int a_ = 0; a_ < 3; a_++)
                for (int b_ = 0; b_ < 3; b_++) {
                    double acc2 = 0;
                    for (int k = 0; k < 3; k++) acc2 += Gf[a_][k] * Rm[k][b_] * sv[b_];
                    gL[a_][b_] = 2 * acc2;
                }
            double gR[3][3];
            for (int b_ = 0; b_ < 3; b_++) {
                double gs = 0;
                for (int a_ = 0; a_ < 3; a_++) { gs += gL[a_][b_] * Rm[a_][b_]; gR[a_][b_] = gL[a_][b_] * sv[b_]; }
                if (d_scales) d_scales[3 * (size_t)i + b_] = gs * I->scale_modifier;
            }
            if (d_rotations) {
                double *dq = d_rotations + 4 * (size_t)i;
                dq[0] = 2 * (-z * gR[0][1] + y * gR[0][2] + z * gR[1][0] - x * gR[1][2] - y * gR[2][0] + x * gR[2][1]);
                dq[1] = 2 * (y * gR[0][1] + z * gR[0][2] + y * gR[1][0] - 2 * x * gR[1][1] - r * gR[1][2] + z * gR[2][0] + r * gR[2][1] - 2 * x * gR[2][2]);
                dq[2] = 2 * (-2 * y * gR[0][0] + x * gR[0][1] + r * gR[0][2] + x * gR[1][0] + z * gR[1][2] - r * gR[2][0] + z * gR[2][1] - 2 * y * gR[2][2]);
                dq[3] = 2 * (-2 * z * gR[0][0] - r * gR[0][1] + x * gR[0][2] + r * gR[1][0] - 2 * z * gR[1][1] + y * gR[1][2] + x * gR[2][0] + y * gR[2][1]);
            }
        }
    }
    if (d_camera) {
#pragma omp critical
        for (int k = 0; k < 35; k++) d_camera[k] += camloc[k];
    }
    } /* omp parallel */
    free(g_xy); free(g_conic); free(g_rgb); free(g_z); free(g_op);
}

/* CPU-baseline legs (bench.py cpu_baseline): stage timings on the host cores.
 * stage 0: preprocess + duplicate + sort + ranges (K1-K5);  stage 1: + blend (K6). */
int64_t gsr_oracle_run_stages(const GsrOracleIn *in, int with_blend)
{
    Ctx *c = gsr_oracle_prepare(in, 1);
    int64_t R = c->R;
    if (with_blend) {
        const size_t P = (size_t)in->W * in->H;
        c->final_T = (real *)calloc(P ? P : 1, sizeof(real));
        c->n_contrib = (uint32_t *)calloc(P ? P : 1, sizeof(uint32_t));
        c->acc = (real *)calloc((P ? P : 1) * 5, sizeof(real));
        c->px_ambig = (uint8_t *)calloc(P ? P : 1, 1);
        blend_all(c);
    }
    gsr_oracle_free(c);
    return R;
}

int gsr_oracle_real_bytes(void) { return (int)sizeof(real); }

/* unit-level entry points so the reference-owned pieces can be pinned against tests/golden */
void gsr_oracle_sh_eval(int deg, const float *sh /*[16,3]*/, const double *dir, double *out /*3, before +0.5/clamp*/)
{
    real d[3] = {(real)dir[0], (real)dir[1], (real)dir[2]}, o[3];
    sh_to_rgb(deg, sh, d, o);
    out[0] = o[0]; out[1] = o[1]; out[2] = o[2];
}

void gsr_oracle_cov3d(const float *scale, float mod, const float *rot, double *out /*6*/)
{
    real c[6];
    cov3d_from_scale_rot(scale, mod, rot, c);
    for (int k = 0; k < 6; k++) out[k] = c[k];
}
