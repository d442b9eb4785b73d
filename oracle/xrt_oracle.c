/*
 * xrt_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C restatement of the reference's (PrincetonUniversity/xicsrt v0.8.13,
 * pure NumPy) per-photon propagation path, written from the reference's
 * behaviour, array-at-a-time in the reference's own call order so that the
 * global np.random stream is consumed exactly as the reference consumes it.
 * Every function cites the reference file:line it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this.  The product (xicsrt_amd/) never does.
 *
 * PARITY PINNING: pinned against golden vectors generated in the build
 * container by importing the reference itself (tests/golden/make_golden.py,
 * numpy 2.2.6 / OpenBLAS as shipped there): integer results (num_out, images,
 * masks) bit-exact; floating-point ray arrays to <= 1e-12 relative (the
 * third-party libm/SVML transcendentals of NumPy differ from glibc in the last
 * ulp).  Third-party arithmetic restated here (not in /root/reference):
 *   - numpy legacy RandomState: MT19937 init_genrand seeding, genrand_res53
 *     doubles, uniform = low + (high-low)*d, polar-method gauss with cache;
 *   - numpy reductions as measured on numpy 2.2.6 (x86-64 SSE2 baseline):
 *       einsum('ij,ij->i'), einsum('ij,j->i'), einsum('ji,ki->kj'):  (p0 + p2) + p1
 *       einsum('ij,ijk->ik'), einsum('ij,ki->kj'), linalg.norm:      (p0 + p1) + p2
 *       np.dot((n,3),(3,)) via OpenBLAS dgemv:   fma(a2,b2, fma(a0,b0, a1*b1))
 *   - np.cross component formulas, np.interp, np.round (half-to-even).
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared (see oracle/Makefile).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <stdio.h>
#include <pthread.h>

#include "../include/xicsrt_hip.h"

/* ------------------------------------------------------------------------ */
/* numpy legacy RandomState                                                   */
/* ------------------------------------------------------------------------ */

typedef struct {
    uint32_t key[624];
    int      pos;
    int      has_gauss;
    double   gauss;
} mt_t;

/* np.random.seed(int) -> mt19937_seed / init_genrand (xicsrt_raytrace.py:111) */
static void mt_seed(mt_t* s, uint32_t seed)
{
    for (int i = 0; i < 624; i++) {
        s->key[i] = seed;
        seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)(i + 1);
    }
    s->pos = 624;
    s->has_gauss = 0;
    s->gauss = 0.0;
}

static void mt_regen(mt_t* s)
{
    uint32_t* mt = s->key;
    int i;
    uint32_t y;
    for (i = 0; i < 624 - 397; i++) {
        y = (mt[i] & 0x80000000u) | (mt[i + 1] & 0x7fffffffu);
        mt[i] = mt[i + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    for (; i < 623; i++) {
        y = (mt[i] & 0x80000000u) | (mt[i + 1] & 0x7fffffffu);
        mt[i] = mt[i + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
    mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    s->pos = 0;
}

static uint32_t mt_u32(mt_t* s)
{
    if (s->pos == 624) mt_regen(s);
    uint32_t y = s->key[s->pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

/* genrand_res53 */
static double mt_double(mt_t* s)
{
    uint32_t a = mt_u32(s) >> 5, b = mt_u32(s) >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
}

/* legacy_gauss: polar Box-Muller with a cached second value */
static double mt_gauss(mt_t* s)
{
    if (s->has_gauss) {
        double t = s->gauss;
        s->has_gauss = 0;
        s->gauss = 0.0;
        return t;
    }
    double f, x1, x2, r2;
    do {
        x1 = 2.0 * mt_double(s) - 1.0;
        x2 = 2.0 * mt_double(s) - 1.0;
        r2 = x1 * x1 + x2 * x2;
    } while (r2 >= 1.0 || r2 == 0.0);
    f = sqrt(-2.0 * log(r2) / r2);
    s->gauss = f * x1;
    s->has_gauss = 1;
    return f * x2;
}

/* numpy legacy random_poisson (numpy/random/src/distributions/distributions.c):
 * PTRS transformed rejection for lam >= 10, multiplication method below */
static double random_loggam(double x)
{
    static const double a[10] = {8.333333333333333e-02, -2.777777777777778e-03, 7.936507936507937e-04,
                                 -5.952380952380952e-04, 8.417508417508418e-04, -1.917526917526918e-03,
                                 6.410256410256410e-03, -2.955065359477124e-02, 1.796443723688307e-01,
                                 -1.39243221690590e+00};
    double x0, x2, lg2pi, gl, gl0;
    long k, n;
    if ((x == 1.0) || (x == 2.0)) return 0.0;
    else if (x < 7.0) n = (long)(7 - x);
    else n = 0;
    x0 = x + n;
    x2 = (1.0 / x0) * (1.0 / x0);
    lg2pi = 1.8378770664093453e+00;
    gl0 = a[9];
    for (k = 8; k >= 0; k--) { gl0 *= x2; gl0 += a[k]; }
    gl = gl0 / x0 + 0.5 * lg2pi + (x0 - 0.5) * log(x0) - x0;
    if (x < 7.0) for (k = 1; k <= n; k++) { gl -= log(x0 - 1.0); x0 -= 1.0; }
    return gl;
}

static int64_t mt_poisson(mt_t* s, double lam)
{
    if (lam >= 10) {
        double slam = sqrt(lam), loglam = log(lam);
        double b = 0.931 + 2.53 * slam;
        double a = -0.059 + 0.02483 * b;
        double invalpha = 1.1239 + 1.1328 / (b - 3.4);
        double vr = 0.9277 - 3.6224 / (b - 2);
        for (;;) {
            double U = mt_double(s) - 0.5;
            double V = mt_double(s);
            double us = 0.5 - fabs(U);
            int64_t k = (int64_t)floor((2 * a / us + b) * U + lam + 0.43);
            if ((us >= 0.07) && (V <= vr)) return k;
            if ((k < 0) || ((us < 0.013) && (V > us))) continue;
            if ((log(V) + log(invalpha) - log(a / (us * us) + b)) <= (-lam + k * loglam - random_loggam(k + 1))) return k;
        }
    } else if (lam == 0) {
        return 0;
    } else {
        double enlam = exp(-lam), prod = 1.0;
        int64_t X = 0;
        for (;;) {
            double U = mt_double(s);
            prod *= U;
            if (prod > enlam) X += 1; else return X;
        }
    }
}

/* ------------------------------------------------------------------------ */
/* small numeric helpers with numpy's evaluation order                        */
/* ------------------------------------------------------------------------ */

/* einsum('ij,ij->i') and friends on 3-vectors: SSE2 2-lane accumulate then tail */
static inline double dot_e(const double a[3], const double b[3])
{
    return (a[0] * b[0] + a[2] * b[2]) + a[1] * b[1];
}
/* np.linalg.norm(axis=1) / einsum over a strided middle index */
static inline double dot_n(const double a[3], const double b[3])
{
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2];
}
/* np.dot((n,3),(3,)) through OpenBLAS dgemv on the build machine */
static inline double dot_blas(const double a[3], const double b[3])
{
    return fma(a[2], b[2], fma(a[0], b[0], a[1] * b[1]));
}
static inline double norm3(const double a[3]) { return sqrt(dot_n(a, a)); }
/* np.cross */
static inline void cross3(const double a[3], const double b[3], double c[3])
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
/* GeometryObject.vector_to_local: einsum('ji,ki->kj') (_GeometryObject.py:163) */
static inline void to_local(const double R[9], const double v[3], double out[3])
{
    for (int j = 0; j < 3; j++) out[j] = dot_e(&R[3 * j], v);
}
/* GeometryObject.vector_to_external: einsum('ij,ki->kj') (_GeometryObject.py:149) */
static inline void to_external(const double R[9], const double v[3], double out[3])
{
    for (int j = 0; j < 3; j++)
        out[j] = (R[0 + j] * v[0] + R[3 + j] * v[1]) + R[6 + j] * v[2];
}

/* np.interp(x, xp, fp) */
static double np_interp(double x, const double* xp, const double* fp, int n)
{
    if (x != x) return x;
    if (x < xp[0]) return fp[0];
    if (x > xp[n - 1]) return fp[n - 1];
    int lo = 0, hi = n - 1;       /* find j with xp[j] <= x < xp[j+1] */
    while (hi - lo > 1) {
        int mid = (lo + hi) / 2;
        if (x >= xp[mid]) lo = mid; else hi = mid;
    }
    int j = lo;
    if (x == xp[n - 1]) return fp[n - 1];
    if (xp[j] == x) return fp[j];
    double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
    double r = slope * (x - xp[j]) + fp[j];
    if (r != r) {
        r = slope * (x - xp[j + 1]) + fp[j + 1];
        if (r != r && fp[j] == fp[j + 1]) r = fp[j];
    }
    return r;
}

/* ------------------------------------------------------------------------ */
/* ray arrays (objects/_RayArray.py:12; 'weight' added at _XicsrtSourceGeneric.py:215) */
/* ------------------------------------------------------------------------ */

typedef struct {
    int64_t  n;
    double*  o;     /* [n][3] */
    double*  d;     /* [n][3] */
    double*  wl;    /* [n]    */
    double*  wt;    /* [n]    */
    uint8_t* mask;  /* [n]    */
    uint8_t* prev;  /* [n] mask before the current optic            */
    uint8_t* hit;   /* [n] ray had an intersection at this optic    */
    /* scratch */
    double*  x;     /* [n][3] intersection */
    double*  nrm;   /* [n][3] normal       */
    double*  loc;   /* [n][3] local coords */
    double*  tmp;   /* [5n]                */
} rays_t;

static int rays_alloc(rays_t* r, int64_t n)
{
    memset(r, 0, sizeof(*r));
    r->n = n;
    size_t m = (size_t)(n > 0 ? n : 1);
    r->o = malloc(m * 3 * sizeof(double));
    r->d = malloc(m * 3 * sizeof(double));
    r->wl = malloc(m * sizeof(double));
    r->wt = malloc(m * sizeof(double));
    r->mask = malloc(m);
    r->prev = malloc(m);
    r->hit = malloc(m);
    r->x = malloc(m * 3 * sizeof(double));
    r->nrm = malloc(m * 3 * sizeof(double));
    r->loc = malloc(m * 3 * sizeof(double));
    r->tmp = malloc(m * 5 * sizeof(double));
    return (r->o && r->d && r->wl && r->wt && r->mask && r->prev && r->hit && r->x && r->nrm && r->loc && r->tmp) ? 0 : -1;
}
static void rays_free(rays_t* r)
{
    free(r->o); free(r->d); free(r->wl); free(r->wt); free(r->mask); free(r->prev); free(r->hit);
    free(r->x); free(r->nrm); free(r->loc); free(r->tmp);
}

/* ------------------------------------------------------------------------ */
/* source                                                                     */
/* ------------------------------------------------------------------------ */

/* np.random.uniform(low, high, n): low + (high-low)*d, array-sequential */
static void fill_uniform(mt_t* mt, double low, double range, double* out, int64_t n)
{
    for (int64_t i = 0; i < n; i++) out[i] = low + range * mt_double(mt);
}

/* tools/xicsrt_spread.py:80-339 -- local unit vectors about +z */
static void vector_distribution(const xrt_source_t* s, mt_t* mt, double* lv /*[n][3]*/,
                                double* t0, double* t1, int64_t n)
{
    switch (s->angular_dist) {
    case XRT_ANG_ISOTROPIC:
        /* xicsrt_spread.py:102-108 */
        fill_uniform(mt, s->ang[0], 1.0 - s->ang[0], t0, n);
        fill_uniform(mt, 0.0, s->two_pi - 0.0, t1, n);
        for (int64_t i = 0; i < n; i++) {
            double z = t0[i], phi = t1[i];
            double st = sqrt(1.0 - z * z);
            lv[3 * i + 0] = st * cos(phi);
            lv[3 * i + 1] = st * sin(phi);
            lv[3 * i + 2] = z;
        }
        break;
    case XRT_ANG_FLAT:
        /* xicsrt_spread.py:235-243 */
        fill_uniform(mt, 0.0, s->ang[0] - 0.0, t0, n);
        fill_uniform(mt, 0.0, s->two_pi - 0.0, t1, n);
        for (int64_t i = 0; i < n; i++) {
            double r = sqrt(t0[i]);
            double a1 = t1[i], a0 = atan(r);
            lv[3 * i + 0] = cos(a1) * sin(a0);
            lv[3 * i + 1] = sin(a1) * sin(a0);
            lv[3 * i + 2] = cos(a0);
        }
        break;
    case XRT_ANG_FLAT_XY:
        /* xicsrt_spread.py:281-292 */
        fill_uniform(mt, s->ang[0], s->ang[1] - s->ang[0], t0, n);
        fill_uniform(mt, s->ang[2], s->ang[3] - s->ang[2], t1, n);
        for (int64_t i = 0; i < n; i++) {
            double x = t0[i], y = t1[i];
            double a0 = atan(sqrt(x * x + y * y));
            double a1 = atan2(y, x);
            lv[3 * i + 0] = cos(a1) * sin(a0);
            lv[3 * i + 1] = sin(a1) * sin(a0);
            lv[3 * i + 2] = cos(a0);
        }
        break;
    case XRT_ANG_ISOTROPIC_XY: {
        /* xicsrt_spread.py:173-194: isotropic batches of n, filtered, until n kept */
        int64_t filled = 0;
        while (filled < n) {
            fill_uniform(mt, s->ang[0], 1.0 - s->ang[0], t0, n);
            fill_uniform(mt, 0.0, s->two_pi - 0.0, t1, n);
            for (int64_t i = 0; i < n && filled < n; i++) {
                double z = t0[i], phi = t1[i];
                double st = sqrt(1.0 - z * z);
                double vx = st * cos(phi), vy = st * sin(phi), vz = z;
                double ax = vx / sqrt(vx * vx + vz * vz);
                double ay = vy / sqrt(vy * vy + vz * vz);
                int ok = (ax > s->ang[1]) && (ax <= s->ang[2]) && (ay > s->ang[3]) && (ay <= s->ang[4]);
                if (ok) {
                    lv[3 * filled + 0] = vx; lv[3 * filled + 1] = vy; lv[3 * filled + 2] = vz;
                    filled++;
                }
            }
        }
        break; }
    }
}

/* sources/_XicsrtSourceGeneric.py:198-227: n rays of one source object written at ray index
 * `base` (a plasma bundle is a focused source whose origin is the bundle centre) */
static void generate_block(const xrt_source_t* s, const double* origin3, mt_t* mt, rays_t* rr, int64_t base, int64_t n)
{
    rays_t view = *rr;
    view.n = n;
    view.o = rr->o + 3 * base; view.d = rr->d + 3 * base; view.wl = rr->wl + base; view.wt = rr->wt + base;
    view.mask = rr->mask + base; view.nrm = rr->nrm + 3 * base; view.loc = rr->loc + 3 * base;
    rays_t* r = &view;
    const double* R = s->orientation;
    const double* xa = &R[0];
    const double* ya = &R[3];
    const double* za = &R[6];
    double* t = r->tmp;

    /* generate_origin (_XicsrtSourceGeneric.py:229-255) */
    if (s->spatial_dist == XRT_SPATIAL_UNIFORM) {
        for (int k = 0; k < 3; k++) {
            double low = -1.0 * s->size[k] / 2.0, high = s->size[k] / 2.0;
            fill_uniform(mt, low, high - low, t + k * n, n);
        }
    } else {
        /* np.random.multivariate_normal(mean, cov, n): standard normals row-major,
         * times the SVD factor the host computed (s->spatial_A), plus mean 0 */
        for (int64_t i = 0; i < n; i++) {
            double z[3];
            for (int k = 0; k < 3; k++) z[k] = mt_gauss(mt);
            for (int k = 0; k < 3; k++) {
                double v = 0.0;
                for (int j = 0; j < 3; j++) v += z[j] * s->spatial_A[3 * j + k];
                t[k * n + i] = v + 0.0;
            }
        }
    }
    for (int64_t i = 0; i < n; i++)
        for (int k = 0; k < 3; k++)
            r->o[3 * i + k] = ((origin3[k] + t[0 * n + i] * xa[k]) + t[1 * n + i] * ya[k]) + t[2 * n + i] * za[k];

    /* generate_direction -> make_normal (:262-266, Directed :46-50, Focused :40-44) */
    double* nv = r->nrm;
    for (int64_t i = 0; i < n; i++) {
        double a[3];
        if (s->kind == XRT_SRC_FOCUSED || s->kind == XRT_SRC_PLASMA)
            for (int k = 0; k < 3; k++) a[k] = s->axis[k] - r->o[3 * i + k];
        else
            for (int k = 0; k < 3; k++) a[k] = s->axis[k];
        double m = norm3(a);
        for (int k = 0; k < 3; k++) nv[3 * i + k] = a[k] / m;
    }
    /* random_direction (:268-293) */
    double* lv = r->loc;
    vector_distribution(s, mt, lv, t, t + n, n);
    for (int64_t i = 0; i < n; i++) {
        const double* nn = &nv[3 * i];
        double c1[3], c2[3], o1[3], o2[3];
        cross3(nn, xa, c1);
        cross3(nn, za, c2);
        for (int k = 0; k < 3; k++) o1[k] = c1[k] + c2[k];
        double m1 = norm3(o1);
        for (int k = 0; k < 3; k++) o1[k] /= m1;
        cross3(nn, o1, o2);
        double m2 = norm3(o2);
        for (int k = 0; k < 3; k++) o2[k] /= m2;
        const double* l = &lv[3 * i];
        for (int k = 0; k < 3; k++)
            r->d[3 * i + k] = (l[0] * o2[k] + l[1] * o1[k]) + l[2] * nn[k];
    }

    /* generate_wavelength (:295-319) */
    switch (s->wavelength_dist) {
    case XRT_WL_CONST:
        for (int64_t i = 0; i < n; i++) r->wl[i] = 1.0 * s->wavelength;
        break;
    case XRT_WL_UNIFORM:
        fill_uniform(mt, s->wl_a, s->wl_b, r->wl, n);
        break;
    case XRT_WL_NORMAL:     /* np.random.normal(loc, sigma, n) (:366) */
        for (int64_t i = 0; i < n; i++) r->wl[i] = s->wavelength + s->wl_a * mt_gauss(mt);
        break;
    case XRT_WL_VOIGT:      /* tools/xicsrt_voigt.py:127-129, then += wavelength (:353) */
        fill_uniform(mt, s->wl_a, s->wl_b, t, n);
        for (int64_t i = 0; i < n; i++)
            r->wl[i] = np_interp(t[i], s->voigt_cdf, s->voigt_x, s->voigt_n) + s->wavelength;
        break;
    }
    if (s->has_velocity) {      /* Doppler (:314-317) */
        for (int64_t i = 0; i < n; i++) {
            double v = dot_e(s->velocity, &r->d[3 * i]);
            r->wl[i] *= 1.0 - (v / s->light_speed);
        }
    }
    for (int64_t i = 0; i < n; i++) { r->wt[i] = 1.0; r->mask[i] = 1; }
}

/* np.interp(x, xp, fp, left=0.0, right=0.0) (sources/_XicsrtPlasmaToroidalDatafile.py:34,43) */
static double np_interp_zero(double x, const double* xp, const double* fp, int n)
{
    if (x != x) return x;
    if (x < xp[0] || x > xp[n - 1]) return 0.0;
    return np_interp(x, xp, fp, n);
}

/* tools/xicsrt_voigt.py:30-106 voigt_cdf_tab(gamma, sigma) with the default gridsize 1000 and cutoff 1e-5:
 * cdf_x = bounds[1:] and cdf, NumPy's element-wise expressions in their order, np.cumsum sequentially.
 * wofz(z).real (scipy.special, Faddeeva function) is taken from Weideman's rational approximation (SIAM J.
 * Numer. Anal. 31 (1994) 1497, coefficients from the host, see xrt_plasma_t): within 3e-16 of SciPy's
 * values on the tables (tests/test_host.py).  Returns 0 when the reference would raise. */
#define VOIGT_GRID 1000
static double faddeeva_re(double u, double a, double L, const double* coef, int n)
{
    const double dr = L + a, di = -u, nr = L - a, ni = u;       /* L - i z, L + i z for z = u + i a */
    const double den = dr * dr + di * di;
    const double Zr = (nr * dr + ni * di) / den, Zi = (ni * dr - nr * di) / den;
    double pr = coef[0], pi = 0.0;
    for (int k = 1; k < n; k++) {
        const double tr = pr * Zr - pi * Zi + coef[k];
        pi = pr * Zi + pi * Zr;
        pr = tr;
    }
    const double d2r = dr * dr - di * di, d2i = 2.0 * dr * di, den2 = d2r * d2r + d2i * d2i;
    return 2.0 * (pr * d2r + pi * d2i) / den2 + 0.5641895835477563 * dr / den;
}

static int voigt_cdf_tab(double gamma, double sigma, const xrt_plasma_t* P, double* cdf_x, double* cdf)
{
    const double fraction = 0.5, cutoff = 1e-5;
    const double gauss_hwfm = sqrt(2.0 * log(1.0 / fraction)) * sigma;
    const double lorentz_hwfm = gamma * sqrt(1.0 / fraction - 1.0);
    const double hwfm_max = sqrt(gauss_hwfm * gauss_hwfm + lorentz_hwfm * lorentz_hwfm);
    const double min_spacing = hwfm_max / 5.0;
    const double value = 100.0 / 2 * min_spacing;
    const double lorentz_cutoff = gamma * sqrt(1.0 / cutoff - 1.0);
    const double gauss_cutoff = sqrt(-1 * (sigma * sigma) * 2 * log(cutoff * sigma * sqrt(2 * 3.141592653589793)));
    const double value_cutoff = lorentz_cutoff > gauss_cutoff ? lorentz_cutoff : gauss_cutoff;
    const double base = exp(1.0 / 10 * log(value_cutoff / value));
    const double step = (value - (-value)) / (double)VOIGT_GRID;
    const double a_im = (gamma / sqrt(2.0)) / sigma, norm = sqrt(2 * 3.141592653589793);
    double lo = 0.0, acc = 0.0, hi_cdf = 0.0;
    int mid = 0;
    for (int i = 0; i <= VOIGT_GRID; i++) {
        const double b0 = (i == VOIGT_GRID) ? value : (double)i * step + (-value);
        const double b = b0 * pow(base, fabs(b0 / value * 10));
        if (i > 0) {
            const double cx = (lo + b) / 2;
            const double u = ((cx - 0.0) / sqrt(2.0)) / sigma;
            const double y = faddeeva_re(u, a_im, P->weideman_L, P->weideman_a, P->n_weideman) / norm / sigma * 1.0;
            const double ydx = y * (b - lo);
            acc = (i == 1) ? ydx : acc + ydx;
            cdf_x[i - 1] = b;
            cdf[i - 1] = acc;
            if (i == 1 || acc > hi_cdf) hi_cdf = acc;
            mid += (acc > 0.25) && (acc < 0.75);
        }
        lo = b;
    }
    return !(mid < 3 || hi_cdf < 0.99);
}

/* What the reference's bundle pipeline assigns to the bundle centred at c (external frame):
 * bundle_filter (filters/_XicsrtBundleFilterSightline.py:31-56), bundle_generate
 * (sources/_XicsrtPlasmaToroidal.py:47-78 with tools/xicsrt_math.py:211-244), setup_bundle_spread
 * and the intensity of create_sources (sources/_XicsrtPlasmaGeneric.py:206-231, :301-319).
 * Fills `sb` (the bundle's focused source) and *lam; returns 0 when the bundle is masked out. */
static int bundle_eval(const xrt_source_t* s, const double* c, xrt_source_t* sb, double* lam, double* vtab)
{
    const xrt_plasma_t* P = s->plasma;
    *sb = *s;
    for (int f = 0; f < P->n_filters; f++) {
        const xrt_bundle_filter_t* F = &P->filters[f];
        double l0[3], l2[3];
        for (int k = 0; k < 3; k++) l0[k] = F->origin[k] - c[k];
        const double proj = dot_e(F->zaxis, l0);                       /* einsum('j,ij->i') */
        for (int k = 0; k < 3; k++) l2[k] = l0[k] - F->zaxis[k] * proj;
        const double distance = sqrt(dot_e(l2, l2));                    /* einsum('ij,ij->i') */
        if (!(F->radius >= distance)) return 0;
    }
    double emis = P->emissivity * P->emissivity_scale;
    if (P->geometry == XRT_PLASMA_TOROIDAL) {
        const double px = c[0] - P->torus_origin[0], py = c[1] - P->torus_origin[1], pz = c[2] - P->torus_origin[2];
        const double d = sqrt(fma(py, py, px * px)) - P->major_radius;  /* norm of a 1-D pair: BLAS ddot */
        const double r0 = sqrt(pz * pz + d * d);                        /* np.power(.,2) is x*x */
        volatile double two = 2.0;
        double flx = pow(r0, two);                                      /* numpy scalar ** 2 calls libm pow */
        flx /= P->minor_radius;
        const double rho = sqrt(flx);
        if (P->n_temperature > 0) {
            const double temp = np_interp_zero(rho, P->temperature_rho, P->temperature_val, P->n_temperature) * P->temperature_scale;
            if (!isfinite(temp)) return 0;
            /* the bundle's wavelength case (_XicsrtSourceGeneric.py:321-367) with linewidth == 0 */
            if (s->wavelength_dist == XRT_WL_NORMAL || s->wavelength_dist == XRT_WL_CONST) {
                if (temp == 0.0) sb->wavelength_dist = XRT_WL_CONST;
                else {
                    sb->wavelength_dist = XRT_WL_NORMAL;
                    sb->wl_a = sqrt(temp / P->mass_number / P->amu_kg / P->c_squared * P->ev_J) * s->wavelength;
                }
            } else if (s->wavelength_dist == XRT_WL_VOIGT && P->voigt_gamma > 0.0) {
                /* linewidth != 0: the bundle's own Voigt profile; a cold bundle gets 1 eV (:335-353) */
                const double tv = (temp == 0.0) ? temp + 1.0 : temp;
                const double sigma = sqrt(tv / P->mass_number / P->amu_kg / P->c_squared * P->ev_J) * s->wavelength;
                if (!voigt_cdf_tab(P->voigt_gamma, sigma, P, vtab, vtab + VOIGT_GRID)) return -9;
                sb->voigt_x = vtab; sb->voigt_cdf = vtab + VOIGT_GRID; sb->voigt_n = VOIGT_GRID;
                double lo = vtab[VOIGT_GRID], hi = vtab[VOIGT_GRID];
                for (int i = 1; i < VOIGT_GRID; i++) {
                    if (vtab[VOIGT_GRID + i] < lo) lo = vtab[VOIGT_GRID + i];
                    if (vtab[VOIGT_GRID + i] > hi) hi = vtab[VOIGT_GRID + i];
                }
                sb->wl_a = lo; sb->wl_b = hi - lo;         /* np.random.uniform(np.min(cdf), np.max(cdf), size) */
            }
        }
        if (P->n_emissivity > 0)
            emis = np_interp_zero(rho, P->emissivity_rho, P->emissivity_val, P->n_emissivity) * P->emissivity_scale;
    }
    double solid_angle = P->solid_angle;
    if (P->has_spread_radius) {
        double v[3] = {c[0] - s->axis[0], c[1] - s->axis[1], c[2] - s->axis[2]};
        const double dist = norm3(v);
        const double spread = atan(P->spread_radius / dist);
        volatile double two = 2.0;
        solid_angle = P->four_pi * pow(sin(spread / 2), two);
        sb->ang[0] = cos(spread);
    }
    double intensity = emis * P->time_resolution * P->bundle_volume * solid_angle / P->four_pi;
    intensity *= P->volume_ratio;
    *lam = intensity;
    return 1;
}

/* sources/_XicsrtSourceGeneric.py:198 for the plain sources; for XRT_SRC_PLASMA
 * XicsrtPlasmaGeneric.generate_rays (sources/_XicsrtPlasmaGeneric.py:384-393):
 * setup_bundles (:176-204), then one focused source per bundle in order (:286-345) whose
 * ray count is np.random.poisson(intensity) (_XicsrtSourceGeneric.py:191-192).
 * Returns the number of rays generated (<= capacity r->n) or -1 on overflow. */
static int64_t generate_rays(const xrt_source_t* s, mt_t* mt, rays_t* r)
{
    if (s->kind == XRT_SRC_EXTERNAL) {      /* the caller's rays (host pointers here), dead ones included */
        const int64_t n = r->n;
        for (int64_t i = 0; i < n; i++) {
            for (int k = 0; k < 3; k++) { r->o[3 * i + k] = s->ext_rays[k * n + i]; r->d[3 * i + k] = s->ext_rays[(3 + k) * n + i]; }
            r->wl[i] = s->ext_rays[6 * n + i];
            r->wt[i] = s->ext_rays[7 * n + i];
            r->mask[i] = s->ext_mask[i] != 0;
        }
        return n;
    }
    if (s->kind != XRT_SRC_PLASMA) {
        generate_block(s, s->origin, mt, r, 0, r->n);
        /* ray_filter (_XicsrtSourceGeneric.py:223, :393-396): XicsrtBundleFilterSightline.filter on the ray
         * dictionary (filters/_XicsrtBundleFilterSightline.py:31-56): mask &= radius >= distance(origin, sightline) */
        for (int f = 0; f < s->n_ray_filters; f++) {
            const xrt_bundle_filter_t* F = &s->ray_filters[f];
            for (int64_t i = 0; i < r->n; i++) {
                double l0[3], l2[3];
                for (int k = 0; k < 3; k++) l0[k] = F->origin[k] - r->o[3 * i + k];
                const double proj = dot_e(F->zaxis, l0);
                for (int k = 0; k < 3; k++) l2[k] = l0[k] - F->zaxis[k] * proj;
                const double distance = sqrt(dot_e(l2, l2));
                r->mask[i] = r->mask[i] && (F->radius >= distance);
            }
        }
        return r->n;
    }
    const int64_t B = s->bundle_count;
    double* off = malloc(sizeof(double) * (3 * (size_t)(B > 0 ? B : 1) + 2 * VOIGT_GRID));
    double* vtab = off + 3 * (size_t)(B > 0 ? B : 1);
    for (int k = 0; k < 3; k++) {
        double low = -1.0 * s->plasma_size[k] / 2.0, high = s->plasma_size[k] / 2.0;
        fill_uniform(mt, low, high - low, off + k * B, B);
    }
    int64_t total = 0;
    for (int64_t b = 0; b < B; b++) {
        double v[3] = {off[0 * B + b], off[1 * B + b], off[2 * B + b]}, c[3];
        to_external(s->orientation, v, c);                 /* point_to_external (:199) */
        for (int k = 0; k < 3; k++) c[k] = c[k] + s->origin[k];
        int64_t nb;
        xrt_source_t sb_store;
        const xrt_source_t* sb = s;
        double lam = s->bundle_intensity;
        if (s->plasma) {
            const int ok = bundle_eval(s, c, &sb_store, &lam, vtab);
            if (ok < 0) { free(off); return -9; }                    /* voigt_cdf_tab raised */
            if (!ok) continue;                                       /* masked bundles draw nothing (:288-289) */
            sb = &sb_store;
        }
        if (s->use_poisson) nb = mt_poisson(mt, lam);
        else {
            if (lam < 1) { free(off); return -2; }                  /* ValueError (_XicsrtSourceGeneric.py:193-194) */
            nb = (int64_t)lam;
        }
        if (total + nb > r->n) { free(off); return -1; }
        generate_block(sb, c, mt, r, total, nb);
        total += nb;
    }
    free(off);
    if (total == 0) return -10;                                      /* 'No rays generated' (_XicsrtPlasmaGeneric.py:368-369) */
    return total;
}

/* ------------------------------------------------------------------------ */
/* optics                                                                     */
/* ------------------------------------------------------------------------ */

/* optics/_ShapeObject.py:79 */
static inline void location_from_distance(const double* o, const double* d, double t, double* x)
{
    for (int k = 0; k < 3; k++) x[k] = o[k] + d[k] * t;
}

/* optics/_ShapePlane.py:25-62 */
static void intersect_plane(const xrt_optic_t* op, rays_t* r)
{
    const double* za = &op->orientation[6];
    for (int64_t i = 0; i < r->n; i++) {
        if (!r->mask[i]) continue;
        const double* o = &r->o[3 * i];
        const double* d = &r->d[3 * i];
        double t;
        if (op->flags & XRT_F_TRACE_LOCAL) {
            const double ez[3] = {0.0, 0.0, 1.0};
            double v[3] = {0.0 - o[0], 0.0 - o[1], 0.0 - o[2]};
            t = dot_blas(v, ez) / dot_blas(d, ez);
        } else {
            double v[3] = {op->origin[0] - o[0], op->origin[1] - o[1], op->origin[2] - o[2]};
            t = dot_blas(v, za) / dot_blas(d, za);
        }
        if (!(t >= 0.0)) { r->mask[i] = 0; continue; }
        location_from_distance(o, d, t, &r->x[3 * i]);
        for (int k = 0; k < 3; k++) r->nrm[3 * i + k] = za[k];
    }
}

/* optics/_ShapeSphere.py:37-106 */
static void intersect_sphere(const xrt_optic_t* op, rays_t* r)
{
    for (int64_t i = 0; i < r->n; i++) {
        if (!r->mask[i]) continue;
        const double* o = &r->o[3 * i];
        const double* d = &r->d[3 * i];
        double L[3] = {op->center[0] - o[0], op->center[1] - o[1], op->center[2] - o[2]};
        double t_ca = dot_e(L, d);
        double dd = sqrt(dot_e(L, L) - t_ca * t_ca);
        if (!(dd <= op->radius)) { r->mask[i] = 0; continue; }
        double t_hc = sqrt(op->radius2 - dd * dd);
        double t0 = t_ca - t_hc, t1 = t_ca + t_hc, t;
        if (op->flags & XRT_F_CONVEX) t = (t0 < t1) ? t0 : t1;
        else                          t = (t0 > t1) ? t0 : t1;
        double* x = &r->x[3 * i];
        location_from_distance(o, d, t, x);
        double c[3] = {op->center[0] - x[0], op->center[1] - x[1], op->center[2] - x[2]};
        double m = norm3(c);
        for (int k = 0; k < 3; k++) r->nrm[3 * i + k] = c[k] / m;
    }
}

/* optics/_ShapeCylinder.py:52-133 */
static void intersect_cylinder(const xrt_optic_t* op, rays_t* r)
{
    const double* pa = op->center;
    const double* va = &op->orientation[0];
    for (int64_t i = 0; i < r->n; i++) {
        if (!r->mask[i]) continue;
        const double* o = &r->o[3 * i];
        const double* d = &r->d[3 * i];
        double dp[3] = {o[0] - pa[0], o[1] - pa[1], o[2] - pa[2]};
        double dDva = dot_e(d, va), dpva = dot_e(dp, va);
        double A1[3], B1[3];
        for (int k = 0; k < 3; k++) { A1[k] = d[k] - dDva * va[k]; B1[k] = dp[k] - dpva * va[k]; }
        double A = dot_e(A1, A1);
        double B = 2.0 * dot_e(A1, B1);
        double C = dot_e(B1, B1) - op->radius2;
        double dis = B * B - 4.0 * A * C;
        if (!(dis >= 0.0)) { r->mask[i] = 0; continue; }
        double sq = sqrt(dis);
        double t0 = (-B - sq) / (2.0 * A), t1 = (-B + sq) / (2.0 * A), t;
        if (op->flags & XRT_F_CONVEX) t = (t0 < t1) ? t0 : t1;
        else                          t = (t0 > t1) ? t0 : t1;
        double* x = &r->x[3 * i];
        location_from_distance(o, d, t, x);
        double q[3] = {pa[0] - x[0], pa[1] - x[1], pa[2] - x[2]};
        double dummy = dot_e(q, va);
        double c[3];
        for (int k = 0; k < 3; k++) c[k] = (pa[k] - dummy * va[k]) - x[k];
        double m = norm3(c);
        for (int k = 0; k < 3; k++) r->nrm[3 * i + k] = c[k] / m;
    }
}

/* tools/xicsrt_quartic.py:54-160 multi_cubic(1, b0, c0, d0, all_roots=False): one real root */
static double cubic_one_root(double a, double b, double c)
{
    const double third = 1. / 3.;
    double a13 = a * third;
    double a2 = a13 * a13;
    double f = third * b - a2;
    double g = a13 * (2 * a2 - b) + c;
    double h = 0.25 * g * g + f * f * f;
    if (f == 0 && g == 0 && h == 0) {                 /* m1: all roots real and equal */
        double cr = (c >= 0) ? pow(c, third) : -pow(-c, third);
        return -cr;
    }
    if (h <= 0) {                                     /* m2: real and distinct */
        double j = sqrt(-f);
        double k = acos(-0.5 * g / (j * j * j));
        double m = cos(third * k);
        return 2 * j * m - a13;
    }
    {                                                 /* m3: one real root */
        double sqrt_h = sqrt(h);
        double x1 = -0.5 * g + sqrt_h, x2 = -0.5 * g - sqrt_h;
        double S = (x1 >= 0) ? pow(x1, third) : -pow(-x1, third);
        double U = (x2 >= 0) ? pow(x2, third) : -pow(-x2, third);
        return (S + U) - a13;
    }
}

/* tools/xicsrt_quartic.py:162-207 multi_quartic with a0 = 1, restated in real arithmetic:
 * the reference works in complex128 and then discards every root whose imaginary part is
 * not exactly zero (_ShapeTorus.py:164-167).  The imaginary parts are exactly zero iff
 * s = sqrt(2p + 2 z0) is real and the quadratic's discriminant is >= 0; complex division
 * by a real s is numpy's Smith form x * (1.0 / s). */
static void quartic_roots(double b0, double c0, double d0, double e0, double roots[4])
{
    double a = b0, b = c0, c = d0, d = e0;            /* division by a0 = 1 is exact */
    double a0 = 0.25 * a;
    double a02 = a0 * a0;
    double p = 3 * a02 - 0.5 * b;
    double q = a * a02 - b * a0 + 0.5 * c;
    double r = 3 * a02 * a02 - b * a02 + c * a0 - d;
    double z0 = cubic_one_root(p, r, p * r - 0.5 * q * q);
    double sarg = 2 * p + 2 * z0;
    roots[0] = roots[1] = roots[2] = roots[3] = NAN;
    if (!(sarg >= 0.0)) return;                       /* s imaginary (or NaN): every root complex */
    double s = sqrt(sarg);
    double t = (s == 0.0) ? (z0 * z0 + r) : (-q) * (1.0 / s);
    {   /* multi_quadratic(1, s, z0 + t) - a0 */
        double h0 = -0.5 * s;
        double delta = h0 * h0 - (z0 + t);
        if (delta >= 0.0) {
            double sd = sqrt(delta);
            roots[0] = (h0 - sd) - a0;
            roots[1] = (h0 + sd) - a0;
        }
    }
    {   /* multi_quadratic(1, -s, z0 - t) - a0 */
        double h0 = -0.5 * (-s);
        double delta = h0 * h0 - (z0 - t);
        if (delta >= 0.0) {
            double sd = sqrt(delta);
            roots[2] = (h0 - sd) - a0;
            roots[3] = (h0 + sd) - a0;
        }
    }
}

/* optics/_ShapeTorus.py:110-216 */
static void intersect_torus(const xrt_optic_t* op, rays_t* r)
{
    const double* R = op->orientation;
    const double* ya = &R[3];
    const double Rt = op->torus_major;
    for (int64_t i = 0; i < r->n; i++) {
        if (!r->mask[i]) continue;
        const double* o = &r->o[3 * i];
        const double* d = &r->d[3 * i];
        double v[3] = {o[0] - op->center[0], o[1] - op->center[1], o[2] - op->center[2]};
        double O[3], D[3];
        to_local(R, v, O);
        to_local(R, d, D);
        double O_mag_sq = dot_e(O, O), dot_OD = dot_e(O, D);
        const double r_sq = op->torus_k[0], two_r_sq = op->torus_k[1], four_R2 = op->torus_k[2],
                     eight_R2 = op->torus_k[3], K = op->torus_k[4];
        double c1 = 4.0 * dot_OD;
        double c2 = ((4.0 * (dot_OD * dot_OD) + 2.0 * O_mag_sq) - two_r_sq) + four_R2 * (D[1] * D[1]);
        double c3 = (4.0 * dot_OD) * (O_mag_sq - r_sq) + (eight_R2 * D[1]) * O[1];
        double c4 = ((O_mag_sq * O_mag_sq - two_r_sq * O_mag_sq) + four_R2 * (O[1] * O[1])) + K;
        double roots[4];
        quartic_roots(c1, c2, c3, c4, roots);
        double t = roots[op->torus_root];
        if (!(isfinite(t) && t > 0.0)) { r->mask[i] = 0; continue; }
        double* x = &r->x[3 * i];
        location_from_distance(o, d, t, x);
        /* intersect_normal (:186-216) */
        double pt[3] = {x[0] - op->center[0], x[1] - op->center[1], x[2] - op->center[2]};
        double dy = dot_e(pt, ya);
        for (int k = 0; k < 3; k++) pt[k] = pt[k] - dy * ya[k];
        double m = norm3(pt);
        double Q[3];
        for (int k = 0; k < 3; k++) Q[k] = op->center[k] + Rt * (pt[k] / m);
        double xn[3] = {x[0] - Q[0], x[1] - Q[1], x[2] - Q[2]};
        double m2 = norm3(xn);
        for (int k = 0; k < 3; k++) r->nrm[3 * i + k] = xn[k] / m2;
    }
}

/* ---- mesh optics (optics/_ShapeMesh.py) ---------------------------------------------- */

/* Moller-Trumbore over every face, later faces overwrite earlier hits (:289-348).
 * np.einsum('i,ji->j', ..., optimize=True) is a BLAS matrix-vector product (fused order),
 * the 'ij,ij->i' forms are numpy's own (p0 + p2) + p1. */
static int mesh_intersect_1(const double* P0, const double* E1, const double* E2, int nf,
                            const double* O, const double* D, double* X, int* hit_face)
{
    const double epsilon = 1e-15;
    int hit = 0;
    for (int ii = 0; ii < nf; ii++) {
        const double* p0 = P0 + 3 * ii; const double* e1 = E1 + 3 * ii; const double* e2 = E2 + 3 * ii;
        double h[3], s[3], q[3];
        cross3(D, e2, h);
        double f = dot_blas(h, e1);
        if ((f > -epsilon) && (f < epsilon)) continue;
        f = 1.0 / f;
        for (int k = 0; k < 3; k++) s[k] = O[k] - p0[k];
        double u = f * dot_e(s, h);
        if ((u < 0.0) || (u > 1.0)) continue;
        cross3(s, e1, q);
        double v = f * dot_e(D, q);
        if ((v < 0.0) || (u + v > 1.0)) continue;
        double t = f * dot_blas(q, e2);
        hit = 1;
        *hit_face = ii;
        for (int k = 0; k < 3; k++) X[k] = O[k] + t * D[k];
    }
    return hit;
}

/* cKDTree(points).query(x)[1]: index of the nearest point (:464-475) */
static int mesh_nearest(const xrt_mesh_t* M, const double* x)
{
    int best = 0;
    double bd = INFINITY;
    for (int i = 0; i < M->n_points; i++) {
        const double* p = M->points + 3 * i;
        double dx = x[0] - p[0], dy = x[1] - p[1], dz = x[2] - p[2];
        double d = (dx * dx + dy * dy) + dz * dz;
        if (d < bd) { bd = d; best = i; }
    }
    return best;
}

/* the <= 8 faces around the nearest point: plane hit + area-sum test, first passing (:350-426) */
static int mesh_intersect_2(const xrt_mesh_t* M, int idx, const double* O, const double* D, double* X, int* hit_face)
{
    for (int k = 0; k < 8; k++) {
        const int f = M->p_faces_idx[k * M->n_points + idx];
        const int valid = M->p_faces_mask[k * M->n_points + idx];
        const double* p0 = M->p0 + 3 * f; const double* p1 = M->p1 + 3 * f; const double* p2 = M->p2 + 3 * f;
        const double* n = M->faces_normal + 3 * f;
        double t0[3] = {p0[0] - O[0], p0[1] - O[1], p0[2] - O[2]};
        double t1 = dot_e(t0, n), t2 = dot_e(D, n);
        double dist = t1 / t2;
        double I[3], a[3], b[3], c[3], bc[3], ca[3], ab[3];
        for (int q = 0; q < 3; q++) I[q] = D[q] * dist + O[q];
        for (int q = 0; q < 3; q++) { a[q] = I[q] - p0[q]; b[q] = I[q] - p1[q]; c[q] = I[q] - p2[q]; }
        cross3(b, c, bc); cross3(c, a, ca); cross3(a, b, ab);
        double diff = ((norm3(bc) + norm3(ca)) + norm3(ab)) - M->faces_area[f];
        if ((diff < 1e-10) && (dist >= 0) && valid) {
            for (int q = 0; q < 3; q++) X[q] = I[q];
            *hit_face = f;
            return 1;
        }
    }
    return 0;
}

/* SciPy CloughTocher2DInterpolator (third party, scipy 1.15.3 interpnd: barycentric walk +
 * _clough_tocher_2d_single), restated; verified against SciPy in tests/tools/ct_check.py */
static void ct_bary(const double* T, const double* x, double* c)
{
    c[2] = 1.0;
    for (int i = 0; i < 2; i++) {
        c[i] = 0.0;
        for (int j = 0; j < 2; j++) c[i] += T[2 * i + j] * (x[j] - T[4 + j]);
        c[2] -= c[i];
    }
}

static int ct_find_simplex(const xrt_mesh_t* M, const double* x, int start, double* c)
{
    const double eps = 100 * 2.220446049250313e-16;
    int s = start;
    if (x[0] != x[0] || x[1] != x[1]) return -1;
    for (int iter = 0; iter < M->n_simplices + 8; iter++) {
        ct_bary(M->ct_transform + 6 * s, x, c);
        int worst = -1;
        double wv = -eps;
        for (int k = 0; k < 3; k++) if (c[k] < wv) { wv = c[k]; worst = k; }
        if (worst < 0) return s;
        int nb = M->ct_neighbors[3 * s + worst];
        if (nb < 0) break;
        s = nb;
    }
    for (s = 0; s < M->n_simplices; s++) {          /* exhaustive fallback */
        ct_bary(M->ct_transform + 6 * s, x, c);
        if (c[0] >= -eps && c[1] >= -eps && c[2] >= -eps) return s;
    }
    return -1;
}

static double ct_eval(const xrt_mesh_t* M, int isimplex, const double* b, int which)
{
    const int* v = M->ct_simplices + 3 * isimplex;
    const double* pts = M->ct_points;
    const double* val = M->ct_values + (size_t)which * M->n_points;
    const double* grd = M->ct_grad + (size_t)which * M->n_points * 2;
    double e12x = pts[2 * v[1]] - pts[2 * v[0]], e12y = pts[2 * v[1] + 1] - pts[2 * v[0] + 1];
    double e23x = pts[2 * v[2]] - pts[2 * v[1]], e23y = pts[2 * v[2] + 1] - pts[2 * v[1] + 1];
    double e31x = pts[2 * v[0]] - pts[2 * v[2]], e31y = pts[2 * v[0] + 1] - pts[2 * v[2] + 1];
    double f1 = val[v[0]], f2 = val[v[1]], f3 = val[v[2]];
    const double* d1 = grd + 2 * v[0]; const double* d2 = grd + 2 * v[1]; const double* d3 = grd + 2 * v[2];
    double df12 = +(d1[0] * e12x + d1[1] * e12y);
    double df21 = -(d2[0] * e12x + d2[1] * e12y);
    double df23 = +(d2[0] * e23x + d2[1] * e23y);
    double df32 = -(d3[0] * e23x + d3[1] * e23y);
    double df31 = +(d3[0] * e31x + d3[1] * e31y);
    double df13 = -(d1[0] * e31x + d1[1] * e31y);
    double c3000 = f1, c2100 = (df12 + 3 * c3000) / 3, c2010 = (df13 + 3 * c3000) / 3;
    double c0300 = f2, c1200 = (df21 + 3 * c0300) / 3, c0210 = (df23 + 3 * c0300) / 3;
    double c0030 = f3, c1020 = (df31 + 3 * c0030) / 3, c0120 = (df32 + 3 * c0030) / 3;
    double c2001 = (c2100 + c2010 + c3000) / 3;
    double c0201 = (c1200 + c0300 + c0210) / 3;
    double c0021 = (c1020 + c0120 + c0030) / 3;
    double g[3];
    for (int k = 0; k < 3; k++) {
        int itri = M->ct_neighbors[3 * isimplex + k];
        if (itri == -1) { g[k] = -1. / 2; continue; }
        const int* w = M->ct_simplices + 3 * itri;
        double y[2], c[3];
        y[0] = (pts[2 * w[0]] + pts[2 * w[1]] + pts[2 * w[2]]) / 3;
        y[1] = (pts[2 * w[0] + 1] + pts[2 * w[1] + 1] + pts[2 * w[2] + 1]) / 3;
        ct_bary(M->ct_transform + 6 * isimplex, y, c);
        if (k == 0)      g[k] = (2 * c[2] + c[1] - 1) / (2 - 3 * c[2] - 3 * c[1]);
        else if (k == 1) g[k] = (2 * c[0] + c[2] - 1) / (2 - 3 * c[0] - 3 * c[2]);
        else             g[k] = (2 * c[1] + c[0] - 1) / (2 - 3 * c[1] - 3 * c[0]);
    }
    double c0111 = (g[0] * (-c0300 + 3 * c0210 - 3 * c0120 + c0030) + (-c0300 + 2 * c0210 - c0120 + c0021 + c0201)) / 2;
    double c1011 = (g[1] * (-c0030 + 3 * c1020 - 3 * c2010 + c3000) + (-c0030 + 2 * c1020 - c2010 + c2001 + c0021)) / 2;
    double c1101 = (g[2] * (-c3000 + 3 * c2100 - 3 * c1200 + c0300) + (-c3000 + 2 * c2100 - c1200 + c2001 + c0201)) / 2;
    double c1002 = (c1101 + c1011 + c2001) / 3;
    double c0102 = (c1101 + c0111 + c0201) / 3;
    double c0012 = (c1011 + c0111 + c0021) / 3;
    double c0003 = (c1002 + c0102 + c0012) / 3;
    double minval = b[0];
    for (int k = 0; k < 3; k++) if (b[k] < minval) minval = b[k];
    double b1 = b[0] - minval, b2 = b[1] - minval, b3 = b[2] - minval, b4 = 3 * minval;
    return (pow(b1, 3) * c3000 + 3 * pow(b1, 2) * b2 * c2100 + 3 * pow(b1, 2) * b3 * c2010 + 3 * pow(b1, 2) * b4 * c2001 +
            3 * b1 * pow(b2, 2) * c1200 + 6 * b1 * b2 * b4 * c1101 + 3 * b1 * pow(b3, 2) * c1020 + 6 * b1 * b3 * b4 * c1011 +
            3 * b1 * pow(b4, 2) * c1002 + pow(b2, 3) * c0300 + 3 * pow(b2, 2) * b3 * c0210 + 3 * pow(b2, 2) * b4 * c0201 +
            3 * b2 * pow(b3, 2) * c0120 + 6 * b2 * b3 * b4 * c0111 + 3 * b2 * pow(b4, 2) * c0102 + pow(b3, 3) * c0030 +
            3 * pow(b3, 2) * b4 * c0021 + 3 * b3 * pow(b4, 2) * c0012 + pow(b4, 3) * c0003);
}

/* ShapeMesh.intersect (:135-170) for rays already in the optic's frame */
static void intersect_mesh(const xrt_optic_t* op, rays_t* r)
{
    const xrt_mesh_t* M = op->mesh;
    for (int64_t i = 0; i < r->n; i++) {
        if (!r->mask[i]) continue;
        const double* o = &r->o[3 * i];
        const double* d = &r->d[3 * i];
        double* x = &r->x[3 * i];
        int face = 0, idx = -1;
        if (M->n_coarse_faces > 0) {
            double xc[3];
            if (!mesh_intersect_1(M->c_p0, M->c_edge1, M->c_edge2, M->n_coarse_faces, o, d, xc, &face)) { r->mask[i] = 0; continue; }
            idx = mesh_nearest(M, xc);
            if (!mesh_intersect_2(M, idx, o, d, x, &face)) { r->mask[i] = 0; continue; }
        } else {
            if (!mesh_intersect_1(M->p0, M->edge1, M->edge2, M->n_faces, o, d, x, &face)) { r->mask[i] = 0; continue; }
        }
        if (M->interpolate) {
            /* mesh_interpolate (:172-196): z and the normal from the C1 interpolants at (x, y) */
            double c[3];
            int start = (idx >= 0) ? M->ct_vertex_simplex[idx] : 0;
            if (start < 0) start = 0;
            int sx = ct_find_simplex(M, x, start, c);
            double nn[3];
            if (sx < 0) { x[2] = NAN; nn[0] = nn[1] = nn[2] = NAN; }
            else {
                x[2] = ct_eval(M, sx, c, 0);
                for (int k = 0; k < 3; k++) nn[k] = ct_eval(M, sx, c, 1 + k);
            }
            double inv = 1.0 / norm3(nn);
            for (int k = 0; k < 3; k++) r->nrm[3 * i + k] = inv * nn[k];
        } else {
            for (int k = 0; k < 3; k++) r->nrm[3 * i + k] = M->faces_normal[3 * face + k];
        }
    }
}

/* tools/xicsrt_aperture.py:108-204: single shape test on local coordinates */
static int aperture_shape(const xrt_aperture_t* a, const double* X)
{
    double x = X[0], y = X[1];
    switch (a->shape) {
    case XRT_AP_NONE: return 1;
    case XRT_AP_CIRCLE: {
        double dx = x - a->origin[0], dy = y - a->origin[1];
        return (dx * dx + dy * dy) < a->size[0] * a->size[0]; }
    case XRT_AP_SQUARE:
        return (fabs(x - a->origin[0]) < a->size[0] / 2.0) && (fabs(y - a->origin[1]) < a->size[0] / 2.0);
    case XRT_AP_RECTANGLE:
        return (fabs(x - a->origin[0]) < a->size[0] / 2.0) && (fabs(y - a->origin[1]) < a->size[1] / 2.0);
    case XRT_AP_ELLIPSE: {
        double ex = (x - a->origin[0]) / a->size[0], ey = (y - a->origin[1]) / a->size[1];
        return (ex * ex + ey * ey) < 1.0; }
    case XRT_AP_TRIANGLE: {
        /* tools/xicsrt_math.py:290-306 */
        const double* v = a->vertices;
        double p0x = v[0], p0y = v[1], p1x = v[2], p1y = v[3], p2x = v[4], p2y = v[5];
        double area = 0.5 * (-p1y * p2x + p0y * (-p1x + p2x) + p0x * (p1y - p2y) + p1x * p2y);
        double ia = 1.0 / (2.0 * area);
        double A = ia * (p0y * p2x - p0x * p2y + (p2y - p0y) * x + (p0x - p2x) * y);
        double B = ia * (p0x * p1y - p0y * p1x + (p0y - p1y) * x + (p1x - p0x) * y);
        double Cc = 1.0 - A - B;
        return (A >= 0.0) && (B >= 0.0) && (Cc >= 0.0); }
    }
    return 1;
}

/* optics/_TraceObject.py:180-232, tools/xicsrt_aperture.py:13-47 */
static void check_bounds(const xrt_optic_t* op, rays_t* r)
{
    for (int64_t i = 0; i < r->n; i++) {
        if (!r->mask[i]) continue;
        double* loc = &r->loc[3 * i];
        const double* x = &r->x[3 * i];
        if (op->flags & XRT_F_TRACE_LOCAL) {
            for (int k = 0; k < 3; k++) loc[k] = x[k];
        } else {
            double v[3] = {x[0] - op->origin[0], x[1] - op->origin[1], x[2] - op->origin[2]};
            to_local(op->orientation, v, loc);
        }
        int m = 1;
        if (op->flags & XRT_F_CHECK_SIZE) {
            if ((op->flags & XRT_F_HAS_XSIZE) && !(fabs(loc[0]) < op->half_size[0])) m = 0;
            if ((op->flags & XRT_F_HAS_YSIZE) && !(fabs(loc[1]) < op->half_size[1])) m = 0;
            if ((op->flags & XRT_F_HAS_ZSIZE) && !(fabs(loc[2]) < op->half_size[2])) m = 0;
        }
        if (m && (op->flags & XRT_F_CHECK_APERTURE) && op->n_apertures > 0) {
            int out = 1;
            for (int a = 0; a < op->n_apertures; a++) {
                int t = aperture_shape(&op->apertures[a], loc);
                switch (op->apertures[a].logic) {
                case XRT_LOGIC_AND:  out = out && t; break;
                case XRT_LOGIC_NOT:  out = out && !t; break;
                case XRT_LOGIC_OR:   out = out || t; break;
                case XRT_LOGIC_NAND: out = !(out && t); break;
                case XRT_LOGIC_NOR:  out = !(out || t); break;
                case XRT_LOGIC_XOR:  out = (out != t); break;
                case XRT_LOGIC_XNOR: out = !(out != t); break;
                }
            }
            m = out;
        }
        r->mask[i] = (uint8_t)m;
    }
}

/* optics/_InteractCrystal.py:96-115 angle_calc for one ray */
static inline void angle_calc(const xrt_optic_t* op, const double* d, const double* nn, double wl, double* bragg, double* inc)
{
    double neg[3] = {-1.0 * nn[0], -1.0 * nn[1], -1.0 * nn[2]};
    *bragg = asin(wl / op->two_d);
    *inc = op->half_pi - acos(fabs(dot_e(d, neg)) / norm3(d));
}

/* optics/_InteractCrystal.py:136-196 reflection probability for one ray */
static inline double rocking_p(const xrt_optic_t* op, double inc, double bragg)
{
    double p;
    if (op->rocking_type == XRT_ROCKING_STEP) p = (fabs(inc - bragg) <= op->rocking_half_fwhm) ? 1.0 : 0.0;
    else { double df = inc - bragg; p = exp(-(df * df) / op->rocking_2sigma2); }
    return p * op->reflectivity;
}

/* optics/_InteractMosaicCrystal.py:53-139.  On entry r->mask = rays on the optic (after bounds),
 * r->nrm = nominal normals.  Reflected rays get their new direction here; mask becomes the
 * reflected set. */
/* returns 0 when no ray is left after the cut-off: the reference then skips the whole block,
 * including the O[:] = xloc update of reflect_vectors (:78 `if np.sum(m) > 0`) */
static int interact_mosaic(const xrt_optic_t* op, rays_t* r, mt_t* mt)
{
    const int64_t n = r->n;
    if (op->mosaic_has_cutoff) {
        for (int64_t i = 0; i < n; i++) {
            if (!r->mask[i]) continue;
            double bragg, inc;
            angle_calc(op, &r->d[3 * i], &r->nrm[3 * i], r->wl[i], &bragg, &inc);
            if (!(fabs(bragg - inc) < op->mosaic_cutoff_angle)) r->mask[i] = 0;
        }
    }
    int64_t alive = 0;
    for (int64_t i = 0; i < n; i++) alive += r->mask[i];
    if (alive == 0) return 0;
    uint8_t* refl = calloc((size_t)n, 1);
    double* nm = malloc(sizeof(double) * 3 * (size_t)n);
    const double* A = op->mosaic_A;
    for (int ii = 0; ii < op->mosaic_depth; ii++) {
        int64_t k = 0;
        for (int64_t i = 0; i < n; i++) k += (r->mask[i] && !refl[i]);
        if (k == 0) break;
        /* mosaic_normals (:109-139): 2k standard normals row-major, then the frame about the nominal normal */
        for (int64_t i = 0; i < n; i++) {
            if (!(r->mask[i] && !refl[i])) continue;
            double z0 = mt_gauss(mt), z1 = mt_gauss(mt);
            double x = z0 * A[0] + z1 * A[2], y = z0 * A[1] + z1 * A[3];
            x += 0.0; y += 0.0;
            double o[3] = {x, y, 1.0};
            double inv = 1.0 / norm3(o);
            double l[3] = {o[0] * inv, o[1] * inv, o[2] * inv};
            const double* nn = &r->nrm[3 * i];
            const double ex[3] = {1.0, 0.0, 0.0}, ez[3] = {0.0, 0.0, 1.0};
            double c1[3], c2[3], R0[3], R1[3];
            cross3(nn, ex, c1);
            cross3(nn, ez, c2);
            for (int q = 0; q < 3; q++) R0[q] = c1[q] + c2[q];
            double m0 = norm3(R0);
            for (int q = 0; q < 3; q++) R0[q] /= m0;
            cross3(nn, R0, R1);
            double m1 = norm3(R1);
            for (int q = 0; q < 3; q++) R1[q] /= m1;
            for (int q = 0; q < 3; q++) nm[3 * i + q] = (l[0] * R0[q] + l[1] * R1[q]) + l[2] * nn[q];
        }
        /* angle_check with the crystallite normals, then reflect the accepted rays */
        for (int64_t i = 0; i < n; i++) {
            if (!(r->mask[i] && !refl[i])) continue;
            int ok = 1;
            if (op->flags & XRT_F_CHECK_BRAGG) {
                double bragg, inc;
                angle_calc(op, &r->d[3 * i], &nm[3 * i], r->wl[i], &bragg, &inc);
                double p = rocking_p(op, inc, bragg);
                double test = 0.0 + (1.0 - 0.0) * mt_double(mt);
                ok = (p >= test);
            }
            if (ok) {
                double* d = &r->d[3 * i];
                double dt = dot_e(d, &nm[3 * i]);
                for (int q = 0; q < 3; q++) d[q] = d[q] - 2.0 * (dt * nm[3 * i + q]);
                refl[i] = 1;
            }
        }
    }
    for (int64_t i = 0; i < n; i++) r->mask[i] = r->mask[i] && refl[i];
    free(refl); free(nm);
    return 1;
}

/* optics/_InteractCrystal.py:96-196: Bragg test, draws in original ray order */
static void angle_check(const xrt_optic_t* op, rays_t* r, mt_t* mt)
{
    for (int64_t i = 0; i < r->n; i++) {
        if (!r->mask[i]) continue;
        const double* d = &r->d[3 * i];
        const double* nn = &r->nrm[3 * i];
        double bragg = asin(r->wl[i] / op->two_d);
        double neg[3] = {-1.0 * nn[0], -1.0 * nn[1], -1.0 * nn[2]};
        double dt = fabs(dot_e(d, neg));
        double inc = op->half_pi - acos(dt / norm3(d));
        double p;
        if (op->rocking_type == XRT_ROCKING_STEP) {
            p = (fabs(inc - bragg) <= op->rocking_half_fwhm) ? 1.0 : 0.0;
        } else {
            double df = inc - bragg;
            p = exp(-(df * df) / op->rocking_2sigma2);
        }
        p *= op->reflectivity;
        double test = 0.0 + (1.0 - 0.0) * mt_double(mt);
        if (!(p >= test)) r->mask[i] = 0;
    }
}

/* optics/_TraceObject.py:157-171 + Interact* */
static void trace_optic(const xrt_optic_t* op, rays_t* r, mt_t* mt)
{
    const int local = (op->flags & XRT_F_TRACE_LOCAL) != 0;
    if (local) {    /* ray_to_local (_GeometryObject.py:126-135), all rays */
        for (int64_t i = 0; i < r->n; i++) {
            double v[3] = {r->o[3 * i] - op->origin[0], r->o[3 * i + 1] - op->origin[1], r->o[3 * i + 2] - op->origin[2]};
            double t[3];
            to_local(op->orientation, v, t);
            memcpy(&r->o[3 * i], t, sizeof(t));
            to_local(op->orientation, &r->d[3 * i], t);
            memcpy(&r->d[3 * i], t, sizeof(t));
        }
    }
    memcpy(r->prev, r->mask, (size_t)r->n);
    switch (op->shape) {
    case XRT_SHAPE_PLANE:    intersect_plane(op, r); break;
    case XRT_SHAPE_SPHERE:   intersect_sphere(op, r); break;
    case XRT_SHAPE_CYLINDER: intersect_cylinder(op, r); break;
    case XRT_SHAPE_TORUS:    intersect_torus(op, r); break;
    case XRT_SHAPE_MESH:     intersect_mesh(op, r); break;
    }
    memcpy(r->hit, r->mask, (size_t)r->n);
    check_bounds(op, r);
    if (op->interact == XRT_INTERACT_CRYSTAL && (op->flags & XRT_F_CHECK_BRAGG))
        angle_check(op, r, mt);
    int touched = 1;
    if (op->interact == XRT_INTERACT_MOSAIC)
        touched = interact_mosaic(op, r, mt);
    for (int64_t i = 0; touched && i < r->n; i++) {
        /* InteractObject.interact / InteractMirror.reflect_vectors (_InteractMirror.py:29-42):
         * O[:] = xloc for every ray (NaN where there was no intersection) */
        double* o = &r->o[3 * i];
        double* d = &r->d[3 * i];
        const double* x = &r->x[3 * i];
        const double* nn = &r->nrm[3 * i];
        if (!r->mask[i]) {
            if (r->prev[i]) for (int k = 0; k < 3; k++) o[k] = r->hit[i] ? x[k] : NAN;
            continue;
        }
        for (int k = 0; k < 3; k++) o[k] = x[k];
        if (op->interact != XRT_INTERACT_NONE && op->interact != XRT_INTERACT_MOSAIC) {
            double dt = dot_e(d, nn);
            for (int k = 0; k < 3; k++) d[k] = d[k] - 2.0 * (dt * nn[k]);
        }
    }
    /* Upstream quirk kept for parity: ShapeMesh.intersect returns a fresh mask array and
     * InteractMosaicCrystal.interact never stores its result into rays['mask'] (plane and sphere
     * shapes only appear to, because their intersect aliases rays['mask']); a mesh mosaic crystal
     * therefore leaves every incoming ray "alive", with a NaN or on-surface origin. */
    if (op->interact == XRT_INTERACT_MOSAIC && op->shape == XRT_SHAPE_MESH) memcpy(r->mask, r->prev, (size_t)r->n);
    if (local) {    /* ray_to_external (_GeometryObject.py:113-124) */
        for (int64_t i = 0; i < r->n; i++) {
            double t[3];
            to_external(op->orientation, &r->o[3 * i], t);
            for (int k = 0; k < 3; k++) r->o[3 * i + k] = t[k] + op->origin[k];
            to_external(op->orientation, &r->d[3 * i], t);
            memcpy(&r->d[3 * i], t, sizeof(t));
        }
    }
}

/* optics/_TraceObject.py:234-293 */
static void make_image(const xrt_optic_t* op, const rays_t* r, uint64_t* img)
{
    if (!(op->flags & XRT_F_IMAGE) || !img) return;
    for (int64_t i = 0; i < r->n; i++) {
        if (!r->mask[i]) continue;
        double v[3] = {r->o[3 * i] - op->origin[0], r->o[3 * i + 1] - op->origin[1], r->o[3 * i + 2] - op->origin[2]};
        double loc[3];
        to_local(op->orientation, v, loc);
        double cx = rint(loc[0] / op->pixel_size + op->pixel_xoff);
        double cy = rint(loc[1] / op->pixel_size + op->pixel_yoff);
        if (!(cx >= 0.0 && cx < (double)op->pixel_nx && cy >= 0.0 && cy < (double)op->pixel_ny)) continue;
        img[op->image_offset + (int64_t)cx * op->pixel_ny + (int64_t)cy] += 1;
    }
}

static void save_history(const rays_t* r, int64_t stride, int e, double* hist, uint8_t* hmask, int all_at_source)
{
    const int64_t n = stride;
    for (int64_t i = 0; i < r->n; i++) {
        hmask[(int64_t)e * n + i] = r->mask[i];
        if (!r->mask[i] && !(e > 0 && r->prev[i]) && !(e == 0 && all_at_source)) continue;   /* alive, or died at this element */
        double* h = hist + (int64_t)e * XRT_HIST_COMPONENTS * n;
        for (int k = 0; k < 3; k++) { h[k * n + i] = r->o[3 * i + k]; h[(3 + k) * n + i] = r->d[3 * i + k]; }
        h[6 * n + i] = r->wl[i];
        h[7 * n + i] = r->wt[i];
    }
}

/* xicsrt_raytrace.raytrace_single (xicsrt_raytrace.py:87-175), keep_history=False stream */
static int run_single(const xrt_scene_t* sc, mt_t* mtp, int n_iter,
                      uint64_t* num_out, uint64_t* images, double* hist, uint8_t* hmask)
{
    mt_t mt = *mtp;
    rays_t r;
    if (rays_alloc(&r, sc->source.intensity)) { rays_free(&r); return -2; }
    for (int it = 0; it < n_iter; it++) {
        r.n = sc->source.intensity;
        int64_t produced = generate_rays(&sc->source, &mt, &r);
        if (produced < 0) { rays_free(&r); return -3; }
        const int64_t stride = r.n;
        r.n = produced;
        const int masked_source = sc->source.kind == XRT_SRC_EXTERNAL || sc->source.n_ray_filters > 0;
        if (masked_source) { for (int64_t i = 0; i < r.n; i++) num_out[0] += r.mask[i]; }
        else num_out[0] += (uint64_t)r.n;
        if (hist) save_history(&r, stride, 0, hist, hmask, masked_source);
        for (int e = 0; e < sc->n_optics; e++) {
            const xrt_optic_t* op = &sc->optics[e];
            trace_optic(op, &r, &mt);
            uint64_t c = 0;
            for (int64_t i = 0; i < r.n; i++) c += r.mask[i];
            num_out[e + 1] += c;
            if (hist) save_history(&r, stride, e + 1, hist, hmask, 0);
            make_image(op, &r, images);
        }
    }
    *mtp = mt;
    rays_free(&r);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* exported                                                                   */
/* ------------------------------------------------------------------------ */

int xrt_oracle_abi_version(void) { return XRT_ABI_VERSION; }
size_t xrt_oracle_sizeof_scene(void) { return sizeof(xrt_scene_t); }

/* known-answer hooks for the RNG restatement */
void xrt_oracle_mt_u32(uint32_t seed, uint32_t* out, int64_t n)
{
    mt_t mt; mt_seed(&mt, seed);
    for (int64_t i = 0; i < n; i++) out[i] = mt_u32(&mt);
}
void xrt_oracle_mt_double(uint32_t seed, double* out, int64_t n)
{
    mt_t mt; mt_seed(&mt, seed);
    for (int64_t i = 0; i < n; i++) out[i] = mt_double(&mt);
}
void xrt_oracle_mt_gauss(uint32_t seed, double* out, int64_t n)
{
    mt_t mt; mt_seed(&mt, seed);
    for (int64_t i = 0; i < n; i++) out[i] = mt_gauss(&mt);
}

typedef struct {
    const xrt_scene_t* sc;
    const uint32_t* seeds;
    int n_runs, n_iter, tid, n_threads;
    uint64_t* num_out;
    uint64_t* images;
    int status;
} job_t;

static void* worker(void* p)
{
    job_t* j = (job_t*)p;
    for (int r = j->tid; r < j->n_runs; r += j->n_threads) {
        mt_t mt;
        mt_seed(&mt, j->seeds[r]);
        int st = run_single(j->sc, &mt, j->n_iter, j->num_out, j->images, NULL, NULL);
        if (st) { j->status = st; break; }
    }
    return NULL;
}

/*
 * xicsrt_raytrace.raytrace / xicsrt_multiprocessing.raytrace
 * (xicsrt_raytrace.py:28-84, xicsrt_multiprocessing.py:12-81): runs are
 * independent; `threads` workers take runs round-robin (Pool analogue) with
 * private accumulators that are summed at the end (combine_raytrace :328-356).
 * All pointers are host memory.  num_out has n_optics+1 entries.
 */
int xrt_oracle_trace(const xrt_scene_t* sc, const uint32_t* seeds, int32_t n_runs, int32_t n_iter,
                     uint64_t* num_out, uint64_t* images, int32_t threads)
{
    if (threads < 1) threads = 1;
    if (threads > n_runs) threads = n_runs > 0 ? n_runs : 1;
    const int ne = sc->n_optics + 1;
    const int64_t nb = sc->image_bins;
    job_t* jobs = calloc((size_t)threads, sizeof(job_t));
    pthread_t* th = calloc((size_t)threads, sizeof(pthread_t));
    int status = 0;
    for (int t = 0; t < threads; t++) {
        jobs[t].sc = sc; jobs[t].seeds = seeds; jobs[t].n_runs = n_runs; jobs[t].n_iter = n_iter;
        jobs[t].tid = t; jobs[t].n_threads = threads;
        jobs[t].num_out = calloc((size_t)ne, sizeof(uint64_t));
        jobs[t].images = images ? calloc((size_t)(nb > 0 ? nb : 1), sizeof(uint64_t)) : NULL;
    }
    if (threads == 1) worker(&jobs[0]);
    else {
        for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, worker, &jobs[t]);
        for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    }
    for (int t = 0; t < threads; t++) {
        if (jobs[t].status) status = jobs[t].status;
        for (int e = 0; e < ne; e++) num_out[e] += jobs[t].num_out[e];
        if (images) for (int64_t b = 0; b < nb; b++) images[b] += jobs[t].images[b];
        free(jobs[t].num_out); free(jobs[t].images);
    }
    free(jobs); free(th);
    return status;
}

/* One iteration from an explicit generator state with the per-element history
 * (Dispatcher deepcopy, _Dispatcher.py:162,187); layout as XRT_HIST_COMPONENTS in
 * xicsrt_hip.h.  state_out receives the generator state after the iteration. */
int xrt_oracle_trace_history(const xrt_scene_t* sc, const xrt_rng_state_t* state_in,
                             uint64_t* num_out, uint64_t* images,
                             double* rays, uint8_t* mask, xrt_rng_state_t* state_out)
{
    mt_t mt;
    memcpy(mt.key, state_in->key, sizeof(mt.key));
    mt.pos = state_in->pos; mt.has_gauss = state_in->has_gauss; mt.gauss = state_in->gauss;
    int st = run_single(sc, &mt, 1, num_out, images, rays, mask);
    if (state_out) {
        memcpy(state_out->key, mt.key, sizeof(mt.key));
        state_out->pos = mt.pos; state_out->has_gauss = mt.has_gauss; state_out->gauss = mt.gauss;
    }
    return st;
}
