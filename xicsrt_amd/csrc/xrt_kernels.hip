// xrt_kernels.hip -- photon propagation for gfx950 (MI355X) and the C ABI of
// include/xicsrt_hip.h.
//
// One workgroup (256 threads, 4 waves) owns one *run* = one MT19937 stream
// (reference: xicsrt_raytrace.raytrace_single, xicsrt/xicsrt_raytrace.py:87) and
// walks the run's rays in tiles of 256 in original ray order.  Everything
// between "draw the random numbers" and "increment the detector pixel" happens
// in registers and LDS; HBM only sees the pixel/counter atomics (and the
// optional history snapshot).
//
//   * The reference draws whole arrays per quantity from ONE stream
//     (x, y, z offsets, cone cosine, cone azimuth, wavelength; then, per Bragg
//     optic, one uniform per ray still alive, in ray order).  Array k therefore
//     lives at stream offset 2*k*N words.  Each needed array gets its own
//     MT19937 "head": a 1024-word ring in LDS holding a sliding window of the
//     state sequence s[n] = f(s[n-624], s[n-623], s[n-227]); 227 words are
//     independent at a time, so a head advances by 512 words (256 doubles) in
//     three barrier-separated phases.  Heads are positioned by walking the
//     stream forward once per iteration (no tempering needed on the way).
//   * The per-ray Bragg uniform is indexed by the ordered live rank, obtained
//     from wave64 ballots + a 4-entry LDS scan; the same rank drives a stable
//     compaction of the live rays into an LDS tile so that the next stage runs
//     on dense lanes.
//   * Arithmetic is IEEE binary64 in the reference's evaluation order
//     (-ffp-contract=off; explicit fma() only where the reference's BLAS call
//     fuses), so masks, counts and images are bit-identical to the CPU result.
//
// Reference functions restated per stage are cited at each device function.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
#include <cmath>
#include <vector>
#include <unordered_map>
#include <memory>
#include <new>

#include "../../include/xicsrt_hip.h"

#define XRT_DEV_MAX_OPTICS XRT_MAX_OPTICS      // (64) the scene lives in device memory: no kernel-argument limit
#define XRT_TILE     256
#define XRT_RING     1024u
#define XRT_RMASK    1023u
#define XRT_MAX_HEADS 7         // 6 source arrays + the stream head
#define XRT_AHEAD    512u       // words every head keeps generated ahead of its read position
#define XRT_STRETCH  20560      // 19937 + 623 words: what one jump reads
#define XRT_TILE_COMP 7         // x,y,z, dx,dy,dz, wavelength
// Records of the fused kernel's circular ray buffer: 63 survivors + (batch - 1) queued candidates + 256 new
// ones, for Bragg batches of 128 or 256 candidates (the host takes 256 when that costs no workgroup per CU)
#define XRT_QCAP_128 448u
#define XRT_QCAP_256 576u

// --------------------------------------------------------------------------
// kernel-argument scene (passed by value: uniform, read through scalar loads)
// --------------------------------------------------------------------------

struct KPlasma {             // device copy of xrt_plasma_t (table pointers are device pointers)
    int32_t geometry, has_spread_radius, n_filters, n_emissivity, n_temperature, pad;
    double  torus_origin[3], major_radius, minor_radius;
    double  emissivity, emissivity_scale, temperature_scale;
    const double *emissivity_rho, *emissivity_val, *temperature_rho, *temperature_val;
    double  spread_radius, solid_angle;
    double  time_resolution, bundle_volume, four_pi, volume_ratio;
    double  mass_number, amu_kg, c_squared, ev_J;
    xrt_bundle_filter_t filters[XRT_MAX_BUNDLE_FILTERS];
    double  voigt_gamma, weideman_L;    // per-bundle Voigt profiles (see xrt_plasma_t)
    const double* weideman_a;           // device
    int32_t n_weideman, pad2;
};

struct KSource {
    int32_t kind, angular_dist, wavelength_dist, has_velocity;
    int64_t n_rays;
    double  origin[3];
    double  xaxis[3], yaxis[3], zaxis[3];
    double  low[3], range[3];
    double  axis[3];
    double  basis[9];           // rows o_2, o_1, normal for GENERIC/DIRECTED
    double  ang[5];
    double  two_pi;
    double  wavelength, wl_a, wl_b;
    double  velocity[3];
    double  light_speed;
    const double* voigt_cdf;    // device
    const double* voigt_x;      // device
    int32_t voigt_n;
    int32_t n_arrays;           // arrays of N doubles the source consumes (5 or 6)
    uint32_t array_used;        // bit k: values of array k are needed
    // plasma sources (staged path only)
    int64_t bundle_count;
    double  plasma_low[3], plasma_range[3];
    double  bundle_intensity;
    double  pois[7];            // sqrt(lam), log(lam), b, a, invalpha, vr, exp(-lam) (numpy's own libm expressions)
    int32_t use_poisson, pad2;
    const KPlasma* plasma;      // device; per-bundle model or null
    const double*  ext_rays;    // XRT_SRC_EXTERNAL: [8][n_rays] device
    const uint8_t* ext_mask;
    int32_t n_ray_filters, pad3;        // XicsrtSourceGeneric.ray_filter (staged path)
    xrt_bundle_filter_t ray_filters[XRT_MAX_BUNDLE_FILTERS];
};

// Mesh tables live in global memory and are only ever read: typed pointers (address space 1 / 4) keep the
// compiler from emitting generic (flat) loads; what a wave walks in lockstep goes through scalar loads.
#define XRT_C4 __attribute__((address_space(4)))
#define XRT_G1 __attribute__((address_space(1)))
typedef const XRT_G1 double*  gdp;
typedef const XRT_G1 int32_t* gip;
typedef double  d4v __attribute__((ext_vector_type(4)));
typedef double  d2v __attribute__((ext_vector_type(2)));
typedef int32_t i4v __attribute__((ext_vector_type(4)));
struct KCellRec { double x, y, z; int32_t idx, next; };      // a point of an x-y bucket; next: chain inside the bucket, -1 ends it
struct KFaceRec { double p0[3], p1[3], p2[3], n[3], area, pad[3]; };   // what mesh_intersect_2 reads of a face, 128 B

struct KMesh {               // device pointers, see xrt_mesh_t
    int32_t n_points, n_faces, n_coarse_faces, interpolate, n_simplices, n_first;
    gdp faces_normal;
    const double* first_rec;        // [n_first + 1][10] p0, edge1, edge2 of the faces mesh_intersect_1 walks (coarse faces, or all)
    const double* plane_rec;        // [n_first + 1][16] plane form of the same faces: n, n.p0, U, u0, V, v0, n.n (+inf: degenerate)
    const XRT_G1 d4v* face_rec;     // [n_faces] KFaceRec, read as four 32-byte vectors
    gip point_faces;                // [n_points][8] the faces around a point, -1 where the reference's mask is False
    // the Clough-Tocher tables, one cache line per simplex / vertex and use:
    const XRT_G1 d4v* ct_srec;      // [n_simplices][16] the three vertices, the edge vectors e12, e23, e31, the neighbour weights g (ct_shared)
    gdp ct_frec;                    // [n_simplices][8] barycentric transform (6) and the three neighbours (ct_find_simplex)
    gdp ct_vrec;                    // [n_points][16] value, d/dx, d/dy of z, normal_x, normal_y, normal_z at a vertex (ct_eval)
    gip face_simplex;               // [n_faces] the simplex with a face's three vertices, -1: none (where the walk starts; null: not built)
    gip ct_vertex_simplex;
    // x-y bucket grid over the points for the exact nearest-point search (built by the library)
    int32_t grid_nx, grid_ny;
    double  grid_x0, grid_y0, grid_hx, grid_hy, grid_ihx, grid_ihy, grid_tiny;
    gdp points;                     // [n_points][3] (search without a grid)
    const XRT_G1 d4v* cells;        // KCellRec [nx*ny] first point of every bucket (idx -1: empty), then the chained ones
    // x-y grid over the faces of the first pass (built by the library; fg_n = 0: none, every face is walked).  A ray
    // crosses the slab fg_zlo .. fg_zhi that holds all those faces along a short stretch; only the faces of the cells
    // under that stretch can be hit.  fg_nbar / fg_sin: every face normal lies within a cone about nbar, and a ray whose
    // direction is at least fg_sin (sine) away from the plane normal to nbar is not nearly parallel to any face.
    int32_t fg_n, fg_pad;
    double  fg_x0, fg_y0, fg_ihx, fg_ihy, fg_zlo, fg_zhi, fg_margin, fg_nbar[3], fg_sin;
    gip fg_start, fg_faces;         // [fg_n * fg_n + 1] first entry of a cell's face list, [..] the lists (ascending face index)
    gdp fg_zcell;                   // [fg_n * fg_n][2] the slab of z that holds the faces of a cell (margin included)
    // The first optic behind a point source sees every ray leave from ONE point O: with m = e2 x e1, E02 = (p0-O) x (p2-O),
    // E10 = (p1-O) x (p0-O) the Moller-Trumbore quantities are f = d.m, u f = d.E02, v f = d.E10 -- three dot products per face.
    const double* pt_rec;           // [n_first + 1][12] m, E02, E10, |m|^2 (+inf: degenerate), 2 spare; or null
    // ... and a grid over the directions from O (dg_n = 0: none): with an orthonormal frame ex, ey, ez about the mean
    // direction to the faces, a ray of direction d lands at (d.ex / d.ez, d.ey / d.ez); a cell of the dg_n x dg_n grid
    // over the faces' images lists the <= 8 faces whose image (with a margin) touches it, ascending; 0xff: no
    // further face, a first byte of 0xfe: more than eight (the ray walks every face).  The first phase of a split launch
    // keeps the cells and pt_rec in LDS.
    int32_t dg_n, dg_big;           // dg_big: 254 faces or more -- 16-bit face numbers (four words per cell, 0xffff / 0xfffe), tables read from global memory
    double  dg_x0, dg_y0, dg_ihx, dg_ihy, dg_ex[3], dg_ey[3], dg_ez[3];
    const uint32_t* dg_cells;       // [dg_n * dg_n][2 or 4]: eight face numbers
    // plane form of EVERY face (as plane_rec) for the second pass around the nearest point; [13]: how far outside (in
    // barycentric units) a point must lie for the reference's area test (diff < 1e-10) to fail for sure
    const XRT_G1 d4v* plane2_rec;   // [n_faces][16] or null (no second pass)
    // A small mesh in LDS (xrt_mesh_rest_lds_kernel; lds_bytes = 0: does not fit / not applicable): the cells above,
    // per point its <= 8 faces and per face the cells that hold its three vertices (p0, p1, p2), as 16-bit numbers
    const uint16_t* lds_pf;         // [n_points][8], 0xffff: none
    const uint16_t* lds_fv;         // [n_faces][4] (the fourth is padding)
    int32_t n_cells, lds_bytes;
    // The faces around a point as a fan (xrt_mesh_star_lds_kernel; null: not built).  Per point 24 16-bit numbers: [0] the
    // entries of the fan (0: this point's rays take the list walk), [1..9] the cells of its spokes -- the other vertices of
    // its faces, counter-clockwise in x-y, the first one again at the end when the fan is closed --, [10..17] the face
    // between entry t and t + 1 (0xffff: none), [18] two bits per such face: which of (point, entry t, entry t + 1) is the
    // face's own first vertex, [19] the point's own cell, [20..23] two floats: the slopes dz/dx, dz/dy of the fan's mean
    // plane.  star_K, star_c2: see mesh_rest_star_lds.
    const uint16_t* lds_star;
    int32_t star_lds_bytes, star_pad;
    double  star_K, star_c2;
};

struct KOptic {
    int32_t shape, interact, flags, rocking_type;
    double  origin[3];
    double  R[9];
    double  half_size[3];
    double  radius, radius2;
    double  center[3];
    double  torus_major;
    double  torus_k[5];
    int32_t torus_root, pad0;
    double  two_d, reflectivity, half_fwhm, two_sigma2, half_pi;
    double  mosaic_cutoff_angle;
    double  mosaic_A[4];
    int32_t mosaic_depth, mosaic_has_cutoff;
    double  pixel_size, pixel_xoff, pixel_yoff;
    int32_t pixel_nx, pixel_ny;
    int64_t image_offset;
    const xrt_aperture_t* apertures;    // device
    const KMesh* mesh;                  // device (XRT_SHAPE_MESH)
    int32_t n_apertures;
    int32_t scr_ok;                     // rocking-curve screen usable (one wavelength for all rays), see bragg_accept
    double  scr_s, scr_a1, scr_a2, scr_a3;      // sin(bragg) and the series of asin(s + d) - asin(s) in d
    double  scr_binv, scr_dmax, scr_ptail;      // 1 / (2 sigma^2), validity radius in d, bound on p outside it
    // screen for rays with their own wavelength: inc - bragg from sin(inc - bragg) = c sqrt(1-s^2) - s sqrt(1-c^2)
    double  scr2_tail, inv_two_d;               // bound on p for |sin(inc - bragg)| >= 0.01; 1 / (2 d)
    int32_t scr2_ok;
    int32_t mesh_lds_bytes;             // host side: KMesh.lds_bytes of `mesh` (which launch finishes a split mesh intersection)
    int32_t mesh_dir_bytes;             // host side: LDS bytes of the mesh's direction grid (KMesh.dg_n), 0: none
    int32_t mesh_ct_lds_bytes;          // host side: bytes of the mesh's Clough-Tocher vertex table when it fits the LDS (xrt_mesh_ct_lds_kernel), else 0
    int32_t mesh_star_lds_bytes;        // host side: KMesh.star_lds_bytes of `mesh` (0: no fans, the list walk for every parked ray; < 0: fans, read from global memory)
    int32_t pad_star;
};

struct KScene {
    KSource src;
    int32_t n_optics;
    int32_t pad;
    KOptic  opt[XRT_DEV_MAX_OPTICS];
};

struct KState {             // xrt_rng_state_t layout
    uint32_t key[624];
    int32_t  pos;
    int32_t  has_gauss;
    double   gauss;
};

// A positioned generator: sliding window of the MT19937 state sequence in a
// 1024-word ring (slot = index & 1023) holding s[gen-1024, gen); `next` is the
// index of the next word to hand out.
struct KStream {
    uint32_t ring[1024];
    uint64_t gen;
    uint64_t next;
    uint64_t pad[2];
};

// --------------------------------------------------------------------------
// MT19937 (numpy legacy RandomState)
// --------------------------------------------------------------------------

__device__ __forceinline__ uint32_t mt_mix(uint32_t a, uint32_t b, uint32_t c)
{
    uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return c ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

__device__ __forceinline__ uint32_t mt_temper(uint32_t y)
{
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

// genrand_res53: (a*2^26 + b) / 2^53, exact in binary64
__device__ __forceinline__ double mt_double(uint32_t w0, uint32_t w1)
{
    uint32_t a = mt_temper(w0) >> 5, b = mt_temper(w1) >> 6;
    return ((double)a * 67108864.0 + (double)b) * (1.0 / 9007199254740992.0);
}

// Values that are the same in every lane but come from memory: tell the compiler
// (scalar registers, scalar branches) with readfirstlane.
__device__ __forceinline__ uint32_t uni32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uni64(uint64_t v)
{
    uint32_t lo = uni32((uint32_t)v), hi = uni32((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// Advance ring `r` (holding s[.. gen)) by `count` words, all threads of the
// workgroup cooperating; at most 227 words are independent.  Ends with a barrier.
__device__ __forceinline__ void ring_advance(uint32_t* r, uint64_t gen, uint32_t count, int tid)
{
    for (uint32_t done = 0; done < count; done += 227u) {
        uint32_t chunk = count - done;
        if (chunk > 227u) chunk = 227u;
        if ((uint32_t)tid < chunk) {
            uint32_t n = (uint32_t)gen + done + (uint32_t)tid;
            r[n & XRT_RMASK] = mt_mix(r[(n - 624u) & XRT_RMASK], r[(n - 623u) & XRT_RMASK],
                                      r[(n - 227u) & XRT_RMASK]);
        }
        __syncthreads();
    }
}

// --------------------------------------------------------------------------
// 3-vector helpers in NumPy's evaluation order (measured on numpy 2.2.6)
// --------------------------------------------------------------------------

struct V3 { double x, y, z; };

__device__ __forceinline__ double dot_e(const V3& a, const V3& b) { return (a.x * b.x + a.z * b.z) + a.y * b.y; }
__device__ __forceinline__ double dot_n(const V3& a, const V3& b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ double dot_blas(const V3& a, const V3& b) { return fma(a.z, b.z, fma(a.x, b.x, a.y * b.y)); }
// Correctly rounded square root.  The library routine first rescales arguments below 2^-767 and picks the
// argument itself for 0 / inf at the end (eight instructions that never act on lengths and discriminants of
// order one); arguments outside [2^-700, 2^700], negative ones and NaN take the library path, the others the
// same v_rsq_f64 + refinement steps without the rescaling.
__device__ __forceinline__ double sqrt_rn(double x)
{
    const uint32_t hi = (uint32_t)__double2hiint(x);
    if (!(hi - 0x14300000u < 0x57800000u)) return sqrt(x);      // exponent field in [0x143, 0x6bb), sign clear
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * 0.5;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    double d = fma(-g, g, x);
    g = fma(d, h, g);
    d = fma(-g, g, x);
    return fma(d, h, g);
}
__device__ __forceinline__ double norm3(const V3& a) { return sqrt_rn(dot_n(a, a)); }

// v / m for the three components of v, each the correctly rounded IEEE quotient: the hardware division
// sequence (v_rcp_f64, two Newton steps on the reciprocal, quotient, one residual correction, v_div_fixup for
// zeros / infinities / NaNs) with the reciprocal shared by the three numerators.  The library's per-division
// v_div_scale only rescales operands near the ends of the exponent range: m outside [2^-511, 2^512) takes the
// plain divisions; the numerators here are components of the vector whose length m is (|v_i| <= m, and sums and
// differences of O(1) coordinates are never within 2^-500 of zero without being zero).
// xrt_selftest_div3 holds this to the / operator on the GPU, tiny and huge operands included.
__device__ __forceinline__ V3 div3_rn(const V3& v, double m)
{
    V3 q;
    if (!(((uint32_t)__double2hiint(m) & 0x7fffffffu) - 0x20000000u < 0x40000000u)) {
        q.x = v.x / m; q.y = v.y / m; q.z = v.z / m;
        return q;
    }
    double r = __builtin_amdgcn_rcp(m);
    double e = fma(-m, r, 1.0);
    r = fma(r, e, r);
    e = fma(-m, r, 1.0);
    r = fma(r, e, r);
    double t = v.x * r; t = fma(fma(-m, t, v.x), r, t); q.x = __builtin_amdgcn_div_fixup(t, m, v.x);
    t = v.y * r; t = fma(fma(-m, t, v.y), r, t); q.y = __builtin_amdgcn_div_fixup(t, m, v.y);
    t = v.z * r; t = fma(fma(-m, t, v.z), r, t); q.z = __builtin_amdgcn_div_fixup(t, m, v.z);
    return q;
}
__device__ __forceinline__ V3 cross3(const V3& a, const V3& b)
{
    V3 c;
    c.x = a.y * b.z - a.z * b.y;
    c.y = a.z * b.x - a.x * b.z;
    c.z = a.x * b.y - a.y * b.x;
    return c;
}
__device__ __forceinline__ V3 ld3(const double* p) { V3 v; v.x = p[0]; v.y = p[1]; v.z = p[2]; return v; }
__device__ __forceinline__ V3 sub3(const V3& a, const V3& b) { V3 c; c.x = a.x - b.x; c.y = a.y - b.y; c.z = a.z - b.z; return c; }
// GeometryObject.vector_to_local (xicsrt/objects/_GeometryObject.py:157-168)
__device__ __forceinline__ V3 to_local(const double* R, const V3& v)
{
    V3 o;
    o.x = dot_e(ld3(R + 0), v);
    o.y = dot_e(ld3(R + 3), v);
    o.z = dot_e(ld3(R + 6), v);
    return o;
}

// GeometryObject.vector_to_external: einsum('ij,ki->kj') (xicsrt/objects/_GeometryObject.py:143-155)
__device__ __forceinline__ V3 to_external(const double* R, const V3& v)
{
    V3 o;
    o.x = (R[0] * v.x + R[3] * v.y) + R[6] * v.z;
    o.y = (R[1] * v.x + R[4] * v.y) + R[7] * v.z;
    o.z = (R[2] * v.x + R[5] * v.y) + R[8] * v.z;
    return o;
}

// np.interp (numpy/core/src/multiarray/compiled_base.c arr_interp)
__device__ double np_interp(double x, const double* xp, const double* fp, int n)
{
    if (x != x) return x;
    if (x < xp[0]) return fp[0];
    if (x > xp[n - 1]) return fp[n - 1];
    int lo = 0, hi = n - 1;
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

// sin and cos of x in [0, 2*pi] (the cone azimuth).  Quadrant reduction with the
// two-piece split of pi/2 and the minimax kernels of Sun's fdlibm (k_sin.c, k_cos.c,
// e_rem_pio2.c medium path; < 1 ulp), evaluated with fused multiply-adds.  The
// general library routine spends most of its instructions on ranges this path
// never sees.
__device__ __forceinline__ void sincos_0_2pi(double x, double* sn, double* cs)
{
    const double invpio2 = 6.36619772367581382433e-01;
    const double pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11;
    const double fn = rint(x * invpio2);
    const int n = (int)fn;
    const double r = fma(-fn, pio2_1, x);       // exact: pio2_1 has 33 significant bits
    const double w = fn * pio2_1t;
    const double y0 = r - w;
    const double y1 = (r - y0) - w;
    const double z = y0 * y0;
    // k_sin
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                 S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double v = z * y0;
    const double rs = fma(z, fma(z, fma(z, fma(z, S6, S5), S4), S3), S2);
    const double sv = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * S1);
    // k_cos
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                 C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double rc = z * fma(z, fma(z, fma(z, fma(z, fma(z, C6, C5), C4), C3), C2), C1);
    const int ix = __double2hiint(y0) & 0x7fffffff;
    double qx = (ix > 0x3fe90000) ? 0.28125 : __hiloint2double(ix - 0x00200000, 0);
    if (ix < 0x3fd33333) qx = 0.0;
    const double hz = 0.5 * z - qx;
    const double a = 1.0 - qx;
    const double cv = a - (hz - (z * rc - y0 * y1));
    const double so = (n & 1) ? cv : sv;
    const double co = (n & 1) ? sv : cv;
    *sn = (n & 2) ? -so : so;
    *cs = ((n + 1) & 2) ? -co : co;
}

// --------------------------------------------------------------------------
// source: XicsrtSourceGeneric.generate_rays (sources/_XicsrtSourceGeneric.py:198-227)
// --------------------------------------------------------------------------

struct Ray {
    V3 o, d;
    double wl;
};

// u[k] = the k-th array's double for this ray (0.0 for arrays whose value cannot matter)
// generate_origin (:229-255): origin + x*xaxis + y*yaxis + z*zaxis, left to right
__device__ __forceinline__ V3 source_origin(const KSource& s, double u0, double u1, double u2)
{
    V3 o;
    double xo = s.low[0] + s.range[0] * u0;
    double yo = s.low[1] + s.range[1] * u1;
    double zo = s.low[2] + s.range[2] * u2;
    o.x = ((s.origin[0] + xo * s.xaxis[0]) + yo * s.yaxis[0]) + zo * s.zaxis[0];
    o.y = ((s.origin[1] + xo * s.xaxis[1]) + yo * s.yaxis[1]) + zo * s.zaxis[1];
    o.z = ((s.origin[2] + xo * s.xaxis[2]) + yo * s.yaxis[2]) + zo * s.zaxis[2];
    return o;
}

// `have_origin`: ray.o is already set (a point source: one origin per run)
template <bool FULL>
__device__ __forceinline__ void source_ray(const KSource& s, const double* u, Ray& ray, bool have_origin = false)
{
    if (!have_origin) ray.o = source_origin(s, u[0], u[1], u[2]);

    // make_normal + basis (:262-285; Directed :46-50; Focused :40-44)
    V3 n, o1, o2;
    if (FULL && s.kind == XRT_SRC_FOCUSED) {
        V3 a = sub3(ld3(s.axis), ray.o);
        n = div3_rn(a, norm3(a));
        V3 c1 = cross3(n, ld3(s.xaxis)), c2 = cross3(n, ld3(s.zaxis));
        o1.x = c1.x + c2.x; o1.y = c1.y + c2.y; o1.z = c1.z + c2.z;
        o1 = div3_rn(o1, norm3(o1));
        o2 = cross3(n, o1);
        o2 = div3_rn(o2, norm3(o2));
    } else {
        o2 = ld3(s.basis + 0); o1 = ld3(s.basis + 3); n = ld3(s.basis + 6);
    }

    // local direction about +z (tools/xicsrt_spread.py:80-294)
    double l0, l1, l2;
    if (!FULL || s.angular_dist == XRT_ANG_ISOTROPIC) {
        double z = s.ang[0] + (1.0 - s.ang[0]) * u[3];
        double phi = 0.0 + (s.two_pi - 0.0) * u[4];
        double st = sqrt_rn(1.0 - z * z);
        double sn, cs;
        sincos_0_2pi(phi, &sn, &cs);
        l0 = st * cs; l1 = st * sn; l2 = z;
    } else if (s.angular_dist == XRT_ANG_FLAT) {
        double r = sqrt(0.0 + (s.ang[0] - 0.0) * u[3]);
        double a1 = 0.0 + (s.two_pi - 0.0) * u[4];
        double a0 = atan(r);
        double s1, c1, s0, c0;
        sincos(a1, &s1, &c1);
        sincos(a0, &s0, &c0);
        l0 = c1 * s0; l1 = s1 * s0; l2 = c0;
    } else {    // XRT_ANG_FLAT_XY
        double x = s.ang[0] + (s.ang[1] - s.ang[0]) * u[3];
        double y = s.ang[2] + (s.ang[3] - s.ang[2]) * u[4];
        double a0 = atan(sqrt(x * x + y * y));
        double a1 = atan2(y, x);
        double s1, c1, s0, c0;
        sincos(a1, &s1, &c1);
        sincos(a0, &s0, &c0);
        l0 = c1 * s0; l1 = s1 * s0; l2 = c0;
    }
    // einsum('ij,ijk->ik', dir_local, [o_2, o_1, normal]) (:287-292)
    ray.d.x = (l0 * o2.x + l1 * o1.x) + l2 * n.x;
    ray.d.y = (l0 * o2.y + l1 * o1.y) + l2 * n.y;
    ray.d.z = (l0 * o2.z + l1 * o1.z) + l2 * n.z;

    // generate_wavelength (:295-319)
    double wl;
    if (s.wavelength_dist == XRT_WL_UNIFORM) {
        wl = s.wl_a + s.wl_b * u[5];
    } else if (FULL && s.wavelength_dist == XRT_WL_VOIGT) {
        double y = s.wl_a + s.wl_b * u[5];
        wl = np_interp(y, s.voigt_cdf, s.voigt_x, s.voigt_n) + s.wavelength;
    } else if (FULL && s.wavelength_dist == XRT_WL_NORMAL) {
        wl = u[5];                      // np.random.normal(wavelength, sigma) value, prepared per ray
    } else {
        wl = 1.0 * s.wavelength;
    }
    if (s.has_velocity) {
        double v = dot_e(ld3(s.velocity), ray.d);
        wl *= 1.0 - (v / s.light_speed);
    }
    ray.wl = wl;
}

// --------------------------------------------------------------------------
// optics
// --------------------------------------------------------------------------

// tools/xicsrt_quartic.py:54-160 multi_cubic(1, b0, c0, d0, all_roots=False): one real root
__device__ double cubic_one_root(double a, double b, double c)
{
    const double third = 1. / 3.;
    double a13 = a * third;
    double a2 = a13 * a13;
    double f = third * b - a2;
    double g = a13 * (2 * a2 - b) + c;
    double h = 0.25 * g * g + f * f * f;
    if (f == 0 && g == 0 && h == 0) {
        double cr = (c >= 0) ? pow(c, third) : -pow(-c, third);
        return -cr;
    }
    if (h <= 0) {
        double j = sqrt(-f);
        double k = acos(-0.5 * g / (j * j * j));
        double m = cos(third * k);
        return 2 * j * m - a13;
    }
    double sqrt_h = sqrt(h);
    double x1 = -0.5 * g + sqrt_h, x2 = -0.5 * g - sqrt_h;
    double S = (x1 >= 0) ? pow(x1, third) : -pow(-x1, third);
    double U = (x2 >= 0) ? pow(x2, third) : -pow(-x2, third);
    return (S + U) - a13;
}

// tools/xicsrt_quartic.py:162-207 multi_quartic (a0 = 1) in real arithmetic; the reference's
// complex128 roots are kept only when their imaginary part is exactly zero
// (optics/_ShapeTorus.py:164-167), i.e. when s = sqrt(2p + 2 z0) is real and the
// quadratic's discriminant is >= 0.  Division of a real by the real s is numpy's
// Smith-form complex division: x * (1.0 / s).  Returns root number `which`.
__device__ double quartic_root(double b0, double c0, double d0, double e0, int which)
{
    double a = b0, b = c0, c = d0, d = e0;
    double a0 = 0.25 * a;
    double a02 = a0 * a0;
    double p = 3 * a02 - 0.5 * b;
    double q = a * a02 - b * a0 + 0.5 * c;
    double r = 3 * a02 * a02 - b * a02 + c * a0 - d;
    double z0 = cubic_one_root(p, r, p * r - 0.5 * q * q);
    double sarg = 2 * p + 2 * z0;
    if (!(sarg >= 0.0)) return __builtin_nan("");
    double s = sqrt(sarg);
    double t = (s == 0.0) ? (z0 * z0 + r) : (-q) * (1.0 / s);
    double h0, delta;
    if (which < 2) { h0 = -0.5 * s;    delta = h0 * h0 - (z0 + t); }
    else           { h0 = -0.5 * (-s); delta = h0 * h0 - (z0 - t); }
    if (!(delta >= 0.0)) return __builtin_nan("");
    double sd = sqrt(delta);
    return ((which & 1) ? (h0 + sd) : (h0 - sd)) - a0;
}

// Shape*.intersect_distance + location_from_distance: the intersection point.
// Returns false when the ray has no intersection (mask &= ... in the reference).
// What a point source's first element needs of the (single) ray origin, evaluated once per run with the
// same operations the per-ray code uses: sphere: L = centre - O and L.L; plane: (origin - O) . zaxis.
struct PointPre { V3 L; double LL, num; };

__device__ __forceinline__ PointPre point_pre(const KOptic& op, const V3& O)
{
    PointPre p;
    p.L = sub3(ld3(op.center), O);
    p.LL = dot_e(p.L, p.L);
    p.num = dot_blas(sub3(ld3(op.origin), O), ld3(op.R + 6));
    return p;
}

template <bool FULL>
__device__ __forceinline__ bool intersect_point(const KOptic& op, const Ray& ray, V3& X, bool pre, const PointPre& pp)
{
    double t;
    if (op.shape == XRT_SHAPE_PLANE) {
        // optics/_ShapePlane.py:32-52 (np.dot -> OpenBLAS dgemv fused order)
        if (FULL && (op.flags & XRT_F_TRACE_LOCAL)) {
            V3 ez; ez.x = 0.0; ez.y = 0.0; ez.z = 1.0;
            V3 v; v.x = 0.0 - ray.o.x; v.y = 0.0 - ray.o.y; v.z = 0.0 - ray.o.z;
            t = dot_blas(v, ez) / dot_blas(ray.d, ez);
        } else {
            V3 za = ld3(op.R + 6);
            const double num = pre ? pp.num : dot_blas(sub3(ld3(op.origin), ray.o), za);
            t = num / dot_blas(ray.d, za);
        }
        if (!(t >= 0.0)) return false;
    } else if (!FULL || op.shape == XRT_SHAPE_SPHERE) {
        // optics/_ShapeSphere.py:52-100
        V3 L = pre ? pp.L : sub3(ld3(op.center), ray.o);
        double t_ca = dot_e(L, ray.d);
        double dd = sqrt_rn((pre ? pp.LL : dot_e(L, L)) - t_ca * t_ca);
        if (!(dd <= op.radius)) return false;
        double t_hc = sqrt_rn(op.radius2 - dd * dd);
        double t0 = t_ca - t_hc, t1 = t_ca + t_hc;
        if (op.flags & XRT_F_CONVEX) t = (t0 < t1) ? t0 : t1;
        else                         t = (t0 > t1) ? t0 : t1;
    } else if (op.shape == XRT_SHAPE_TORUS) {
        // optics/_ShapeTorus.py:110-183: quartic in the torus frame (axis = local y)
        V3 O = to_local(op.R, sub3(ray.o, ld3(op.center)));
        V3 D = to_local(op.R, ray.d);
        double O_mag_sq = dot_e(O, O), dot_OD = dot_e(O, D);
        const double r_sq = op.torus_k[0], two_r_sq = op.torus_k[1], four_R2 = op.torus_k[2],
                     eight_R2 = op.torus_k[3], K = op.torus_k[4];
        double c1 = 4.0 * dot_OD;
        double c2 = ((4.0 * (dot_OD * dot_OD) + 2.0 * O_mag_sq) - two_r_sq) + four_R2 * (D.y * D.y);
        double c3 = (4.0 * dot_OD) * (O_mag_sq - r_sq) + (eight_R2 * D.y) * O.y;
        double c4 = ((O_mag_sq * O_mag_sq - two_r_sq * O_mag_sq) + four_R2 * (O.y * O.y)) + K;
        t = quartic_root(c1, c2, c3, c4, op.torus_root);
        if (!(isfinite(t) && t > 0.0)) return false;
    } else {
        // optics/_ShapeCylinder.py:52-110
        V3 pa = ld3(op.center), va = ld3(op.R + 0);
        V3 dp = sub3(ray.o, pa);
        double dDva = dot_e(ray.d, va), dpva = dot_e(dp, va);
        V3 A1, B1;
        A1.x = ray.d.x - dDva * va.x; A1.y = ray.d.y - dDva * va.y; A1.z = ray.d.z - dDva * va.z;
        B1.x = dp.x - dpva * va.x;    B1.y = dp.y - dpva * va.y;    B1.z = dp.z - dpva * va.z;
        double A = dot_e(A1, A1);
        double B = 2.0 * dot_e(A1, B1);
        double C = dot_e(B1, B1) - op.radius2;
        double dis = B * B - 4.0 * A * C;
        if (!(dis >= 0.0)) return false;
        double sq = sqrt(dis);
        double t0 = (-B - sq) / (2.0 * A), t1 = (-B + sq) / (2.0 * A);
        if (op.flags & XRT_F_CONVEX) t = (t0 < t1) ? t0 : t1;
        else                         t = (t0 > t1) ? t0 : t1;
    }
    // ShapeObject.location_from_distance (optics/_ShapeObject.py:79)
    X.x = ray.o.x + ray.d.x * t; X.y = ray.o.y + ray.d.y * t; X.z = ray.o.z + ray.d.z * t;
    return true;
}

// Shape*.intersect_normal at an intersection point (only needed for rays that
// reflect, so it runs after the bounds test / compaction)
template <bool FULL>
__device__ __forceinline__ V3 surface_normal(const KOptic& op, const V3& X)
{
    V3 nrm;
    if (op.shape == XRT_SHAPE_PLANE) {
        nrm = ld3(op.R + 6);                        // optics/_ShapePlane.py:56-62
    } else if (!FULL || op.shape == XRT_SHAPE_SPHERE) {
        V3 q = sub3(ld3(op.center), X);             // optics/_ShapeSphere.py:102-106
        nrm = div3_rn(q, norm3(q));
    } else if (op.shape == XRT_SHAPE_TORUS) {
        // optics/_ShapeTorus.py:186-216: away from the nearest point of the major circle
        V3 C = ld3(op.center), ya = ld3(op.R + 3);
        V3 pt = sub3(X, C);
        double dy = dot_e(pt, ya);
        pt.x = pt.x - dy * ya.x; pt.y = pt.y - dy * ya.y; pt.z = pt.z - dy * ya.z;
        double m = norm3(pt);
        V3 Q;
        Q.x = C.x + op.torus_major * (pt.x / m); Q.y = C.y + op.torus_major * (pt.y / m); Q.z = C.z + op.torus_major * (pt.z / m);
        V3 xn = sub3(X, Q);
        double m2 = norm3(xn);
        nrm.x = xn.x / m2; nrm.y = xn.y / m2; nrm.z = xn.z / m2;
    } else {
        V3 pa = ld3(op.center), va = ld3(op.R + 0); // optics/_ShapeCylinder.py:112-133
        V3 q = sub3(pa, X);
        double dummy = dot_e(q, va);
        V3 c;
        c.x = (pa.x - dummy * va.x) - X.x; c.y = (pa.y - dummy * va.y) - X.y; c.z = (pa.z - dummy * va.z) - X.z;
        double m = norm3(c);
        nrm.x = c.x / m; nrm.y = c.y / m; nrm.z = c.z / m;
    }
    return nrm;
}

#include "xrt_mesh.inc"

// tools/xicsrt_aperture.py:108-204
__device__ bool aperture_shape(const xrt_aperture_t& a, double x, double y)
{
    switch (a.shape) {
    case XRT_AP_CIRCLE: {
        double dx = x - a.origin[0], dy = y - a.origin[1];
        return (dx * dx + dy * dy) < a.size[0] * a.size[0]; }
    case XRT_AP_SQUARE:
        return (fabs(x - a.origin[0]) < a.size[0] / 2.0) && (fabs(y - a.origin[1]) < a.size[0] / 2.0);
    case XRT_AP_RECTANGLE:
        return (fabs(x - a.origin[0]) < a.size[0] / 2.0) && (fabs(y - a.origin[1]) < a.size[1] / 2.0);
    case XRT_AP_ELLIPSE: {
        double ex = (x - a.origin[0]) / a.size[0], ey = (y - a.origin[1]) / a.size[1];
        return (ex * ex + ey * ey) < 1.0; }
    case XRT_AP_TRIANGLE: {
        const double* v = a.vertices;
        double p0x = v[0], p0y = v[1], p1x = v[2], p1y = v[3], p2x = v[4], p2y = v[5];
        double area = 0.5 * (-p1y * p2x + p0y * (-p1x + p2x) + p0x * (p1y - p2y) + p1x * p2y);
        double ia = 1.0 / (2.0 * area);
        double A = ia * (p0y * p2x - p0x * p2y + (p2y - p0y) * x + (p0x - p2x) * y);
        double B = ia * (p0x * p1y - p0y * p1x + (p0y - p1y) * x + (p1x - p0x) * y);
        double Cc = 1.0 - A - B;
        return (A >= 0.0) && (B >= 0.0) && (Cc >= 0.0); }
    default: return true;
    }
}

// TraceObject.check_bounds (optics/_TraceObject.py:180-232, tools/xicsrt_aperture.py:13-47)
template <bool FULL>
__device__ __forceinline__ bool check_bounds(const KOptic& op, const V3& X)
{
    V3 loc = (FULL && (op.flags & XRT_F_TRACE_LOCAL)) ? X : to_local(op.R, sub3(X, ld3(op.origin)));
    bool m = true;
    if (op.flags & XRT_F_CHECK_SIZE) {
        if ((op.flags & XRT_F_HAS_XSIZE) && !(fabs(loc.x) < op.half_size[0])) m = false;
        if ((op.flags & XRT_F_HAS_YSIZE) && !(fabs(loc.y) < op.half_size[1])) m = false;
        if ((op.flags & XRT_F_HAS_ZSIZE) && !(fabs(loc.z) < op.half_size[2])) m = false;
    }
    if (FULL && m && (op.flags & XRT_F_CHECK_APERTURE) && op.n_apertures > 0) {
        bool out = true;
        for (int a = 0; a < op.n_apertures; a++) {
            const xrt_aperture_t ap = op.apertures[a];
            bool t = aperture_shape(ap, loc.x, loc.y);
            switch (ap.logic) {
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
    return m;
}

// InteractCrystal.angle_calc + rocking_curve_filter (optics/_InteractCrystal.py:96-196)
// angle_calc (optics/_InteractCrystal.py:96-115): |bragg - incident| for the mosaic cutoff
__device__ __forceinline__ double bragg_offset(const KOptic& op, const Ray& ray, const V3& nrm)
{
    double bragg = asin(ray.wl / op.two_d);
    V3 neg; neg.x = -1.0 * nrm.x; neg.y = -1.0 * nrm.y; neg.z = -1.0 * nrm.z;
    double dt = fabs(dot_e(ray.d, neg));
    double inc = op.half_pi - acos(dt / norm3(ray.d));
    return fabs(bragg - inc);
}

// ---- rocking-curve screen ----------------------------------------------------------------------------------
// The decision is p >= test with p = R exp(-(inc - bragg)^2 / 2 sigma^2) (or the step function) and test a
// uniform deviate.  When every ray has the same wavelength, inc - bragg = asin(c) - asin(s) is a short series
// in d = c - s (constants from the host, relative error <= 1e-6 inside |d| < scr_dmax), and a single-precision
// exponential gives p to ~2e-5.  Unless `test` falls within 1e-3 (relative) of that estimate the outcome of
// the exact evaluation is already known; the exact evaluation then only runs for about one candidate in a
// thousand (and for test == 0, NaNs, scenes without the screen).  `c`: the cosine of the angle between the ray
// and the surface normal, exact or good to a few ulp (the margins cover 1e-9).  Sets `decided`.
__device__ __forceinline__ bool screen_df(const KOptic& op, double dfa, double test, bool& decided)
{
    decided = true;
    if (op.rocking_type == XRT_ROCKING_STEP) {
        const double a = fabs(dfa), m = fma(op.half_fwhm, 1e-5, 1e-12);
        if (a < op.half_fwhm - m) return op.reflectivity >= test;       // p = 1.0 * R exactly
        if (a > op.half_fwhm + m) return false;                         // p = 0.0 * R < test
    } else {
        const double xa = (dfa * dfa) * op.scr_binv;
        const float xf = fminf((float)xa, 80.0f);
        const double pa = (double)__builtin_amdgcn_exp2f(xf * -1.44269504f) * op.reflectivity;
        if (test > pa * 1.001) return false;
        if (xa < 80.0 && test < pa * 0.999) return true;
    }
    decided = false;
    return false;
}

__device__ __forceinline__ bool bragg_screen(const KOptic& op, double c, double test, bool& decided)
{
    const double d = c - op.scr_s;
    if (fabs(d) < op.scr_dmax) return screen_df(op, d * fma(d, fma(d, op.scr_a3, op.scr_a2), op.scr_a1), test, decided);
    decided = test > op.scr_ptail;
    return false;
}

// The same for a ray with its own wavelength (s = wl / 2d): sin(inc - bragg) = c sqrt(1 - s^2) - s sqrt(1 - c^2)
// from approximate square roots (absolute error ~1e-15), inc - bragg = asin(y) = y + y^3/6 to 1e-9 relative for
// |y| < 0.01; beyond that |inc - bragg| >= |y| bounds p (scr2_tail).
__device__ __forceinline__ double sqrt_approx(double u)
{
    double y = __builtin_amdgcn_rsq(u);
    y = y * fma(-0.5 * u * y, y, 1.5);
    return u * y;
}
__device__ __forceinline__ bool bragg_screen_wl(const KOptic& op, double c, double wl, double test, bool& decided)
{
    const double s = wl * op.inv_two_d;
    const double y = c * sqrt_approx(fma(-s, s, 1.0)) - s * sqrt_approx(fma(-c, c, 1.0));
    if (fabs(y) < 0.01) return screen_df(op, y * fma(y * y, 0.16666666666666666, 1.0), test, decided);
    decided = (fabs(y) >= 0.01) && (test > op.scr2_tail);       // (a NaN y decides nothing)
    return false;
}

__device__ __forceinline__ bool bragg_accept(const KOptic& op, const Ray& ray, const V3& nrm, double test,
                                             bool have_bragg, double bragg_shared)
{
    // |d . (-n)| == |d . n| bit for bit (negation commutes with every rounding), so the negated normal of
    // the reference (:104) is not formed
    double dt = fabs(dot_e(ray.d, nrm));
    const double c = dt / norm3(ray.d);
    if (op.scr_ok && have_bragg && test > 0.0) {
        bool decided;
        const bool acc = bragg_screen(op, c, test, decided);
        if (decided) return acc;
    }
#ifndef XRT_DEV_NO_WL_SCREEN
    else if (op.scr2_ok && !have_bragg && test > 0.0) {
        // (a ray with its own wavelength: the same screen through sin(inc - bragg), see bragg_screen_wl)
        bool decided;
        const bool acc = bragg_screen_wl(op, c, ray.wl, test, decided);
        if (decided) return acc;
    }
#endif
    // ---- exact evaluation, in the reference's order -----------------------------------------------------
    // a monochromatic source gives every ray the same asin argument: evaluated once per run
    double bragg = have_bragg ? bragg_shared : asin(ray.wl / op.two_d);
    double inc = op.half_pi - acos(c);
    double p;
    if (op.rocking_type == XRT_ROCKING_STEP) {
        p = (fabs(inc - bragg) <= op.half_fwhm) ? 1.0 : 0.0;
    } else {
        double df = inc - bragg;
        p = exp(-(df * df) / op.two_sigma2);
    }
    p *= op.reflectivity;
    return p >= test;
}

// The direction of the surface normal at X without its normalisation, for the shapes where that is cheap
// (what the screen needs is only the cosine of the incidence angle to a few ulp)
__device__ __forceinline__ bool normal_direction(const KOptic& op, const V3& X, V3& nu)
{
    if (op.shape == XRT_SHAPE_PLANE) { nu = ld3(op.R + 6); return true; }
    if (op.shape == XRT_SHAPE_SPHERE) { nu = sub3(ld3(op.center), X); return true; }
    return false;
}

// TraceObject.make_image (optics/_TraceObject.py:234-293): one hit -> one pixel
__device__ __forceinline__ void image_hit(const KOptic& op, const V3& X, unsigned long long* images)
{
    V3 loc = to_local(op.R, sub3(X, ld3(op.origin)));
    double cx = rint(loc.x / op.pixel_size + op.pixel_xoff);
    double cy = rint(loc.y / op.pixel_size + op.pixel_yoff);
    if (cx >= 0.0 && cx < (double)op.pixel_nx && cy >= 0.0 && cy < (double)op.pixel_ny)
        atomicAdd(&images[op.image_offset + (long long)cx * op.pixel_ny + (long long)cy], 1ULL);
}

// The same pixel, counted in the workgroup's own LDS copy of the bins: two 16-bit counters per word
// (fused kernel variant 4, flushed to the u64 bins before a counter can reach 2^16).
__device__ __forceinline__ void image_hit_lds(const KOptic& op, const V3& X, uint32_t* lbins)
{
    V3 loc = to_local(op.R, sub3(X, ld3(op.origin)));
    double cx = rint(loc.x / op.pixel_size + op.pixel_xoff);
    double cy = rint(loc.y / op.pixel_size + op.pixel_yoff);
    if (cx >= 0.0 && cx < (double)op.pixel_nx && cy >= 0.0 && cy < (double)op.pixel_ny) {
        const uint32_t p = (uint32_t)(op.image_offset + (long long)cx * op.pixel_ny + (long long)cy);
        atomicAdd(&lbins[p >> 1], 1u << ((p & 1u) << 4));
    }
}

// The same 16-bit counters where nothing bounds how many hits a counter takes between two flushes (xrt_mosaic_kernel: the
// reflected rays of a whole run): the lane whose add takes a counter from 0x7fff to 0x8000 -- exactly one per crossing -- moves
// 0x8000 hits on into the u64 bin.  (Up to 255 more adds may land before its subtraction does: no carry into the neighbour.)
__device__ __forceinline__ void image_hit_lds_spill(const KOptic& op, const V3& X, uint32_t* lbins, unsigned long long* images)
{
    V3 loc = to_local(op.R, sub3(X, ld3(op.origin)));
    double cx = rint(loc.x / op.pixel_size + op.pixel_xoff);
    double cy = rint(loc.y / op.pixel_size + op.pixel_yoff);
    if (cx >= 0.0 && cx < (double)op.pixel_nx && cy >= 0.0 && cy < (double)op.pixel_ny) {
        typedef __attribute__((address_space(3))) uint32_t l32;
        const uint32_t p = (uint32_t)(op.image_offset + (long long)cx * op.pixel_ny + (long long)cy);
        const uint32_t sh = (p & 1u) << 4;
        l32* const w = (l32*)lbins + (p >> 1);
        const uint32_t old = __hip_atomic_fetch_add(w, 1u << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (((old >> sh) & 0xffffu) == 0x7fffu) {
            __hip_atomic_fetch_sub(w, 0x8000u << sh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            atomicAdd(&images[p], 0x8000ULL);
        }
    }
}

// --------------------------------------------------------------------------
// workgroup scan: ordered rank of `flag` among the 256 threads (wave64 ballot,
// mbcnt, 4 wave totals through LDS).  One barrier; `slot` alternates so that
// back-to-back scans need no second one.
// --------------------------------------------------------------------------

// Workgroup barrier for data that went through LDS only: waits for the wave's LDS operations, not for its
// global loads / stores / atomics in flight (__syncthreads() does: a pixel atomic or a prefetched record in front of
// a barrier then costs a round trip to L2 / HBM).  Used by the propagation kernel, whose waves talk through LDS.
__device__ __forceinline__ void lds_barrier()
{
#ifdef XRT_FULL_BARRIERS
    __syncthreads();
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

// `rotw`: the ranks follow the virtual thread order (tid + 64 * rotw) mod 256, i.e. wave `rotw` holds
// virtual threads 192.. and wave (4 - rotw) mod 4 comes first (see the wave rotation of the fused kernel).
template <bool LDS_ONLY = false>
__device__ __forceinline__ uint32_t wg_rank(bool flag, uint32_t* wave_tot /*[2][4]*/, int& slot,
                                            int tid, uint32_t& total, uint32_t rotw = 0u)
{
    unsigned long long b = __ballot(flag);
    uint32_t lane_rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0u));
    uint32_t wave = (((uint32_t)tid >> 6) + rotw) & 3u;
    uint32_t* wt = wave_tot + slot * 4;
    if ((tid & 63) == 0) wt[wave] = (uint32_t)__popcll(b);
    if (LDS_ONLY) lds_barrier(); else __syncthreads();
    uint32_t t0 = wt[0], t1 = wt[1], t2 = wt[2], t3 = wt[3];
    uint32_t base = (wave > 0 ? t0 : 0u) + (wave > 1 ? t1 : 0u) + (wave > 2 ? t2 : 0u);
    total = uni32(t0 + t1 + t2 + t3);
    slot ^= 1;
    return base + lane_rank;
}

// --------------------------------------------------------------------------
// history snapshot (objects/_Dispatcher.py:162,187), layout in xicsrt_hip.h
// --------------------------------------------------------------------------

__device__ __forceinline__ void hist_write(double* hist, uint8_t* hmask, int64_t n, int e, uint32_t id,
                                           const V3& o, const V3& d, double wl, bool alive)
{
    double* h = hist + (int64_t)e * XRT_HIST_COMPONENTS * n;
    h[0 * n + id] = o.x; h[1 * n + id] = o.y; h[2 * n + id] = o.z;
    h[3 * n + id] = d.x; h[4 * n + id] = d.y; h[5 * n + id] = d.z;
    h[6 * n + id] = wl;  h[7 * n + id] = 1.0;
    hmask[(int64_t)e * n + id] = alive ? 1 : 0;
}

// --------------------------------------------------------------------------
// generator set-up kernels
// --------------------------------------------------------------------------


// explicit numpy state -> stream (xrt_trace_history).  The heads of the fused kernel refill their rings in fixed
// half-ring steps and for that read up to 112 words BEHIND a head's position: with `pos` < 112 those lie in front of the
// imported block.  MT19937 runs backwards as well: s[n+624] = s[n+397] ^ twist(msb(s[n]) | low31(s[n+1])) gives the top bit
// of s[n] and the low 31 bits of s[n+1], so every earlier word follows from later ones; the 400 free slots of the ring
// are filled with them (they are the words numpy generated before this block whenever there was such a block).
__device__ __forceinline__ uint32_t mt_untwist(uint32_t t)      // y with twist(y) = t, y = msb(s[n]) | low31(s[n+1])
{
    return (t & 0x80000000u) ? (((t ^ 0x9908b0dfu) << 1) | 1u) : (t << 1);
}

__global__ void xrt_import_state_kernel(const KState* in, KStream* out, KState* gauss_state)
{
    for (int i = threadIdx.x; i < 624; i += blockDim.x) out->ring[i] = in->key[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int n = -1; n >= -400; n--) {
            const uint32_t hi = mt_untwist(out->ring[(uint32_t)(n + 624) & XRT_RMASK] ^ out->ring[(uint32_t)(n + 397) & XRT_RMASK]);
            const uint32_t lo = mt_untwist(out->ring[(uint32_t)(n + 623) & XRT_RMASK] ^ out->ring[(uint32_t)(n + 396) & XRT_RMASK]);
            out->ring[(uint32_t)n & XRT_RMASK] = (hi & 0x80000000u) | (lo & 0x7fffffffu);
        }
        out->gen = 624; out->next = (uint64_t)in->pos;
        gauss_state[0].has_gauss = in->has_gauss;
        gauss_state[0].gauss = in->gauss;
    }
}

// One wave walks one run's stream forward (no tempering, no workgroup barriers:
// LDS operations of a single wave execute in order) and snapshots the window
// wherever a source array starts: array k of N doubles begins 2*k*N words after
// the iteration's first word.  The stream itself is
// left at the first word after the source arrays (the Bragg uniforms).
__device__ __forceinline__ void wave_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ void wave_walk(uint32_t* ring, uint64_t& gen, uint64_t target, int lane)
{
    // full chunks of 192 (<= 227 words are independent): all nine reads in flight
    // together, then the three results are written
    while (gen < target && target - gen >= 192ull) {
        const uint32_t n0 = (uint32_t)gen + (uint32_t)lane, n1 = n0 + 64u, n2 = n0 + 128u;
        const uint32_t a0 = ring[(n0 - 624u) & XRT_RMASK], b0 = ring[(n0 - 623u) & XRT_RMASK], c0 = ring[(n0 - 227u) & XRT_RMASK];
        const uint32_t a1 = ring[(n1 - 624u) & XRT_RMASK], b1 = ring[(n1 - 623u) & XRT_RMASK], c1 = ring[(n1 - 227u) & XRT_RMASK];
        const uint32_t a2 = ring[(n2 - 624u) & XRT_RMASK], b2 = ring[(n2 - 623u) & XRT_RMASK], c2 = ring[(n2 - 227u) & XRT_RMASK];
        ring[n0 & XRT_RMASK] = mt_mix(a0, b0, c0);
        ring[n1 & XRT_RMASK] = mt_mix(a1, b1, c1);
        ring[n2 & XRT_RMASK] = mt_mix(a2, b2, c2);
        wave_fence();
        gen += 192ull;
    }
    if (gen < target) {
        const uint32_t m = (uint32_t)(target - gen);
#pragma unroll
        for (uint32_t j = 0; j < 3; j++) {
            uint32_t o = j * 64u + (uint32_t)lane;
            if (o < m) {
                uint32_t n = (uint32_t)gen + o;
                ring[n & XRT_RMASK] = mt_mix(ring[(n - 624u) & XRT_RMASK], ring[(n - 623u) & XRT_RMASK],
                                             ring[(n - 227u) & XRT_RMASK]);
            }
        }
        wave_fence();
        gen = target;
    }
}

// The same walk over `count` words with 32-bit positions (inside the propagation kernel: a wave walks one ring)
__device__ __forceinline__ void wave_walk_n(uint32_t* ring, uint32_t gen, uint64_t count, int lane)
{
    while (count >= 192ull) {
        const uint32_t n0 = gen + (uint32_t)lane, n1 = n0 + 64u, n2 = n0 + 128u;
        const uint32_t a0 = ring[(n0 - 624u) & XRT_RMASK], b0 = ring[(n0 - 623u) & XRT_RMASK], c0 = ring[(n0 - 227u) & XRT_RMASK];
        const uint32_t a1 = ring[(n1 - 624u) & XRT_RMASK], b1 = ring[(n1 - 623u) & XRT_RMASK], c1 = ring[(n1 - 227u) & XRT_RMASK];
        const uint32_t a2 = ring[(n2 - 624u) & XRT_RMASK], b2 = ring[(n2 - 623u) & XRT_RMASK], c2 = ring[(n2 - 227u) & XRT_RMASK];
        ring[n0 & XRT_RMASK] = mt_mix(a0, b0, c0);
        ring[n1 & XRT_RMASK] = mt_mix(a1, b1, c1);
        ring[n2 & XRT_RMASK] = mt_mix(a2, b2, c2);
        wave_fence();
        gen += 192u; count -= 192ull;
    }
    if (count > 0ull) {
        const uint32_t m = (uint32_t)count;
#pragma unroll
        for (uint32_t j = 0; j < 3; j++) {
            const uint32_t o = j * 64u + (uint32_t)lane;
            if (o < m) {
                const uint32_t n = gen + o;
                ring[n & XRT_RMASK] = mt_mix(ring[(n - 624u) & XRT_RMASK], ring[(n - 623u) & XRT_RMASK], ring[(n - 227u) & XRT_RMASK]);
            }
        }
        wave_fence();
    }
}

// An imported generator state (np.random.get_state(): block of 624 words + position) brought to the form
// the jump-ahead wants: exactly 624 words generated ahead of `next`, i.e. `pos` more words are generated.
__global__ __launch_bounds__(64)
void xrt_advance_kernel(KStream* st)
{
    __shared__ uint32_t ring[XRT_RING];
    const int lane = threadIdx.x;
    for (int i = lane; i < (int)XRT_RING; i += 64) ring[i] = st->ring[i];
    uint64_t gen = uni64(st->gen);
    const uint64_t next = uni64(st->next);
    wave_fence();
    wave_walk(ring, gen, next + 624ull, lane);
    for (int i = lane; i < (int)XRT_RING; i += 64) st->ring[i] = ring[i];
    if (lane == 0) st->gen = gen;
}

// np.random.seed(int) -> init_genrand (xicsrt_raytrace.py:111), one wave per run: lane 0 runs the
// serial recurrence into LDS, the wave then generates the 512 words of the canonical form
// (what the propagation kernel leaves behind, and what the jump expects) and stores the ring coalesced
__global__ __launch_bounds__(64)
void xrt_seed_kernel(const uint32_t* seeds, KStream* streams, KState* gauss_state, int n_runs)
{
    __shared__ uint32_t ring[XRT_RING];
    const int r = blockIdx.x, lane = threadIdx.x;
    if (r >= n_runs) return;
    if (lane == 0) {
        gauss_state[r].has_gauss = 0;
        gauss_state[r].gauss = 0.0;
        uint32_t s = seeds[r];
        for (int i = 0; i < 624; i++) {
            ring[i] = s;
            s = 1812433253u * (s ^ (s >> 30)) + (uint32_t)(i + 1);
        }
    }
    wave_fence();
    uint64_t gen = 624;
    wave_walk(ring, gen, 624ull + (uint64_t)XRT_AHEAD, lane);
    KStream* st = streams + r;
    for (int i = lane; i < (int)XRT_RING; i += 64) st->ring[i] = ring[i];
    if (lane == 0) { st->gen = 624 + XRT_AHEAD; st->next = 624; }
}

__global__ __launch_bounds__(256)
void xrt_seek_kernel(KStream* streams, KStream* heads, int n_runs, int n_arrays, uint32_t array_used,
                     int n_src_heads, int64_t n_rays)
{
    __shared__ uint32_t rings[4][XRT_RING];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int run = blockIdx.x * 4 + wave;
    if (run >= n_runs) return;
    uint32_t* ring = rings[wave];
    KStream* st = streams + run;
    for (int i = lane; i < (int)XRT_RING; i += 64) ring[i] = st->ring[i];
    uint64_t gen = uni64(st->gen);
    const uint64_t next = uni64(st->next);
    wave_fence();
    int h = 0;
    for (int k = 0; k <= n_arrays; k++) {
        const uint64_t target = next + 2ull * (uint64_t)k * (uint64_t)n_rays;
        const bool snap = (k < n_arrays) && ((array_used >> k) & 1u);
        if (!snap && k < n_arrays) continue;
        wave_walk(ring, gen, target, lane);
        KStream* out = (k < n_arrays) ? (heads + (size_t)run * n_src_heads + h) : st;
        for (int i = lane; i < (int)XRT_RING; i += 64) out->ring[i] = ring[i];
        if (lane == 0) { out->gen = gen; out->next = target; }
        if (k < n_arrays) h++;
    }
}

// MT19937 jump-ahead (polynomial method, see mt_jump.inc): the window of 624
// state words J steps ahead is the XOR of the windows at every j with g_j = 1,
// g(t) = t^J mod phi(t).  One workgroup per run builds the stretch
// S[i] = s[gen-624+i], i < 19937+623, in LDS once and forms, for every needed
// source array k and for the stream head, the window that ends at the array's
// first word (J_k = 2kN - 512 from the canonical stream form gen = next + 512).
// Cost is independent of N; the sequential xrt_seek_kernel remains the general path.
#define XRT_JUMP_THREADS 1024
#include "xrt_jump.inc"

#define XRT_PLASMA_PART 1      // what the fused kernel needs of plasma sources (the scout kernel comes behind xrt_staged.inc)
#include "xrt_plasma.inc"
#undef XRT_PLASMA_PART

// --------------------------------------------------------------------------
// the propagation kernel
// --------------------------------------------------------------------------

struct KArgs {
    KStream* streams;                   // [n_runs] stream heads, updated in place
    const KStream* heads;               // [n_runs][n_src_heads] positioned source heads
    int32_t n_runs, n_src_heads;
    unsigned long long* num_out;        // [n_optics+1]
    unsigned long long* images;         // may be null
    uint32_t* run_counter;              // dynamic run dispenser
    double*  hist;                      // HIST only
    uint8_t* hmask;
    // segmented runs (SEG kernels): a run's rays are split into n_seg segments of seg_len rays
    // (a multiple of the tile), one work unit each; `heads` is then [n_runs][n_seg][n_src_heads].
    int32_t  n_seg;
    int32_t  mode;                      // 1: count the Bragg candidates of every unit; 2: propagate
    int64_t  seg_len;
    uint32_t* unit_count;               // [n_runs][n_seg] candidates per unit (written by mode 1)
    const KStream* chunk_heads;         // per run n_seg stream heads, chunk_words apart, from the first Bragg uniform on
    int64_t  chunk_words;
    int64_t  run_stride;                // SEG: heads / chunk_heads of consecutive runs are this many KStreams apart
    // A segment is shared by n_sub work units: all load the segment's heads, unit j first walks them (raw
    // generation only) over the j * sub_len rays in front of its own
    int32_t  n_sub, pad_sub;
    int64_t  sub_len;
    // SEG == 2 (one pass, look-back): every unit parks its Bragg candidates, stably compacted in ray order, in its part
    // of the run's structure-of-arrays in HBM (record i of the unit at index ray_lo + i), publishes their number + 1 in
    // unit_flag and, once the units in front of it in the run have published theirs, knows where its Bragg uniforms start
    double*   cand;                     // [n_runs][6 or 7][cand_cap]
    uint32_t* cand_id;                  // [n_runs][cand_cap] ray indices (HIST)
    uint32_t* cand_aux;                 // [n_runs][cand_cap] hit face (variant 2)
    int64_t   cand_cap;
    uint32_t* unit_flag;                // [n_runs][n_seg * n_sub], zero before the launch
    // SEG == 3 / 4 (the phases of SEG == 2 as launches of their own around xrt_mesh_rest_kernel, a mesh crystal): that
    // kernel finishes the intersection of every parked ray and leaves per ray the hit point, the normal and in cand_aux
    // XRT_CAND_DEAD for a ray that is out; per 64 rays and per unit the number left alive (= Bragg draws)
    uint32_t dir_lds_bytes, pad_dir;    // SEG == 3: bytes of the direction grid's tables behind the workgroup's other LDS (0: none; KMesh.dg_n)
    uint32_t* batch_alive;              // [n_runs][cand_cap / 64]
    uint32_t* unit_alive;               // [n_runs][n_seg * n_sub], zero before the launch
    // ... a mesh whose faces are known as fans around its points (KMesh.lds_star): xrt_mesh_star_lds_kernel settles the parked
    // rays it can settle from the fan of the nearest point alone and lists the others per unit (slow_q: record numbers within
    // the unit, unit_slow: how many); the launch that walks the face lists then takes only those (slow_pass != 0)
    uint32_t* slow_q;                   // [n_runs][cand_cap]: the fan launch lists per 64 records (a wave: no atomics), slow_cnt how many;
    uint32_t* slow_cnt;                 // [n_runs][cand_cap / 64]   xrt_mesh_slow_compact_kernel closes the gaps per unit (slow_q2, unit_slow),
    uint32_t* slow_q2;                  // [n_runs][cand_cap]         and the list walk reads that (its slow_q = this slow_q2)
    uint32_t* unit_slow;                // [n_runs][n_seg * n_sub]
    // The work of the launches with tables in LDS (one 1024-thread workgroup per CU): the blocks of 1024 records that hold
    // rays, as a list (unit, block, records of the unit) made by xrt_mesh_items_kernel -- a workgroup reads its next item
    // instead of looking through the units' counts for it, a round trip to memory per block looked at.
    const uint4* items;                 // [n_runs * n_seg * n_sub * blocks per unit]
    const uint32_t* n_items;            // [1]
    uint32_t  slow_pass;
    // What a parked ray's record holds between the launches.  A mesh that is not interpolated (split_interp == 0): its
    // (local-frame) origin and direction [, wavelength] as the first phase left them, and in cand_aux the face its
    // intersection ended on (XRT_CAND_DEAD: the ray is out) -- the hit point is not kept: the few rays the crystal reflects
    // form it again from origin, direction and face (mesh_hit_point), and the normal is the face's.  An interpolated mesh
    // (split_interp != 0): the hit point in place of the origin (its height from the interpolation) and the interpolated
    // normal in components 6 - 8 (between the two middle launches component 6 holds the nearest point).
    uint32_t  split_interp;
    // ... and a mesh that is not interpolated straight behind a point source (unit_o below): the records hold nothing but the
    // direction [and the wavelength] -- components 0 - 2 [3], 24 - 32 B instead of 48 - 56
    uint32_t  split_lean;
    // A mesh crystal straight behind a point source: every parked ray of the call has the same (local-frame) origin.  The first
    // phase then leaves it ONCE per unit (unit_o[unit][3], from the unit's first parked ray -- the device's own arithmetic) and
    // not in the records (24 of their 52 bytes); the middle launches and the second phase take it from there.  Null: per ray.
    double*   unit_o;
    unsigned long long* dbg;            // development: [units][8] wall-clock stamps of a unit's phases (null: none)
    // Pixel bins of the fused kernel: `images` may point to image_rep replicas, image_stride bins apart, which the library
    // sums into the caller's bins behind the last launch.  Scattered 8-byte atomics execute at the memory side, and all
    // workgroups adding into the same few hot pixels of a small image serialise there; workgroup w adds into replica w mod R.
    uint32_t image_rep, pad_rep;
    uint64_t image_stride;
    unsigned long long* images_rep;     // host side: the replicas (launch_variant points `images` at them)
    // Gaussian wavelengths (np.random.normal) prepared by xrt_gauss_kernel: the values per run and ray, and
    // the words the rejection sampler consumed in front of the Bragg uniforms (null: none)
    const double*   wl_array;           // [n_runs][n_rays]
    const uint64_t* base_words;         // [n_runs]
    // circular ray buffer in LDS: capacity in records and the Bragg batch (128 or 256 candidates)
    uint32_t qcap, bragg_batch;
    unsigned long long* progress;       // tiles done by all workgroups (null: no priority feedback)
    uint32_t lbins_words, pad_lbins;    // variant 4: words of the workgroup's LDS copy of the pixel bins (two bins each)
    KPlasmaRays plasma;                 // XRT_SRC_PLASMA: what xrt_plasma_scout_kernel left per run slot (run index = slot)
};

#define XRT_CAND_DEAD 0xfffffffeu
#define XRT_MESH_LDS_MAX (150u * 1024u)     // tables of xrt_mesh_rest_lds_kernel: one 1024-thread workgroup per CU
#ifndef XRT_WAVES_PER_EU
#define XRT_WAVES_PER_EU 4
#endif
#ifndef XRT_ABLATE
#define XRT_ABLATE 0        // development: > 0 cuts the fused kernel short behind a stage (instruction accounting)
#endif

// The flattened scene lives in device memory (workspace) and is read through the constant
// address space: uniform scalar loads (s_load), no kernarg-size limit on the number of optics.
// Such loads are invariant, so LLVM would hoist all of them out of the tile loop and then spill
// ~150 scalar registers into vector-register lanes (v_writelane / v_readlane are VALU work in the
// hot loop).  scene_fresh() hands the same pointer back through an empty asm: loads behind it
// cannot move above it, which keeps the live ranges of the scene constants inside one stage.
__device__ __forceinline__ const KScene* scene_fresh(const KScene* p)
{
    uint64_t a = (uint64_t)p;
    asm volatile("" : "+s"(a));
    return (const KScene*)(const XRT_C4 KScene*)a;
}
#define SC  (*scl)
#define SRC (scl->src)

// HIST: write the per-element history.  VARIANT 0: the lean variant (isotropic
// cone, shared cone axis, constant or uniform wavelength, plane/sphere, no
// apertures) with lower register use; 1: every source / analytic shape /
// aperture feature; 2: 1 + optics traced in their local frame and mesh optics.
// SEG != 0: few runs of many rays -- the work unit is a part of a run (a segment, or one of the n_sub parts of a
// segment; see KArgs).  The Bragg uniform of a ray is indexed by its ordered live rank in the whole run:
//   SEG == 2 (one pass): a unit takes its rays up to the Bragg element once, parks the candidates in HBM (stably
//     compacted, 48 - 60 B each), publishes their number and picks up the numbers of the units in front of it in the run
//     (the dispenser hands units out in order, so those are always under way); that sum positions its stream head, and
//     the unit runs the Bragg test and the elements behind it over its parked candidates;
//   SEG == 1 (two passes; candidate buffers beyond the workspace budget): a first launch (mode 1) only counts every
//     unit's candidates and the second (mode 2) starts each unit at the position the counts of the earlier units give.
template <bool HIST, int VARIANT, int SEG>
__global__ __launch_bounds__(XRT_TILE, ((VARIANT == 2 && SEG < 3) ? 2 : XRT_WAVES_PER_EU))
void xrt_trace_kernel(const KScene* __restrict__ scene_g, const KArgs args)
{
    constexpr bool FULL = VARIANT == 1 || VARIANT == 2;
    constexpr bool EXT = VARIANT == 2;
    constexpr bool LEANWL = VARIANT == 3;       // the lean geometry with a prepared wavelength per ray
    constexpr bool LBINS = VARIANT == 4;        // the lean geometry without a Bragg test, pixel bins pre-aggregated in LDS
    // SEG >= 2: the candidates of the Bragg element are parked in HBM.  3 / 4 (a mesh crystal, variant 2 only): the two phases
    // as launches of their own -- 3: rays up to the first pass over the mesh's faces, parked with the face; then
    // xrt_mesh_rest_kernel finishes the intersection of every parked ray (a thread per ray, at the occupancy its own
    // registers allow); 4: Bragg test and the elements behind over what is left.  Neither compiles any of the mesh code
    // but the first pass: four workgroups per CU instead of two.
    constexpr bool PH_A = SEG != 4, PH_B = SEG != 3, SPLIT = SEG >= 3;
    // (3 alone also in front of xrt_mosaic_kernel, variants 0 / 1: the rays on a mosaic crystal are parked and that is all)
    static_assert(!SPLIT || !HIST, "split phases: without histories only");
    static_assert(SEG != 4 || VARIANT == 2, "second phase of the split: mesh variant only");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    // One circular structure-of-arrays buffer of XRT_QCAP ray records serves as
    //  (a) the FIFO queue of rays waiting for the Bragg test (filled tile by tile in ray
    //      order, drained 128 at a time so that the test always runs on two full waves), and
    //  (b) scratch for the stable compactions between the other elements (its free part).
    // A record is the point (3), the direction (3), the wavelength unless all rays share one, and (HIST) the
    // ray index; a mesh / local-frame variant also keeps the hit face.
    const KScene* scl0 = scene_fresh(scene_g);
    const bool q_has_wl = !(scl0->src.wavelength_dist == XRT_WL_CONST && !scl0->src.has_velocity);
    const uint32_t qcap = args.qcap, bbatch = args.bragg_batch;
    double*   qbuf   = reinterpret_cast<double*>(lds_raw);                             // [6 or 7][qcap]
    uint32_t* qid    = reinterpret_cast<uint32_t*>(qbuf + (q_has_wl ? 7u : 6u) * qcap); // [qcap] (HIST only)
    uint32_t* qaux   = qid + (HIST ? qcap : 0u);                                       // [qcap] (EXT only)
    uint32_t* rings  = qaux + (EXT ? qcap : 0u);                                       // [nh+1][1024]
    const int tid = threadIdx.x;
    const KScene* scl = scene_fresh(scene_g);
    const int64_t N = SRC.n_rays;
    const int nh = args.n_src_heads;
    uint32_t* stream = rings + nh * XRT_RING;                                          // (variant 4 has none: no Bragg draws)
    uint32_t* small = stream + ((LBINS && nh > 0) ? 0u : XRT_RING);     // (at least one ring: variant 4 borrows it at the end of a segmented run)
    uint32_t* wave_tot = small;                                                        // [2][4]
    unsigned long long* cnt = reinterpret_cast<unsigned long long*>(small + 8);        // [XRT_DEV_MAX_OPTICS+1]
    uint32_t* bcast = small + 8 + 2 * (XRT_DEV_MAX_OPTICS + 2);
    // variant 4: image-heavy scenes (no rocking curve in front of the imaged elements: most rays end in a pixel or two)
    // are bound by the rate of scattered 8-byte atomics (~20 G/s).  Every workgroup counts into a private copy of the
    // bins in LDS instead, two 16-bit counters per word, and adds it to the u64 bins every 255 tiles (a counter grows
    // by at most 256 per tile) and at the end of its unit: 3.5 - 50 x fewer global atomics.
    uint32_t* lbins = bcast + 8;                                                        // [args.lbins_words]
    unsigned long long* const img = args.images ? args.images + (size_t)(blockIdx.x % (args.image_rep ? args.image_rep : 1u)) * args.image_stride
                                                : nullptr;
    uint32_t lb_tiles = 0;
    if constexpr (LBINS) {
        for (uint32_t i = (uint32_t)threadIdx.x; i < args.lbins_words; i += XRT_TILE) lbins[i] = 0u;
        lds_barrier();
    }
    auto lbins_flush = [&]() __attribute__((always_inline)) {
        if constexpr (LBINS) {
            lds_barrier();
            for (uint32_t i = (uint32_t)threadIdx.x; i < args.lbins_words; i += XRT_TILE) {
                const uint32_t v = lbins[i];
                if (v) {
                    lbins[i] = 0u;
                    if (v & 0xffffu) atomicAdd(&img[2u * i], (unsigned long long)(v & 0xffffu));
                    if (v >> 16) atomicAdd(&img[2u * i + 1u], (unsigned long long)(v >> 16));
                }
            }
            lds_barrier();
            lb_tiles = 0;
        }
    };
    auto pixel = [&](const KOptic& op, const V3& X) __attribute__((always_inline)) {
        if constexpr (LBINS) image_hit_lds(op, X, lbins);
        else image_hit(op, X, img);
    };

    // Fair sharing of a CU.  The SIMD arbiter prefers the oldest wave, so of the workgroups that share a CU the
    // first-dispatched one runs ahead of the others for the whole launch, finishes early and leaves its slot
    // empty while the others are still at work (measured: waves alive for 78 % of a 1024-run launch; vector
    // issue 72 % busy against 82 % when finished slots are refilled).  Every 32 tiles a workgroup adds its
    // progress to a global counter and compares itself with the mean over all workgroups: ahead of the
    // field -> issue priority 0, behind -> 2, which the arbiter ranks above age.
    uint32_t tiles_done = 0;
    const uint32_t n_wg = gridDim.x;
    int slot = 0;
    // Wave rotation.  The stages behind a compaction work on n < 256 dense rays, i.e. on the first
    // ceil(n / 64) *virtual* waves; which hardware wave is virtual wave 0 moves on with every such stage
    // so that the partial-width stages (Bragg test: 128 rays, elements behind it: a few rays) load the
    // four SIMDs of the CU evenly instead of always the ones that host waves 0 and 1.
    uint32_t rotw = 0;
    auto vtid = [&]() __attribute__((always_inline)) -> uint32_t { return ((uint32_t)tid + 64u * rotw) & 255u; };

    // the (single) element that makes a Bragg test on this path, or -1
    int be = -1;
    for (int e = 0; e < SC.n_optics; e++)
        if (SC.opt[e].interact == XRT_INTERACT_CRYSTAL && (SC.opt[e].flags & XRT_F_CHECK_BRAGG)) be = e;
    if constexpr (SEG == 3 && !EXT) {       // (in front of xrt_mosaic_kernel: the scene's one mosaic crystal)
        for (int e = 0; e < SC.n_optics; e++) if (SC.opt[e].interact == XRT_INTERACT_MOSAIC) be = e;
    }
    // A mesh crystal that makes the Bragg test: only the exhaustive first pass runs in a tile's own lanes (about half
    // of a cone of rays misses the mesh altogether, and a wave of unrelated rays executes the union of its lanes'
    // work); the rays that hit a face wait in the Bragg queue and get the rest -- nearest point, second pass,
    // interpolation, bounds -- on the dense lanes of a Bragg batch, in ray order, just in front of their Bragg test.
    // SEG == 3: the direction grid of a mesh behind a point source and its face records (KMesh.dg_n, pt_rec) in LDS
    const __attribute__((address_space(3))) double* l_dpt = nullptr;
    const __attribute__((address_space(3))) uint32_t* l_dcells = nullptr;
    if constexpr (SEG == 3) {
        if (args.dir_lds_bytes > 0u && be >= 0) {
            MeshRef Mh = *(const XRT_C4 KMesh*)uniform_u64((uint64_t)SC.opt[be].mesh);
            double* tp = reinterpret_cast<double*>(lbins);
            const uint32_t n_pt = 12u * ((uint32_t)Mh.n_first + 1u), n_c = 2u * (uint32_t)(Mh.dg_n * Mh.dg_n);
            uint32_t* tc = reinterpret_cast<uint32_t*>(tp + n_pt);
            const XRT_G1 double* gp = (const XRT_G1 double*)(uint64_t)Mh.pt_rec;
            const XRT_G1 uint32_t* gc = (const XRT_G1 uint32_t*)(uint64_t)Mh.dg_cells;
            for (uint32_t i = (uint32_t)tid; i < n_pt; i += XRT_TILE) tp[i] = gp[i];
            for (uint32_t i = (uint32_t)tid; i < n_c; i += XRT_TILE) tc[i] = gc[i];
            l_dpt = (const __attribute__((address_space(3))) double*)tp;
            l_dcells = (const __attribute__((address_space(3))) uint32_t*)tc;
            lds_barrier();
        }
    }
    bool mesh_pre = false;
    // (SEG == 2 counts a unit's candidates in its first phase: the whole intersection runs there)
    if constexpr (EXT && SEG != 2) mesh_pre = be >= 0 && SC.opt[be].shape == XRT_SHAPE_MESH;      // (SEG >= 3: always, by the host's choice of route)

    double wl_run = 0.0;        // the wavelength of every ray when it is not part of the records
    auto q_store = [&](uint32_t i, const V3& o, const V3& d, double wl, uint32_t id) __attribute__((always_inline)) {
        qbuf[0 * qcap + i] = o.x; qbuf[1 * qcap + i] = o.y; qbuf[2 * qcap + i] = o.z;
        qbuf[3 * qcap + i] = d.x; qbuf[4 * qcap + i] = d.y; qbuf[5 * qcap + i] = d.z;
        if (q_has_wl) qbuf[6 * qcap + i] = wl;
        if (HIST) qid[i] = id;
    };
    auto q_load = [&](uint32_t i, V3& o, V3& d, double& wl, uint32_t& id) __attribute__((always_inline)) {
        o.x = qbuf[0 * qcap + i]; o.y = qbuf[1 * qcap + i]; o.z = qbuf[2 * qcap + i];
        d.x = qbuf[3 * qcap + i]; d.y = qbuf[4 * qcap + i]; d.z = qbuf[5 * qcap + i];
        wl = q_has_wl ? qbuf[6 * qcap + i] : wl_run;
        id = HIST ? qid[i] : 0u;
    };
    auto q_wrap = [&](uint32_t i) __attribute__((always_inline)) -> uint32_t { return i >= qcap ? i - qcap : i; };
    // SEG == 2: the same record in the run's candidate arrays in HBM (`cbase`: first double of the run's arrays)
    double* cbase = nullptr;
    size_t crun = 0;                // first ray-index / face word of the run
    // (blocks of 256 records, component-major inside a block: what a batch reads lies in one 12 - 14 KB stretch)
    // (split phases: + the surface normal at the hit point, components 6 - 8, which xrt_mesh_rest_kernel leaves)
    const bool q_lean = SPLIT && args.split_lean != 0u;           // (records without the origin: KArgs.split_lean)
    const int q_d0 = q_lean ? 0 : 3;                               // first component of the direction
    const int q_wlc = q_lean ? 3 : ((SPLIT && args.split_interp) ? 9 : 6);
    const int q_ncomp = q_wlc + (q_has_wl ? 1 : 0);
    auto cand_store = [&](int64_t i, const V3& o, const V3& d, double wl, uint32_t id, int aux) __attribute__((always_inline)) {
        if constexpr (SEG == 3 && !EXT) {
            // (in front of xrt_mosaic_kernel: a record in one piece -- its layers read single rays of a thinning list)
            double* r = cbase + i * (int64_t)q_ncomp;
            r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = d.x; r[4] = d.y; r[5] = d.z;
            if (q_has_wl) r[6] = wl;
            return;
        }
        double* c = cbase + (i >> 8) * (int64_t)(q_ncomp * 256) + (i & 255);
        bool shared_o = false;
        if constexpr (SPLIT) shared_o = args.unit_o != nullptr;
        if (!shared_o) { c[0 * 256] = o.x; c[1 * 256] = o.y; c[2 * 256] = o.z; }
        c[q_d0 * 256] = d.x; c[(q_d0 + 1) * 256] = d.y; c[(q_d0 + 2) * 256] = d.z;
        if (q_has_wl) c[q_wlc * 256] = wl;
        if (HIST) args.cand_id[crun + i] = id;
        if (EXT) args.cand_aux[crun + i] = (uint32_t)aux;
    };
    auto cand_load = [&](int64_t i, V3& o, V3& d, double& wl, uint32_t& id, int& aux) __attribute__((always_inline)) {
        const double* c = cbase + (i >> 8) * (int64_t)(q_ncomp * 256) + (i & 255);
        // (split phases, a mesh that is not interpolated behind a point source: components 0 - 2 hold nothing, see KArgs.unit_o)
        bool skip_o = false;
        if constexpr (SPLIT) skip_o = args.unit_o != nullptr && !args.split_interp;
        if (!skip_o) { o.x = c[0 * 256]; o.y = c[1 * 256]; o.z = c[2 * 256]; }
        d.x = c[q_d0 * 256]; d.y = c[(q_d0 + 1) * 256]; d.z = c[(q_d0 + 2) * 256];
        wl = q_has_wl ? c[q_wlc * 256] : wl_run;
        id = HIST ? args.cand_id[crun + i] : 0u;
        aux = EXT ? (int)args.cand_aux[crun + i] : 0;
    };
    // (split phases: the surface normal of a parked ray that is still there -- the face's, or the interpolated one with the
    //  interpolated height, see KArgs.split_interp)
    auto cand_normal = [&](int64_t i, int aux) __attribute__((always_inline)) -> V3 {
        V3 n;
        n.x = n.y = n.z = 0.0;
        if constexpr (SPLIT) {
            if ((uint32_t)aux == XRT_CAND_DEAD) return n;
            if (args.split_interp) {
                const double* c = cbase + (i >> 8) * (int64_t)(q_ncomp * 256) + (i & 255);
                n.x = c[6 * 256]; n.y = c[7 * 256]; n.z = c[8 * 256];
            } else {
                MeshRef Mh = *(const XRT_C4 KMesh*)uniform_u64((uint64_t)SC.opt[be].mesh);
                const gdp fn = Mh.faces_normal + 3 * (size_t)aux;
                n.x = fn[0]; n.y = fn[1]; n.z = fn[2];
            }
        }
        return n;
    };

    for (;;) {
        // ---- next run ------------------------------------------------------
        if (tid == 0) bcast[0] = atomicAdd(args.run_counter, 1u);
        lds_barrier();
        const uint32_t unit = uni32(bcast[0]);
        uint32_t run = unit, seg = 0, sub = 0, upr = 1, uidx = 0;      // (upr: units per run, uidx: this unit's place in its run)
        if (SEG) {
            upr = (uint32_t)args.n_seg * (uint32_t)args.n_sub;
            run = unit / upr; uidx = unit - run * upr;
            seg = uidx / (uint32_t)args.n_sub; sub = uidx - seg * (uint32_t)args.n_sub;
        }
        if (run >= (uint32_t)args.n_runs) break;
        const bool counting = SEG == 1 && args.mode == 1;
        const bool last_unit = !SEG || uidx + 1u == upr;
        if constexpr (SEG >= 2) {
            crun = (size_t)run * (size_t)args.cand_cap;
            cbase = args.cand + crun * (size_t)q_ncomp;
        }
        // a plasma run has as many rays as its bundles drew (xrt_plasma_scout_kernel), N is the capacity
        const bool plasma = FULL && !SEG && SRC.kind == XRT_SRC_PLASMA;
        const int64_t N_run = plasma ? (int64_t)uni64((uint64_t)args.plasma.n_src[run]) : N;
        // ray range of this unit
        int64_t ray_lo = 0, ray_hi = N_run;
        if (SEG) {
            const int64_t seg_lo = (int64_t)seg * args.seg_len;
            const int64_t seg_hi = (seg_lo + args.seg_len < N) ? seg_lo + args.seg_len : N;
            ray_lo = seg_lo + (int64_t)sub * args.sub_len;
            if (ray_lo > seg_hi) ray_lo = seg_hi;
            ray_hi = (ray_lo + args.sub_len < seg_hi) ? ray_lo + args.sub_len : seg_hi;
        }

        // ---- load the positioned heads and the stream head ----------------
        // The source heads advance in lockstep (512 words per tile each), so they share one position:
        // every ring is rotated while it is loaded so that the head's next word sits in slot 0.  A tile's
        // words then occupy one half of every ring ([0,512) or [512,1024), alternating), all heads use the
        // same LDS offsets (the ring number goes into the instruction's immediate offset), and the next
        // tile's half is generated in three fixed chunks of 171, 171 and 170 words (<= 227 are independent).
        if constexpr (PH_A) {
            uint32_t havail[6];
            int h = 0;
#pragma unroll
            for (int k = 0; k < 6; k++) {
                havail[k] = XRT_AHEAD;
                if ((SRC.array_used >> k) & 1u) {
                    const KStream* src = SEG ? args.heads + (size_t)run * args.run_stride + (size_t)seg * nh + h
                                             : args.heads + (size_t)run * nh + h;
                    uint32_t* r = rings + h * XRT_RING;
                    const uint32_t nx = uni32((uint32_t)src->next), gn = uni32((uint32_t)src->gen);
                    for (int i = tid; i < (int)XRT_RING; i += XRT_TILE) r[((uint32_t)i - nx) & XRT_RMASK] = src->ring[i];
                    havail[k] = gn - nx;
                    h++;
                }
            }
            lds_barrier();
            // A head arrives with 0 (jump-ahead, sequential walk) to 624 words generated beyond its position;
            // each is brought to at least one tile's worth from where IT stands (its ring holds nothing older
            // than 1024 words behind its own front).
            bool again = true;
            while (again) {
                again = false;
                int hh = 0;
#pragma unroll
                for (int k = 0; k < 6; k++) {           // (array k <-> ring hh; unused arrays keep havail = XRT_AHEAD)
                    if (havail[k] < XRT_AHEAD) {
                        uint32_t chunk = XRT_AHEAD - havail[k];
                        if (chunk > 227u) chunk = 227u;
                        if ((uint32_t)tid < chunk) {
                            uint32_t* r = rings + hh * XRT_RING;
                            const uint32_t n = havail[k] + (uint32_t)tid;
                            r[n & XRT_RMASK] = mt_mix(r[(n - 624u) & XRT_RMASK], r[(n - 623u) & XRT_RMASK], r[(n - 227u) & XRT_RMASK]);
                        }
                        havail[k] += chunk;
                        again = again || (havail[k] < XRT_AHEAD);
                    }
                    if ((SRC.array_used >> k) & 1u) hh++;
                }
                lds_barrier();
            }
        }
        uint32_t hslot = 512u;      // slot of the current tile's first word, 0 or 512 (flipped when a tile starts)
        uint32_t gstep = 3;         // chunks of the next tile's half that are generated (3 = all)
        KStream* st = args.streams + run;
        // the stream head: the run's own, or (SEG) the chunk head at or before this unit's first
        // Bragg uniform, which lies 2 * (candidates of the earlier segments) words into the draws
        uint64_t seg_skip = 0;
        const KStream* st_in = st;
        uint64_t s_next0 = 0, s_gen0 = 0, s_used = 0;
        uint32_t sgen = XRT_AHEAD, spos = 0;       // (until the head is opened: nothing to generate)
        // (mt_step keeps XRT_AHEAD = 512 words generated beyond the stream head's position, and no more: the step in front of
        //  a batch's survivor scan runs while slower waves still read the batch's 512 words, and a word's slot is that of the
        //  word 1024 further on)
        constexpr uint32_t s_ahead = XRT_AHEAD;
        // Opens the unit's stream head.  SEG: `before` Bragg candidates of this run lie in front of the unit's; the draws
        // start behind the source arrays (+ the words a Gaussian wavelength array consumed), at or behind a chunk head.
        auto stream_open = [&](bool positioned, unsigned long long before) __attribute__((always_inline)) {
            if (SEG && positioned) {
                const uint64_t words = 2ull * before + (args.base_words ? uni64(args.base_words[run]) : 0ull);
                const uint64_t chunk = words / (uint64_t)args.chunk_words;
                seg_skip = words - chunk * (uint64_t)args.chunk_words;
                st_in = args.chunk_heads + (size_t)run * args.run_stride + chunk;
            }
            if constexpr (!LBINS) for (int i = tid; i < (int)XRT_RING; i += XRT_TILE) stream[i] = st_in->ring[i];
            s_next0 = uni64(st_in->next); s_gen0 = uni64(st_in->gen);
            // (variant 4: nothing is drawn from the stream head, it passes through memory untouched and always counts as generated)
            sgen = LBINS ? (uint32_t)s_next0 + XRT_AHEAD : (uint32_t)s_gen0; spos = (uint32_t)s_next0;
        };
        // SEG == 2 with a Bragg element: the head is opened behind the unit's first phase, when the units in front have
        // published their candidate counts
        const bool deferred = SEG >= 2 && be >= 0;
        if (!deferred) {
            // SEG == 1, second launch, with a Bragg optic: the candidates of the earlier units of this run, summed by the
            // whole workgroup.  Without a Bragg optic only the run's last unit needs the stream (to store the run's new head).
            unsigned long long before = 0;
            if (SEG == 1 && args.mode == 2 && be >= 0) {
                if (tid == 0) cnt[0] = 0ULL;
                lds_barrier();
                unsigned long long part = 0;
                for (uint32_t q = (uint32_t)tid; q < uidx; q += XRT_TILE) part += args.unit_count[(size_t)run * upr + q];
                if (part) atomicAdd(&cnt[0], part);
                lds_barrier();
                before = uni64(cnt[0]);
                lds_barrier();
            }
            stream_open(SEG != 0 && !counting && (be >= 0 || last_unit), before);
        }
        uint32_t qhead = 0, qcount = 0, bcount = 0;
        uint32_t n_candidates = 0;
        if (tid < XRT_DEV_MAX_OPTICS + 1) cnt[tid] = 0ULL;
        lds_barrier();

        // One generation step: every head (and the stream head) that has fewer
        // than 512 words ready extends its window by at most 227 words.  Steps
        // must be separated by a barrier; they are placed in front of barriers
        // the tile needs anyway.
        auto mt_step = [&]() __attribute__((always_inline)) {
            if (gstep < 3u) {
                const uint32_t cnt = gstep < 2u ? 171u : 170u;
                if ((uint32_t)tid < cnt) {
                    const uint32_t w = (hslot ^ 512u) + 171u * gstep + (uint32_t)tid;       // never wraps
                    const uint32_t a = (w + 400u) & XRT_RMASK, c = (w + 797u) & XRT_RMASK;  // n - 624, n - 227
                    const uint32_t b = (w + 401u) & XRT_RMASK;                              // n - 623
#pragma unroll
                    for (int h = 0; h < 6; h++) {
                        if (h < nh) {
                            uint32_t* r = rings + h * XRT_RING;
                            r[w] = mt_mix(r[a], r[b], r[c]);
                        }
                    }
                }
                gstep++;
            }
            uint32_t avail = sgen - spos;
            if (avail < s_ahead) {
                uint32_t chunk = s_ahead - avail;
                if (chunk > 227u) chunk = 227u;
                if ((uint32_t)tid < chunk) {
                    uint32_t n = sgen + (uint32_t)tid;
                    stream[n & XRT_RMASK] = mt_mix(stream[(n - 624u) & XRT_RMASK], stream[(n - 623u) & XRT_RMASK],
                                                   stream[(n - 227u) & XRT_RMASK]);
                }
                sgen += chunk;
            }
        };

        // Walks the stream head to this unit's first Bragg uniform (seg_skip < chunk_words words) and, for a later part
        // of a segment, the source heads over the `tiles` tiles of the parts in front.  Raw generation only, a wave per
        // ring and without workgroup barriers (LDS operations of one wave execute in order; <= 227 words are independent,
        // 192 = three per lane are produced per step), at raised issue priority: the walk is a chain of LDS round trips.
        auto heads_skip = [&](int64_t tiles) __attribute__((always_inline)) {
            if (seg_skip == 0 && tiles == 0) return;
            lds_barrier();
            __builtin_amdgcn_s_setprio(3);
            const int wave = tid >> 6, lane = tid & 63;
            if (tiles > 0) {
                // (a head holds its words [0, 512) when a tile loop starts: word w in slot w & 1023)
                for (int h = wave; h < nh; h += 4) wave_walk_n(rings + h * XRT_RING, XRT_AHEAD, 512ull * (uint64_t)tiles, lane);
                hslot = (tiles & 1) ? 0u : 512u;
                gstep = 3;
            }
            if (seg_skip > 0) {
                const uint32_t have = sgen - spos;          // words generated beyond the position
                const uint64_t want = seg_skip + (uint64_t)XRT_AHEAD;
                if (want > (uint64_t)have) {
                    if (wave == 3) wave_walk_n(stream, sgen, want - (uint64_t)have, lane);
                    sgen += (uint32_t)(want - (uint64_t)have);
                }
                spos += (uint32_t)seg_skip; s_used += seg_skip; seg_skip = 0;
            }
            __builtin_amdgcn_s_setprio(SEG == 2 ? 2 : 0);
            lds_barrier();
        };
        auto stamp = [&](int k) __attribute__((always_inline)) {
#ifdef XRT_UNIT_CLOCKS
            if (SEG && args.dbg && tid == 0) args.dbg[(size_t)unit * 8 + k] = wall_clock64();
#else
            (void)k;
#endif
        };
        stamp(0);
        if constexpr (SEG == 2) __builtin_amdgcn_s_setprio(2);       // (first phase: above the units in their second phase)
        if constexpr (SEG != 0 && PH_A) heads_skip((ray_lo < ray_hi) ? (int64_t)sub * (args.sub_len / XRT_TILE) : 0);
        stamp(1);

        // A point source (no spatial array in use: every offset is -0 + 0 u = 0) has one origin per run;
        // it and what the first element derives from it alone are evaluated here instead of per ray
        const bool point = !EXT && (SRC.array_used & 7u) == 0u && SRC.kind != XRT_SRC_FOCUSED && SRC.kind != XRT_SRC_PLASMA &&
                           SC.n_optics > 0;
        V3 O_run;
        PointPre pre0;
        O_run.x = O_run.y = O_run.z = 0.0;
        pre0.L = O_run; pre0.LL = 0.0; pre0.num = 0.0;
        if (point) {
            O_run = source_origin(SRC, 0.0, 0.0, 0.0);
            pre0 = point_pre(SC.opt[0], O_run);
        }

        // Bragg angle shared by all rays when the wavelength is one constant
        const bool wl_shared = (SRC.wavelength_dist == XRT_WL_CONST) && !SRC.has_velocity;
        double bragg_shared = 0.0;
        if (wl_shared && be >= 0) bragg_shared = asin((1.0 * SRC.wavelength) / SC.opt[be].two_d);
        wl_run = 1.0 * SRC.wavelength;

        // Elements without a Bragg test, starting at element `e` with rays held by the threads
        // tid < n_in (`fresh` = the rays still have to be intersected with element e; otherwise
        // `alive` already says which of them left element e).  Runs up to (not including) the
        // Bragg element, where it returns true with the candidates' state for the caller to queue.
        // `scratch`: first free record of the circular buffer, used for the compactions.
        // `rot_in`: rotation the incoming rays are laid out with (0 for a source tile: ray order = thread order).
        auto plain_elements = [&](int e, bool fresh, uint32_t n_in, bool& have, bool& alive, Ray& ray, V3& X,
                                  uint32_t& id, int& aux, uint32_t scratch, uint32_t rot_in) __attribute__((always_inline)) -> int {
            for (; e < SC.n_optics && n_in > 0; e++) {
                scl = scene_fresh(scene_g);
                const KOptic& op = SC.opt[e];
                if (fresh) {
                    alive = false;
                    // (the EXT parts are discarded statements in the other variants: even dead code here
                    //  perturbs the register allocation of the lean kernel measurably)
                    if constexpr (EXT) {
                        const bool local = (op.flags & XRT_F_TRACE_LOCAL) != 0;
                        const bool is_mesh = (op.shape == XRT_SHAPE_MESH);
                        if (have) {
                            if (local) {    // TraceObject.trace_global -> ray_to_local (optics/_TraceObject.py:146-148)
                                ray.o = to_local(op.R, sub3(ray.o, ld3(op.origin)));
                                ray.d = to_local(op.R, ray.d);
                            }
                            bool hit = false;
                            if (is_mesh && e == be && mesh_pre && !counting) {
                                // queued with its (local-frame) origin and the face of the first pass; see mesh_pre
                                bool by_grid = false;
                                if constexpr (SEG == 3) {
                                    if (l_dpt) {
                                        aux = mesh_first_dir(op.mesh, l_dcells, l_dpt, ray.o.x, ray.o.y, ray.o.z, ray.d.x, ray.d.y, ray.d.z);
                                        by_grid = true;
                                    }
                                }
                                if (!by_grid) aux = mesh_first(op.mesh, ray.o.x, ray.o.y, ray.o.z, ray.d.x, ray.d.y, ray.d.z);      // (many faces: through the direction grid in global memory, if there is one)
                                hit = aux >= 0;
                                X = ray.o;
                                alive = hit;
                            } else {
                            bool whole_mesh = false;
                            if constexpr (!SPLIT) {        // (split phases: the Bragg element is the scene's only mesh)
                                if (is_mesh) {
                                    const MeshHit h = mesh_hit(op.mesh, ray.o.x, ray.o.y, ray.o.z, ray.d.x, ray.d.y, ray.d.z);
                                    hit = h.hit != 0; X.x = h.x; X.y = h.y; X.z = h.z; aux = h.aux;
                                    whole_mesh = true;
                                }
                            }
                            if (!whole_mesh) hit = intersect_point<FULL>(op, ray, X, false, pre0);
                            alive = hit && check_bounds<FULL>(op, X);
                            }
                            if (HIST && !alive && !counting) {
                                V3 xo = X, dd = ray.d;
                                if (!hit) { xo.x = xo.y = xo.z = __builtin_nan(""); }
                                if (local) {
                                    xo = to_external(op.R, xo);
                                    xo.x += op.origin[0]; xo.y += op.origin[1]; xo.z += op.origin[2];
                                    dd = to_external(op.R, dd);
                                }
                                hist_write(args.hist, args.hmask, N, e + 1, id, xo, dd, ray.wl, false);
                            }
                        }
                        if (e == be) return e;                   // candidates for the Bragg queue
                        if (alive) {
                            ray.o = X;
                            if (op.interact != XRT_INTERACT_NONE) {
                                V3 nrm;
                                if constexpr (!SPLIT) nrm = is_mesh ? mesh_normal(op.mesh, X.x, X.y, aux) : surface_normal<FULL>(op, X);
                                else nrm = surface_normal<FULL>(op, X);
                                double dt = dot_e(ray.d, nrm);
                                ray.d.x = ray.d.x - 2.0 * (dt * nrm.x);
                                ray.d.y = ray.d.y - 2.0 * (dt * nrm.y);
                                ray.d.z = ray.d.z - 2.0 * (dt * nrm.z);
                            }
                            if (local) {    // ray_to_external (optics/_TraceObject.py:152-154)
                                ray.o = to_external(op.R, ray.o);
                                ray.o.x += op.origin[0]; ray.o.y += op.origin[1]; ray.o.z += op.origin[2];
                                ray.d = to_external(op.R, ray.d);
                            }
                            if (HIST && !counting) hist_write(args.hist, args.hmask, N, e + 1, id, ray.o, ray.d, ray.wl, true);
                        }
                    } else {
                        if (have) {
                            // (e == 0 is only ever reached with rays straight from the source)
                            bool hit = intersect_point<FULL>(op, ray, X, point && e == 0, pre0);
                            alive = hit && check_bounds<FULL>(op, X);
                            if (HIST && !alive && !counting) {
                                V3 xo = X;
                                if (!hit) { xo.x = xo.y = xo.z = __builtin_nan(""); }
                                hist_write(args.hist, args.hmask, N, e + 1, id, xo, ray.d, ray.wl, false);
                            }
                        }
                        if (e == be) return e;                   // candidates for the Bragg queue
                        // InteractObject / InteractMirror.reflect_vectors (optics/_InteractMirror.py:29-42)
                        if (alive) {
                            ray.o = X;
                            if (op.interact != XRT_INTERACT_NONE) {
                                V3 nrm = surface_normal<FULL>(op, X);
                                double dt = dot_e(ray.d, nrm);
                                ray.d.x = ray.d.x - 2.0 * (dt * nrm.x);
                                ray.d.y = ray.d.y - 2.0 * (dt * nrm.y);
                                ray.d.z = ray.d.z - 2.0 * (dt * nrm.z);
                            }
                            if (HIST && !counting) hist_write(args.hist, args.hmask, N, e + 1, id, ray.o, ray.d, ray.wl, true);
                        }
                    }
                }
                fresh = true;
                // stable compaction of the survivors for the next element; the pixel of element e
                // is found after it, on dense lanes
                uint32_t n_out;
                mt_step();
                uint32_t rank = wg_rank<true>(alive, wave_tot, slot, tid, n_out, rot_in);
                (void)rank;
                if (tid == 0 && !counting) cnt[e + 1] += n_out;
                const bool image = (op.flags & XRT_F_IMAGE) && args.images && !counting;
                if constexpr (LBINS) {
                    // no compaction between the elements (and so no record buffer): the rays stay in their lanes
                    have = alive;
                    if (image && alive) pixel(op, ray.o);
                } else
                if (e + 1 < SC.n_optics && n_out > 0) {
                    if (alive) q_store(q_wrap(scratch + rank), ray.o, ray.d, ray.wl, id);
                    mt_step();
                    lds_barrier();
                    // the survivors move on to the waves behind the ones that held the incoming rays
                    rotw = (rotw + ((n_in + 63u) >> 6)) & 3u;
                    rot_in = rotw;
                    const uint32_t vt = vtid();
                    have = vt < n_out;
                    if (have) {
                        q_load(q_wrap(scratch + vt), ray.o, ray.d, ray.wl, id);
                        if (image) pixel(op, ray.o);
                    }
                    // the next write into the buffer happens behind the next scan's barrier
                } else if (image && alive) {
                    pixel(op, ray.o);
                }
                n_in = n_out;
            }
            return -1;
        };

        // The survivors of the Bragg test wait in a second queue of the same circular buffer (bcount records
        // right below the candidates' head) until 64 of them are there: the elements behind the crystal then
        // run on a full wave once per ~64 reflected rays instead of on a handful of lanes after every batch
        // (that stage is a long dependent chain -- pixel index, plane intersection, bounds, pixel index --
        // during which the other waves of the workgroup sit at barriers).  Their order does not matter:
        // counters and pixel bins are sums, history rows are written by ray index.
        auto drain_survivors = [&](bool all) __attribute__((always_inline)) {
            while (bcount >= 64u || (all && bcount > 0u)) {
                const uint32_t n = bcount < 64u ? bcount : 64u;
                mt_step();
                lds_barrier();                                  // survivor records visible
                scl = scene_fresh(scene_g);
                const KOptic& opb = SC.opt[be];
                const uint32_t vt = vtid();
                bool have_b = vt < n, alive_b = false;
                Ray rb;
                V3 Xb;
                uint32_t idb = 0;
                int auxb = 0;
                Xb.x = Xb.y = Xb.z = 0.0;
                rb.o = Xb; rb.d = Xb; rb.wl = 0.0;
                if (have_b) {
                    q_load(q_wrap(qhead + qcap - bcount + vt), rb.o, rb.d, rb.wl, idb);
                    if constexpr (!EXT) {
                        // InteractCrystal.interact -> reflect_vectors (optics/_InteractMirror.py:29-42), deferred from the Bragg stage
                        const V3 nrm = surface_normal<FULL>(opb, rb.o);
                        const double dt = dot_e(rb.d, nrm);
                        rb.d.x = rb.d.x - 2.0 * (dt * nrm.x);
                        rb.d.y = rb.d.y - 2.0 * (dt * nrm.y);
                        rb.d.z = rb.d.z - 2.0 * (dt * nrm.z);
                        if (HIST) hist_write(args.hist, args.hmask, N, be + 1, idb, rb.o, rb.d, rb.wl, true);
                    }
                    if ((opb.flags & XRT_F_IMAGE) && args.images) pixel(opb, rb.o);
                }
                bcount -= n;
                plain_elements(be + 1, true, n, have_b, alive_b, rb, Xb, idb, auxb, q_wrap(qhead + qcount), rotw);
                rotw = (rotw + 1u) & 3u;
            }
        };

        // ---- Bragg test of a batch (128 or 256) of candidates in ray order: the first nb records of the queue
        auto bragg_batch = [&](uint32_t nb, bool drain_all) __attribute__((always_inline)) {
            scl = scene_fresh(scene_g);
            const KOptic& op = SC.opt[be];
            // n uniforms from the stream head (np.random.uniform(0,1,n_live), optics/_InteractCrystal.py:189)
            while ((sgen - spos) < 2u * nb) { mt_step(); lds_barrier(); }
            mt_step();
            lds_barrier();                                  // queue records visible
            const uint32_t brot = rotw;
            const uint32_t vt = vtid();
            bool have = vt < nb, alive = false;
            Ray ray;
            V3 X;
            uint32_t id = 0;
            X.x = X.y = X.z = 0.0;
            ray.o = X; ray.d = X; ray.wl = 0.0;
            uint32_t n_draws = nb;       // Bragg uniforms the batch consumes: one per record (mesh_pre: per ray left)
            if constexpr (EXT) {
                const bool local = (op.flags & XRT_F_TRACE_LOCAL) != 0;
                uint32_t draw = vt;           // which of the batch's Bragg uniforms is this ray's
                int baux = 0;
                if (have) {
                    q_load(q_wrap(qhead + vt), X, ray.d, ray.wl, id);
                    baux = (int)qaux[q_wrap(qhead + vt)];
                }
                if (mesh_pre) {
                    // the records are rays that hit a face in the first pass: the rest of ShapeMesh.intersect and the
                    // bounds here, on dense lanes; the Bragg uniforms go to those that are left, in ray order
                    bool cand = false;
                    if (have) {
                        ray.o = X;
                        const MeshHit h = mesh_rest(op.mesh, ray.o.x, ray.o.y, ray.o.z, ray.d.x, ray.d.y, ray.d.z, baux);
                        X.x = h.x; X.y = h.y; X.z = h.z; baux = h.aux;
                        cand = (h.hit != 0) && check_bounds<FULL>(op, X);
                        if (HIST && !cand) {
                            V3 xo = X, dd = ray.d;
                            if (h.hit == 0) { xo.x = xo.y = xo.z = __builtin_nan(""); }
                            if (local) {
                                xo = to_external(op.R, xo);
                                xo.x += op.origin[0]; xo.y += op.origin[1]; xo.z += op.origin[2];
                                dd = to_external(op.R, dd);
                            }
                            hist_write(args.hist, args.hmask, N, be + 1, id, xo, dd, ray.wl, false);
                        }
                    }
                    draw = wg_rank<true>(cand, wave_tot, slot, tid, n_draws, brot);
                    have = cand;
                }
                if (have) {
                    V3 nrm;
                    if (op.shape == XRT_SHAPE_MESH) nrm = mesh_normal(op.mesh, X.x, X.y, baux);
                    else nrm = surface_normal<FULL>(op, X);
                    uint32_t n = spos + 2u * draw;
                    double test = 0.0 + (1.0 - 0.0) * mt_double(stream[n & XRT_RMASK], stream[(n + 1u) & XRT_RMASK]);
                    alive = bragg_accept(op, ray, nrm, test, wl_shared, bragg_shared);
                    if (HIST && !alive) {
                        V3 xo = X, dd = ray.d;
                        if (local) {
                            xo = to_external(op.R, xo);
                            xo.x += op.origin[0]; xo.y += op.origin[1]; xo.z += op.origin[2];
                            dd = to_external(op.R, dd);
                        }
                        hist_write(args.hist, args.hmask, N, be + 1, id, xo, dd, ray.wl, false);
                    }
                    if (alive) {
                        ray.o = X;
                        double dt = dot_e(ray.d, nrm);
                        ray.d.x = ray.d.x - 2.0 * (dt * nrm.x);
                        ray.d.y = ray.d.y - 2.0 * (dt * nrm.y);
                        ray.d.z = ray.d.z - 2.0 * (dt * nrm.z);
                        if (local) {
                            ray.o = to_external(op.R, ray.o);
                            ray.o.x += op.origin[0]; ray.o.y += op.origin[1]; ray.o.z += op.origin[2];
                            ray.d = to_external(op.R, ray.d);
                        }
                        if (HIST) hist_write(args.hist, args.hmask, N, be + 1, id, ray.o, ray.d, ray.wl, true);
                    }
                }
            } else {
                // The reflection itself (exact normal: a square root and three divisions) waits until the
                // survivors are drained, on dense lanes; here the screen decides nearly every candidate from
                // the incidence cosine alone, |d . nu| / (|nu| |d|) with the un-normalised normal direction nu of a sphere.
                if (have) {
                    q_load(q_wrap(qhead + vt), X, ray.d, ray.wl, id);
                    uint32_t n = spos + 2u * vt;
                    double test = 0.0 + (1.0 - 0.0) * mt_double(stream[n & XRT_RMASK], stream[(n + 1u) & XRT_RMASK]);
                    bool decided = false;
                    V3 nu;
                    if (test > 0.0 && ((op.scr_ok && wl_shared) || op.scr2_ok) && normal_direction(op, X, nu)) {
                        // (a plane's normal is the optic's z axis as the reference holds it -- |z| - 1 ~ 1e-9 for axes typed in
                        //  with 8 digits -- and the reference does NOT divide by its length: neither does the screen)
                        const double pp = (op.shape == XRT_SHAPE_PLANE ? 1.0 : dot_n(nu, nu)) * dot_n(ray.d, ray.d);
                        double y = __builtin_amdgcn_rsq(pp);
                        y = y * fma(-0.5 * pp * y, y, 1.5);
                        const double ca = fabs(dot_n(ray.d, nu)) * y;
                        alive = (op.scr_ok && wl_shared) ? bragg_screen(op, ca, test, decided)
                                                         : bragg_screen_wl(op, ca, ray.wl, test, decided);
                    }
                    if (!decided) alive = bragg_accept(op, ray, surface_normal<FULL>(op, X), test, wl_shared, bragg_shared);
#if XRT_ABLATE == 1
                    alive = alive && (id == 0xffffffffu);
#endif
                    if (HIST && !alive) hist_write(args.hist, args.hmask, N, be + 1, id, X, ray.d, ray.wl, false);
                    if (alive) ray.o = X;
                }
            }
            spos += 2u * n_draws;
            s_used += 2ull * n_draws;
            const uint32_t head_old = qhead;
            qhead = q_wrap(qhead + nb);
            qcount -= nb;
            // ---- the reflected rays join the survivor queue, which sits right below the candidates' head:
            // [qhead - bcount, qhead).  The batch just freed [head_old, qhead); the new survivors go to its
            // upper end by ordered rank, and the (< 64) survivors left from earlier batches are moved up
            // behind them by a wave that has nothing else to do in this stage (virtual wave 3): it reads
            // them in front of the scan's barrier and writes them behind it.
            V3 mo, md;
            double mwl = 0.0;
            uint32_t mid = 0;
            mo.x = mo.y = mo.z = 0.0; md = mo;
            const bool mover = (vt >= 192u) && (vt - 192u < bcount);
            if (mover) q_load(q_wrap(head_old + qcap - bcount + (vt - 192u)), mo, md, mwl, mid);
            uint32_t n_s;
            mt_step();
            const uint32_t srank = wg_rank<true>(alive, wave_tot, slot, tid, n_s, brot);
            if (tid == 0) cnt[be + 1] += n_s;
            if (alive) q_store(q_wrap(qhead + qcap - n_s + srank), ray.o, ray.d, ray.wl, id);
            if (mover && n_s < nb) q_store(q_wrap(qhead + qcap - n_s - bcount + (vt - 192u)), mo, md, mwl, mid);
            bcount += n_s;
            rotw = (rotw + ((nb + 63u) >> 6)) & 3u;
            // ---- survivors: 64 at a time (everything that is left behind the run's last batch) through
            // the elements behind the Bragg element, on one full wave
            drain_survivors(drain_all);
        };

        // ---- tiles of 256 rays in original order ----------------------------
        for (int64_t i0 = ray_lo; PH_A && i0 < ray_hi; i0 += XRT_TILE) {
            const int64_t left = ray_hi - i0;
            const uint32_t n_tile = left < XRT_TILE ? (uint32_t)left : (uint32_t)XRT_TILE;
            scl = scene_fresh(scene_g);
            if constexpr (LBINS) {
                if (lb_tiles == 255u) lbins_flush();       // 255 tiles x 256 rays: no 16-bit counter can have wrapped
                lb_tiles++;
            }
            if (args.progress) {
                if ((tiles_done & 31u) == 31u && tid == 0) {
                    const unsigned long long all = atomicAdd(args.progress, 32ULL) + 32ULL;
                    const long long ahead = (long long)(tiles_done + 1u) - (long long)(all / n_wg);
                    bcast[1] = ahead > 48 ? 0u : (ahead < -48 ? 2u : 1u);
                }
                if ((tiles_done & 31u) == 1u && tiles_done > 32u) {       // (a barrier lies between the two)
                    // (one-pass units: a step higher, above the units that are in their second phase and wait for the first
                    //  phase of the ones in front of them)
                    const uint32_t pr = uni32(bcast[1]);
                    if (pr == 0u) __builtin_amdgcn_s_setprio(SEG == 2 ? 1 : 0);
                    else if (pr == 1u) __builtin_amdgcn_s_setprio(SEG == 2 ? 2 : 1);
                    else __builtin_amdgcn_s_setprio(SEG == 2 ? 3 : 2);
                }
                tiles_done++;
            }

            // everything this tile consumes must be generated: normally already
            // done behind the previous tile's barriers
            while (gstep < 3u) { mt_step(); lds_barrier(); }
            hslot ^= 512u;

            double u[6];
            {
                int h = 0;
                const uint32_t pair = (hslot >> 1) + (uint32_t)tid;     // words hslot + 2 tid, + 1: one 8-byte read
#pragma unroll
                for (int k = 0; k < 6; k++) {
                    u[k] = 0.0;
                    if ((SRC.array_used >> k) & 1u) {
                        const uint2 w = reinterpret_cast<const uint2*>(rings + h * XRT_RING)[pair];
                        u[k] = mt_double(w.x, w.y);
                        h++;
                    }
                }
            }
            gstep = 0;              // mt_step now fills the other half with the next tile's words

            Ray ray;
            V3 X;
            X.x = X.y = X.z = 0.0;
            uint32_t id = (uint32_t)(i0 + tid);
            int aux = 0;
            bool have = (uint32_t)tid < n_tile, alive = false;
            if constexpr (SEG && FULL) {
                if (args.wl_array) u[5] = have ? args.wl_array[(size_t)run * (size_t)N + (size_t)id] : 0.0;
            }
#if XRT_ABLATE >= 4
            have = have && (u[3] + u[4] < -1.0);
            if (have)
#endif
            if (point) ray.o = O_run;
            if constexpr (FULL && !SEG) {
                if (plasma) {
                    if (have) plasma_ray(SRC, args.plasma, (size_t)run, id, ray);
                    else { ray.o.x = ray.o.y = ray.o.z = 0.0; ray.d = ray.o; ray.wl = 0.0; }
                } else source_ray<FULL>(SRC, u, ray, point);
            } else source_ray<FULL>(SRC, u, ray, point);
#if XRT_ABLATE == 3
            have = have && (id == 0xffffffffu);
#endif
            if constexpr (SEG && LEANWL) ray.wl = have ? args.wl_array[(size_t)run * (size_t)N + (size_t)id] : 0.0;
            if (tid == 0 && !counting) cnt[0] += n_tile;
            if (HIST && have && !counting) hist_write(args.hist, args.hmask, N, 0, id, ray.o, ray.d, ray.wl, true);

            // elements in config order (objects/_Dispatcher.py:166-196) up to the Bragg element
            if (counting) {
                // mode 1: only the number of rays that reach the Bragg test matters
                const int stop = plain_elements(0, true, n_tile, have, alive, ray, X, id, aux, 0u, 0u);
                if (stop >= 0) {
                    uint32_t n_a;
                    mt_step();
                    wg_rank<true>(alive, wave_tot, slot, tid, n_a);
                    n_candidates += n_a;
                }
                continue;
            }
            const int stop = plain_elements(0, true, n_tile, have, alive, ray, X, id, aux, q_wrap(qhead + qcount), 0u);
#if XRT_ABLATE == 2
            alive = alive && (id == 0xffffffffu);
#endif
            if (stop >= 0) {
                // queue the candidates in ray order: ordered live rank -> FIFO position
                // (a locally traced element queues its local-frame point and direction)
                // (elements in front of the Bragg element may have moved the rays to rotated waves)
                const uint32_t rot_here = (stop > 0) ? rotw : 0u;
                uint32_t n_a;
                mt_step();
                uint32_t rank = wg_rank<true>(alive, wave_tot, slot, tid, n_a, rot_here);
                if constexpr (SEG >= 2) {
                    if (alive) cand_store(ray_lo + (int64_t)(n_candidates + rank), X, ray.d, ray.wl, id, aux);
                    if constexpr (SEG == 3) {
                        if (args.unit_o && alive && n_candidates + rank == 0u) {        // (the unit's first parked ray)
                            double* uo = args.unit_o + 3 * (size_t)unit;
                            uo[0] = X.x; uo[1] = X.y; uo[2] = X.z;
                        }
                    }
                    n_candidates += n_a;
                } else if constexpr (EXT) {
                    if (alive) {
                        q_store(q_wrap(qhead + qcount + rank), X, ray.d, ray.wl, id);
                        qaux[q_wrap(qhead + qcount + rank)] = (uint32_t)aux;
                    }
                    qcount += n_a;
                } else {
                    if (alive) q_store(q_wrap(qhead + qcount + rank), X, ray.d, ray.wl, id);
                    qcount += n_a;
                }
            }

            // ---- Bragg test, a batch (128 or 256) of queued rays at a time ------------------
            if constexpr (SEG < 2) {
                const bool last_tile = (i0 + XRT_TILE >= ray_hi);
                while (be >= 0 && (qcount >= bbatch || (last_tile && qcount > 0u))) {
                    const uint32_t nb = qcount < bbatch ? qcount : bbatch;
                    bragg_batch(nb, last_tile && qcount == nb);
                }
                if (be >= 0 && last_tile) drain_survivors(true);
            }
        }

        // ---- SEG == 2: the unit's parked candidates.  Publish their number, wait for the numbers of the units in front
        // (handed out earlier by the dispenser: under way or done, and waiting for nothing behind them), open the stream
        // head at the position their sum gives, then Bragg test + the elements behind it, a batch at a time.
        if constexpr (SEG >= 2) {
            if (be >= 0) {
                unsigned long long* acc = reinterpret_cast<unsigned long long*>(bcast + 2);
                stamp(2);
                __builtin_amdgcn_s_setprio(0);
                if constexpr (PH_A) {
                    __syncthreads();                                // (the candidates of the last tile are written: global stores complete)
                    if (tid == 0) __hip_atomic_store(&args.unit_flag[unit], n_candidates + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
              if constexpr (PH_B) {
                // (split phases: the launches in front are complete -- the unit's candidates as its first phase counted them,
                //  the Bragg draws of a unit = the candidates xrt_mesh_rest_kernel left alive)
                if constexpr (SEG == 4) n_candidates = uni32(args.unit_flag[unit]) - 1u;
                uint32_t n_unit_draws = n_candidates;
                if constexpr (SEG == 4) n_unit_draws = uni32(args.unit_alive[unit]);
                if (tid == 0) *acc = 0ULL;
                lds_barrier();
                unsigned long long part = 0;
                for (uint32_t q = (uint32_t)tid; q < uidx; q += XRT_TILE) {
                    if constexpr (SEG == 4) part += (unsigned long long)args.unit_alive[(size_t)run * upr + q];
                    else {
                        const uint32_t* f = &args.unit_flag[(size_t)run * upr + q];
                        uint32_t v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        while (v == 0u) {
                            __builtin_amdgcn_s_sleep(8);
                            v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        part += (unsigned long long)(v - 1u);
                    }
                }
                if (part) atomicAdd(acc, part);
                lds_barrier();
                const unsigned long long before = uni64(*acc);
                stamp(3);
#ifdef XRT_UNIT_CLOCKS
                if (args.dbg && tid == 0) args.dbg[(size_t)unit * 8 + 7] = ((unsigned long long)n_candidates << 32) | (before & 0xffffffffull);
#endif
                // ---- second phase, wave by wave and without workgroup barriers (a barrier in front of every dependent step of
                // a 256-candidate batch -- uniforms, test, survivor scan, drain -- left the vector units idle two thirds of the
                // time).  Each of the four waves takes a quarter of the unit's candidates (a little less for the later ones, which
                // first walk their copy of the stream head's ring over the quarters in front: raw generation, 192 words a step),
                // 64 at a time, and keeps its own queue of reflected rays, which go through the elements behind the crystal 64 at
                // a time.
                if (n_candidates > 0u || last_unit) {
                    const uint64_t words = 2ull * before + (args.base_words ? uni64(args.base_words[run]) : 0ull);
                    const uint64_t chunk = words / (uint64_t)args.chunk_words;
                    const uint64_t skip = words - chunk * (uint64_t)args.chunk_words;
                    const KStream* ch = args.chunk_heads + (size_t)run * args.run_stride + chunk;
                    const uint64_t c_next = uni64(ch->next), c_gen = uni64(ch->gen);
                    const uint64_t first = c_next + skip;                          // the unit's first Bragg uniform (stream word)
                    const int wave = tid >> 6, lane = tid & 63;
                    // the workgroup's LDS in front of `small`, re-carved: four rings, four survivor queues
                    uint32_t* wring = reinterpret_cast<uint32_t*>(lds_raw) + wave * XRT_RING;
                    // (a queued ray: point, direction[, wavelength] -- whatever the parked records hold, see KArgs.split_lean)
                    const uint32_t wq_comps = q_has_wl ? 7u : 6u;
                    const uint32_t rec_bytes = 8u * wq_comps + (HIST ? 4u : 0u);
                    const uint32_t pool = ((uint32_t)(reinterpret_cast<unsigned char*>(small) - lds_raw) - 4u * 4u * XRT_RING) / 4u & ~7u;
                    uint32_t wq_cap = pool / rec_bytes;
                    if (wq_cap > 128u) wq_cap = 128u;
                    wq_cap &= ~1u;
                    double* wq = reinterpret_cast<double*>(lds_raw + 4u * 4u * XRT_RING + (size_t)wave * pool);       // [wq_comps][wq_cap]
                    uint32_t* wqid = reinterpret_cast<uint32_t*>(wq + (size_t)wq_comps * wq_cap);                        // [wq_cap] (HIST)
                    // batches of 64 candidates [kb, ke) of this wave: 26 / 25.5 / 24.5 / 24 % of them
                    const uint32_t n_batches = (n_candidates + 63u) >> 6;
                    const uint32_t cut[5] = {0u, (n_batches * 133u + 256u) >> 9, (n_batches * 264u + 256u) >> 9, (n_batches * 389u + 256u) >> 9, n_batches};
                    const uint32_t kb = cut[wave], ke = cut[wave + 1];
                    uint64_t g64 = c_gen;       // words of the stream this wave's ring holds: [g64 - 1024, g64)
                    auto ring_to = [&](uint64_t need) __attribute__((always_inline)) {      // whole steps of 192 words
                        if (g64 < need) {
                            const uint64_t steps = (need - g64 + 191ull) / 192ull;
                            wave_walk_n(wring, (uint32_t)g64, 192ull * steps, lane);
                            g64 += 192ull * steps;
                        }
                    };
                    lds_barrier();              // (everyone is done with the first phase's rings and record buffer)
                    // wave 0 walks the chunk head to the unit's first uniform, the others take copies and walk on to theirs
                    if (wave == 0) {
                        __builtin_amdgcn_s_setprio(3);
                        for (int i = lane; i < (int)XRT_RING; i += 64) wring[i] = ch->ring[i];
                        wave_fence();
                        ring_to(first + 128ull);
                        if (lane == 0) *acc = g64;
                        __builtin_amdgcn_s_setprio(0);
                    }
                    lds_barrier();
                    if (wave != 0) {
                        const uint32_t* r0 = reinterpret_cast<const uint32_t*>(lds_raw);
                        for (int i = lane; i < (int)XRT_RING; i += 64) wring[i] = r0[i];
                        g64 = uni64(*acc);
                        wave_fence();
                    }
                    lds_barrier();
                    // the wave's first Bragg uniform: one per candidate in front (split phases: per candidate left alive in front)
                    uint64_t upos = first + 128ull * (uint64_t)kb;
                    if constexpr (SEG == 4) {
                        const uint32_t* ba = args.batch_alive + ((crun + (size_t)ray_lo) >> 6);
                        uint32_t in_front = 0;
                        for (uint32_t q = (uint32_t)lane; q < kb; q += 64u) in_front += ba[q];
                        for (int o = 32; o > 0; o >>= 1) in_front += __shfl_xor(in_front, o);
                        upos = first + 2ull * (uint64_t)uni32(in_front);
                    }
                    if (kb < ke) ring_to(upos + 128ull);
                    uint32_t qn = 0;            // records in this wave's queue
                    scl = scene_fresh(scene_g);
                    const bool img_b = (SC.opt[be].flags & XRT_F_IMAGE) && args.images;
                    // 64 (or the last qn) queued rays: the crystal's reflection where it was deferred, its pixel, the elements behind
                    auto wave_drain = [&]() __attribute__((always_inline)) {
                        const uint32_t n = qn < 64u ? qn : 64u;
                        qn -= n;
                        bool have_b = (uint32_t)lane < n;
                        Ray rb;
                        V3 Xb;
                        uint32_t idb = 0;
                        int auxb = 0;
                        Xb.x = Xb.y = Xb.z = 0.0;
                        rb.o = Xb; rb.d = Xb; rb.wl = wl_run;
                        scl = scene_fresh(scene_g);
                        if (have_b) {
                            const double* r = wq + qn + lane;
                            rb.o.x = r[0 * wq_cap]; rb.o.y = r[1 * wq_cap]; rb.o.z = r[2 * wq_cap];
                            rb.d.x = r[3 * wq_cap]; rb.d.y = r[4 * wq_cap]; rb.d.z = r[5 * wq_cap];
                            if (q_has_wl) rb.wl = r[6 * wq_cap];
                            if (HIST) idb = wqid[qn + lane];
                            const KOptic& opb = SC.opt[be];
                            if constexpr (!EXT) {
                                // InteractCrystal.interact -> reflect_vectors (optics/_InteractMirror.py:29-42), deferred from the test
                                const V3 nrm = surface_normal<FULL>(opb, rb.o);
                                const double dt = dot_e(rb.d, nrm);
                                rb.d.x = rb.d.x - 2.0 * (dt * nrm.x);
                                rb.d.y = rb.d.y - 2.0 * (dt * nrm.y);
                                rb.d.z = rb.d.z - 2.0 * (dt * nrm.z);
                                if (HIST) hist_write(args.hist, args.hmask, N, be + 1, idb, rb.o, rb.d, rb.wl, true);
                            }
                            if (img_b) pixel(opb, rb.o);
                        }
                        // elements behind the crystal (objects/_Dispatcher.py:166-196), the rays staying in their lanes
                        for (int e = be + 1; e < SC.n_optics; e++) {
                            scl = scene_fresh(scene_g);
                            const KOptic& op = SC.opt[e];
                            bool alive_b = false;
                            if constexpr (EXT) {
                                const bool local = (op.flags & XRT_F_TRACE_LOCAL) != 0;
                                const bool is_mesh = (op.shape == XRT_SHAPE_MESH);
                                if (have_b) {
                                    if (local) {    // TraceObject.trace_global -> ray_to_local (optics/_TraceObject.py:146-148)
                                        rb.o = to_local(op.R, sub3(rb.o, ld3(op.origin)));
                                        rb.d = to_local(op.R, rb.d);
                                    }
                                    bool hit = false, whole_mesh = false;
                                    if constexpr (!SPLIT) {        // (split phases: the Bragg element is the scene's only mesh)
                                        if (is_mesh) {
                                            const MeshHit h = mesh_hit(op.mesh, rb.o.x, rb.o.y, rb.o.z, rb.d.x, rb.d.y, rb.d.z);
                                            hit = h.hit != 0; Xb.x = h.x; Xb.y = h.y; Xb.z = h.z; auxb = h.aux;
                                            whole_mesh = true;
                                        }
                                    }
                                    if (!whole_mesh) hit = intersect_point<FULL>(op, rb, Xb, false, pre0);
                                    alive_b = hit && check_bounds<FULL>(op, Xb);
                                    if (HIST && !alive_b) {
                                        V3 xo = Xb, dd = rb.d;
                                        if (!hit) { xo.x = xo.y = xo.z = __builtin_nan(""); }
                                        if (local) {
                                            xo = to_external(op.R, xo);
                                            xo.x += op.origin[0]; xo.y += op.origin[1]; xo.z += op.origin[2];
                                            dd = to_external(op.R, dd);
                                        }
                                        hist_write(args.hist, args.hmask, N, e + 1, idb, xo, dd, rb.wl, false);
                                    }
                                    if (alive_b) {
                                        rb.o = Xb;
                                        if (op.interact != XRT_INTERACT_NONE) {
                                            V3 nrm;
                                            if constexpr (!SPLIT) nrm = is_mesh ? mesh_normal(op.mesh, Xb.x, Xb.y, auxb) : surface_normal<FULL>(op, Xb);
                                            else nrm = surface_normal<FULL>(op, Xb);
                                            double dt = dot_e(rb.d, nrm);
                                            rb.d.x = rb.d.x - 2.0 * (dt * nrm.x);
                                            rb.d.y = rb.d.y - 2.0 * (dt * nrm.y);
                                            rb.d.z = rb.d.z - 2.0 * (dt * nrm.z);
                                        }
                                        if (local) {    // ray_to_external (optics/_TraceObject.py:152-154)
                                            rb.o = to_external(op.R, rb.o);
                                            rb.o.x += op.origin[0]; rb.o.y += op.origin[1]; rb.o.z += op.origin[2];
                                            rb.d = to_external(op.R, rb.d);
                                        }
                                        if (HIST) hist_write(args.hist, args.hmask, N, e + 1, idb, rb.o, rb.d, rb.wl, true);
                                    }
                                }
                            } else {
                                if (have_b) {
                                    bool hit = intersect_point<FULL>(op, rb, Xb, false, pre0);
                                    alive_b = hit && check_bounds<FULL>(op, Xb);
                                    if (HIST && !alive_b) {
                                        V3 xo = Xb;
                                        if (!hit) { xo.x = xo.y = xo.z = __builtin_nan(""); }
                                        hist_write(args.hist, args.hmask, N, e + 1, idb, xo, rb.d, rb.wl, false);
                                    }
                                    // InteractObject / InteractMirror.reflect_vectors (optics/_InteractMirror.py:29-42)
                                    if (alive_b) {
                                        rb.o = Xb;
                                        if (op.interact != XRT_INTERACT_NONE) {
                                            V3 nrm = surface_normal<FULL>(op, Xb);
                                            double dt = dot_e(rb.d, nrm);
                                            rb.d.x = rb.d.x - 2.0 * (dt * nrm.x);
                                            rb.d.y = rb.d.y - 2.0 * (dt * nrm.y);
                                            rb.d.z = rb.d.z - 2.0 * (dt * nrm.z);
                                        }
                                        if (HIST) hist_write(args.hist, args.hmask, N, e + 1, idb, rb.o, rb.d, rb.wl, true);
                                    }
                                }
                            }
                            const unsigned long long ob = __ballot(alive_b);
                            if (ob == 0ULL) break;
                            if (lane == 0) atomicAdd(&cnt[e + 1], (unsigned long long)__popcll(ob));
                            if (alive_b && (op.flags & XRT_F_IMAGE) && args.images) pixel(op, rb.o);
                            have_b = alive_b;
                        }
                    };

                    // ---- this wave's batches: candidates 64 k .. 64 k + 63, k = wave, wave + 4, ...
                    V3 pf_x, pf_d, pf_n;
                    double pf_wl = wl_run;
                    uint32_t pf_id = 0;
                    int pf_aux = 0;
                    pf_x.x = pf_x.y = pf_x.z = 0.0; pf_d = pf_x; pf_n = pf_x;
                    if (kb < ke && 64u * kb + (uint32_t)lane < n_candidates) {
                        cand_load(ray_lo + (int64_t)(64u * kb + (uint32_t)lane), pf_x, pf_d, pf_wl, pf_id, pf_aux);
                        if constexpr (SPLIT) pf_n = cand_normal(ray_lo + (int64_t)(64u * kb + (uint32_t)lane), pf_aux);
                    }
                    for (uint32_t k = kb; k < ke; k++) {
                        const uint32_t c = 64u * k + (uint32_t)lane;
                        bool have = c < n_candidates;
                        bool alive = false;
                        Ray ray;
                        V3 X = pf_x;
                        uint32_t id = pf_id;
                        int baux = pf_aux;
                        ray.o = X; ray.d = pf_d; ray.wl = pf_wl;
                        const V3 nrm_rec = pf_n;
                        // which of the batch's uniforms is this candidate's (split phases: the candidates left alive draw, in ray order)
                        uint32_t draw = (uint32_t)lane, n_draws = 64u;
                        if constexpr (SEG == 4) {
                            have = have && (uint32_t)baux != XRT_CAND_DEAD;
                            const unsigned long long hb = __ballot(have);
                            draw = __builtin_amdgcn_mbcnt_hi((uint32_t)(hb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)hb, 0u));
                            n_draws = (uint32_t)__popcll(hb);
                        }
#ifndef XRT_ABL_NOLOAD
                        if (k + 1u < ke && c + 64u < n_candidates) {       // the wave's next batch
                            cand_load(ray_lo + (int64_t)(c + 64u), pf_x, pf_d, pf_wl, pf_id, pf_aux);
                            if constexpr (SPLIT) pf_n = cand_normal(ray_lo + (int64_t)(c + 64u), pf_aux);
                        }
#endif
                        // this batch's uniforms: words upos .. upos + 2 n_draws (np.random.uniform(0,1,n_live), optics/_InteractCrystal.py:189)
#ifndef XRT_ABL_NORING
                        ring_to(upos + 128ull);
#endif
                        scl = scene_fresh(scene_g);
                        const KOptic& op = SC.opt[be];
#ifdef XRT_ABL_NOTEST
                        if (have && id == 0xffffffffu) {
#else
                        if (have) {
#endif
                            const uint32_t n = (uint32_t)upos + 2u * draw;
                            const double test = 0.0 + (1.0 - 0.0) * mt_double(wring[n & XRT_RMASK], wring[(n + 1u) & XRT_RMASK]);
                            if constexpr (EXT) {
                                const bool local = (op.flags & XRT_F_TRACE_LOCAL) != 0;
                                V3 nrm;
                                if constexpr (SPLIT) nrm = nrm_rec;
                                else {
                                    if (op.shape == XRT_SHAPE_MESH) nrm = mesh_normal(op.mesh, X.x, X.y, baux);
                                    else nrm = surface_normal<FULL>(op, X);
                                }
                                alive = bragg_accept(op, ray, nrm, test, wl_shared, bragg_shared);
                                if (HIST && !alive) {
                                    V3 xo = X, dd = ray.d;
                                    if (local) {
                                        xo = to_external(op.R, xo);
                                        xo.x += op.origin[0]; xo.y += op.origin[1]; xo.z += op.origin[2];
                                        dd = to_external(op.R, dd);
                                    }
                                    hist_write(args.hist, args.hmask, N, be + 1, id, xo, dd, ray.wl, false);
                                }
                                if (alive) {
                                    if constexpr (SPLIT) {
                                        // (a mesh that is not interpolated: the record holds the ray's origin -- its hit point on the
                                        //  face, formed as the middle launches formed it)
                                        if (!args.split_interp) {
                                            if (args.unit_o) { const double* uo = args.unit_o + 3 * (size_t)unit; ray.o.x = uo[0]; ray.o.y = uo[1]; ray.o.z = uo[2]; }
                                            X = mesh_hit_point(op.mesh, baux, ray);
                                        }
                                    }
                                    ray.o = X;
                                    double dt = dot_e(ray.d, nrm);
                                    ray.d.x = ray.d.x - 2.0 * (dt * nrm.x);
                                    ray.d.y = ray.d.y - 2.0 * (dt * nrm.y);
                                    ray.d.z = ray.d.z - 2.0 * (dt * nrm.z);
                                    if (local) {
                                        ray.o = to_external(op.R, ray.o);
                                        ray.o.x += op.origin[0]; ray.o.y += op.origin[1]; ray.o.z += op.origin[2];
                                        ray.d = to_external(op.R, ray.d);
                                    }
                                    if (HIST) hist_write(args.hist, args.hmask, N, be + 1, id, ray.o, ray.d, ray.wl, true);
                                }
                            } else {
                                // (the screen decides nearly every candidate from the incidence cosine alone; see bragg_batch)
                                bool decided = false;
                                V3 nu;
                                if (test > 0.0 && ((op.scr_ok && wl_shared) || op.scr2_ok) && normal_direction(op, X, nu)) {
                                    const double pp = (op.shape == XRT_SHAPE_PLANE ? 1.0 : dot_n(nu, nu)) * dot_n(ray.d, ray.d);
                                    double y = __builtin_amdgcn_rsq(pp);
                                    y = y * fma(-0.5 * pp * y, y, 1.5);
                                    const double ca = fabs(dot_n(ray.d, nu)) * y;
                                    alive = (op.scr_ok && wl_shared) ? bragg_screen(op, ca, test, decided)
                                                                     : bragg_screen_wl(op, ca, ray.wl, test, decided);
                                }
                                if (!decided) alive = bragg_accept(op, ray, surface_normal<FULL>(op, X), test, wl_shared, bragg_shared);
                                if (HIST && !alive) hist_write(args.hist, args.hmask, N, be + 1, id, X, ray.d, ray.wl, false);
                                if (alive) ray.o = X;
                            }
                        }
#ifdef XRT_ABL_NOSURV
                        alive = alive && (id == 0xffffffffu);
#endif
                        upos += 2ull * (uint64_t)n_draws;
                        // the reflected rays join the wave's queue
                        const unsigned long long sb = __ballot(alive);
                        if (sb != 0ULL) {
                            const uint32_t n_new = (uint32_t)__popcll(sb);
                            if (qn + n_new > wq_cap) wave_drain();
                            if (alive) {
                                const uint32_t at = qn + __builtin_amdgcn_mbcnt_hi((uint32_t)(sb >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sb, 0u));
                                double* r = wq + at;
                                r[0 * wq_cap] = ray.o.x; r[1 * wq_cap] = ray.o.y; r[2 * wq_cap] = ray.o.z;
                                r[3 * wq_cap] = ray.d.x; r[4 * wq_cap] = ray.d.y; r[5 * wq_cap] = ray.d.z;
                                if (q_has_wl) r[6 * wq_cap] = ray.wl;
                                if (HIST) wqid[at] = id;
                            }
                            if (lane == 0) atomicAdd(&cnt[be + 1], (unsigned long long)n_new);
                            qn += n_new;
                            wave_fence();
                            while (qn >= 64u) wave_drain();
                        }
                    }
                    while (qn > 0u) wave_drain();
                    // ---- the run's last unit hands the stream on: position behind the run's last Bragg uniform, 512 words
                    // generated ahead (the canonical form; a few more may be in the ring, which holds the 624 words in front all the same)
                    if (last_unit) {
                        const uint64_t end = first + 2ull * (uint64_t)n_unit_draws;
                        if (wave == 0) ring_to(end + (uint64_t)XRT_AHEAD);
                        lds_barrier();
                        const uint32_t* r0 = reinterpret_cast<const uint32_t*>(lds_raw);
                        for (int i = tid; i < (int)XRT_RING; i += XRT_TILE) st->ring[i] = r0[i];
                        if (tid == 0) { st->next = end; st->gen = end + (uint64_t)XRT_AHEAD; }
                    }
                }
                stamp(4);
              }
            }
        }

        stamp(5);
#ifdef XRT_UNIT_CLOCKS
        if (SEG && args.dbg && tid == 0) args.dbg[(size_t)unit * 8 + 6] = cnt[be >= 0 ? be + 1 : 0];
#endif

        // ---- run done: counters out, stream head back to memory ---------------
        // canonical form: exactly 512 words generated ahead (what xrt_jump_kernel expects)
        lds_barrier();
        if (counting) {
            if (tid == 0) args.unit_count[unit] = n_candidates;
            lds_barrier();
            continue;
        }
        if (!HIST && last_unit) while ((sgen - spos) < XRT_AHEAD) { mt_step(); lds_barrier(); }
        lbins_flush();
        if (tid <= SC.n_optics && cnt[tid] != 0ULL) atomicAdd(&args.num_out[tid], cnt[tid]);
        // the stream head goes back to memory: always for a whole run; of a segmented run only the
        // last segment knows where the run's stream ends
        if constexpr (LBINS) {
            // the stream head was positioned behind the source arrays and nothing was drawn from it: it goes back as the
            // run's new head (of a segmented run: chunk head 0, by the last segment), brought to the canonical form
            // -- 512 words generated ahead, what the next iteration's jump expects -- in the ring of a source head
            // that is done
            if (last_unit) {
                uint32_t* sr = rings;
                lds_barrier();
                for (int i = tid; i < (int)XRT_RING; i += XRT_TILE) sr[i] = st_in->ring[i];
                uint32_t g = (uint32_t)s_gen0;
                lds_barrier();
                while ((g - spos) < XRT_AHEAD) {
                    uint32_t chunk = XRT_AHEAD - (g - spos);
                    if (chunk > 227u) chunk = 227u;
                    if ((uint32_t)tid < chunk) {
                        const uint32_t n = g + (uint32_t)tid;
                        sr[n & XRT_RMASK] = mt_mix(sr[(n - 624u) & XRT_RMASK], sr[(n - 623u) & XRT_RMASK], sr[(n - 227u) & XRT_RMASK]);
                    }
                    g += chunk;
                    lds_barrier();
                }
                for (int i = tid; i < (int)XRT_RING; i += XRT_TILE) st->ring[i] = sr[i];
                if (tid == 0) { st->next = s_next0; st->gen = s_gen0 + (uint64_t)(uint32_t)(g - (uint32_t)s_gen0); }
            }
        } else
        if (last_unit && !deferred) {
            for (int i = tid; i < (int)XRT_RING; i += XRT_TILE) st->ring[i] = stream[i];
            if (tid == 0) {
                st->next = s_next0 + s_used;
                st->gen = s_next0 + s_used + (uint64_t)(sgen - spos);
            }
        }
        lds_barrier();
    }
}

// Between the two phases of a split one-pass launch (xrt_trace_kernel<false, 2, 3> and <.., 4>): the rest of
// ShapeMesh.intersect (optics/_ShapeMesh.py:289-432 behind the first pass: hit point of the face, nearest fine point, the
// <= 8 faces around it, interpolation), the bounds and the normal for every ray the first phase parked, a thread per ray.
// Nothing here depends on a neighbour or on a stream: the grid is (unit, block of 256 of its rays), the registers are
// those of the mesh code alone.  DEFER (an interpolated mesh): only up to the hit face; the hit point, the nearest point's
// index (in the slot of the normal's first component) and the face are left for xrt_mesh_ct_kernel / xrt_mesh_ct_lds_kernel,
// which interpolate, check the bounds and count.
template <bool DEFER>
__global__ __launch_bounds__(XRT_TILE)
void xrt_mesh_rest_kernel(const KScene* __restrict__ scene_g, const KArgs args, int be, uint32_t blocks_per_unit)
{
    const KScene* scl = scene_fresh(scene_g);
    const uint32_t unit = blockIdx.x / blocks_per_unit, blk = blockIdx.x - unit * blocks_per_unit;
    // (slow_pass: only the rays xrt_mesh_star_lds_kernel listed, see KArgs.slow_q)
    const uint32_t n_unit = args.slow_pass ? uni32(args.unit_slow[unit]) : uni32(args.unit_flag[unit]) - 1u;
    if (256u * blk >= n_unit) return;
    const uint32_t upr = (uint32_t)args.n_seg * (uint32_t)args.n_sub;
    const uint32_t run = unit / upr, uidx = unit - run * upr;
    const uint32_t seg = uidx / (uint32_t)args.n_sub, sub = uidx - seg * (uint32_t)args.n_sub;
    const int64_t N = SRC.n_rays;
    const int64_t seg_lo = (int64_t)seg * args.seg_len;
    const int64_t seg_hi = (seg_lo + args.seg_len < N) ? seg_lo + args.seg_len : N;
    int64_t ray_lo = seg_lo + (int64_t)sub * args.sub_len;
    if (ray_lo > seg_hi) ray_lo = seg_hi;
    const bool q_has_wl = !(SRC.wavelength_dist == XRT_WL_CONST && !SRC.has_velocity);
    const int q_d0 = args.split_lean ? 0 : 3;                    // (records without the origin: KArgs.split_lean)
    const int q_ncomp = (args.split_lean ? 3 : (args.split_interp ? 9 : 6)) + (q_has_wl ? 1 : 0);
    const size_t crun = (size_t)run * (size_t)args.cand_cap;
    const bool have = 256u * blk + (uint32_t)threadIdx.x < n_unit;
    uint32_t kk = 256u * blk + (uint32_t)threadIdx.x;          // the record's number within its unit
    if (args.slow_pass && have) kk = args.slow_q[crun + (size_t)ray_lo + kk];
    const int64_t i = ray_lo + (int64_t)kk;
    double* c = args.cand + crun * (size_t)q_ncomp + (i >> 8) * (int64_t)(q_ncomp * 256) + (i & 255);
    const KOptic& op = SC.opt[be];
    bool alive = false;
    if (have) {
        const int face = (int)args.cand_aux[crun + (size_t)i];
        int idx = -1;
        // (the origin: the ray's own, or the one every parked ray of the unit has -- KArgs.unit_o)
        const double* po = args.unit_o ? args.unit_o + 3 * (size_t)unit : nullptr;
        const double ox = po ? po[0] : c[0 * 256], oy = po ? po[1] : c[1 * 256], oz = po ? po[2] : c[2 * 256];
        const MeshHit h = mesh_rest_impl<false>(op.mesh, ox, oy, oz, c[q_d0 * 256], c[(q_d0 + 1) * 256], c[(q_d0 + 2) * 256], face, &idx);
        V3 X;
        X.x = h.x; X.y = h.y; X.z = h.z;
        // (what is left of the ray: the face it ended on, or XRT_CAND_DEAD -- see KArgs.split_interp)
        if constexpr (DEFER) {
            if (h.hit != 0) {
                c[0 * 256] = X.x; c[1 * 256] = X.y; c[2 * 256] = X.z;
                c[6 * 256] = __hiloint2double(0, idx);          // (the nearest point: where the interpolation's walk may start)
            }
            args.cand_aux[crun + (size_t)i] = h.hit != 0 ? (uint32_t)h.aux : XRT_CAND_DEAD;
        } else {
            alive = (h.hit != 0) && check_bounds<true>(op, X);
            args.cand_aux[crun + (size_t)i] = alive ? (uint32_t)h.aux : XRT_CAND_DEAD;
        }
    }
    if constexpr (DEFER) return;
    const unsigned long long ab = __ballot(alive);
    if (args.slow_pass) {       // (the listed rays lie anywhere in the unit: they add to what the first launch counted)
        if (alive) atomicAdd(&args.batch_alive[((crun + (size_t)ray_lo) >> 6) + (kk >> 6)], 1u);
        return;
    }
    if ((threadIdx.x & 63u) == 0u) {
        const uint32_t n = (uint32_t)__popcll(ab);
        args.batch_alive[((crun + (size_t)ray_lo) >> 6) + 4u * blk + (threadIdx.x >> 6)] = n;
    }
}

// The same for a mesh whose tables fit the LDS (KMesh.lds_bytes > 0; see mesh_rest_lds): one 1024-thread workgroup per
// CU keeps the tables and takes blocks of 1024 parked rays round robin.
#define XRT_MESH_LDS_THREADS 1024
// DEFER (an interpolated mesh): only up to the hit face; the hit point and the nearest point's index (in the slot of the
// normal's first component) are left for xrt_mesh_ct_kernel, which interpolates, checks the bounds and counts.
template <bool DEFER>
__global__ __launch_bounds__(XRT_MESH_LDS_THREADS)
void xrt_mesh_rest_lds_kernel(const KScene* __restrict__ scene_g, const KArgs args, int be)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const KScene* scl = scene_fresh(scene_g);
    const KOptic& op = SC.opt[be];
    const KMesh* Mp = op.mesh;
    MeshRef M = *(const XRT_C4 KMesh*)uniform_u64((uint64_t)Mp);
    const int tid = threadIdx.x;
    // tables: cells | point faces | face vertices | faces of the first pass
    const uint32_t nc = (uint32_t)M.n_cells, np = (uint32_t)M.n_points, nf = (uint32_t)M.n_faces, n1 = (uint32_t)M.n_first + 1u;
    d4v* l_cells = reinterpret_cast<d4v*>(lds_raw);
    uint32_t* l_pf = reinterpret_cast<uint32_t*>(l_cells + nc);                // [np][4] words
    uint32_t* l_fv = l_pf + 4u * np;                                           // [nf][2] words
    double* l_first = reinterpret_cast<double*>(l_fv + 2u * nf);
    {
        const XRT_G1 d4v* g = M.cells;
        for (uint32_t i = (uint32_t)tid; i < nc; i += XRT_MESH_LDS_THREADS) l_cells[i] = g[i];
        const XRT_G1 uint32_t* gp = (const XRT_G1 uint32_t*)(uint64_t)M.lds_pf;
        for (uint32_t i = (uint32_t)tid; i < 4u * np; i += XRT_MESH_LDS_THREADS) l_pf[i] = gp[i];
        const XRT_G1 uint32_t* gf = (const XRT_G1 uint32_t*)(uint64_t)M.lds_fv;
        for (uint32_t i = (uint32_t)tid; i < 2u * nf; i += XRT_MESH_LDS_THREADS) l_fv[i] = gf[i];
        const XRT_G1 double* g1 = (const XRT_G1 double*)(uint64_t)M.first_rec;
        for (uint32_t i = (uint32_t)tid; i < 10u * n1; i += XRT_MESH_LDS_THREADS) l_first[i] = g1[i];
    }
    __syncthreads();
    MeshLds L;
    L.cells = (lcell)l_cells;
    L.pf = (const XRT_LDS3 uint16_t*)l_pf;
    L.fv = (const XRT_LDS3 uint16_t*)l_fv;
    L.first = (const XRT_LDS3 double*)l_first;
    const int64_t N = SRC.n_rays;
    const bool q_has_wl = !(SRC.wavelength_dist == XRT_WL_CONST && !SRC.has_velocity);
    const int q_d0 = args.split_lean ? 0 : 3;                    // (records without the origin: KArgs.split_lean)
    const int q_ncomp = (args.split_lean ? 3 : (args.split_interp ? 9 : 6)) + (q_has_wl ? 1 : 0);
    const uint32_t upr = (uint32_t)args.n_seg * (uint32_t)args.n_sub;
    // (unit, block of 1024 rays) from the list of the blocks that hold rays; slow_pass: the rays xrt_mesh_star_lds_kernel listed
    const uint32_t n_it = uni32(*args.n_items);
    for (uint32_t ii = blockIdx.x; ii < n_it; ii += gridDim.x) {
        const uint4 item = args.items[ii];
        const uint32_t unit = uni32(item.x), blk = uni32(item.y), n_unit = uni32(item.z);
        const uint32_t run = unit / upr, uidx = unit - run * upr;
        const uint32_t seg = uidx / (uint32_t)args.n_sub, sub = uidx - seg * (uint32_t)args.n_sub;
        const int64_t seg_lo = (int64_t)seg * args.seg_len;
        const int64_t seg_hi = (seg_lo + args.seg_len < N) ? seg_lo + args.seg_len : N;
        int64_t ray_lo = seg_lo + (int64_t)sub * args.sub_len;
        if (ray_lo > seg_hi) ray_lo = seg_hi;
        const size_t crun = (size_t)run * (size_t)args.cand_cap;
        const uint32_t k = 1024u * blk + (uint32_t)tid;
        const bool have = k < n_unit;
        uint32_t kk = k;                                        // the record's number within its unit
        if (args.slow_pass && have) kk = args.slow_q[crun + (size_t)ray_lo + k];
        const int64_t i = ray_lo + (int64_t)kk;
        double* c = args.cand + crun * (size_t)q_ncomp + (i >> 8) * (int64_t)(q_ncomp * 256) + (i & 255);
        bool alive = false;
        if (have) {
            const int face = (int)args.cand_aux[crun + (size_t)i];
            int idx;
            V3 nrm;
            nrm.x = nrm.y = nrm.z = 0.0;
            const double* po = args.unit_o ? args.unit_o + 3 * (size_t)unit : nullptr;
            const double ox = po ? po[0] : c[0 * 256], oy = po ? po[1] : c[1 * 256], oz = po ? po[2] : c[2 * 256];
            const MeshHit h = mesh_rest_lds(Mp, L, ox, oy, oz, c[q_d0 * 256], c[(q_d0 + 1) * 256], c[(q_d0 + 2) * 256], face, idx, nrm);
            V3 X;
            X.x = h.x; X.y = h.y; X.z = h.z;
            if constexpr (DEFER) {
                if (h.hit != 0) {
                    c[0 * 256] = X.x; c[1 * 256] = X.y; c[2 * 256] = X.z;
                    c[6 * 256] = __hiloint2double(0, idx);
                }
                args.cand_aux[crun + (size_t)i] = h.hit != 0 ? (uint32_t)h.aux : XRT_CAND_DEAD;
            } else {
                alive = (h.hit != 0) && check_bounds<true>(op, X);
                args.cand_aux[crun + (size_t)i] = alive ? (uint32_t)h.aux : XRT_CAND_DEAD;
            }
        }
        if constexpr (DEFER) continue;
        const unsigned long long ab = __ballot(alive);
        if (args.slow_pass) {       // (the listed rays lie anywhere in the unit: they add to what the first launch counted)
            if (alive) atomicAdd(&args.batch_alive[((crun + (size_t)ray_lo) >> 6) + (kk >> 6)], 1u);
            continue;
        }
        if ((tid & 63) == 0 && 64u * (k >> 6) < ((n_unit + 63u) & ~63u)) {
            const uint32_t n = (uint32_t)__popcll(ab);
            args.batch_alive[((crun + (size_t)ray_lo) >> 6) + (k >> 6)] = n;
        }
    }
}

// The fan launch (KMesh.lds_star, mesh_rest_star_lds): the parked rays that one face of the nearest point's fan settles are
// finished here -- DEFER as above --, the others keep their records untouched and are listed per unit for the launch that
// walks the lists (xrt_mesh_rest_lds_kernel / xrt_mesh_rest_kernel with slow_pass).  Tables in LDS: the buckets of the
// points (also the vertex table), the fans, the faces of the first pass.
template <bool DEFER, class LT>
__device__ __forceinline__ void mesh_star_blocks(const KScene* __restrict__ scene_g, const KArgs& args, int be, const LT& L)
{
    const KScene* scl = scene_fresh(scene_g);
    const KOptic& op = SC.opt[be];
    const KMesh* Mp = op.mesh;
    const int tid = threadIdx.x;
    const int64_t N = SRC.n_rays;
    const bool q_has_wl = !(SRC.wavelength_dist == XRT_WL_CONST && !SRC.has_velocity);
    const int q_d0 = args.split_lean ? 0 : 3;                    // (records without the origin: KArgs.split_lean)
    const int q_ncomp = (args.split_lean ? 3 : (args.split_interp ? 9 : 6)) + (q_has_wl ? 1 : 0);
    const uint32_t upr = (uint32_t)args.n_seg * (uint32_t)args.n_sub;
    // One block (1024 records, 64 per wave) per round, in this order: (1) the item of the NEXT block -- asked for a round ago --
    // is taken up, (2) what the round BEFORE found is stored, (3) the next block's records and the item of the block after
    // are asked for, (4) this block's arithmetic.  Loads and stores count in one vmcnt, in order, and how many stores a round
    // issues depends on its rays: a round can only wait for "all" -- here all of it was asked for a block's arithmetic ago.
    // (Stores behind the arithmetic and the next item / records in front of it cost three memory round trips per round:
    // 0.55 - 0.65 of the vector issue rate with two blocks per round.)
    struct Block { uint32_t unit, k, n_unit; size_t crun; int64_t ray_lo; double* c; bool have; };
    struct Rec { double r[6]; int face; };
    struct Found { bool settled, alive, slow; uint32_t aux; int idx; V3 X; unsigned long long sl, ab; };
    const uint32_t items = uni32(*args.n_items);
    auto item_ask = [&](uint32_t it) __attribute__((always_inline)) -> uint4 {
        return args.items[it < items ? it : (items > 0u ? items - 1u : 0u)];
    };
    auto locate = [&](uint32_t it, const uint4& item, Block& B) __attribute__((always_inline)) {
        // item `it` of the list of the blocks that hold rays (beyond its end: nothing to do)
        B.have = false; B.unit = 0; B.k = 0; B.n_unit = 0; B.crun = 0; B.ray_lo = 0; B.c = args.cand;
        if (it < items) {
            const uint32_t unit = uni32(item.x), blk = uni32(item.y), n_unit = uni32(item.z);
            const uint32_t run = unit / upr, uidx = unit - run * upr;
            const uint32_t seg = uidx / (uint32_t)args.n_sub, sub = uidx - seg * (uint32_t)args.n_sub;
            const int64_t seg_lo = (int64_t)seg * args.seg_len;
            const int64_t seg_hi = (seg_lo + args.seg_len < N) ? seg_lo + args.seg_len : N;
            int64_t ray_lo = seg_lo + (int64_t)sub * args.sub_len;
            if (ray_lo > seg_hi) ray_lo = seg_hi;
            B.unit = unit; B.n_unit = n_unit; B.ray_lo = ray_lo;
            B.crun = (size_t)run * (size_t)args.cand_cap;
            B.k = 1024u * blk + (uint32_t)tid;
            const int64_t i = ray_lo + (int64_t)B.k;
            B.c = args.cand + B.crun * (size_t)q_ncomp + (i >> 8) * (int64_t)(q_ncomp * 256) + (i & 255);
            B.have = B.k < n_unit;
        }
    };
    const int q0 = args.unit_o ? 3 : 0;                // (behind a point source the origin is the unit's, KArgs.unit_o)
    auto ask = [&](const Block& B, Rec& R) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 6; q++) R.r[q] = 0.0;
        R.face = 0;
        if (B.have) {
#pragma unroll
            for (int q = 0; q < 6; q++) if (q >= q0) R.r[q] = B.c[(q < 3 ? q : q - 3 + q_d0) * 256];
            R.face = (int)args.cand_aux[B.crun + (size_t)(B.ray_lo + (int64_t)B.k)];
            if (args.unit_o) { const double* uo = args.unit_o + 3 * (size_t)B.unit; R.r[0] = uo[0]; R.r[1] = uo[1]; R.r[2] = uo[2]; }
        }
    };
    auto work = [&](const Block& B, const Rec& R) __attribute__((always_inline)) -> Found {
        Found F;
        F.settled = false; F.alive = false; F.slow = false; F.aux = 0u; F.idx = 0;
        F.X.x = F.X.y = F.X.z = 0.0;
        if (B.have) {
            V3 nrm;
            nrm.x = nrm.y = nrm.z = 0.0;
            bool slow = false;
            const MeshHit h = mesh_rest_star_lds(Mp, L, R.r[0], R.r[1], R.r[2], R.r[3], R.r[4], R.r[5], R.face, F.idx, nrm, slow);
            F.slow = slow;
            if (!slow) {
                F.settled = true;
                F.X.x = h.x; F.X.y = h.y; F.X.z = h.z;
                F.aux = (uint32_t)h.aux;
                if constexpr (!DEFER) {
                    F.alive = check_bounds<true>(op, F.X);
                    if (!F.alive) F.aux = XRT_CAND_DEAD;
                }
            }
        }
        // the rays left for the list walk: their numbers within the unit
#ifdef XRT_DEV_NO_SLOWQ
        F.sl = 0ULL;
#else
        F.sl = __ballot(F.slow);
#endif
        F.ab = __ballot(F.alive);
        return F;
    };
    auto put = [&](const Block& B, const Found& F) __attribute__((always_inline)) {
        const uint32_t k = B.k, n_unit = B.n_unit;
        const size_t crun = B.crun;
        const int64_t ray_lo = B.ray_lo, i = ray_lo + (int64_t)k;
        if (F.settled) {
            if constexpr (DEFER) {
                double* c = B.c;
                c[0 * 256] = F.X.x; c[1 * 256] = F.X.y; c[2 * 256] = F.X.z;
                c[6 * 256] = __hiloint2double(0, F.idx);
            }
            args.cand_aux[crun + (size_t)i] = F.aux;
        }
        // (per 64 records -- this wave's -- and without atomics: a counter per unit that every wave of the chip adds to and
        //  waits for cost 4 ms of the launch's 27)
        if (F.slow) args.slow_q[crun + (size_t)ray_lo + (k & ~63u) + __builtin_amdgcn_mbcnt_hi((uint32_t)(F.sl >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)F.sl, 0u))] = k;
        const bool in_unit = n_unit > 0u && 64u * (k >> 6) < ((n_unit + 63u) & ~63u);
        if ((tid & 63) == 0 && in_unit) args.slow_cnt[((crun + (size_t)ray_lo) >> 6) + (k >> 6)] = (uint32_t)__popcll(F.sl);
        if constexpr (!DEFER) {
            if ((tid & 63) == 0 && in_unit) {
                const uint32_t n = (uint32_t)__popcll(F.ab);
                args.batch_alive[((crun + (size_t)ray_lo) >> 6) + (k >> 6)] = n;
            }
        }
    };
    uint32_t it = blockIdx.x;
    if (it >= items) return;
    Block Bc, Bp;
    Rec Rc;
    Found Fp;
    uint4 raw_n;
    {
        const uint4 raw_c = item_ask(it);
        raw_n = item_ask(it + gridDim.x);
        locate(it, raw_c, Bc);
        ask(Bc, Rc);
    }
    locate(items, raw_n, Bp);                            // (nothing found yet: a block without rays)
    Fp = work(Bp, Rc);
    for (; it < items; it += gridDim.x) {
        Block Bn;
        Rec Rn;
        locate(it + gridDim.x, raw_n, Bn);
        put(Bp, Fp);
        ask(Bn, Rn);
        raw_n = item_ask(it + 2u * gridDim.x);
        Fp = work(Bc, Rc);
        Bp = Bc; Bc = Bn; Rc = Rn;
    }
    put(Bp, Fp);
}

template <bool DEFER>
__global__ __launch_bounds__(XRT_MESH_LDS_THREADS)
void xrt_mesh_star_lds_kernel(const KScene* __restrict__ scene_g, const KArgs args, int be)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const KScene* scl = scene_fresh(scene_g);
    const KOptic& op = SC.opt[be];
    const KMesh* Mp = op.mesh;
    MeshRef M = *(const XRT_C4 KMesh*)uniform_u64((uint64_t)Mp);
    const int tid = threadIdx.x;
    const uint32_t nc = (uint32_t)M.n_cells, np = (uint32_t)M.n_points, n1 = (uint32_t)M.n_first + 1u;
    d4v* l_cells = reinterpret_cast<d4v*>(lds_raw);
    uint32_t* l_star = reinterpret_cast<uint32_t*>(l_cells + nc);              // [np][12] words
    double* l_first = reinterpret_cast<double*>(l_star + 12u * np);
    {
        const XRT_G1 d4v* g = M.cells;
        for (uint32_t i = (uint32_t)tid; i < nc; i += XRT_MESH_LDS_THREADS) l_cells[i] = g[i];
        const XRT_G1 uint32_t* gs = (const XRT_G1 uint32_t*)(uint64_t)M.lds_star;
        for (uint32_t i = (uint32_t)tid; i < 12u * np; i += XRT_MESH_LDS_THREADS) l_star[i] = gs[i];
        const XRT_G1 double* g1 = (const XRT_G1 double*)(uint64_t)M.first_rec;
        for (uint32_t i = (uint32_t)tid; i < 10u * n1; i += XRT_MESH_LDS_THREADS) l_first[i] = g1[i];
    }
    __syncthreads();
    MeshStarLds L;
    L.cells = (lcell)l_cells;
    L.star = (const XRT_LDS3 uint16_t*)l_star;
    L.first = (const XRT_LDS3 double*)l_first;
    mesh_star_blocks<DEFER>(scene_g, args, be, L);
}

// The same for a mesh whose tables do not fit the LDS (81 x 81 points: 210 KB of buckets alone): the fans, the buckets and the
// first pass' faces read where they lie in global memory (~1 MB: they stay in the L2).  A gather costs the L1 a tag lookup per
// lane and 16 bytes, so a ray is dearer here than with the tables in LDS -- but it is ONE face's test instead of the walk over
// the <= 8 faces of the list, and the rays the fans cannot settle are few.
template <bool DEFER>
__global__ __launch_bounds__(XRT_MESH_LDS_THREADS)
void xrt_mesh_star_kernel(const KScene* __restrict__ scene_g, const KArgs args, int be)
{
    const KScene* scl = scene_fresh(scene_g);
    MeshRef M = *(const XRT_C4 KMesh*)uniform_u64((uint64_t)SC.opt[be].mesh);
    MeshStarT<TabGlobal> L;
    L.cells = M.cells;
    L.star = (const XRT_G1 uint16_t*)(uint64_t)M.lds_star;
    L.first = (const XRT_G1 double*)(uint64_t)M.first_rec;
    mesh_star_blocks<DEFER>(scene_g, args, be, L);
}

// The blocks of 1024 records that hold rays (KArgs.items), units in order: counts[u] - minus records in unit u.
__global__ __launch_bounds__(1024)
void xrt_mesh_items_kernel(const uint32_t* counts, uint32_t minus, uint32_t n_units, uint4* items, uint32_t* n_items)
{
    __shared__ uint32_t part[1024];
    __shared__ uint32_t running;
    const uint32_t tid = threadIdx.x;
    if (tid == 0) running = 0u;
    __syncthreads();
    for (uint32_t u0 = 0; u0 < n_units; u0 += 1024u) {
        const uint32_t u = u0 + tid;
        const uint32_t n = u < n_units ? counts[u] - minus : 0u;
        const uint32_t nb = (n + 1023u) >> 10;
        part[tid] = nb;
        __syncthreads();
        for (uint32_t off = 1u; off < 1024u; off <<= 1) {
            const uint32_t v = tid >= off ? part[tid - off] : 0u;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        const uint32_t at = running + part[tid] - nb;
        for (uint32_t b = 0; b < nb; b++) { uint4 it; it.x = u; it.y = b; it.z = n; it.w = 0u; items[at + b] = it; }
        __syncthreads();
        if (tid == 1023u) running += part[1023];
        __syncthreads();
    }
    if (tid == 0) *n_items = running;
}

// The parked rays a unit has left when the middle launches are through (what the second phase reads as the number of its Bragg
// draws, KArgs.unit_alive): the sum of the unit's counts per 64 records.  A launch of its own, a workgroup per unit -- as an
// atomic per wave of the middle launches it was 8e6 adds on a thousand addresses, 2.9 ms of the fan launch's 21.
__global__ __launch_bounds__(256)
void xrt_mesh_unit_alive_kernel(const KScene* __restrict__ scene_g, const KArgs args, uint32_t n_units)
{
    __shared__ uint32_t part[4];
    const KScene* scl = scene_fresh(scene_g);
    const uint32_t upr = (uint32_t)args.n_seg * (uint32_t)args.n_sub;
    const int64_t N = SRC.n_rays;
    for (uint32_t unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
        const uint32_t n_unit = uni32(args.unit_flag[unit]) - 1u;
        const uint32_t run = unit / upr, uidx = unit - run * upr;
        const uint32_t seg = uidx / (uint32_t)args.n_sub, sub = uidx - seg * (uint32_t)args.n_sub;
        const int64_t seg_lo = (int64_t)seg * args.seg_len;
        const int64_t seg_hi = (seg_lo + args.seg_len < N) ? seg_lo + args.seg_len : N;
        int64_t ray_lo = seg_lo + (int64_t)sub * args.sub_len;
        if (ray_lo > seg_hi) ray_lo = seg_hi;
        const size_t crun = (size_t)run * (size_t)args.cand_cap;
        const uint32_t* ba = args.batch_alive + ((crun + (size_t)ray_lo) >> 6);
        const uint32_t groups = (n_unit + 63u) >> 6;
        uint32_t sum = 0;
        for (uint32_t g = threadIdx.x; g < groups; g += 256u) sum += ba[g];
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        if ((threadIdx.x & 63u) == 0u) part[threadIdx.x >> 6] = sum;
        __syncthreads();
        if (threadIdx.x == 0) args.unit_alive[unit] = part[0] + part[1] + part[2] + part[3];
        __syncthreads();
    }
}


// The fan launch's lists (per 64 records of a unit) without their gaps: per unit one dense list for the launch that walks the
// face lists, and its length.  A workgroup per unit, 256 runs of 64 records a step.
__global__ __launch_bounds__(256)
void xrt_mesh_slow_compact_kernel(const KScene* __restrict__ scene_g, const KArgs args, uint32_t n_units)
{
    __shared__ uint32_t part[256];
    __shared__ uint32_t running;
    const KScene* scl = scene_fresh(scene_g);
    const int64_t N = SRC.n_rays;
    const uint32_t upr = (uint32_t)args.n_seg * (uint32_t)args.n_sub;
    const uint32_t tid = threadIdx.x;
    for (uint32_t unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
        const uint32_t n_unit = uni32(args.unit_flag[unit]) - 1u;
        const uint32_t run = unit / upr, uidx = unit - run * upr;
        const uint32_t seg = uidx / (uint32_t)args.n_sub, sub = uidx - seg * (uint32_t)args.n_sub;
        const int64_t seg_lo = (int64_t)seg * args.seg_len;
        const int64_t seg_hi = (seg_lo + args.seg_len < N) ? seg_lo + args.seg_len : N;
        int64_t ray_lo = seg_lo + (int64_t)sub * args.sub_len;
        if (ray_lo > seg_hi) ray_lo = seg_hi;
        const size_t base = (size_t)run * (size_t)args.cand_cap + (size_t)ray_lo;
        const uint32_t nb = (n_unit + 63u) >> 6;
        if (tid == 0) running = 0u;
        __syncthreads();
        for (uint32_t b0 = 0; b0 < nb; b0 += 256u) {
            const uint32_t b = b0 + tid;
            const uint32_t cnt = b < nb ? args.slow_cnt[(base >> 6) + b] : 0u;
            part[tid] = cnt;
            __syncthreads();
            for (uint32_t off = 1u; off < 256u; off <<= 1) {
                const uint32_t v = tid >= off ? part[tid - off] : 0u;
                __syncthreads();
                part[tid] += v;
                __syncthreads();
            }
            const uint32_t at = running + part[tid] - cnt;
            for (uint32_t j = 0; j < cnt; j++) args.slow_q2[base + at + j] = args.slow_q[base + 64u * b + j];
            __syncthreads();
            if (tid == 255u) running += part[255];
            __syncthreads();
        }
        if (tid == 0) args.unit_slow[unit] = running;
        __syncthreads();
    }
}

// Behind xrt_mesh_rest_lds_kernel<true>: the interpolated height and normal (SciPy CloughTocher2DInterpolator, see
// xrt_mesh.inc), the bounds and the counts for every parked ray that hit a face, a thread per ray.
__global__ __launch_bounds__(XRT_TILE)
void xrt_mesh_ct_kernel(const KScene* __restrict__ scene_g, const KArgs args, int be, uint32_t blocks_per_unit)
{
    const KScene* scl = scene_fresh(scene_g);
    const uint32_t unit = blockIdx.x / blocks_per_unit, blk = blockIdx.x - unit * blocks_per_unit;
    const uint32_t n_unit = uni32(args.unit_flag[unit]) - 1u;
    if (256u * blk >= n_unit) return;
    const uint32_t upr = (uint32_t)args.n_seg * (uint32_t)args.n_sub;
    const uint32_t run = unit / upr, uidx = unit - run * upr;
    const uint32_t seg = uidx / (uint32_t)args.n_sub, sub = uidx - seg * (uint32_t)args.n_sub;
    const int64_t N = SRC.n_rays;
    const int64_t seg_lo = (int64_t)seg * args.seg_len;
    const int64_t seg_hi = (seg_lo + args.seg_len < N) ? seg_lo + args.seg_len : N;
    int64_t ray_lo = seg_lo + (int64_t)sub * args.sub_len;
    if (ray_lo > seg_hi) ray_lo = seg_hi;
    const bool q_has_wl = !(SRC.wavelength_dist == XRT_WL_CONST && !SRC.has_velocity);
    const int q_ncomp = (args.split_lean ? 3 : (args.split_interp ? 9 : 6)) + (q_has_wl ? 1 : 0);
    const size_t crun = (size_t)run * (size_t)args.cand_cap;
    const int64_t i = ray_lo + (int64_t)(256u * blk + (uint32_t)threadIdx.x);
    double* c = args.cand + crun * (size_t)q_ncomp + (i >> 8) * (int64_t)(q_ncomp * 256) + (i & 255);
    uint32_t face = XRT_CAND_DEAD;
    if (256u * blk + (uint32_t)threadIdx.x < n_unit) face = args.cand_aux[crun + (size_t)i];
    const bool have = face != XRT_CAND_DEAD;
    const KOptic& op = SC.opt[be];
    bool alive = false;
    if (have) {
        MeshHit h;
        h.x = c[0 * 256]; h.y = c[1 * 256]; h.z = c[2 * 256]; h.aux = 0; h.hit = 1;
        CtShared G;
        mesh_rest_ct(op.mesh, __double2loint(c[6 * 256]), (int)face, h, G);
        V3 X;
        X.x = h.x; X.y = h.y; X.z = h.z;
        alive = check_bounds<true>(op, X);
        if (alive) {
            const V3 nrm = mesh_normal_kept(op.mesh, G, h.aux);
            c[2 * 256] = X.z;
            c[6 * 256] = nrm.x; c[7 * 256] = nrm.y; c[8 * 256] = nrm.z;
        } else args.cand_aux[crun + (size_t)i] = XRT_CAND_DEAD;
    }
    const unsigned long long ab = __ballot(alive);
    if ((threadIdx.x & 63u) == 0u) {
        const uint32_t n = (uint32_t)__popcll(ab);
        args.batch_alive[((crun + (size_t)ray_lo) >> 6) + 4u * blk + (threadIdx.x >> 6)] = n;
    }
}

// The same with the vertex table (value and gradient of height and normal components: 96 bytes per point) in LDS, for a
// mesh whose table fits (41 x 41 points: 161 376 of the 163 840 bytes): 24 of a ray's ~40 gathers then cost an LDS read
// instead of an L1 tag lookup per lane.  One 1024-thread workgroup per CU, blocks of 1024 parked rays round robin.
__global__ __launch_bounds__(XRT_MESH_LDS_THREADS)
void xrt_mesh_ct_lds_kernel(const KScene* __restrict__ scene_g, const KArgs args, int be)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const KScene* scl = scene_fresh(scene_g);
    const KOptic& op = SC.opt[be];
    const KMesh* Mp = op.mesh;
    MeshRef M = *(const XRT_C4 KMesh*)uniform_u64((uint64_t)Mp);
    const int tid = threadIdx.x;
    double* l_v = reinterpret_cast<double*>(lds_raw);                       // [n_points][12]
    {
        const uint32_t np = (uint32_t)M.n_points;
        const gdp g = M.ct_vrec;
        for (uint32_t i = (uint32_t)tid; i < 12u * np; i += XRT_MESH_LDS_THREADS) { const uint32_t v = i / 12u; l_v[i] = g[16u * v + (i - 12u * v)]; }
    }
    __syncthreads();
    const XRT_LDS3 double* lv = (const XRT_LDS3 double*)l_v;
    const int64_t N = SRC.n_rays;
    const bool q_has_wl = !(SRC.wavelength_dist == XRT_WL_CONST && !SRC.has_velocity);
    const int q_ncomp = (args.split_lean ? 3 : (args.split_interp ? 9 : 6)) + (q_has_wl ? 1 : 0);
    const uint32_t upr = (uint32_t)args.n_seg * (uint32_t)args.n_sub;
    const uint32_t n_it = uni32(*args.n_items);
    for (uint32_t ii = blockIdx.x; ii < n_it; ii += gridDim.x) {
        const uint4 item = args.items[ii];
        const uint32_t unit = uni32(item.x), blk = uni32(item.y), n_unit = uni32(item.z);
        const uint32_t run = unit / upr, uidx = unit - run * upr;
        const uint32_t seg = uidx / (uint32_t)args.n_sub, sub = uidx - seg * (uint32_t)args.n_sub;
        const int64_t seg_lo = (int64_t)seg * args.seg_len;
        const int64_t seg_hi = (seg_lo + args.seg_len < N) ? seg_lo + args.seg_len : N;
        int64_t ray_lo = seg_lo + (int64_t)sub * args.sub_len;
        if (ray_lo > seg_hi) ray_lo = seg_hi;
        const size_t crun = (size_t)run * (size_t)args.cand_cap;
        const uint32_t k = 1024u * blk + (uint32_t)tid;
        const int64_t i = ray_lo + (int64_t)k;
        double* c = args.cand + crun * (size_t)q_ncomp + (i >> 8) * (int64_t)(q_ncomp * 256) + (i & 255);
        uint32_t face = XRT_CAND_DEAD;
        if (k < n_unit) face = args.cand_aux[crun + (size_t)i];
        const bool have = face != XRT_CAND_DEAD;
        bool alive = false;
        if (have) {
            MeshHit h;
            h.x = c[0 * 256]; h.y = c[1 * 256]; h.z = c[2 * 256]; h.aux = 0; h.hit = 1;
            CtShared G;
            mesh_rest_ct_at(Mp, lv, 12, __double2loint(c[6 * 256]), (int)face, h, G);
            V3 X;
            X.x = h.x; X.y = h.y; X.z = h.z;
            alive = check_bounds<true>(op, X);
            if (alive) {
                const V3 nrm = mesh_normal_kept_at(lv, 12, G, h.aux);
                c[2 * 256] = X.z;
                c[6 * 256] = nrm.x; c[7 * 256] = nrm.y; c[8 * 256] = nrm.z;
            } else args.cand_aux[crun + (size_t)i] = XRT_CAND_DEAD;
        }
        const unsigned long long ab = __ballot(alive);
        if ((tid & 63) == 0 && 64u * (k >> 6) < ((n_unit + 63u) & ~63u)) {
            const uint32_t n = (uint32_t)__popcll(ab);
            args.batch_alive[((crun + (size_t)ray_lo) >> 6) + (k >> 6)] = n;
        }
    }
}

#undef SC
#undef SRC

#include "xrt_staged.inc"
#include "xrt_mosaic.inc"

#define XRT_PLASMA_PART 2
#include "xrt_plasma.inc"
#undef XRT_PLASMA_PART


#include "xrt_gauss.inc"

// ==========================================================================
// host side of the C ABI
// ==========================================================================

#include "mt_jump.inc"

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, const char* a = "")
{
    snprintf(g_err, sizeof(g_err), fmt, a);
    return code;
}

#define HIP_TRY(expr)                                                              \
    do {                                                                           \
        hipError_t e_ = (expr);                                                    \
        if (e_ != hipSuccess) return fail(-10, #expr ": %s", hipGetErrorString(e_)); \
    } while (0)

// Which device routes the calls of this thread took since the last xrt_last_path(1): tests of the
// fallback switches (environment variables, read at every call) assert on it.
static thread_local uint32_t g_paths = 0;
static bool env_on(const char* name) { return getenv(name) != nullptr; }

static bool timing_on = false;
static double timing_ms = 0.0;
static int64_t timing_launches = 0;
static const int TIMING_MAX = 64;
static hipEvent_t timing_ev[TIMING_MAX][2];
static int timing_n = 0;

extern "C" int xrt_abi_version(void) { return XRT_ABI_VERSION; }
extern "C" const char* xrt_last_error(void) { return g_err; }
extern "C" size_t xrt_sizeof_scene(void) { return sizeof(xrt_scene_t); }

extern "C" int xrt_device_count(int* count)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return fail(-10, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return 0;
}

extern "C" int xrt_scene_check(const xrt_scene_t* sc)
{
    if (!sc) return fail(-1, "%s", "scene is NULL");
    if (sc->source.plasma) {
        const xrt_plasma_t* P = sc->source.plasma;
        if (sc->source.kind != XRT_SRC_PLASMA) return fail(-2, "%s", "per-bundle plasma model on a non-plasma source");
        if (P->n_filters < 0 || P->n_filters > XRT_MAX_BUNDLE_FILTERS) return fail(-2, "%s", "bad bundle filter count");
        if (P->geometry != XRT_PLASMA_BOX && P->geometry != XRT_PLASMA_TOROIDAL) return fail(-3, "%s", "plasma geometry is not implemented on the device path");
        if ((P->n_emissivity > 0 && (!P->emissivity_rho || !P->emissivity_val)) ||
            (P->n_temperature > 0 && (!P->temperature_rho || !P->temperature_val)))
            return fail(-2, "%s", "plasma profile tables missing");
        if ((P->n_emissivity > 0 || P->n_temperature > 0) && P->geometry != XRT_PLASMA_TOROIDAL)
            return fail(-2, "%s", "plasma profiles need a flux geometry");
        if (P->n_temperature > 0 && sc->source.wavelength_dist == XRT_WL_VOIGT &&
            (!(P->voigt_gamma > 0.0) || P->n_weideman < 8 || P->n_weideman > 64 || !P->weideman_a))
            return fail(-2, "%s", "a temperature profile with a natural linewidth needs voigt_gamma and the Weideman coefficients");
    }
    if (sc->n_optics < 0 || sc->n_optics > XRT_DEV_MAX_OPTICS)
        return fail(-2, "%s", "device path supports at most 64 optics (XRT_MAX_OPTICS)");
    const xrt_source_t& s = sc->source;
    if (s.intensity < 0) return fail(-2, "%s", "negative intensity");
    if (s.spatial_dist != XRT_SPATIAL_UNIFORM && s.spatial_dist != XRT_SPATIAL_GAUSSIAN)
        return fail(-2, "%s", "unknown spatial_dist");
    if (s.angular_dist < XRT_ANG_ISOTROPIC || s.angular_dist > XRT_ANG_FLAT_XY)
        return fail(-2, "%s", "unknown angular_dist");
    if (s.wavelength_dist < XRT_WL_CONST || s.wavelength_dist > XRT_WL_VOIGT)
        return fail(-2, "%s", "unknown wavelength_dist");
    if (s.kind < XRT_SRC_GENERIC || s.kind > XRT_SRC_EXTERNAL) return fail(-2, "%s", "unknown source kind");
    if (s.n_ray_filters < 0 || s.n_ray_filters > XRT_MAX_BUNDLE_FILTERS) return fail(-2, "%s", "bad ray filter count");
    if (s.n_ray_filters > 0 && (s.kind == XRT_SRC_PLASMA || s.kind == XRT_SRC_EXTERNAL))
        return fail(-2, "%s", "ray filters belong to the non-plasma sources (a plasma filters its bundles)");
    if (s.kind == XRT_SRC_EXTERNAL && s.intensity > 0 && (!s.ext_rays || !s.ext_mask))
        return fail(-2, "%s", "external rays missing");
    if (s.kind == XRT_SRC_PLASMA) {
        if (s.bundle_count < 1) return fail(-2, "%s", "plasma bundle_count < 1");
        if (s.spatial_dist != XRT_SPATIAL_UNIFORM || s.angular_dist != XRT_ANG_ISOTROPIC)
            return fail(-3, "%s", "plasma bundles support uniform voxels and isotropic cones only");
        if (!(s.bundle_intensity >= 0.0)) return fail(-2, "%s", "bad plasma bundle intensity");
    }
    if (s.wavelength_dist == XRT_WL_VOIGT && (s.voigt_n < 2 || !s.voigt_cdf || !s.voigt_x))
        return fail(-2, "%s", "voigt table missing");
    for (int e = 0; e < sc->n_optics; e++) {
        const xrt_optic_t& o = sc->optics[e];
        if (o.shape < XRT_SHAPE_PLANE || o.shape > XRT_SHAPE_MESH)
            return fail(-3, "%s", "optic shape is not implemented on the device path");
        if (o.shape == XRT_SHAPE_MESH) {
            const xrt_mesh_t* m = o.mesh;
            if (!m || m->n_points < 3 || m->n_faces < 1 || !m->points || !m->p0 || !m->faces_normal || !m->p_faces_idx)
                return fail(-2, "%s", "mesh tables missing");
            if (m->interpolate && (m->n_simplices < 1 || !m->ct_simplices || !m->ct_grad)) return fail(-2, "%s", "mesh interpolation tables missing");

        }
        if (o.shape == XRT_SHAPE_TORUS && (o.torus_root < 0 || o.torus_root > 3))
            return fail(-2, "%s", "torus root index out of range");
        if (o.interact < XRT_INTERACT_NONE || o.interact > XRT_INTERACT_MOSAIC)
            return fail(-3, "%s", "optic interaction is not implemented on the device path");
        if (o.interact == XRT_INTERACT_MOSAIC && (o.mosaic_depth < 0 || o.mosaic_depth > 1000))
            return fail(-2, "%s", "mosaic_depth out of range");
        if (o.interact == XRT_INTERACT_MOSAIC && sc->source.intensity >= (1ll << 29))
            return fail(-2, "%s", "mosaic crystals take at most 2^29 - 1 rays per iteration (three ray-index bits carry flags)");
        if (o.n_apertures < 0 || o.n_apertures > XRT_MAX_APERTURES) return fail(-2, "%s", "bad aperture count");
        if ((o.flags & XRT_F_IMAGE) && (o.pixel_nx < 0 || o.pixel_ny < 0 || o.image_offset < 0 ||      // (0 x n images exist: np.round(0.5) = 0)
                                        o.image_offset + (int64_t)o.pixel_nx * o.pixel_ny > sc->image_bins))
            return fail(-2, "%s", "image layout inconsistent");
    }
    return 0;
}

// workspace layout: [run counter 256 B][KScene][apertures][voigt tables][KState in][seeds][streams][heads]
static int count_heads(const xrt_scene_t* sc)
{
    if (sc->source.kind == XRT_SRC_PLASMA) return 0;    // fused path: the draws come from the scout's stream dump
    int n = 2;      // the two angular arrays are always needed
    for (int i = 0; i < 3; i++) if (sc->source.size[i] != 0.0) n++;
    if (sc->source.wavelength_dist == XRT_WL_UNIFORM || sc->source.wavelength_dist == XRT_WL_VOIGT) n++;
    return n;
}
static size_t al256(size_t v) { return (v + 255) & ~(size_t)255; }
static size_t ws_off_scene() { return 256; }
static size_t ws_off_apertures() { return ws_off_scene() + al256(sizeof(KScene)); }
static size_t ws_off_voigt() { return ws_off_apertures() + sizeof(xrt_aperture_t) * XRT_MAX_APERTURES * XRT_DEV_MAX_OPTICS; }
static size_t ws_off_state(const xrt_scene_t* sc)
{
    return al256(ws_off_voigt() + 2 * sizeof(double) * (size_t)(sc->source.voigt_n > 0 ? sc->source.voigt_n : 0));
}
static size_t ws_off_seeds(const xrt_scene_t* sc) { return al256(ws_off_state(sc) + sizeof(KState)); }
static size_t ws_off_streams(const xrt_scene_t* sc, int n_runs) { return al256(ws_off_seeds(sc) + sizeof(uint32_t) * (size_t)n_runs); }
static size_t ws_off_heads(const xrt_scene_t* sc, int n_runs) { return al256(ws_off_streams(sc, n_runs) + sizeof(KStream) * (size_t)n_runs); }
static size_t ws_off_polys(const xrt_scene_t* sc, int n_runs)
{
    return al256(ws_off_heads(sc, n_runs) + sizeof(KStream) * (size_t)n_runs * (size_t)count_heads(sc));
}

// The staged path (xrt_staged.inc) is needed when the stream is consumed in a data-dependent
// way before the Bragg uniforms, or by more than one Bragg optic.
// np.random.normal wavelengths can be prepared as an array for the fused path (xrt_gauss_kernel) when the
// ray count is even (no cached second value is left over) and large enough for the jump-ahead
static bool gauss_prepared(const xrt_source_t& s)
{
    return s.wavelength_dist == XRT_WL_NORMAL && (s.intensity % 2) == 0 && s.intensity >= 2 * XRT_TILE &&
           !getenv("XICSRT_NO_JUMP") && !getenv("XICSRT_STAGED_GAUSS");
}

// Optics side of needs_staged: whole-array passes per mosaic layer, more than one Bragg element
static bool optics_need_staged(const xrt_scene_t* sc)
{
    int n_bragg = 0;
    for (int e = 0; e < sc->n_optics; e++) {
        if (sc->optics[e].interact == XRT_INTERACT_MOSAIC) return true;
        if (sc->optics[e].interact == XRT_INTERACT_CRYSTAL && (sc->optics[e].flags & XRT_F_CHECK_BRAGG)) n_bragg++;
    }
    return n_bragg > 1;
}
// A plasma scene whose optics the fused kernel can take: scout kernel + fused kernel (xrt_plasma.inc); only
// xrt_trace_history (one run, history) still sends such a scene through the staged kernel.
static size_t plasma_dump_words(const xrt_scene_t* sc);
static bool plasma_fused(const xrt_scene_t* sc)
{
    return sc->source.kind == XRT_SRC_PLASMA && !optics_need_staged(sc) && !getenv("XICSRT_PLASMA_STAGED") &&
           plasma_dump_words(sc) < (1ull << 31);       // the scout indexes its stream with 32 bits
}

static bool needs_staged(const xrt_scene_t* sc)
{
    const xrt_source_t& s = sc->source;
    if (s.kind == XRT_SRC_PLASMA || s.kind == XRT_SRC_EXTERNAL) return true;
    if (s.n_ray_filters > 0) return true;                  // rays switched off at the source
    if (s.spatial_dist == XRT_SPATIAL_GAUSSIAN || s.angular_dist == XRT_ANG_ISOTROPIC_XY) return true;
    if (s.wavelength_dist == XRT_WL_NORMAL && !gauss_prepared(s)) return true;
    int n_bragg = 0;
    for (int e = 0; e < sc->n_optics; e++) {
        if (sc->optics[e].interact == XRT_INTERACT_MOSAIC) return true;     // whole-array passes per layer
        if (sc->optics[e].interact == XRT_INTERACT_CRYSTAL && (sc->optics[e].flags & XRT_F_CHECK_BRAGG)) n_bragg++;
    }
    return n_bragg > 1;
}
// Staged path: one workgroup per slot, one per CU (XRT_ST_SLOTS = 256 CUs), fewer when the per-slot
// ray arrays of a scene would take more than XRT_ST_BUDGET bytes of workspace in total.
#define XRT_ST_SLOTS 256
// Workspace budgets.  The big regions of a workspace -- the slots of the staged path, the parked candidates of the one-pass
// routes -- are sized to budgets, beyond which a call works through its runs in batches or takes a leaner route.  The
// budgets are constants for a 288 GB part, scaled down on a device with less memory (a share of its TOTAL memory: the
// figure must not move between xrt_workspace_bytes and xrt_trace), and capped by xrt_set_workspace_budget (a caller whose
// allocation failed asks again with what is free) or XICSRT_WORKSPACE_BUDGET_MB.
#define XRT_ST_BUDGET_MAX   (96ull << 30)
#define XRT_CAND_BUDGET_MAX (48ull << 30)
static size_t g_budget_cap = 0;             // xrt_set_workspace_budget: 0 = none
static size_t device_total_bytes()
{
    static thread_local int c_dev = -2;
    static thread_local size_t c_total = 0;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return 0; }       // (no device: the constants)
    if (dev != c_dev) {
        size_t total = 0;
        if (hipDeviceTotalMem(&total, dev) != hipSuccess) { (void)hipGetLastError(); total = 0; }
        c_dev = dev; c_total = total;
    }
    return c_total;
}
static size_t budget_of(size_t most, double share)
{
    size_t b = most;
    const size_t total = device_total_bytes();
    if (total > 0 && (size_t)((double)total * share) < b) b = (size_t)((double)total * share);
    if (g_budget_cap > 0 && g_budget_cap < b) b = g_budget_cap;
    if (const char* e = getenv("XICSRT_WORKSPACE_BUDGET_MB")) { const long long v = atoll(e); if (v > 0 && ((size_t)v << 20) < b) b = (size_t)v << 20; }
    if (b < (1ull << 20)) b = 1ull << 20;
    return b;
}
#define XRT_ST_BUDGET   budget_of(XRT_ST_BUDGET_MAX, 0.40)
#define XRT_CAND_BUDGET budget_of(XRT_CAND_BUDGET_MAX, 0.20)
static int staged_slots_for(int n_runs, size_t per_slot_bytes, int per_cu)
{
    size_t s = (size_t)XRT_ST_SLOTS * (size_t)per_cu;
    if (per_slot_bytes > 0 && s * per_slot_bytes > XRT_ST_BUDGET) s = XRT_ST_BUDGET / per_slot_bytes;
    if (s < 1) s = 1;
    if ((size_t)n_runs < s) s = (size_t)(n_runs < 1 ? 1 : n_runs);
    return (int)s;
}
static size_t ws_off_gauss(const xrt_scene_t* sc, int n_runs) { return al256(ws_off_polys(sc, n_runs) + sizeof(uint32_t) * 624 * (XRT_MAX_HEADS + 1)); }
static size_t ws_off_staged(const xrt_scene_t* sc, int n_runs) { return al256(ws_off_gauss(sc, n_runs) + sizeof(KState) * (size_t)n_runs); }
// bytes of one mesh's tables on the device (each array 256-byte aligned) + its KMesh header
static size_t mesh_bytes(const xrt_mesh_t* m)
{
    if (!m) return 0;
    size_t b = al256(sizeof(KMesh));
    const size_t P = (size_t)m->n_points, F = (size_t)m->n_faces, Cn = (size_t)m->n_coarse_faces, T = (size_t)m->n_simplices;
    const size_t n_first = Cn > 0 ? Cn : F;
    b += al256(F * 24) + al256((n_first + 1) * 80) + al256((n_first + 1) * 128) + al256(F * sizeof(KFaceRec)) + al256(P * 32) + al256(P * 24);
    if (m->interpolate) b += al256(T * 128) + al256(T * 64) + al256(P * 128) + al256(P * 4) + al256(F * 4);
    b += al256((2 * P + 1) * sizeof(KCellRec));                       // bucket grid: <= P buckets + <= P chained points
    b += al256((64 * 64 + 1) * 4) + al256(n_first * 16 * 4 + 64) + al256(64 * 64 * 16);   // face grid of the first pass: cell starts, lists (<= 16 cells per face on average), cell slabs
    b += al256((n_first + 1) * 96);                                   // point-source form of the first pass
    if (Cn > 0) b += al256(F * 128);                                  // plane form of every face (second pass)
    b += al256(P * 16) + al256(F * 8);                                // 16-bit tables of the LDS form
    b += al256(P * 48);                                               // the fans (KMesh.lds_star)
    b += al256(128 * 128 * 16);                                       // direction grid of the point-source form
    return b;
}
static size_t meshes_bytes(const xrt_scene_t* sc)
{
    size_t b = 0;
    for (int e = 0; e < sc->n_optics; e++) if (sc->optics[e].shape == XRT_SHAPE_MESH) b += mesh_bytes(sc->optics[e].mesh);
    return b;
}

// device copy of the per-bundle plasma model: header + the two profile tables
static size_t plasma_bytes(const xrt_scene_t* sc)
{
    const xrt_plasma_t* P = sc->source.plasma;
    if (!P) return 0;
    return al256(sizeof(KPlasma)) + 2 * al256(sizeof(double) * (size_t)(P->n_emissivity > 0 ? P->n_emissivity : 1))
                                  + 2 * al256(sizeof(double) * (size_t)(P->n_temperature > 0 ? P->n_temperature : 1))
                                  + al256(sizeof(double) * (size_t)(P->n_weideman > 0 ? P->n_weideman : 1));
}

static size_t staged_slot_bytes(const xrt_scene_t* sc)
{
    const size_t n = (size_t)(sc->source.intensity > 0 ? sc->source.intensity : 1);
    const size_t nb = (size_t)(sc->source.bundle_count > 0 ? sc->source.bundle_count : 0);
    const size_t voigt = (sc->source.plasma && sc->source.plasma->voigt_gamma > 0.0) ? 2 * XRT_VOIGT_GRID * sizeof(double) : 0;
    return n * (XRT_ST_ARRAYS * sizeof(double) + 3 * sizeof(uint32_t)) + nb * XRT_ST_BUNDLE_ROWS * sizeof(double) + voigt + 16;
}
// source and optics run as separate launches over batches of up to 4 x 256 run slots
static int staged_slots(const xrt_scene_t* sc, int n_runs)
{
    if (plasma_fused(sc)) n_runs = 1;      // the staged kernel only serves xrt_trace_history then
    return staged_slots_for(n_runs, staged_slot_bytes(sc), sc->source.kind == XRT_SRC_EXTERNAL ? 1 : 4);
}
// plasma on the fused path: per run slot the scout's bundle tables, ray -> bundle map, normal wavelengths and
// the stream dump (every word of the source stage: 10-12 per ray, the Poisson trials, the rejected normal
// candidates); as many slots as runs within the budget
static size_t plasma_dump_words(const xrt_scene_t* sc)
{
    const size_t n = (size_t)(sc->source.intensity > 0 ? sc->source.intensity : 1);
    const size_t nb = (size_t)(sc->source.bundle_count > 0 ? sc->source.bundle_count : 0);
    return (16 * n + 86 * nb + 8192 + 63) / 64 * 64;
}
static size_t plasma_slot_bytes(const xrt_scene_t* sc)
{
    const size_t n = (size_t)(sc->source.intensity > 0 ? sc->source.intensity : 1);
    const size_t nb = (size_t)(sc->source.bundle_count > 0 ? sc->source.bundle_count : 0);
    return al256(nb * XRT_PB_ROWS * 8) + al256(nb * XRT_PP_ROWS * 8) + al256(n * 4) + al256(n * 8) + al256(plasma_dump_words(sc) * 4);
}
static int plasma_slots(const xrt_scene_t* sc, int n_runs)
{
    size_t s = XRT_ST_BUDGET / plasma_slot_bytes(sc);
    if (const char* e = getenv("XICSRT_PLASMA_SLOTS")) { const int v = atoi(e); if (v >= 1 && (size_t)v < s) s = (size_t)v; }   // tests: several batches
    if (s < 1) s = 1;
    if ((size_t)n_runs < s) s = (size_t)(n_runs < 1 ? 1 : n_runs);
    return (int)s;
}
static size_t plasma_ws_bytes(const xrt_scene_t* sc, int n_runs)
{
    if (!plasma_fused(sc)) return 0;
    return al256((size_t)plasma_slots(sc, n_runs) * (plasma_slot_bytes(sc) + 16) + 512);
}
static size_t staged_bytes(const xrt_scene_t* sc, int n_runs)
{
    // (a scene with Gaussian wavelengths can be sent to the staged path at run time: a cached gauss value)
    if (!needs_staged(sc) && sc->source.wavelength_dist != XRT_WL_NORMAL) return 0;
    return al256((size_t)staged_slots(sc, n_runs) * staged_slot_bytes(sc) + 256);
}

// Few runs of many rays: split every run into work units so that the whole chip has work (the unit of parallelism is
// otherwise one workgroup per run).  A run is cut into n_seg segments, each with its own jump-positioned source heads
// (a jump costs about as much as tracing 15 000 rays), and a segment into n_sub parts that share those heads: part j
// first walks them over the rays in front of its own (raw generation only, ~20 x faster than tracing them).  The
// stream behind the source arrays gets chunk heads, chunk_words apart; a unit's first Bragg uniform lies at most that
// far behind one of them.  Lengths are multiples of the tile.
struct SegPlan {
    int n_seg; int64_t seg_len;     // n_seg == 1 && seg_len == 0: one unit per run, unsegmented kernels
    int n_sub; int64_t sub_len;     // seg_len = n_sub * sub_len
    int n_chunk_heads; int64_t chunk_words;
    int n_gchunks; int64_t gpairs;  // Gaussian wavelengths: chunks of the candidate stream, candidate pairs per chunk
    bool mesh_split;                // a mesh crystal: first phase, xrt_mesh_rest_kernel, second phase as launches of their own
};
static int count_heads(const xrt_scene_t* sc);
static bool needs_ext(const xrt_scene_t* sc);
static int bragg_element(const xrt_scene_t* sc)
{
    int be = -1;
    for (int e = 0; e < sc->n_optics; e++)
        if (sc->optics[e].interact == XRT_INTERACT_CRYSTAL && (sc->optics[e].flags & XRT_F_CHECK_BRAGG)) be = e;
    return be;
}
// A mesh crystal that makes the Bragg test, and the scene's only mesh: the one-pass route in three launches (see
// xrt_trace_kernel, SEG == 3 / 4), whole runs included when there are many
static bool mesh_split_ok(const xrt_scene_t* sc)
{
    if (env_on("XICSRT_NO_MESH_SPLIT") || env_on("XICSRT_SEG_TWO_PASS") || getenv("XICSRT_NO_JUMP")) return false;
    const int be = bragg_element(sc);
    if (be < 0 || sc->optics[be].shape != XRT_SHAPE_MESH || sc->source.kind == XRT_SRC_PLASMA) return false;
    for (int e = 0; e < sc->n_optics; e++)
        if (e != be && sc->optics[e].shape == XRT_SHAPE_MESH) return false;
    return sc->source.intensity >= 4096;
}
// Cost model of a plan in microseconds (measured on MI355X).  The jump kernel takes a run's jobs in
// groups of eight: 20 us for a run's stretch + 255 us of one CU per group, spread evenly over the CUs.  A walk costs
// latency: the unit at the end of the longest walk finishes last.
#define XRT_COST_STRETCH_US 20.0
#define XRT_COST_GROUP_US   255.0
#define XRT_COST_SKIP_US    1.0     // a workgroup walking its source heads over one tile of rays (0.54 us) + what the late start costs the unit
#define XRT_COST_SSKIP_US   0.6     // a wave walking the stream head over 512 words
static double jump_cost_us(int n_runs, int jobs_per_run)
{
    // (the groups are spread evenly over the 256 CUs in shares of 1/8 - 1 group; a CU rebuilds the stretch for every run it touches)
    const double groups = (double)n_runs * (double)((jobs_per_run + 7) / 8);
    const double per_cu = groups / 256.0;
    const double runs_per_cu = per_cu < 1.0 ? 1.0 : (double)n_runs / 256.0 + 1.0;
    return XRT_COST_STRETCH_US * runs_per_cu + XRT_COST_GROUP_US * (per_cu < 0.125 ? 0.125 : per_cu) * 1.1;
}
static SegPlan plan_segments(const xrt_scene_t* sc, int n_runs)
{
    SegPlan p = {1, 0, 1, 0, 0, 0, 0, 0, false};
    if (needs_staged(sc)) return p;
    const bool msplit = mesh_split_ok(sc);
    const int64_t N = sc->source.intensity;
    const bool gauss = gauss_prepared(sc->source);
    int want = 0, want_sub = 0;                 // units per run, parts per segment (0: by the cost model)
    // (up to 512 runs: two units per run and more fill the chip's 1024 workgroup slots; round 3 stopped at 255, and 256 - 511
    //  runs left half of the slots empty)
    int seg_below = 513;
    if (const char* e = getenv("XICSRT_SEG_BELOW")) { const int v = atoi(e); if (v >= 1) seg_below = v; }
    int64_t min_len = 4096;                     // below this the set-up of a unit outweighs its rays
    if (const char* e = getenv("XICSRT_SEGMENTS")) { want = atoi(e); want_sub = 1; min_len = XRT_TILE; }
    else if (n_runs < seg_below) {
        // one round of units, all resident together and of one size: as many as workgroups fit on the chip (a second
        // round for a few units more would double the time), four per CU (two for the mesh / local-frame variant)
        int target = 256 * ((needs_ext(sc) && !msplit) ? 2 : 4);
        if (const char* t = getenv("XICSRT_TARGET_UNITS")) target = atoi(t) > 0 ? atoi(t) : target;
        want = target / n_runs;
    }
    if (const char* e = getenv("XICSRT_SUBUNITS")) want_sub = atoi(e) > 0 ? atoi(e) : want_sub;
    int want_ch = 0;
    if (const char* e = getenv("XICSRT_CHUNK_HEADS")) want_ch = atoi(e) > 0 ? atoi(e) : 0;
    // Gaussian wavelengths: chunks of the candidate stream
    uint64_t gauss_words = 0;
    if (gauss) {
        // candidate pairs that certainly yield N/2 accepted ones (acceptance pi/4; mean + ~8 sigma + slack)
        const double need = (double)(N / 2);
        const int64_t cand = (int64_t)(need / 0.7853981633974483 + 8.0 * sqrt(need) + 1024.0);
        int chunks = (512 + n_runs - 1) / n_runs;
        const int64_t most = (cand + 1023) / 1024;                       // at least 1024 pairs per chunk
        if (chunks > most) chunks = (int)most;
        if (chunks < 1) chunks = 1;
        int64_t per = (cand + chunks - 1) / chunks;
        per = (per + 63) / 64 * 64;
        p.gpairs = per;
        p.n_gchunks = (int)((cand + per - 1) / per);
        gauss_words = 4ull * (uint64_t)p.gpairs * (uint64_t)p.n_gchunks;
    }
    // chunk heads cover the words behind the source arrays that Bragg draws can start at: [0, gauss_words + 2 N]
    const uint64_t W = gauss_words + 2ull * (uint64_t)N;
    const bool bragg = bragg_element(sc) >= 0;
    const int nh = count_heads(sc) > 0 ? count_heads(sc) : 1;
    int64_t units = 1, sublen = 0;
    const bool split = !(want <= 1 || N < 2 * XRT_TILE || getenv("XICSRT_NO_JUMP"));
    if (split) {
        sublen = (N + want - 1) / want;
        if (sublen < min_len) sublen = min_len;
        sublen = (sublen + XRT_TILE - 1) / XRT_TILE * XRT_TILE;
        units = (N + sublen - 1) / sublen;
    }
    // parts per segment x chunk heads: the pair with the least jump time + longest walks
    int64_t sub = 1;
    int n_ch = 1;
    {
        double best = 1e300;
        const int sub_lo = want_sub > 0 ? want_sub : 1, sub_hi = want_sub > 0 ? want_sub : 16;
        for (int64_t m = sub_lo; m <= sub_hi && m <= (units > 1 ? units : 1); m++) {
            const int64_t segs = (units + m - 1) / m;
            const double walk = (double)(m - 1) * (double)(sublen / XRT_TILE) * XRT_COST_SKIP_US;
            for (double c = 1.0; c <= 2048.0; c = (c < 8.0 ? c + 1.0 : floor(c * 1.25))) {
                int ch = bragg ? (want_ch > 0 ? want_ch : (int)c) : 1;
                const double swalk = bragg && units > 1 ? 0.75 * ((double)W / (double)ch / 512.0) * XRT_COST_SSKIP_US : 0.0;
                const double cost = jump_cost_us(n_runs, (int)(segs * nh) + ch + p.n_gchunks) + walk + swalk;
                if (cost < best) { best = cost; sub = m; n_ch = ch; }
                if (!bragg || want_ch > 0) break;
            }
        }
    }
    int64_t n = 1, len = 0;
    if (split) {
        if (sub > units) sub = units;
        len = sub * sublen;
        n = (N + len - 1) / len;
        if (n * sub <= 1 || n > 4096) { n = 1; len = 0; sub = 1; sublen = 0; }
    } else sub = 1;
    if (n == 1 && sub == 1 && !gauss && !msplit) { p.gpairs = 0; p.n_gchunks = 0; return p; }
    p.mesh_split = msplit;
    if (n == 1 && sub == 1) { len = (N + XRT_TILE - 1) / XRT_TILE * XRT_TILE; sublen = len; }      // whole runs through the SEG kernels
    p.n_seg = (int)n; p.seg_len = len; p.n_sub = (int)sub; p.sub_len = sublen;
    if (!bragg) {
        // no Bragg draws: only the run's last unit opens the stream, to hand it on (behind the Gaussian words)
        p.n_chunk_heads = 1; p.chunk_words = (int64_t)(W + 2);
    } else {
        uint64_t CH = W / (uint64_t)n_ch + 1;
        CH = (CH + 1023) / 1024 * 1024;
        p.chunk_words = (int64_t)CH;
        p.n_chunk_heads = (int)(W / CH) + 1;
    }
    return p;
}
static bool seg_active(const SegPlan& p) { return p.n_seg > 1 || p.n_sub > 1 || p.n_gchunks > 0 || p.mesh_split; }
// jobs of a run: [segment][source head], the chunk heads, the chunk heads of the Gaussian candidate stream
static int seg_jobs(const xrt_scene_t* sc, const SegPlan& p) { return p.n_seg * count_heads(sc) + p.n_chunk_heads + p.n_gchunks; }
// segmented runs: [dst heads n_runs x n_jobs][polys][offsets][unit counts / flags][gauss: values, chunk counts, words]
// the jump kernel cuts a run's groups of eight jobs into 1 - 8 shares so that every CU gets at least two work items
static int jump_shares(int n_runs, int n_jobs)
{
    const long long groups = (long long)n_runs * ((n_jobs + 7) / 8);
    int shares = 1;
    while (shares < 8 && groups * shares < 512) shares *= 2;
    return shares;
}
static size_t jump_partial_bytes(int n_runs, size_t nj)
{
    const int sh = jump_shares(n_runs, (int)nj);
    return al256(sh > 1 ? sizeof(uint32_t) * 624 * nj * (size_t)n_runs * (size_t)sh : 256);
}
static size_t seg_bytes(const xrt_scene_t* sc, int n_runs)
{
    const SegPlan p = plan_segments(sc, n_runs);
    if (!seg_active(p)) return 0;
    const size_t nj = (size_t)seg_jobs(sc, p);
    size_t b = al256(sizeof(KStream) * nj * (size_t)n_runs) + jump_partial_bytes(n_runs, nj) + al256(sizeof(uint64_t) * nj)
               + al256(sizeof(uint32_t) * (size_t)n_runs * (size_t)p.n_seg * (size_t)p.n_sub) + 256;
    if (p.n_gchunks > 0)
        b += al256(sizeof(double) * (size_t)n_runs * (size_t)sc->source.intensity) + al256(sizeof(uint32_t) * (size_t)n_runs * (size_t)p.n_gchunks)
             + al256(sizeof(uint64_t) * (size_t)n_runs);
    return b;
}
// One-pass segmented runs park every unit's Bragg candidates in HBM: per run and ray of capacity 7 doubles, the ray
// index and the hit face (64 B).  At the end of the workspace; beyond the budget the two-pass route is taken.
#define XRT_CAND_BUDGET_TAIL (2ull << 30)       // for the < 256 runs an unsegmented launch leaves to a second pass
static size_t cand_capacity(const xrt_scene_t* sc) { return ((size_t)(sc->source.intensity > 0 ? sc->source.intensity : 1) + 255) & ~(size_t)255; }
// doubles of a parked ray's record between the launches of a split trace (KArgs.split_interp): origin, direction
// [, interpolated normal] [, wavelength]
// (the mesh straight behind a source without extent: one origin for every parked ray of a unit, KArgs.unit_o)
static bool split_shared_origin(const xrt_scene_t* sc)
{
    const xrt_source_t& src = sc->source;
    return bragg_element(sc) == 0 && (src.kind == XRT_SRC_GENERIC || src.kind == XRT_SRC_DIRECTED) && src.spatial_dist == XRT_SPATIAL_UNIFORM &&
           src.size[0] == 0.0 && src.size[1] == 0.0 && src.size[2] == 0.0 && !env_on("XICSRT_NO_SHARED_ORIGIN");
}
static int split_comps(const xrt_scene_t* sc)
{
    const int be = bragg_element(sc);
    const bool interp = be >= 0 && sc->optics[be].mesh && sc->optics[be].mesh->interpolate;
    const bool has_wl = !(sc->source.wavelength_dist == XRT_WL_CONST && !sc->source.has_velocity);
    const bool lean = !interp && split_shared_origin(sc);       // (KArgs.split_lean: the records hold no origin)
    return (lean ? 3 : (interp ? 9 : 6)) + (has_wl ? 1 : 0);
}
static size_t mesh_split_off_aux(const xrt_scene_t* sc, int n_runs) { return al256((size_t)n_runs * cand_capacity(sc) * 8 * (size_t)split_comps(sc)); }
static size_t mesh_split_off_batch_alive(const xrt_scene_t* sc, int n_runs) { return mesh_split_off_aux(sc, n_runs) + al256((size_t)n_runs * cand_capacity(sc) * 4); }
static size_t mesh_split_off_unit_alive(const xrt_scene_t* sc, int n_runs) { return mesh_split_off_batch_alive(sc, n_runs) + al256((size_t)n_runs * (cand_capacity(sc) / 64) * 4); }
// (behind the units' counts: the rays the fan launch leaves to the list walk, per unit, and how many -- KArgs.slow_q)
static size_t mesh_split_off_slow_q(const xrt_scene_t* sc, int n_runs, const SegPlan& p) { return mesh_split_off_unit_alive(sc, n_runs) + al256(sizeof(uint32_t) * (size_t)n_runs * (size_t)p.n_seg * (size_t)p.n_sub); }
static size_t mesh_split_off_slow_cnt(const xrt_scene_t* sc, int n_runs, const SegPlan& p) { return mesh_split_off_slow_q(sc, n_runs, p) + al256((size_t)n_runs * cand_capacity(sc) * 4); }
static size_t mesh_split_off_slow_q2(const xrt_scene_t* sc, int n_runs, const SegPlan& p) { return mesh_split_off_slow_cnt(sc, n_runs, p) + al256((size_t)n_runs * (cand_capacity(sc) / 64) * 4); }
static size_t mesh_split_off_unit_slow(const xrt_scene_t* sc, int n_runs, const SegPlan& p) { return mesh_split_off_slow_q2(sc, n_runs, p) + al256((size_t)n_runs * cand_capacity(sc) * 4); }
// (behind those: the list of the blocks of 1024 records that hold rays, KArgs.items, twice -- all parked rays, and the rays
//  left to the list walk -- 16 bytes per block, and the lists' lengths)
static size_t mesh_split_items_max(const xrt_scene_t* sc, int n_runs, const SegPlan& p)
{
    const size_t bpu4 = ((size_t)(p.sub_len / XRT_TILE) + 3) / 4;
    return (size_t)n_runs * (size_t)p.n_seg * (size_t)p.n_sub * bpu4;
}
static size_t mesh_split_off_items(const xrt_scene_t* sc, int n_runs, const SegPlan& p)
{
    return mesh_split_off_unit_slow(sc, n_runs, p) + al256(sizeof(uint32_t) * (size_t)n_runs * (size_t)p.n_seg * (size_t)p.n_sub);
}
// (... and the units' shared origins, KArgs.unit_o)
static size_t mesh_split_off_unit_o(const xrt_scene_t* sc, int n_runs, const SegPlan& p)
{
    return mesh_split_off_items(sc, n_runs, p) + 2 * al256(16 * mesh_split_items_max(sc, n_runs, p)) + 256;
}
static size_t mesh_split_end(const xrt_scene_t* sc, int n_runs, const SegPlan& p)
{
    return mesh_split_off_unit_o(sc, n_runs, p) + al256(24 * (size_t)n_runs * (size_t)p.n_seg * (size_t)p.n_sub) + 256;
}
static size_t cand_bytes(const xrt_scene_t* sc, int n_runs, size_t budget)
{
    const SegPlan p = plan_segments(sc, n_runs);
    if (!seg_active(p) || bragg_element(sc) < 0 || env_on("XICSRT_SEG_TWO_PASS")) return 0;
    if (p.mesh_split) {
        // split phases: 10 doubles (+ the normal) and the face per ray, the rays left alive per 64 and per unit.  (A call whose
        // runs would take more than the budget goes through them in batches, mesh_batch_runs; what is left over the budget
        // here is a single run of > 6e8 rays.)
        // A history call of the same scene (xrt_trace_history: one run) takes the one-pass route WITHOUT the split (64 B per ray
        // of capacity: 7 doubles, ray index, face) over the same workspace: a single run's region holds whichever is larger --
        // records without the origin (KArgs.split_lean) make the split's the smaller one.
        size_t b = mesh_split_end(sc, n_runs, p) + 256;
        if (b > budget) return 0;                       // (the budget is the split route's: what the history call needs on top is one run's)
        const size_t plain = al256((size_t)n_runs * cand_capacity(sc) * 64) + 256;
        if (n_runs == 1 && p.n_seg * p.n_sub > 1 && plain > b) b = plain;
        return b;
    }
    if (p.n_seg * p.n_sub <= 1) return 0;
    const size_t b = al256((size_t)n_runs * cand_capacity(sc) * 64) + 256;
    return b <= budget ? b : 0;
}

static size_t ws_off_plasma_rays(const xrt_scene_t* sc, int n_runs)
{
    return al256(ws_off_staged(sc, n_runs) + staged_bytes(sc, n_runs) + meshes_bytes(sc) + plasma_bytes(sc) + seg_bytes(sc, n_runs) + 256);
}
// Replicas of the pixel bins for the fused kernel (see KArgs.image_rep): as many as fit 64 MiB, at most 32; behind
// the plasma arrays.  One replica = none (the kernel adds into the caller's bins).
static int image_replicas(const xrt_scene_t* sc)
{
    if (sc->image_bins <= 0) return 1;
    int r = 16;
    if (const char* e = getenv("XICSRT_IMAGE_REPLICAS")) { const int v = atoi(e); if (v >= 1 && v <= 64) r = v; }
    while (r > 1 && (size_t)r * (size_t)sc->image_bins * 8 > (64ull << 20)) r >>= 1;
    return r;
}
static size_t image_rep_bytes(const xrt_scene_t* sc)
{
    const int r = image_replicas(sc);
    return r > 1 ? al256((size_t)r * (size_t)sc->image_bins * 8) + 256 : 0;
}
static size_t ws_off_image_rep(const xrt_scene_t* sc, int n_runs) { return al256(ws_off_plasma_rays(sc, n_runs) + plasma_ws_bytes(sc, n_runs) + 256); }
// everything but the candidate arrays of the one-pass segmented route (they come last)
static size_t ws_base_bytes(const xrt_scene_t* sc, int n_runs) { return ws_off_image_rep(sc, n_runs) + image_rep_bytes(sc) + 256; }

__global__ void xrt_image_reduce_kernel(unsigned long long* rep, int n_rep, uint64_t bins, unsigned long long* out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= bins) return;
    unsigned long long s = 0;
    for (int r = 0; r < n_rep; r++) { s += rep[(size_t)r * bins + i]; rep[(size_t)r * bins + i] = 0ULL; }
    if (s) out[i] += s;
}
// the fused kernels of the calls in between add into the replicas (zeroed once per xrt_trace / xrt_trace_history)
static int image_rep_begin(const xrt_scene_t* sc, char* ws, int n_runs, KArgs* a, hipStream_t stream)
{
    a->image_rep = 1; a->image_stride = 0;
    const int r = image_replicas(sc);
    if (!a->images || r <= 1) return 0;
    unsigned long long* rep = reinterpret_cast<unsigned long long*>(ws + ws_off_image_rep(sc, n_runs));
    HIP_TRY(hipMemsetAsync(rep, 0, (size_t)r * (size_t)sc->image_bins * 8, stream));
    a->image_rep = (uint32_t)r; a->image_stride = (uint64_t)sc->image_bins; a->images_rep = rep;
    return 0;
}
static int image_rep_end(const xrt_scene_t* sc, char* ws, int n_runs, const KArgs& a, unsigned long long* out, hipStream_t stream)
{
    if (a.image_rep <= 1 || !out) return 0;
    unsigned long long* rep = reinterpret_cast<unsigned long long*>(ws + ws_off_image_rep(sc, n_runs));
    hipLaunchKernelGGL(xrt_image_reduce_kernel, dim3((unsigned)((sc->image_bins + 255) / 256)), dim3(256), 0, stream,
                       rep, (int)a.image_rep, (uint64_t)sc->image_bins, out);
    HIP_TRY(hipGetLastError());
    return 0;
}
static bool tail_split_possible(const xrt_scene_t* sc);
// A mesh crystal's launches park 60 - 92 bytes per ray of capacity between them (split_comps): a call whose runs would take more than the budget
// goes through them in equal batches (every batch a call of its own over the same workspace: the sums are sums).  A batch of
// fewer than 1024 runs is cut into work units by plan_segments (up to 512 runs), so a batch that cannot hold 1024 runs holds
// at most 512.  Returns the runs per batch (n_runs: no batches).
static bool mesh_split_ok(const xrt_scene_t* sc);
static int mesh_batch_runs(const xrt_scene_t* sc, int n_runs)
{
    if (n_runs < 2 || needs_staged(sc) || !mesh_split_ok(sc)) return n_runs;
    const size_t budget = XRT_CAND_BUDGET;
    const size_t per_run = cand_capacity(sc) * (8 * (size_t)split_comps(sc) + 12) + cand_capacity(sc) / 8 + 64;
    if (per_run * (size_t)n_runs + (1ull << 20) <= budget) return n_runs;
    long long most = (long long)((budget - (budget < (2ull << 20) ? 0 : (1ull << 20))) / per_run);
    if (most < 1) return n_runs;                    // (a single run beyond the budget: cand_bytes picks the route)
    if (most < 1024 && most > 512) most = 512;
    if (most >= n_runs) return n_runs;
    const int n_batches = (int)((n_runs + most - 1) / most);
    return (n_runs + n_batches - 1) / n_batches;
}
extern "C" size_t xrt_workspace_bytes(const xrt_scene_t* sc, int32_t n_runs)
{
    if (!sc || n_runs < 0) return 0;
    if (n_runs < 1) n_runs = 1;
    {
        const int per = mesh_batch_runs(sc, n_runs);
        if (per < n_runs) {
            // the batches' layouts: `per` runs, and what is left for the last one
            const int last = n_runs - (n_runs - 1) / per * per;
            size_t need = ws_base_bytes(sc, per) + cand_bytes(sc, per, XRT_CAND_BUDGET);
            const size_t b = ws_base_bytes(sc, last) + cand_bytes(sc, last, XRT_CAND_BUDGET);
            return b > need ? b : need;
        }
    }
    size_t need = ws_base_bytes(sc, n_runs) + cand_bytes(sc, n_runs, XRT_CAND_BUDGET);
    // xrt_trace may leave the last < 256 runs of an unsegmented launch to a second pass over the same workspace,
    // which takes the segmented route with a layout of its own: the workspace holds that as well
    if (n_runs > 256 && tail_split_possible(sc))
        for (int t = 1; t < 256; t++) {
            const size_t b = ws_base_bytes(sc, t) + cand_bytes(sc, t, XRT_CAND_BUDGET_TAIL);
            if (b > need) need = b;
        }
    return need;
}
static size_t ws_off_seg(const xrt_scene_t* sc, int n_runs)
{
    return al256(ws_off_staged(sc, n_runs) + staged_bytes(sc, n_runs) + meshes_bytes(sc) + plasma_bytes(sc));
}

// fused-kernel variant 2: an optic traced in its local frame, or a mesh
static bool needs_ext(const xrt_scene_t* sc)
{
    for (int e = 0; e < sc->n_optics; e++)
        if ((sc->optics[e].flags & XRT_F_TRACE_LOCAL) || sc->optics[e].shape == XRT_SHAPE_MESH) return true;
    return false;
}

static bool needs_full(const xrt_scene_t* sc)
{
    const xrt_source_t& s = sc->source;
    if (s.kind == XRT_SRC_FOCUSED || s.kind == XRT_SRC_PLASMA || s.angular_dist != XRT_ANG_ISOTROPIC || s.wavelength_dist == XRT_WL_VOIGT) return true;
    if (s.wavelength_dist == XRT_WL_NORMAL && s.has_velocity) return true;     // Doppler shift of a prepared wavelength
    for (int e = 0; e < sc->n_optics; e++) {
        const xrt_optic_t& o = sc->optics[e];
        if (o.shape == XRT_SHAPE_CYLINDER || o.shape == XRT_SHAPE_TORUS) return true;
        if ((o.flags & XRT_F_CHECK_APERTURE) && o.n_apertures > 0) return true;
    }
    return false;
}

// Constants of the rocking-curve screen in bragg_accept (not part of the reference's arithmetic: they only
// decide which candidates need the exact evaluation).  Usable when all rays share one wavelength.
static void bragg_screen(const xrt_source_t& src, KOptic& q)
{
    q.scr_ok = 0;
    q.scr_s = q.scr_a1 = q.scr_a2 = q.scr_a3 = q.scr_binv = q.scr_dmax = 0.0;
    q.scr_ptail = INFINITY;
    q.scr2_ok = 0; q.scr2_tail = INFINITY; q.inv_two_d = 0.0;
    // (a mosaic crystal's layers make the same test with the crystallite's normal: optics/_InteractMosaicCrystal.py:96-103)
    if ((q.interact != XRT_INTERACT_CRYSTAL && q.interact != XRT_INTERACT_MOSAIC) || !(q.flags & XRT_F_CHECK_BRAGG)) return;
    const double R = q.reflectivity;
    if (!(R >= 0.0) || !std::isfinite(R)) return;
    // rays with their own wavelength: |sin(inc - bragg)| >= 0.01 must put p below every non-zero deviate
    if (q.two_d > 0.0 && std::isfinite(q.two_d)) {
        q.inv_two_d = 1.0 / q.two_d;
        if (q.rocking_type == XRT_ROCKING_STEP) {
            if (q.half_fwhm >= 0.0 && q.half_fwhm < 0.0099) { q.scr2_ok = 1; q.scr2_tail = 0.0; }
        } else if (q.two_sigma2 > 0.0 && std::isfinite(q.two_sigma2) && 1e-4 / q.two_sigma2 >= 46.0) {
            q.scr_binv = 1.0 / q.two_sigma2;
            q.scr2_ok = 1; q.scr2_tail = R * exp(-45.0) * 1.001;
        }
    }
    if (src.wavelength_dist != XRT_WL_CONST || src.has_velocity) return;
    const double s0 = (1.0 * src.wavelength) / q.two_d;
    if (!(fabs(s0) < 0.995)) return;
    const double u = 1.0 - s0 * s0;
    const double a1 = 1.0 / sqrt(u), a2 = s0 / (2.0 * u * sqrt(u)), a3 = (1.0 + 2.0 * s0 * s0) / (6.0 * u * u * sqrt(u));
    auto rel_err = [&](double d) {
        const double exact = asin(s0 + d) - asin(s0), ser = d * (a1 + d * (a2 + d * a3));
        return fabs(ser - exact) / fabs(exact);
    };
    double dmax = 0.05 * u;
    for (int it = 0; it < 200; it++) {
        double worst = 0.0;
        for (int j = 4; j <= 8; j++) {
            const double d = dmax * j / 8.0;
            if (fabs(s0 + d) >= 1.0 || fabs(s0 - d) >= 1.0) { worst = 1.0; break; }
            const double e1 = rel_err(d), e2 = rel_err(-d);
            if (e1 > worst) worst = e1;
            if (e2 > worst) worst = e2;
        }
        if (worst <= 1e-6) break;
        dmax *= 0.8;
        if (it == 199) return;
    }
    if (q.rocking_type == XRT_ROCKING_STEP) {
        if (!(q.half_fwhm >= 0.0) || !std::isfinite(q.half_fwhm)) return;
        // outside the radius |inc - bragg| >= |d| >= dmax: p = 0 when that is beyond the half width
        q.scr_ptail = (dmax > q.half_fwhm * (1.0 + 1e-6) + 1e-12) ? 0.0 : INFINITY;
    } else {
        if (!(q.two_sigma2 > 0.0) || !std::isfinite(q.two_sigma2)) return;
        q.scr_binv = 1.0 / q.two_sigma2;
        if (dmax * dmax * q.scr_binv >= 46.0) q.scr_ptail = R * exp(-45.0) * 1.001;
    }
    q.scr_s = s0; q.scr_a1 = a1; q.scr_a2 = a2; q.scr_a3 = a3; q.scr_dmax = dmax;
    q.scr_ok = 1;
}

// the device form of one optic (table pointers are set by the caller)
static void fill_koptic(const xrt_optic_t& o, KOptic& q)
{
    memset(&q, 0, sizeof(q));
    q.shape = o.shape; q.interact = o.interact; q.flags = o.flags; q.rocking_type = o.rocking_type;
    for (int i = 0; i < 3; i++) { q.origin[i] = o.origin[i]; q.half_size[i] = o.half_size[i]; q.center[i] = o.center[i]; }
    for (int i = 0; i < 9; i++) q.R[i] = o.orientation[i];
    q.radius = o.radius; q.radius2 = o.radius2;
    q.torus_major = o.torus_major; q.torus_root = o.torus_root;
    for (int i = 0; i < 5; i++) q.torus_k[i] = o.torus_k[i];
    q.two_d = o.two_d; q.reflectivity = o.reflectivity; q.half_fwhm = o.rocking_half_fwhm;
    q.two_sigma2 = o.rocking_2sigma2; q.half_pi = o.half_pi;
    q.mosaic_depth = o.mosaic_depth; q.mosaic_has_cutoff = o.mosaic_has_cutoff;
    q.mosaic_cutoff_angle = o.mosaic_cutoff_angle;
    for (int i = 0; i < 4; i++) q.mosaic_A[i] = o.mosaic_A[i];
    q.pixel_size = o.pixel_size; q.pixel_xoff = o.pixel_xoff; q.pixel_yoff = o.pixel_yoff;
    q.pixel_nx = o.pixel_nx; q.pixel_ny = o.pixel_ny; q.image_offset = o.image_offset;
    q.n_apertures = o.n_apertures;
    q.scr_ptail = INFINITY;
}

static void build_kscene(const xrt_scene_t* sc, char* ws, KScene* k)
{
    memset(k, 0, sizeof(*k));
    const xrt_source_t& s = sc->source;
    KSource& d = k->src;
    d.kind = s.kind; d.angular_dist = s.angular_dist; d.wavelength_dist = s.wavelength_dist;
    d.has_velocity = s.has_velocity;
    d.n_rays = s.intensity;
    for (int i = 0; i < 3; i++) {
        d.origin[i] = s.origin[i];
        d.xaxis[i] = s.orientation[i]; d.yaxis[i] = s.orientation[3 + i]; d.zaxis[i] = s.orientation[6 + i];
        // np.random.uniform(-1*size/2, size/2, n) (_XicsrtSourceGeneric.py:233-235)
        const double low = -1.0 * s.size[i] / 2.0, high = s.size[i] / 2.0;
        d.low[i] = low; d.range[i] = high - low;
        d.axis[i] = s.axis[i]; d.velocity[i] = s.velocity[i];
    }
    for (int i = 0; i < 9; i++) d.basis[i] = s.basis[i];
    for (int i = 0; i < 5; i++) d.ang[i] = s.ang[i];
    d.two_pi = s.two_pi; d.wavelength = s.wavelength; d.wl_a = s.wl_a; d.wl_b = s.wl_b;
    d.light_speed = s.light_speed;
    d.voigt_n = s.voigt_n;
    d.voigt_cdf = reinterpret_cast<const double*>(ws + ws_off_voigt());
    d.voigt_x = d.voigt_cdf + (s.voigt_n > 0 ? s.voigt_n : 0);
    d.bundle_count = s.bundle_count;
    d.bundle_intensity = s.bundle_intensity;
    d.use_poisson = s.use_poisson;
    d.ext_rays = s.ext_rays; d.ext_mask = s.ext_mask;
    d.n_ray_filters = s.n_ray_filters;
    for (int i = 0; i < XRT_MAX_BUNDLE_FILTERS; i++) d.ray_filters[i] = s.ray_filters[i];
    for (int i = 0; i < 3; i++) {
        const double low = -1.0 * s.plasma_size[i] / 2.0, high = s.plasma_size[i] / 2.0;
        d.plasma_low[i] = low; d.plasma_range[i] = high - low;
    }
    {   // constants of random_poisson_ptrs / _mult, with the host libm numpy itself uses
        const double lam = s.bundle_intensity;
        const double slam = sqrt(lam > 0 ? lam : 0.0), loglam = log(lam > 0 ? lam : 1.0);
        const double b = 0.931 + 2.53 * slam;
        const double a = -0.059 + 0.02483 * b;
        d.pois[0] = slam; d.pois[1] = loglam; d.pois[2] = b; d.pois[3] = a;
        d.pois[4] = 1.1239 + 1.1328 / (b - 3.4);
        d.pois[5] = 0.9277 - 3.6224 / (b - 2);
        d.pois[6] = exp(-lam);
    }
    const bool wl_array = (s.wavelength_dist == XRT_WL_UNIFORM || s.wavelength_dist == XRT_WL_VOIGT);
    d.n_arrays = wl_array ? 6 : 5;
    d.array_used = 0;
    for (int i = 0; i < 3; i++) if (s.size[i] != 0.0) d.array_used |= 1u << i;
    d.array_used |= (1u << 3) | (1u << 4);
    if (wl_array) d.array_used |= 1u << 5;
    if (s.kind == XRT_SRC_PLASMA) d.array_used = 0;     // no positioned heads: see xrt_plasma.inc

    k->n_optics = sc->n_optics;
    for (int e = 0; e < sc->n_optics; e++) {
        KOptic& q = k->opt[e];
        fill_koptic(sc->optics[e], q);
        q.apertures = reinterpret_cast<const xrt_aperture_t*>(ws + ws_off_apertures()) + (size_t)e * XRT_MAX_APERTURES;
        bragg_screen(s, q);
    }
}

static size_t lds_bytes(int n_src_heads, bool ext, bool hist, bool has_wl, uint32_t qcap, uint32_t lbins_words = 0)
{
    size_t b = sizeof(double) * (has_wl ? 7 : 6) * qcap + sizeof(uint32_t) * qcap * ((ext ? 1 : 0) + (hist ? 1 : 0));
    b += sizeof(uint32_t) * XRT_RING * (size_t)(n_src_heads + 1);
    b += sizeof(uint32_t) * (8 + 2 * (XRT_DEV_MAX_OPTICS + 2) + 8);
    b += sizeof(uint32_t) * (size_t)lbins_words;
    return (b + 15) & ~(size_t)15;
}

// Pixel bins pre-aggregated in LDS (fused kernel variant 4) pay when most rays end in a pixel and the bins fit:
// a lean scene without a Bragg test (nothing thins the rays out in front of the imaged elements), images wanted,
// no history, at most 2^14 bins (32 KiB: three workgroups per CU stay resident).
#define XRT_LBINS_MAX 16384
static bool lds_bins_wanted(const xrt_scene_t* sc, bool images, bool hist)
{
    if (!images || hist || env_on("XICSRT_NO_LDS_BINS")) return false;
    if (sc->image_bins <= 0 || sc->image_bins > XRT_LBINS_MAX) return false;
    for (int e = 0; e < sc->n_optics; e++)
        if (sc->optics[e].interact == XRT_INTERACT_CRYSTAL && (sc->optics[e].flags & XRT_F_CHECK_BRAGG)) return false;
    return true;
}
// Bragg batches of 256 candidates (all four waves busy, half as many batches) when the larger ray buffer
// does not cost a workgroup per CU (160 KiB of LDS), else 128
static size_t plan_queue(const KScene& ks, int n_src_heads, bool ext, bool hist, KArgs* a, uint32_t lbins_words = 0, int wmax_given = 0)
{
    const bool has_wl = !(ks.src.wavelength_dist == XRT_WL_CONST && !ks.src.has_velocity);
    a->lbins_words = lbins_words;
    if (lbins_words > 0) {          // variant 4: no Bragg queue, no compaction buffer, no stream-head ring
        a->qcap = 0u; a->bragg_batch = 64u;
        return lds_bytes(n_src_heads > 0 ? n_src_heads - 1 : 0, false, false, has_wl, 0u, lbins_words);
    }
    const size_t l128 = lds_bytes(n_src_heads, ext, hist, has_wl, XRT_QCAP_128);
    const size_t l256 = lds_bytes(n_src_heads, ext, hist, has_wl, XRT_QCAP_256);
    const size_t cu = 160u * 1024u;
    size_t w128 = cu / l128, w256 = cu / l256;
    const size_t wmax = wmax_given > 0 ? (size_t)wmax_given : (ext ? 2 : 4);            // what the register budget of the variant allows anyway
    if (w128 > wmax) w128 = wmax;
    if (w256 > wmax) w256 = wmax;
    const bool big = w256 >= w128 && w256 >= 1 && !getenv("XICSRT_BRAGG_BATCH_128");
    a->qcap = big ? XRT_QCAP_256 : XRT_QCAP_128;
    a->bragg_batch = big ? 256u : 128u;
    return big ? l256 : l128;
}

// device copy of every mesh's tables behind the staged region; fills KOptic.mesh
// ---- the packed tables of a mesh, kept between calls -------------------------------------------------------------------
// Everything upload_meshes derives from a mesh -- face records, plane forms, the point-source form and the direction grid,
// the bucket grid, the 16-bit tables and the fans, the Clough-Tocher records -- depends on the mesh, on the optic's frame and
// on the source point only: milliseconds of host work per call (3.5 ms for an 81 x 81 mesh), behind a stream synchronisation,
// for every batch of runs and every second pass.  The region's bytes are therefore kept (in pinned host memory, keyed by a
// 128-bit fingerprint of all inputs and switches); a later call copies them, asynchronously on its stream, to wherever its
// layout puts the region, with the header's pointers moved along.
struct MeshFp { uint64_t a, b; };
static void fp_mix(MeshFp& f, const void* data, size_t bytes)
{
    if (!data || !bytes) { f.a = (f.a ^ 0x51ull) * 0x9E3779B97F4A7C15ull; f.b += 0x7full; return; }
    const unsigned char* c = reinterpret_cast<const unsigned char*>(data);
    uint64_t a = f.a, b = f.b ^ (uint64_t)bytes;
    size_t i = 0;
    for (; i + 8 <= bytes; i += 8) {
        uint64_t x;
        memcpy(&x, c + i, 8);
        a = (a ^ x) * 0x9E3779B97F4A7C15ull; a ^= a >> 29;
        b = (b + x) * 0xC2B2AE3D27D4EB4Full; b ^= b >> 31;
    }
    if (i < bytes) {
        uint64_t x = 0;
        memcpy(&x, c + i, bytes - i);
        a = (a ^ x) * 0x9E3779B97F4A7C15ull; a ^= a >> 29;
        b = (b + x) * 0xC2B2AE3D27D4EB4Full; b ^= b >> 31;
    }
    f.a = a; f.b = b;
}
static MeshFp mesh_fingerprint(const xrt_scene_t* sc, int e)
{
    const xrt_mesh_t* m = sc->optics[e].mesh;
    const size_t P = (size_t)m->n_points, F = (size_t)m->n_faces, Cn = (size_t)m->n_coarse_faces, T = (size_t)m->n_simplices;
    MeshFp f = {0x243F6A8885A308D3ull, 0x13198A2E03707344ull};
    const int32_t head[8] = {m->n_points, m->n_faces, m->n_coarse_faces, m->interpolate, m->n_simplices, e, sc->source.kind, sc->source.spatial_dist};
    fp_mix(f, head, sizeof(head));
    uint32_t sw = 0;
    const char* names[] = {"XICSRT_NO_DIR_GRID", "XICSRT_NO_MESH_LDS", "XICSRT_NO_FACE_GRID", "XICSRT_NO_POINT_FORM", "XICSRT_NO_MESH_FANS"};
    for (int i = 0; i < 5; i++) if (env_on(names[i])) sw |= 1u << i;
    fp_mix(f, &sw, sizeof(sw));
    fp_mix(f, sc->optics[e].origin, 24); fp_mix(f, sc->optics[e].orientation, 72);
    fp_mix(f, sc->source.origin, 24); fp_mix(f, sc->source.size, 24);
    fp_mix(f, m->points, P * 24);
    fp_mix(f, m->p0, F * 24); fp_mix(f, m->p1, F * 24); fp_mix(f, m->p2, F * 24);
    fp_mix(f, m->edge1, F * 24); fp_mix(f, m->edge2, F * 24);
    fp_mix(f, m->faces_normal, F * 24); fp_mix(f, m->faces_area, F * 8);
    fp_mix(f, m->p_faces_idx, P * 32); fp_mix(f, m->p_faces_mask, P * 8);
    fp_mix(f, m->c_p0, Cn * 24); fp_mix(f, m->c_edge1, Cn * 24); fp_mix(f, m->c_edge2, Cn * 24);
    if (m->interpolate) {
        fp_mix(f, m->ct_simplices, T * 12); fp_mix(f, m->ct_neighbors, T * 12); fp_mix(f, m->ct_transform, T * 48);
        fp_mix(f, m->ct_points, P * 16); fp_mix(f, m->ct_values, P * 32); fp_mix(f, m->ct_grad, P * 64);
        fp_mix(f, m->ct_vertex_simplex, P * 4);
    }
    return f;
}
struct MeshCacheEntry {
    MeshFp fp;
    int dev;
    char* pinned;               // the region's bytes (hipHostMalloc)
    size_t bytes;
    uintptr_t base0;            // the device address the header's pointers were made for
    KMesh k;
    int32_t lds_bytes, star_lds_bytes, ct_lds_bytes, dir_bytes;
    uint64_t used;
};
static std::mutex g_mesh_mu;
static std::vector<MeshCacheEntry> g_mesh_cache;
static uint64_t g_mesh_tick = 0;
// the header's device pointers, moved by `delta` bytes
static void kmesh_relocate(KMesh& k, ptrdiff_t delta)
{
    auto mv = [&](auto& ptr) { if (ptr) ptr = reinterpret_cast<std::remove_reference_t<decltype(ptr)>>(reinterpret_cast<uintptr_t>(ptr) + (uintptr_t)delta); };
    mv(k.faces_normal); mv(k.first_rec); mv(k.plane_rec); mv(k.face_rec); mv(k.point_faces);
    mv(k.ct_srec); mv(k.ct_frec); mv(k.ct_vrec); mv(k.face_simplex); mv(k.ct_vertex_simplex);
    mv(k.points); mv(k.cells); mv(k.fg_start); mv(k.fg_faces); mv(k.fg_zcell);
    mv(k.pt_rec); mv(k.dg_cells); mv(k.plane2_rec); mv(k.lds_pf); mv(k.lds_fv); mv(k.lds_star);
}

static int upload_meshes(const xrt_scene_t* sc, char* ws, int n_runs, KScene* ks, hipStream_t stream)
{
    char* base = ws + ws_off_staged(sc, n_runs) + staged_bytes(sc, n_runs);
    int dev_now = 0;
    for (int e = 0; e < sc->n_optics; e++) {
        ks->opt[e].mesh = nullptr;
        ks->opt[e].mesh_lds_bytes = 0;
        ks->opt[e].mesh_dir_bytes = 0;
        ks->opt[e].mesh_ct_lds_bytes = 0;
        ks->opt[e].mesh_star_lds_bytes = 0;
        if (sc->optics[e].shape != XRT_SHAPE_MESH) continue;
        const xrt_mesh_t* m = sc->optics[e].mesh;
        const size_t region = mesh_bytes(m);
        HIP_TRY(hipGetDevice(&dev_now));
        const MeshFp fp = mesh_fingerprint(sc, e);
        const bool use_cache = !env_on("XICSRT_NO_MESH_CACHE");
        if (use_cache) {
            // the tables of an earlier call: the region's bytes from pinned memory, stream-ordered (an earlier call on `stream`
            // may still be reading this part of the workspace with another layout: the copy queues behind it)
            std::unique_lock<std::mutex> lock(g_mesh_mu);
            MeshCacheEntry* hit = nullptr;
            for (MeshCacheEntry& c : g_mesh_cache)
                if (c.fp.a == fp.a && c.fp.b == fp.b && c.dev == dev_now && c.bytes == region) hit = &c;
            if (hit) {
                hit->used = ++g_mesh_tick;
                KMesh k = hit->k;
                kmesh_relocate(k, (ptrdiff_t)((uintptr_t)base - hit->base0));
                const size_t hdr = al256(sizeof(KMesh));
                const char* pinned = hit->pinned;
                const int32_t l0 = hit->lds_bytes, l1 = hit->star_lds_bytes, l2 = hit->ct_lds_bytes, l3 = hit->dir_bytes;
                lock.unlock();
                HIP_TRY(hipMemcpyAsync(base + hdr, pinned + hdr, region - hdr, hipMemcpyHostToDevice, stream));
                HIP_TRY(hipMemcpyAsync(base, &k, sizeof(KMesh), hipMemcpyHostToDevice, stream));     // (pageable: staged by the runtime before it returns)
                ks->opt[e].mesh = reinterpret_cast<const KMesh*>(base);
                ks->opt[e].mesh_lds_bytes = l0; ks->opt[e].mesh_star_lds_bytes = l1; ks->opt[e].mesh_ct_lds_bytes = l2; ks->opt[e].mesh_dir_bytes = l3;
                base += region;
                continue;
            }
        }
        KMesh k;
        memset(&k, 0, sizeof(k));
        k.n_points = m->n_points; k.n_faces = m->n_faces; k.n_coarse_faces = m->n_coarse_faces;
        k.interpolate = m->interpolate; k.n_simplices = m->n_simplices;
        char* p = base + al256(sizeof(KMesh));
        // the region is put together on the host (device addresses in the header, the bytes at the same offsets in `blob`)
        std::vector<char> blob(region, 0);
        hipError_t put_err = hipSuccess;
        auto put = [&](const void* src, size_t bytes) -> uintptr_t {
            char* dst = p;
            p += al256(bytes > 0 ? bytes : 8);
            if ((size_t)(p - base) > region) { put_err = hipErrorOutOfMemory; return (uintptr_t)dst; }
            if (src && bytes) memcpy(&blob[(size_t)(dst - base)], src, bytes);
            return (uintptr_t)dst;
        };
        const size_t P = (size_t)m->n_points, F = (size_t)m->n_faces, Cn = (size_t)m->n_coarse_faces, T = (size_t)m->n_simplices;
        k.faces_normal = (gdp)put(m->faces_normal, F * 24);
        {   // faces of the first (exhaustive) Moller-Trumbore pass, one 80-byte record each (+1: the walk reads one ahead)
            const size_t n_first = Cn > 0 ? Cn : F;
            const double *q0 = Cn > 0 ? m->c_p0 : m->p0, *q1 = Cn > 0 ? m->c_edge1 : m->edge1, *q2 = Cn > 0 ? m->c_edge2 : m->edge2;
            std::vector<double> rec((n_first + 1) * 10, 0.0);
            for (size_t i = 0; i < n_first; i++)
                for (int a = 0; a < 3; a++) { rec[10 * i + a] = q0[3 * i + a]; rec[10 * i + 3 + a] = q1[3 * i + a]; rec[10 * i + 6 + a] = q2[3 * i + a]; }
            k.n_first = (int32_t)n_first;
            k.first_rec = (const double*)put(rec.data(), rec.size() * 8);
            // plane form of the same faces for the classification pass: P - p0 = u e1 + v e2 (+ w n), n = e1 x e2,
            // u = (P - p0).U with U = (e2 x n) / (e1.(e2 x n)), v = (P - p0).V with V = (n x e1) / (e2.(n x e1))
            std::vector<double> pr((n_first + 1) * 16, 0.0);
            auto plane_form = [](const double* p0, const double* e1, const double* e2, double* o) {
                o[12] = HUGE_VAL;                                  // degenerate until shown otherwise
                const double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
                const double a[3] = {e2[1] * n[2] - e2[2] * n[1], e2[2] * n[0] - e2[0] * n[2], e2[0] * n[1] - e2[1] * n[0]};   // e2 x n
                const double b[3] = {n[1] * e1[2] - n[2] * e1[1], n[2] * e1[0] - n[0] * e1[2], n[0] * e1[1] - n[1] * e1[0]};   // n x e1
                const double da = e1[0] * a[0] + e1[1] * a[1] + e1[2] * a[2], db = e2[0] * b[0] + e2[1] * b[1] + e2[2] * b[2];
                const double nn = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
                bool ok = std::isfinite(nn) && nn > 0.0 && std::isfinite(da) && std::isfinite(db) && da != 0.0 && db != 0.0;
                double U[3], V[3];
                for (int c = 0; c < 3 && ok; c++) { U[c] = a[c] / da; V[c] = b[c] / db; ok = std::isfinite(U[c]) && std::isfinite(V[c]) && std::isfinite(p0[c]); }
                if (!ok) return;
                o[0] = n[0]; o[1] = n[1]; o[2] = n[2]; o[3] = n[0] * p0[0] + n[1] * p0[1] + n[2] * p0[2];
                o[4] = U[0]; o[5] = U[1]; o[6] = U[2]; o[7] = -(U[0] * p0[0] + U[1] * p0[1] + U[2] * p0[2]);
                o[8] = V[0]; o[9] = V[1]; o[10] = V[2]; o[11] = -(V[0] * p0[0] + V[1] * p0[1] + V[2] * p0[2]);
                o[12] = nn;
                if (!std::isfinite(o[3]) || !std::isfinite(o[7]) || !std::isfinite(o[11])) { for (int c = 0; c < 12; c++) o[c] = 0.0; o[12] = HUGE_VAL; }
            };
            if (Cn > 0) {
                // every face for the second pass; [13] = 5e-10 / area (ten times the excess of the area sum the reference tolerates)
                std::vector<double> p2(F * 16, 0.0);
                for (size_t i = 0; i < F; i++) {
                    plane_form(m->p0 + 3 * i, m->edge1 + 3 * i, m->edge2 + 3 * i, &p2[16 * i]);
                    const double area = m->faces_area[i];
                    p2[16 * i + 13] = (std::isfinite(area) && area > 0.0) ? 5e-10 / area : HUGE_VAL;
                    if (!(p2[16 * i + 13] < 0.25)) p2[16 * i + 12] = HUGE_VAL;       // (tiny faces: always the exact test)
                }
                k.plane2_rec = (const XRT_G1 d4v*)put(p2.data(), p2.size() * 8);
            } else k.plane2_rec = nullptr;
            for (size_t i = 0; i <= n_first; i++) {
                double* o = &pr[16 * i];
                o[12] = HUGE_VAL;                                  // degenerate until shown otherwise (the spare record too)
                if (i == n_first) break;
                const double *p0 = q0 + 3 * i, *e1 = q1 + 3 * i, *e2 = q2 + 3 * i;
                const double n[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
                const double a[3] = {e2[1] * n[2] - e2[2] * n[1], e2[2] * n[0] - e2[0] * n[2], e2[0] * n[1] - e2[1] * n[0]};   // e2 x n
                const double b[3] = {n[1] * e1[2] - n[2] * e1[1], n[2] * e1[0] - n[0] * e1[2], n[0] * e1[1] - n[1] * e1[0]};   // n x e1
                const double da = e1[0] * a[0] + e1[1] * a[1] + e1[2] * a[2], db = e2[0] * b[0] + e2[1] * b[1] + e2[2] * b[2];
                const double nn = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
                bool ok = std::isfinite(nn) && nn > 0.0 && std::isfinite(da) && std::isfinite(db) && da != 0.0 && db != 0.0;
                double U[3], V[3];
                for (int c = 0; c < 3 && ok; c++) { U[c] = a[c] / da; V[c] = b[c] / db; ok = std::isfinite(U[c]) && std::isfinite(V[c]) && std::isfinite(p0[c]); }
                if (!ok) continue;
                o[0] = n[0]; o[1] = n[1]; o[2] = n[2]; o[3] = n[0] * p0[0] + n[1] * p0[1] + n[2] * p0[2];
                o[4] = U[0]; o[5] = U[1]; o[6] = U[2]; o[7] = -(U[0] * p0[0] + U[1] * p0[1] + U[2] * p0[2]);
                o[8] = V[0]; o[9] = V[1]; o[10] = V[2]; o[11] = -(V[0] * p0[0] + V[1] * p0[1] + V[2] * p0[2]);
                o[12] = nn;
                if (!std::isfinite(o[3]) || !std::isfinite(o[7]) || !std::isfinite(o[11])) { for (int c = 0; c < 12; c++) o[c] = 0.0; o[12] = HUGE_VAL; }
            }
            k.plane_rec = (const double*)put(pr.data(), pr.size() * 8);
            // ---- point-source form (see KMesh.pt_rec): the mesh is the first optic behind a source without extent -------
            k.pt_rec = nullptr;
            {
                const xrt_source_t& src = sc->source;
                const bool point = e == 0 && (src.kind == XRT_SRC_GENERIC || src.kind == XRT_SRC_DIRECTED) && src.spatial_dist == XRT_SPATIAL_UNIFORM &&
                                   src.size[0] == 0.0 && src.size[1] == 0.0 && src.size[2] == 0.0 && !env_on("XICSRT_NO_POINT_FORM");
                if (point) {
                    const xrt_optic_t& o = sc->optics[e];
                    double Ol[3];           // the source point in the optic's frame (ray_to_local, optics/_TraceObject.py:146-148)
                    for (int r = 0; r < 3; r++) {
                        Ol[r] = 0.0;
                        for (int c = 0; c < 3; c++) Ol[r] += o.orientation[3 * r + c] * (src.origin[c] - o.origin[c]);
                    }
                    std::vector<double> pt((n_first + 1) * 12, 0.0);
                    for (size_t i = 0; i <= n_first; i++) {
                        double* w = &pt[12 * i];
                        w[9] = HUGE_VAL;
                        if (i == n_first) break;
                        const double *p0 = q0 + 3 * i, *e1 = q1 + 3 * i, *e2 = q2 + 3 * i;
                        const double a[3] = {p0[0] - Ol[0], p0[1] - Ol[1], p0[2] - Ol[2]};
                        const double b1[3] = {a[0] + e1[0], a[1] + e1[1], a[2] + e1[2]}, b2[3] = {a[0] + e2[0], a[1] + e2[1], a[2] + e2[2]};
                        const double m[3] = {e2[1] * e1[2] - e2[2] * e1[1], e2[2] * e1[0] - e2[0] * e1[2], e2[0] * e1[1] - e2[1] * e1[0]};
                        const double E02[3] = {a[1] * b2[2] - a[2] * b2[1], a[2] * b2[0] - a[0] * b2[2], a[0] * b2[1] - a[1] * b2[0]};
                        const double E10[3] = {b1[1] * a[2] - b1[2] * a[1], b1[2] * a[0] - b1[0] * a[2], b1[0] * a[1] - b1[1] * a[0]};
                        const double mm = m[0] * m[0] + m[1] * m[1] + m[2] * m[2];
                        bool fin = std::isfinite(mm) && mm > 0.0;
                        for (int c = 0; c < 3 && fin; c++) fin = std::isfinite(E02[c]) && std::isfinite(E10[c]);
                        if (!fin) continue;
                        for (int c = 0; c < 3; c++) { w[c] = m[c]; w[3 + c] = E02[c]; w[6 + c] = E10[c]; }
                        w[9] = mm;
                    }
                    k.pt_rec = (const double*)put(pt.data(), pt.size() * 8);
                    // ---- direction grid (see KMesh.dg_n) ----
                    k.dg_n = 0;
                    k.dg_big = n_first >= 0xfe ? 1 : 0;
                    if (n_first >= 8 && n_first < 0xfffe && !env_on("XICSRT_NO_DIR_GRID")) {
                        double ez[3] = {0, 0, 0};
                        bool ok = true;
                        for (size_t i = 0; i < n_first; i++)
                            for (int c = 0; c < 3; c++) ez[c] += (q0[3 * i + c] + (q1[3 * i + c] + q2[3 * i + c]) / 3.0) - Ol[c];
                        const double zl = sqrt(ez[0] * ez[0] + ez[1] * ez[1] + ez[2] * ez[2]);
                        ok = std::isfinite(zl) && zl > 0.0;
                        double ex[3] = {0, 0, 0}, ey[3] = {0, 0, 0};
                        if (ok) {
                            for (int c = 0; c < 3; c++) ez[c] /= zl;
                            const int least = fabs(ez[0]) <= fabs(ez[1]) ? (fabs(ez[0]) <= fabs(ez[2]) ? 0 : 2) : (fabs(ez[1]) <= fabs(ez[2]) ? 1 : 2);
                            double t[3] = {0, 0, 0};
                            t[least] = 1.0;
                            ex[0] = t[1] * ez[2] - t[2] * ez[1]; ex[1] = t[2] * ez[0] - t[0] * ez[2]; ex[2] = t[0] * ez[1] - t[1] * ez[0];
                            const double xl = sqrt(ex[0] * ex[0] + ex[1] * ex[1] + ex[2] * ex[2]);
                            for (int c = 0; c < 3; c++) ex[c] /= xl;
                            ey[0] = ez[1] * ex[2] - ez[2] * ex[1]; ey[1] = ez[2] * ex[0] - ez[0] * ex[2]; ey[2] = ez[0] * ex[1] - ez[1] * ex[0];
                        }
                        // images of the vertices; every vertex well in front of O along ez, no face seen edge-on
                        std::vector<double> img(n_first * 6);
                        double lo2[2] = {HUGE_VAL, HUGE_VAL}, hi2[2] = {-HUGE_VAL, -HUGE_VAL};
                        for (size_t i = 0; i < n_first && ok; i++) {
                            for (int vtx = 0; vtx < 3 && ok; vtx++) {
                                double w[3];
                                for (int c = 0; c < 3; c++) w[c] = q0[3 * i + c] + (vtx == 1 ? q1[3 * i + c] : vtx == 2 ? q2[3 * i + c] : 0.0) - Ol[c];
                                const double wl = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
                                const double wz = w[0] * ez[0] + w[1] * ez[1] + w[2] * ez[2];
                                ok = std::isfinite(wl) && wl > 0.0 && wz > 0.2 * wl;
                                if (!ok) break;
                                const double px = (w[0] * ex[0] + w[1] * ex[1] + w[2] * ex[2]) / wz, py = (w[0] * ey[0] + w[1] * ey[1] + w[2] * ey[2]) / wz;
                                img[6 * i + 2 * vtx] = px; img[6 * i + 2 * vtx + 1] = py;
                                if (px < lo2[0]) lo2[0] = px;
                                if (px > hi2[0]) hi2[0] = px;
                                if (py < lo2[1]) lo2[1] = py;
                                if (py > hi2[1]) hi2[1] = py;
                            }
                            if (!ok) break;
                            const double* g = &img[6 * i];
                            const double ax = g[2] - g[0], ay = g[3] - g[1], bx = g[4] - g[0], by = g[5] - g[1];
                            const double area2 = fabs(ax * by - ay * bx), len2 = fmax(ax * ax + ay * ay, bx * bx + by * by);
                            ok = area2 > 1e-3 * len2;                   // (an image that is a sliver: a face seen nearly edge-on)
                        }
                        const double extx = hi2[0] - lo2[0], exty = hi2[1] - lo2[1];
                        ok = ok && std::isfinite(extx) && std::isfinite(exty) && extx > 0.0 && exty > 0.0;
                        if (ok) {
                            int G = 4 * (int)ceil(sqrt((double)n_first));
                            if (G < 8) G = 8;
                            if (G > (k.dg_big ? 128 : 32)) G = k.dg_big ? 128 : 32;
                            const int CW = k.dg_big ? 4 : 2, IDB = k.dg_big ? 16 : 8;          // words per cell, bits per face number
                            const uint32_t IDM = k.dg_big ? 0xffffu : 0xffu;
                            const double mx = 1e-6 * extx, my = 1e-6 * exty;       // (a miss by this much in the image is a miss by >= 1e-6 in barycentric units)
                            const double x0 = lo2[0] - 2 * mx, y0 = lo2[1] - 2 * my, hx = (extx + 4 * mx) / G, hy = (exty + 4 * my) / G;
                            std::vector<uint32_t> cellsd((size_t)G * G * CW, 0xffffffffu);
                            std::vector<int> cnt((size_t)G * G, 0);
                            for (size_t i = 0; i < n_first; i++) {              // ascending face index within a cell
                                const double* g = &img[6 * i];
                                const double fx0 = fmin(g[0], fmin(g[2], g[4])) - mx, fx1 = fmax(g[0], fmax(g[2], g[4])) + mx;
                                const double fy0 = fmin(g[1], fmin(g[3], g[5])) - my, fy1 = fmax(g[1], fmax(g[3], g[5])) + my;
                                int cx0 = (int)floor((fx0 - x0) / hx), cx1 = (int)floor((fx1 - x0) / hx);
                                int cy0 = (int)floor((fy0 - y0) / hy), cy1 = (int)floor((fy1 - y0) / hy);
                                if (cx0 < 0) cx0 = 0;
                                if (cy0 < 0) cy0 = 0;
                                if (cx1 > G - 1) cx1 = G - 1;
                                if (cy1 > G - 1) cy1 = G - 1;
                                for (int cy = cy0; cy <= cy1; cy++)
                                    for (int cx = cx0; cx <= cx1; cx++) {
                                        // the cell (with the margin) against the three edges of the image: wholly beyond one -> no overlap
                                        const double bx0 = x0 + cx * hx - mx, bx1 = x0 + (cx + 1) * hx + mx, by0 = y0 + cy * hy - my, by1 = y0 + (cy + 1) * hy + my;
                                        bool apart = false;
                                        for (int ed = 0; ed < 3 && !apart; ed++) {
                                            const double *pa = g + 2 * ed, *pb = g + 2 * ((ed + 1) % 3), *pc = g + 2 * ((ed + 2) % 3);
                                            const double nxe = pb[1] - pa[1], nye = pa[0] - pb[0];                    // a normal of edge a-b
                                            const double side = nxe * (pc[0] - pa[0]) + nye * (pc[1] - pa[1]);       // where the third vertex lies
                                            const double sg = side >= 0.0 ? 1.0 : -1.0;
                                            const double tolr = 1e-9 * (fabs(nxe) * (fabs(bx0) + fabs(bx1) + fabs(pa[0])) + fabs(nye) * (fabs(by0) + fabs(by1) + fabs(pa[1])));
                                            const double corners[4][2] = {{bx0, by0}, {bx1, by0}, {bx0, by1}, {bx1, by1}};
                                            bool all_out = true;
                                            for (int q = 0; q < 4; q++)
                                                if (sg * (nxe * (corners[q][0] - pa[0]) + nye * (corners[q][1] - pa[1])) >= -tolr) all_out = false;
                                            apart = all_out;
                                        }
                                        if (apart) continue;
                                        const size_t c = (size_t)cy * G + cx;
                                        if (cnt[c] < 8) {
                                            const int per = 32 / IDB;
                                            uint32_t& wd = cellsd[CW * c + (cnt[c] / per)];
                                            const int sh = IDB * (cnt[c] % per);
                                            wd = (wd & ~(IDM << sh)) | ((uint32_t)i << sh);
                                        }
                                        cnt[c]++;
                                    }
                            }
                            for (size_t c = 0; c < cnt.size(); c++) if (cnt[c] > 8) cellsd[CW * c] = 0xfffffffeu;
                            k.dg_n = G; k.dg_x0 = x0; k.dg_y0 = y0; k.dg_ihx = 1.0 / hx; k.dg_ihy = 1.0 / hy;
                            for (int c = 0; c < 3; c++) { k.dg_ex[c] = ex[c]; k.dg_ey[c] = ey[c]; k.dg_ez[c] = ez[c]; }
                            k.dg_cells = (const uint32_t*)put(cellsd.data(), cellsd.size() * 4);
                        }
                    }
                }
            }
            // ---- x-y grid over these faces (see KMesh.fg_n) ----------------------------------------------------
            k.fg_n = 0;
            {
                // (a ray's faces through the grid are gathered per lane, 128 bytes each; the plain walk reads every face once per
                //  wave through scalar loads: measured on the 32 coarse faces of cfg5, the walk is 1.75 x faster)
                bool ok = n_first > 128 && !env_on("XICSRT_NO_FACE_GRID");
                double lo[3] = {HUGE_VAL, HUGE_VAL, HUGE_VAL}, hi[3] = {-HUGE_VAL, -HUGE_VAL, -HUGE_VAL}, nb[3] = {0, 0, 0};
                std::vector<double> fb(n_first * 6), fn(n_first * 3);
                for (size_t i = 0; i < n_first && ok; i++) {
                    const double *p0 = q0 + 3 * i, *e1 = q1 + 3 * i, *e2 = q2 + 3 * i;
                    if (pr[16 * i + 12] == HUGE_VAL) { ok = false; break; }         // a degenerate face: every ray takes the whole walk
                    for (int c = 0; c < 3; c++) {
                        const double a = p0[c], b1 = p0[c] + e1[c], b2 = p0[c] + e2[c];
                        double mn = a < b1 ? a : b1, mx = a > b1 ? a : b1;
                        if (b2 < mn) mn = b2;
                        if (b2 > mx) mx = b2;
                        fb[6 * i + c] = mn; fb[6 * i + 3 + c] = mx;
                        if (mn < lo[c]) lo[c] = mn;
                        if (mx > hi[c]) hi[c] = mx;
                    }
                    const double nl = sqrt(pr[16 * i + 12]);
                    for (int c = 0; c < 3; c++) { fn[3 * i + c] = pr[16 * i + c] / nl; }
                }
                if (ok) {
                    // the cone of the face normals: orientation-free (a face and its flipped twin count alike)
                    for (size_t i = 0; i < n_first; i++) {
                        const double sgn = (fn[3 * i] * fn[0] + fn[3 * i + 1] * fn[1] + fn[3 * i + 2] * fn[2]) < 0.0 ? -1.0 : 1.0;
                        for (int c = 0; c < 3; c++) nb[c] += sgn * fn[3 * i + c];
                    }
                    const double nl = sqrt(nb[0] * nb[0] + nb[1] * nb[1] + nb[2] * nb[2]);
                    ok = std::isfinite(nl) && nl > 0.0;
                    double cmin = 1.0;
                    for (size_t i = 0; i < n_first && ok; i++) {
                        const double c = fabs(fn[3 * i] * nb[0] + fn[3 * i + 1] * nb[1] + fn[3 * i + 2] * nb[2]) / nl;
                        if (c < cmin) cmin = c;
                    }
                    const double ext = (hi[0] - lo[0]) > (hi[1] - lo[1]) ? (hi[0] - lo[0]) : (hi[1] - lo[1]);
                    ok = ok && cmin > 0.5 && std::isfinite(ext) && ext > 0.0 && (hi[0] - lo[0]) > 0.0 && (hi[1] - lo[1]) > 0.0;
                    if (ok) {
                        // |d . nbar| / |d| >= sin(alpha) + 1e-3 keeps |d . n| / |d| >= ~1e-3 for every face normal n (alpha: the cone's half angle)
                        k.fg_sin = sqrt(1.0 - cmin * cmin) + 1e-3;
                        for (int c = 0; c < 3; c++) k.fg_nbar[c] = nb[c] / nl;
                        const double margin = 1e-5 * ext;               // (classification tolerance 1e-7 in barycentric units, rounding: far inside)
                        int g = (int)ceil(sqrt((double)n_first / 2.0));
                        if (g < 1) g = 1;
                        if (g > 64) g = 64;
                        k.fg_n = g; k.fg_margin = margin;
                        k.fg_x0 = lo[0] - margin; k.fg_y0 = lo[1] - margin;
                        k.fg_ihx = (double)g / ((hi[0] - lo[0]) + 2.0 * margin); k.fg_ihy = (double)g / ((hi[1] - lo[1]) + 2.0 * margin);
                        k.fg_zlo = lo[2] - margin; k.fg_zhi = hi[2] + margin;
                        std::vector<std::vector<int32_t>> cells((size_t)g * g);
                        auto cell_of = [&](double v, double v0, double ih) { int c = (int)floor((v - v0) * ih); return c < 0 ? 0 : (c > g - 1 ? g - 1 : c); };
                        for (size_t i = 0; i < n_first; i++) {
                            const int cx0 = cell_of(fb[6 * i] - margin, k.fg_x0, k.fg_ihx), cx1 = cell_of(fb[6 * i + 3] + margin, k.fg_x0, k.fg_ihx);
                            const int cy0 = cell_of(fb[6 * i + 1] - margin, k.fg_y0, k.fg_ihy), cy1 = cell_of(fb[6 * i + 4] + margin, k.fg_y0, k.fg_ihy);
                            for (int cy = cy0; cy <= cy1; cy++) for (int cx = cx0; cx <= cx1; cx++) cells[(size_t)cy * g + cx].push_back((int32_t)i);
                        }
                        std::vector<int32_t> start((size_t)g * g + 1), list;
                        std::vector<double> zc((size_t)g * g * 2);
                        for (size_t c = 0; c < (size_t)g * g; c++) {
                            start[c] = (int32_t)list.size(); list.insert(list.end(), cells[c].begin(), cells[c].end());
                            double zl = HUGE_VAL, zh = -HUGE_VAL;
                            for (int32_t f : cells[c]) { if (fb[6 * f + 2] < zl) zl = fb[6 * f + 2]; if (fb[6 * f + 5] > zh) zh = fb[6 * f + 5]; }
                            zc[2 * c] = zl - margin; zc[2 * c + 1] = zh + margin;
                        }
                        start[(size_t)g * g] = (int32_t)list.size();
                        if (list.size() > n_first * 16) k.fg_n = 0;     // (more than the space set aside: a pathological mesh)
                        else {
                            k.fg_start = (gip)put(start.data(), start.size() * 4);
                            k.fg_faces = (gip)put(list.empty() ? nullptr : list.data(), list.size() * 4);
                            k.fg_zcell = (gdp)put(zc.data(), zc.size() * 8);
                        }
                    }
                }
            }
        }
        {   // per face what the second pass reads, per point its <= 8 faces
            std::vector<KFaceRec> fr(F);
            for (size_t i = 0; i < F; i++) {
                memset(&fr[i], 0, sizeof(KFaceRec));
                for (int a = 0; a < 3; a++) {
                    fr[i].p0[a] = m->p0[3 * i + a]; fr[i].p1[a] = m->p1[3 * i + a]; fr[i].p2[a] = m->p2[3 * i + a];
                    fr[i].n[a] = m->faces_normal[3 * i + a];
                }
                fr[i].area = m->faces_area[i];
            }
            k.face_rec = (const XRT_G1 d4v*)put(fr.data(), F * sizeof(KFaceRec));
            std::vector<int32_t> pf(P * 8);
            for (size_t i = 0; i < P; i++)
                for (int j = 0; j < 8; j++)
                    pf[8 * i + j] = m->p_faces_mask[(size_t)j * P + i] ? m->p_faces_idx[(size_t)j * P + i] : -1;
            k.point_faces = (gip)put(pf.data(), P * 32);
        }
        k.points = (gdp)put(m->points, P * 24);
        if (m->interpolate) {
            k.ct_vertex_simplex = (gip)put(m->ct_vertex_simplex, P * 4);
            {
                std::vector<double> fr(T * 8, 0.0), vr(P * 16, 0.0);
                for (size_t si = 0; si < T; si++) {
                    for (int c = 0; c < 6; c++) fr[8 * si + c] = m->ct_transform[6 * si + c];
                    const int32_t nb4[4] = {m->ct_neighbors[3 * si], m->ct_neighbors[3 * si + 1], m->ct_neighbors[3 * si + 2], 0};
                    memcpy(&fr[8 * si + 6], nb4, 16);
                }
                for (size_t pi = 0; pi < P; pi++)
                    for (int wq = 0; wq < 4; wq++) {
                        vr[16 * pi + 3 * wq] = m->ct_values[(size_t)wq * P + pi];
                        vr[16 * pi + 3 * wq + 1] = m->ct_grad[((size_t)wq * P + pi) * 2];
                        vr[16 * pi + 3 * wq + 2] = m->ct_grad[((size_t)wq * P + pi) * 2 + 1];
                    }
                k.ct_frec = (gdp)put(fr.data(), fr.size() * 8);
                k.ct_vrec = (gdp)put(vr.data(), vr.size() * 8);
            }
            {   // where the walk to a hit point's simplex starts: at the simplex that has the hit face's vertices (the faces
                // are the reference's Delaunay triangles of the same points), if there is one
                struct Key3 { uint64_t a, b, c; bool operator==(const Key3& o) const { return a == o.a && b == o.b && c == o.c; } };
                struct Key3Hash { size_t operator()(const Key3& q) const { return (size_t)(q.a * 0x9e3779b97f4a7c15ull ^ (q.b + 0x7f4a7c15ull) * 0xbf58476d1ce4e5b9ull ^ (q.c * 0x94d049bb133111ebull)); } };
                std::unordered_map<Key3, int32_t, Key3Hash> pt_of, simplex_of;
                pt_of.reserve(P * 2);
                auto key_of = [](const double* v) { Key3 q; memcpy(&q.a, v, 8); memcpy(&q.b, v + 1, 8); memcpy(&q.c, v + 2, 8); return q; };
                for (size_t i = 0; i < P; i++) pt_of.emplace(key_of(m->points + 3 * i), (int32_t)i);
                auto sorted3 = [](uint64_t a, uint64_t b, uint64_t c) { Key3 q; if (a > b) std::swap(a, b); if (b > c) std::swap(b, c); if (a > b) std::swap(a, b); q.a = a; q.b = b; q.c = c; return q; };
                simplex_of.reserve(T * 2);
                for (size_t si = 0; si < T; si++)
                    simplex_of.emplace(sorted3((uint64_t)m->ct_simplices[3 * si], (uint64_t)m->ct_simplices[3 * si + 1], (uint64_t)m->ct_simplices[3 * si + 2]), (int32_t)si);
                std::vector<int32_t> fs(F, -1);
                std::vector<std::vector<int32_t>> around;          // the simplices at every point (made when first needed)
                for (size_t i = 0; i < F; i++) {
                    const double* vs[3] = {m->p0 + 3 * i, m->p1 + 3 * i, m->p2 + 3 * i};
                    uint64_t pi[3];
                    bool all = true;
                    for (int j = 0; j < 3 && all; j++) {
                        const auto it = pt_of.find(key_of(vs[j]));
                        if (it == pt_of.end()) all = false; else pi[j] = (uint64_t)it->second;
                    }
                    if (!all) continue;
                    const auto it = simplex_of.find(sorted3(pi[0], pi[1], pi[2]));
                    if (it != simplex_of.end()) { fs[i] = it->second; continue; }
                    // Another triangulation of the same points (half of the quads of a regular grid get the other diagonal in x-y):
                    // of the simplices at the face's vertices the one that holds the face's centroid -- a ray that ended on the
                    // face is then in it or in the one across the diagonal, one step of the walk away.
                    if (around.empty()) {
                        around.resize(P);
                        for (size_t si = 0; si < T; si++) for (int j = 0; j < 3; j++) around[(size_t)m->ct_simplices[3 * si + j]].push_back((int32_t)si);
                    }
                    const double gx = (vs[0][0] + vs[1][0] + vs[2][0]) / 3.0, gy = (vs[0][1] + vs[1][1] + vs[2][1]) / 3.0;
                    int32_t best_s = m->ct_vertex_simplex[pi[0]];
                    double best_c = -HUGE_VAL;
                    for (int j = 0; j < 3; j++)
                        for (int32_t si : around[pi[j]]) {
                            const double* Tm = m->ct_transform + 6 * (size_t)si;
                            const double c0 = Tm[0] * (gx - Tm[4]) + Tm[1] * (gy - Tm[5]), c1 = Tm[2] * (gx - Tm[4]) + Tm[3] * (gy - Tm[5]);
                            const double cm = fmin(fmin(c0, c1), 1.0 - c0 - c1);
                            if (cm > best_c) { best_c = cm; best_s = si; }
                        }
                    fs[i] = best_s;
                }
                k.face_simplex = (gip)put(fs.data(), F * 4);
            }
            // per simplex what the evaluation needs of it alone (SciPy _clough_tocher_2d_single: the edge vectors and, per
            // neighbour, the weight g from the neighbour's centroid in this simplex' barycentric coordinates), with the
            // operations ct_shared used to make per ray (this file is compiled without contraction, host and device)
            std::vector<double> sr(T * 16, 0.0);
            for (size_t si = 0; si < T; si++) {
                const int32_t* v = m->ct_simplices + 3 * si;
                const double* pts = m->ct_points;
                double* o = &sr[16 * si];
                int32_t iv[4] = {v[0], v[1], v[2], 0};
                memcpy(o, iv, 16);
                o[2] = pts[2 * v[1]] - pts[2 * v[0]]; o[3] = pts[2 * v[1] + 1] - pts[2 * v[0] + 1];
                o[4] = pts[2 * v[2]] - pts[2 * v[1]]; o[5] = pts[2 * v[2] + 1] - pts[2 * v[1] + 1];
                o[6] = pts[2 * v[0]] - pts[2 * v[2]]; o[7] = pts[2 * v[0] + 1] - pts[2 * v[2] + 1];
                const double* Tm = m->ct_transform + 6 * si;
                for (int kk = 0; kk < 3; kk++) {
                    const int itri = m->ct_neighbors[3 * si + kk];
                    if (itri == -1) { o[8 + kk] = -1. / 2; continue; }
                    const int32_t* w = m->ct_simplices + 3 * (size_t)itri;
                    volatile double y0 = (pts[2 * w[0]] + pts[2 * w[1]] + pts[2 * w[2]]) / 3;
                    volatile double y1 = (pts[2 * w[0] + 1] + pts[2 * w[1] + 1] + pts[2 * w[2] + 1]) / 3;
                    double c[3];
                    c[2] = 1.0;
                    c[0] = 0.0; c[0] += Tm[0] * (y0 - Tm[4]); c[0] += Tm[1] * (y1 - Tm[5]); c[2] -= c[0];
                    c[1] = 0.0; c[1] += Tm[2] * (y0 - Tm[4]); c[1] += Tm[3] * (y1 - Tm[5]); c[2] -= c[1];
                    if (kk == 0)      o[8 + kk] = (2 * c[2] + c[1] - 1) / (2 - 3 * c[2] - 3 * c[1]);
                    else if (kk == 1) o[8 + kk] = (2 * c[0] + c[2] - 1) / (2 - 3 * c[0] - 3 * c[2]);
                    else              o[8 + kk] = (2 * c[1] + c[0] - 1) / (2 - 3 * c[1] - 3 * c[0]);
                }
            }
            k.ct_srec = (const XRT_G1 d4v*)put(sr.data(), sr.size() * 8);
        }
        if (m->n_coarse_faces > 0 && P > 0) {
            // x-y bucket grid for the nearest-point search, about one point per bucket
            double lo[2] = {m->points[0], m->points[1]}, hi[2] = {m->points[0], m->points[1]};
            for (size_t i = 0; i < P; i++)
                for (int a = 0; a < 2; a++) {
                    const double v = m->points[3 * i + a];
                    if (v < lo[a]) lo[a] = v;
                    if (v > hi[a]) hi[a] = v;
                }
            int g = (int)floor(sqrt((double)P));
            if (g < 1) g = 1;
            int nx = g, ny = g;
            while ((size_t)nx * ny > P) ny--;
            double ext[2] = {hi[0] - lo[0], hi[1] - lo[1]};
            if (!(ext[0] > 0.0)) { ext[0] = 1.0; nx = 1; }
            if (!(ext[1] > 0.0)) { ext[1] = 1.0; ny = 1; }
            bool finite = std::isfinite(ext[0]) && std::isfinite(ext[1]);
            if (finite) {
                k.grid_nx = nx; k.grid_ny = ny;
                k.grid_x0 = lo[0]; k.grid_y0 = lo[1];
                k.grid_hx = ext[0] / nx; k.grid_hy = ext[1] / ny;
                k.grid_ihx = 1.0 / k.grid_hx; k.grid_ihy = 1.0 / k.grid_hy;
                k.grid_tiny = 1e-9 * (k.grid_hx > k.grid_hy ? k.grid_hx : k.grid_hy);
                const size_t NC = (size_t)nx * ny;
                std::vector<KCellRec> cells(NC);
                for (size_t c = 0; c < NC; c++) { cells[c].x = cells[c].y = cells[c].z = 0.0; cells[c].idx = -1; cells[c].next = -1; }
                std::vector<int32_t> tail(NC, -1);                // last record of every bucket's chain
                std::vector<int32_t> pslot(P, -1);                // where a point's record went
                for (size_t i = 0; i < P; i++) {                  // ascending point index along a chain
                    double fx = floor((m->points[3 * i] - k.grid_x0) * k.grid_ihx), fy = floor((m->points[3 * i + 1] - k.grid_y0) * k.grid_ihy);
                    if (!(fx >= 0.0)) fx = 0.0;
                    if (!(fy >= 0.0)) fy = 0.0;
                    const int cx = fx > (double)(nx - 1) ? nx - 1 : (int)fx, cy = fy > (double)(ny - 1) ? ny - 1 : (int)fy;
                    const size_t c = (size_t)cy * nx + cx;
                    KCellRec r;
                    r.x = m->points[3 * i]; r.y = m->points[3 * i + 1]; r.z = m->points[3 * i + 2]; r.idx = (int32_t)i; r.next = -1;
                    if (tail[c] < 0) { cells[c] = r; tail[c] = (int32_t)c; }
                    else { cells[(size_t)tail[c]].next = (int32_t)cells.size(); tail[c] = (int32_t)cells.size(); cells.push_back(r); }
                    pslot[i] = tail[c];
                }
                k.cells = (const XRT_G1 d4v*)put(cells.data(), cells.size() * sizeof(KCellRec));
                // ---- the LDS form (see KMesh.lds_pf): every vertex of every face must be one of the points ----------
                k.n_cells = (int32_t)cells.size();
                const size_t n_first = (size_t)k.n_first;
                const size_t lds_need = cells.size() * sizeof(KCellRec) + P * 16 + F * 8 + (n_first + 1) * 80 + 64;
                // (the tables of 16-bit numbers: for the LDS forms, and -- the fans -- also for a mesh that does not fit the LDS,
                //  read from global memory then: star_lds_bytes < 0)
                const bool lds_ok = lds_need <= XRT_MESH_LDS_MAX && !env_on("XICSRT_NO_MESH_LDS");
                if (F < 0xffff && cells.size() < 0xffff && (lds_ok || !env_on("XICSRT_NO_MESH_FANS"))) {
                    struct Key { uint64_t a, b, c; bool operator==(const Key& o) const { return a == o.a && b == o.b && c == o.c; } };
                    struct KeyHash { size_t operator()(const Key& q) const { return (size_t)(q.a * 0x9e3779b97f4a7c15ull ^ (q.b + 0x7f4a7c15ull) * 0xbf58476d1ce4e5b9ull ^ (q.c * 0x94d049bb133111ebull)); } };
                    std::unordered_map<Key, int32_t, KeyHash> where;
                    where.reserve(P * 2);
                    auto key_of = [](const double* v) { Key q; memcpy(&q.a, v, 8); memcpy(&q.b, v + 1, 8); memcpy(&q.c, v + 2, 8); return q; };
                    for (size_t i = 0; i < P; i++) where.emplace(key_of(m->points + 3 * i), pslot[i]);
                    std::vector<uint16_t> fv(F * 4, 0), pf16(P * 8, 0xffff);
                    bool all = true;
                    for (size_t i = 0; i < F && all; i++) {
                        const double* vs[3] = {m->p0 + 3 * i, m->p1 + 3 * i, m->p2 + 3 * i};
                        for (int j = 0; j < 3 && all; j++) {
                            const auto it = where.find(key_of(vs[j]));
                            if (it == where.end() || it->second < 0) all = false;
                            else fv[4 * i + j] = (uint16_t)it->second;
                        }
                    }
                    for (size_t i = 0; i < P; i++)
                        for (int j = 0; j < 8; j++)
                            if (m->p_faces_mask[(size_t)j * P + i]) {
                                const int32_t f = m->p_faces_idx[(size_t)j * P + i];
                                if (f < 0 || (size_t)f >= F) all = false; else pf16[8 * i + j] = (uint16_t)f;
                            }
                    if (all && lds_ok) {
                        k.lds_pf = (const uint16_t*)put(pf16.data(), pf16.size() * 2);
                        k.lds_fv = (const uint16_t*)put(fv.data(), fv.size() * 2);
                        k.lds_bytes = (int32_t)lds_need;
                    }
                    // ---- the faces around every point as a fan (KMesh.lds_star, mesh_rest_star_lds) ------------------
                    const size_t star_need = cells.size() * sizeof(KCellRec) + P * 48 + (n_first + 1) * 80 + 64;
                    if (all && !env_on("XICSRT_NO_MESH_FANS")) {
                        auto sub = [](const double* a, const double* b, double* o) { o[0] = a[0] - b[0]; o[1] = a[1] - b[1]; o[2] = a[2] - b[2]; };
                        auto crs = [](const double* a, const double* b, double* o) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; };
                        auto dot = [](const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
                        auto len = [&](const double* a) { return sqrt(dot(a, a)); };
                        // the smallest altitude, edge and sin(angle / 2) of the mesh's faces
                        double h_min = HUGE_VAL, l_min = HUGE_VAL, s_min = 1.0;
                        bool faces_ok = true;
                        for (size_t i = 0; i < F && faces_ok; i++) {
                            const double* v[3] = {m->p0 + 3 * i, m->p1 + 3 * i, m->p2 + 3 * i};
                            double e[3][3], l[3], nrm[3];
                            for (int j = 0; j < 3; j++) { sub(v[(j + 1) % 3], v[j], e[j]); l[j] = len(e[j]); }
                            crs(e[0], e[1], nrm);
                            const double a2 = len(nrm), lmax = fmax(l[0], fmax(l[1], l[2])), lmn = fmin(l[0], fmin(l[1], l[2]));
                            faces_ok = std::isfinite(a2) && a2 > 0.0 && std::isfinite(lmax) && lmn > 0.0;
                            if (!faces_ok) break;
                            h_min = fmin(h_min, a2 / lmax); l_min = fmin(l_min, lmn);
                            for (int j = 0; j < 3; j++) {           // the angle at vertex j + 1, between -e[j] and e[j + 1]
                                const double cs = -dot(e[j], e[(j + 1) % 3]) / (l[j] * l[(j + 1) % 3]);
                                const double half = sqrt(fmax(0.0, 0.5 * (1.0 - fmin(1.0, fmax(-1.0, cs)))));       // sin(angle / 2)
                                s_min = fmin(s_min, half);
                            }
                        }
                        const double starK = faces_ok ? 1e-10 / (h_min * l_min * s_min) : HUGE_VAL;
                        if (std::isfinite(starK) && starK > 0.0 && starK < 1e-3) {
                            std::vector<uint16_t> star(P * 24, 0);
                            double spread = 0.0;
                            size_t n_fans = 0;
                            for (size_t i = 0; i < P; i++) {
                                if (pslot[i] < 0) continue;
                                const uint16_t cs = (uint16_t)pslot[i];
                                const KCellRec& pc = cells[cs];
                                int fl[8], nf = 0;
                                for (int j = 0; j < 8; j++) if (pf16[8 * i + j] != 0xffff) fl[nf++] = pf16[8 * i + j];
                                if (nf == 0) continue;
                                struct Spoke { uint16_t slot; double ang; };
                                std::vector<Spoke> sp;
                                bool ok = true;
                                for (int q = 0; q < nf && ok; q++) {
                                    int own = 0;
                                    for (int vtx = 0; vtx < 3; vtx++) {
                                        const uint16_t sl = fv[4 * (size_t)fl[q] + vtx];
                                        if (sl == cs) { own++; continue; }
                                        bool seen = false;
                                        for (const Spoke& t : sp) seen = seen || t.slot == sl;
                                        if (!seen) sp.push_back({sl, atan2(cells[sl].y - pc.y, cells[sl].x - pc.x)});
                                    }
                                    ok = own == 1;
                                    for (int r = 0; r < q && ok; r++) ok = fl[r] != fl[q];
                                }
                                const int ms = (int)sp.size();
                                if (!ok || ms < 2 || ms > 9) continue;
                                for (const Spoke& t : sp) ok = ok && std::isfinite(t.ang);
                                if (!ok) continue;
                                std::sort(sp.begin(), sp.end(), [](const Spoke& a, const Spoke& b) { return a.ang < b.ang; });
                                // the face between consecutive spokes (counter-clockwise); every face of the point must be one
                                std::vector<int> sect(ms, -1);
                                int used = 0;
                                for (int t = 0; t < ms; t++) {
                                    const uint16_t sa = sp[t].slot, sb = sp[(t + 1) % ms].slot;
                                    if (ms == 2 && t == 1) break;               // (two spokes: one sector)
                                    for (int q = 0; q < nf; q++) {
                                        const uint16_t* fq = &fv[4 * (size_t)fl[q]];
                                        const bool ha = fq[0] == sa || fq[1] == sa || fq[2] == sa, hb = fq[0] == sb || fq[1] == sb || fq[2] == sb;
                                        if (ha && hb && sa != sb) { if (sect[t] < 0) { sect[t] = fl[q]; used++; } else ok = false; }
                                    }
                                }
                                if (!ok || used != nf) continue;
                                // a sector with a face: counter-clockwise in x-y and narrower than pi; the fan's normals close together
                                double nsum[3] = {0, 0, 0};
                                std::vector<double> nrm((size_t)ms * 3, 0.0);
                                const double pp[3] = {pc.x, pc.y, pc.z};
                                for (int t = 0; t < ms && ok; t++) {
                                    if (sect[t] < 0) continue;
                                    const KCellRec &qa = cells[sp[t].slot], &qb = cells[sp[(t + 1) % ms].slot];
                                    const double a[3] = {qa.x, qa.y, qa.z}, b[3] = {qb.x, qb.y, qb.z};
                                    double e1[3], e2[3];
                                    sub(a, pp, e1); sub(b, pp, e2);
                                    double turn = sp[(t + 1) % ms].ang - sp[t].ang;
                                    if (turn < 0.0) turn += 6.283185307179586;
                                    crs(e1, e2, &nrm[3 * t]);
                                    const double nl = len(&nrm[3 * t]);
                                    ok = nrm[3 * t + 2] > 0.0 && turn > 1e-6 && turn < 3.1 && std::isfinite(nl) && nl > 0.0;
                                    if (!ok) break;
                                    for (int c = 0; c < 3; c++) { nrm[3 * t + c] /= nl; nsum[c] += nrm[3 * t + c]; }
                                }
                                if (!ok) continue;
                                double worst = 0.0;
                                for (int t = 0; t < ms; t++)
                                    for (int u = t + 1; u < ms; u++)
                                        if (sect[t] >= 0 && sect[u] >= 0) worst = fmax(worst, acos(fmin(1.0, fmax(-1.0, dot(&nrm[3 * t], &nrm[3 * u])))));
                                if (!(worst < 0.2)) continue;
                                // entries: a closed fan repeats its first spoke at the end; an open one starts behind a gap
                                int start = 0;
                                bool closed = true;
                                for (int t = 0; t < ms; t++) if (sect[t] < 0) { closed = false; start = (t + 1) % ms; }
                                if (ms == 2) { closed = false; start = 0; }
                                const int n_ent = closed ? ms + 1 : ms;
                                if (n_ent > 9) continue;
                                uint16_t* rec = &star[24 * i];
                                rec[0] = (uint16_t)n_ent;
                                uint16_t codes = 0;
                                for (int t = 0; t < n_ent; t++) rec[1 + t] = sp[(start + t) % ms].slot;
                                for (int t = 0; t < 8; t++) rec[10 + t] = 0xffff;
                                for (int t = 0; t + 1 < n_ent; t++) {
                                    const int st = (start + t) % ms;
                                    if (sect[st] < 0) continue;
                                    rec[10 + t] = (uint16_t)sect[st];
                                    const uint16_t v0 = fv[4 * (size_t)sect[st]];
                                    const uint16_t code = v0 == cs ? 0 : (v0 == sp[st].slot ? 1 : 2);
                                    codes |= (uint16_t)(code << (2 * t));
                                }
                                rec[18] = codes;
                                rec[19] = cs;
                                {   // the fan's mean plane through the point, z - z_p = gx (x - x_p) + gy (y - y_p), single precision
                                    const float gx = (float)(-nsum[0] / nsum[2]), gy = (float)(-nsum[1] / nsum[2]);
                                    if (!(std::isfinite(gx) && std::isfinite(gy))) { rec[0] = 0; continue; }
                                    memcpy(&rec[20], &gx, 4);
                                    memcpy(&rec[22], &gy, 4);
                                }
                                spread = fmax(spread, worst);
                                n_fans++;
                            }
                            if (n_fans > 0) {
                                k.lds_star = (const uint16_t*)put(star.data(), star.size() * 2);
                                k.star_lds_bytes = (lds_ok && star_need <= XRT_MESH_LDS_MAX) ? (int32_t)star_need : -1;
                                k.star_K = starK;
                                k.star_c2 = (2.0 * spread + 0.05) * (2.0 * spread + 0.05);
                            }
                        }
                    }
                }
            }
        }
        if (put_err != hipSuccess) return fail(-10, "%s", "the mesh tables do not fit their region of the workspace");
        memcpy(&blob[0], &k, sizeof(KMesh));
        ks->opt[e].mesh = reinterpret_cast<const KMesh*>(base);
        ks->opt[e].mesh_lds_bytes = k.lds_bytes;
        ks->opt[e].mesh_star_lds_bytes = k.star_lds_bytes;
        ks->opt[e].mesh_ct_lds_bytes = (m->interpolate && (size_t)m->n_points * 96 + 64 <= 160u * 1024u && !env_on("XICSRT_NO_MESH_LDS"))
                                           ? (int32_t)((size_t)m->n_points * 96) : 0;
        ks->opt[e].mesh_dir_bytes = (k.dg_n > 0 && !k.dg_big) ? (int32_t)(((size_t)k.n_first + 1) * 96 + (size_t)k.dg_n * k.dg_n * 8) : 0;
        // into pinned memory (kept for later calls), from there to the device on the stream; without pinned memory: a copy
        // from the temporary, behind whatever the stream still has to do with this part of the workspace
        char* pinned = nullptr;
        if (use_cache && hipHostMalloc(reinterpret_cast<void**>(&pinned), region, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); pinned = nullptr; }
        if (pinned) {
            memcpy(pinned, blob.data(), region);
            HIP_TRY(hipMemcpyAsync(base, pinned, region, hipMemcpyHostToDevice, stream));
            MeshCacheEntry c;
            c.fp = fp; c.dev = dev_now; c.pinned = pinned; c.bytes = region; c.base0 = (uintptr_t)base; c.k = k;
            c.lds_bytes = ks->opt[e].mesh_lds_bytes; c.star_lds_bytes = ks->opt[e].mesh_star_lds_bytes;
            c.ct_lds_bytes = ks->opt[e].mesh_ct_lds_bytes; c.dir_bytes = ks->opt[e].mesh_dir_bytes;
            std::lock_guard<std::mutex> lock(g_mesh_mu);
            c.used = ++g_mesh_tick;
            if (g_mesh_cache.size() >= 8) {
                // (the least recently used entry makes room; its pinned bytes may still feed a copy another thread queued a
                //  moment ago: they are released only after that device's work has drained -- rare, and off the common path)
                size_t old = 0;
                for (size_t i = 1; i < g_mesh_cache.size(); i++) if (g_mesh_cache[i].used < g_mesh_cache[old].used) old = i;
                if (g_mesh_cache[old].dev != dev_now) {       // (the entry of another device -- another host thread's: its work, not ours)
                    (void)hipSetDevice(g_mesh_cache[old].dev);
                    (void)hipDeviceSynchronize();
                    (void)hipSetDevice(dev_now);
                } else (void)hipDeviceSynchronize();
                (void)hipHostFree(g_mesh_cache[old].pinned);
                g_mesh_cache[old] = c;
            } else g_mesh_cache.push_back(c);
        } else {
            HIP_TRY(hipStreamSynchronize(stream));
            HIP_TRY(hipMemcpy(base, blob.data(), region, hipMemcpyHostToDevice));
        }
        base += region;
    }
    return 0;
}

// device copy of the per-bundle plasma model behind the mesh tables; fills KSource.plasma
static int upload_plasma(const xrt_scene_t* sc, char* ws, int n_runs, KScene* ks, hipStream_t stream)
{
    ks->src.plasma = nullptr;
    const xrt_plasma_t* P = sc->source.plasma;
    if (!P) return 0;
    char* base = ws + ws_off_staged(sc, n_runs) + staged_bytes(sc, n_runs) + meshes_bytes(sc);
    static thread_local KPlasma k;
    memset(&k, 0, sizeof(k));
    k.geometry = P->geometry; k.has_spread_radius = P->has_spread_radius; k.n_filters = P->n_filters;
    k.n_emissivity = P->n_emissivity; k.n_temperature = P->n_temperature;
    for (int i = 0; i < 3; i++) k.torus_origin[i] = P->torus_origin[i];
    k.major_radius = P->major_radius; k.minor_radius = P->minor_radius;
    k.emissivity = P->emissivity; k.emissivity_scale = P->emissivity_scale; k.temperature_scale = P->temperature_scale;
    k.spread_radius = P->spread_radius; k.solid_angle = P->solid_angle;
    k.time_resolution = P->time_resolution; k.bundle_volume = P->bundle_volume; k.four_pi = P->four_pi;
    k.volume_ratio = P->volume_ratio;
    k.mass_number = P->mass_number; k.amu_kg = P->amu_kg; k.c_squared = P->c_squared; k.ev_J = P->ev_J;
    for (int i = 0; i < XRT_MAX_BUNDLE_FILTERS; i++) k.filters[i] = P->filters[i];
    char* p = base + al256(sizeof(KPlasma));
    auto put = [&](const double* src, int n) -> const double* {
        char* dst = p;
        p += al256(sizeof(double) * (size_t)(n > 0 ? n : 1));
        if (src && n > 0) (void)hipMemcpyAsync(dst, src, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, stream);
        return reinterpret_cast<const double*>(dst);
    };
    k.emissivity_rho = put(P->emissivity_rho, P->n_emissivity);
    k.emissivity_val = put(P->emissivity_val, P->n_emissivity);
    k.temperature_rho = put(P->temperature_rho, P->n_temperature);
    k.temperature_val = put(P->temperature_val, P->n_temperature);
    k.voigt_gamma = P->voigt_gamma; k.weideman_L = P->weideman_L; k.n_weideman = P->n_weideman;
    k.weideman_a = put(P->weideman_a, P->n_weideman);
    HIP_TRY(hipMemcpyAsync(base, &k, sizeof(KPlasma), hipMemcpyHostToDevice, stream));
    ks->src.plasma = reinterpret_cast<const KPlasma*>(base);
    return 0;
}

// uploads of the small tables (pageable host memory -> staged copies, ordered on the stream)
static int upload_tables(const xrt_scene_t* sc, char* ws, hipStream_t stream)
{
    for (int e = 0; e < sc->n_optics; e++)
        if (sc->optics[e].n_apertures > 0)
            HIP_TRY(hipMemcpyAsync(ws + ws_off_apertures() + sizeof(xrt_aperture_t) * XRT_MAX_APERTURES * (size_t)e,
                                   sc->optics[e].apertures, sizeof(xrt_aperture_t) * (size_t)sc->optics[e].n_apertures,
                                   hipMemcpyHostToDevice, stream));
    if (sc->source.wavelength_dist == XRT_WL_VOIGT) {
        size_t nb = sizeof(double) * (size_t)sc->source.voigt_n;
        HIP_TRY(hipMemcpyAsync(ws + ws_off_voigt(), sc->source.voigt_cdf, nb, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(ws + ws_off_voigt() + nb, sc->source.voigt_x, nb, hipMemcpyHostToDevice, stream));
    }
    return 0;
}

// The flattened scene goes to the workspace through the kernel-argument path (copied by the runtime
// at launch, ordered on the stream, no host buffer whose lifetime would matter), 2 KB at a time.
struct KBlob { uint32_t w[512]; };
__global__ void xrt_put_kernel(uint32_t* dst, const KBlob blob, int n_words)
{
    for (int i = threadIdx.x; i < n_words; i += blockDim.x) dst[i] = blob.w[i];
}
static int upload_scene(const KScene& ks, char* ws, hipStream_t stream)
{
    static_assert(sizeof(KScene) % 4 == 0, "scene size");
    const uint32_t* src = reinterpret_cast<const uint32_t*>(&ks);
    uint32_t* dst = reinterpret_cast<uint32_t*>(ws + ws_off_scene());
    // (the source and the optics in use: 0.5 KB each of the 64 the structure has room for)
    const int total = (int)((offsetof(KScene, opt) + sizeof(KOptic) * (size_t)(ks.n_optics > 0 ? ks.n_optics : 0) + 3) / 4);
    for (int o = 0; o < total; o += 512) {
        KBlob b;
        const int n = (total - o) < 512 ? (total - o) : 512;
        memcpy(b.w, src + o, (size_t)n * 4);
        hipLaunchKernelGGL(xrt_put_kernel, dim3(1), dim3(256), 0, stream, dst + o, b, n);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}
static const KScene* device_scene(char* ws) { return reinterpret_cast<const KScene*>(ws + ws_off_scene()); }

// Runs just above a multiple of the resident workgroups: a run is one sequential stream, so the last few would each
// hold a workgroup slot for a full run time while the rest of the chip idles (1024 runs: 20.0 ms, 1025: 35.2 ms).
// xrt_trace lets the unsegmented launch leave out such a tail (< 256 runs) and traces it as a call of its own, which
// takes the segmented route.  t_tail: -1 not allowed here, 0 allowed (set to the number of runs left out), > 0 decided.
static thread_local int t_tail = -1;
static bool tail_split_possible(const xrt_scene_t* sc)
{
    return sc->source.intensity >= 200000 && !needs_staged(sc) && !env_on("XICSRT_NO_TAIL_SPLIT");
}

template <bool HIST, int VARIANT, int SEG = 0>
static int launch_variant(const KScene* ks, const KArgs& a_in, int n_runs, size_t lds, hipStream_t stream, bool may_leave_tail = false)
{
    KArgs a = a_in;
    if (a.image_rep > 1u && a.images) a.images = a.images_rep;
#ifdef XRT_DEV_ONLY_LEAN     // development builds: only the lean kernels (-DXRT_DEV_VARIANT=2: the mesh / local-frame ones) are compiled
#ifndef XRT_DEV_VARIANT
#define XRT_DEV_VARIANT 0
#endif
    if constexpr (HIST || VARIANT != XRT_DEV_VARIANT) return fail(-3, "%s", "development build: one kernel variant only");
    else {
#endif
    auto kern = xrt_trace_kernel<HIST, VARIANT, SEG>;
    // attribute and occupancy queries cost ~0.1 ms each: once per (device, LDS size) and instantiation
    static thread_local int c_dev = -1, c_cus = 256, c_per_cu = 1;
    static thread_local size_t c_lds = 0;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != c_dev || lds != c_lds) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_TRY(hipDeviceGetAttribute(&c_cus, hipDeviceAttributeMultiprocessorCount, dev));
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&c_per_cu, kern, XRT_TILE, lds));
        if (c_per_cu < 1) c_per_cu = 1;
        if (c_per_cu > 8) c_per_cu = 8;
        c_dev = dev; c_lds = lds;
    }
    int cus = c_cus, per_cu = c_per_cu;
    if (const char* cap = getenv("XICSRT_MAX_WG_PER_CU")) { int c = atoi(cap); if (c >= 1 && c < per_cu) per_cu = c; }
    int grid = cus * per_cu;
    if (!SEG && !HIST && may_leave_tail && t_tail >= 0 && !env_on("XICSRT_NO_TAIL_SPLIT")) {
        if (t_tail == 0 && n_runs > grid && (n_runs % grid) < 256) t_tail = n_runs % grid;
        if (t_tail > 0) { n_runs -= t_tail; a.n_runs = n_runs; }
    }
    const int units = SEG ? n_runs * a.n_seg * a.n_sub : n_runs;
    if (grid > units) grid = units;
    if (grid < 1) grid = 1;
    int ti = -1;
    if (timing_on && timing_n < TIMING_MAX) {
        ti = timing_n++;
        HIP_TRY(hipEventCreate(&timing_ev[ti][0]));
        HIP_TRY(hipEventCreate(&timing_ev[ti][1]));
        HIP_TRY(hipEventRecord(timing_ev[ti][0], stream));
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(XRT_TILE), lds, stream, ks, a);
    HIP_TRY(hipGetLastError());
    if (ti >= 0) HIP_TRY(hipEventRecord(timing_ev[ti][1], stream));
    return 0;
#ifdef XRT_DEV_ONLY_LEAN
    }
#endif
}

// one iteration of every run: position the heads (jump-ahead when every stream is in the
// canonical form and the arrays are long enough, else the sequential walk), then propagate
// The unsegmented routes: every run's source heads at the first words of their arrays and the run's stream head behind the
// source arrays (jump-ahead when every stream is in the canonical form, else the sequential walk).
static int position_heads(const xrt_scene_t* sc, const KScene& ks, char* ws, int n_runs, int nh, int64_t N, bool canonical,
                          KStream* streams, KStream* heads, hipStream_t stream)
{
    if (canonical && !env_on("XICSRT_NO_JUMP") && N >= (int64_t)XRT_AHEAD / 2) {
        g_paths |= XRT_PATH_JUMP;
        static thread_local std::vector<uint32_t> hpolys;
        hpolys.clear();
        int n_polys = 0;
        for (int k = 1; k <= ks.src.n_arrays; k++) {
            if (k < ks.src.n_arrays && !((ks.src.array_used >> k) & 1u)) continue;
            hpolys.resize((size_t)(n_polys + 1) * 624);
            const uint64_t J = 2ull * (uint64_t)k * (uint64_t)N - (uint64_t)XRT_AHEAD;
            if (!mtjump::jump_poly(J, hpolys.data() + (size_t)n_polys * 624))
                return fail(-5, "%s", "MT19937 characteristic polynomial could not be derived");
            n_polys++;
        }
        uint32_t* d_polys = reinterpret_cast<uint32_t*>(ws + ws_off_polys(sc, n_runs));
        HIP_TRY(hipMemcpyAsync(d_polys, hpolys.data(), sizeof(uint32_t) * 624 * (size_t)n_polys, hipMemcpyHostToDevice, stream));
        const size_t jl = sizeof(uint32_t) * ((size_t)XRT_STRETCH + 48 + 624 + 624 * (size_t)n_polys);
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(xrt_jump_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)jl));
        int dev = 0, cus = 256;
        HIP_TRY(hipGetDevice(&dev));
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        int grid = n_runs < cus ? n_runs : cus;
        hipLaunchKernelGGL(xrt_jump_kernel, dim3(grid), dim3(XRT_JUMP_THREADS), jl, stream, streams, heads, d_polys, n_runs,
                           ks.src.n_arrays, ks.src.array_used, nh, n_polys, N);
        HIP_TRY(hipGetLastError());
    } else {
        g_paths |= XRT_PATH_SEEK;
        hipLaunchKernelGGL(xrt_seek_kernel, dim3((n_runs + 3) / 4), dim3(256), 0, stream,
                           streams, heads, n_runs, ks.src.n_arrays, ks.src.array_used, nh, N);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

// A mosaic crystal's layers over parked rays (xrt_mosaic.inc) instead of the staged kernel: one mosaic crystal of an analytic
// shape in the global frame, no other Bragg test, no mesh and no local frame anywhere, an ordinary source whose draws have a
// fixed length; the parked rays and the layers' draws in the staged path's slots, a run per slot, the runs in batches of the slots.
static bool mosaic_fused_ok(const xrt_scene_t* sc, int n_runs, bool hist)
{
    if (hist || env_on("XICSRT_NO_MOSAIC_FUSED")) return false;
    const xrt_source_t& s = sc->source;
    if (s.kind == XRT_SRC_PLASMA || s.kind == XRT_SRC_EXTERNAL || s.n_ray_filters > 0) return false;
    if (s.spatial_dist == XRT_SPATIAL_GAUSSIAN || s.angular_dist == XRT_ANG_ISOTROPIC_XY || s.wavelength_dist == XRT_WL_NORMAL) return false;
    // (runs of few rays: the staged kernel's few passes are as good, and a record block of 256 per run does not fit its slot;
    //  XICSRT_MOSAIC_FUSED_MIN: the tests send small scenes this way)
    long long least = 65536;
    if (const char* e = getenv("XICSRT_MOSAIC_FUSED_MIN")) { const long long v = atoll(e); if (v >= 1) least = v; }
    if (s.intensity < least) return false;
    {
        const bool has_wl = !(s.wavelength_dist == XRT_WL_CONST && !s.has_velocity);
        if (cand_capacity(sc) * (8 * (size_t)(has_wl ? 7 : 6) + 32) + 16 > staged_slot_bytes(sc)) return false;
    }
    int n_mosaic = 0;
    for (int e = 0; e < sc->n_optics; e++) {
        const xrt_optic_t& o = sc->optics[e];
        if (o.interact == XRT_INTERACT_MOSAIC) n_mosaic++;
        if (o.interact == XRT_INTERACT_CRYSTAL && (o.flags & XRT_F_CHECK_BRAGG)) return false;
    }
    (void)n_runs;           // (more runs than the staged path has slots: batches)
    return n_mosaic == 1 && !needs_ext(sc);
}

// `ahead`: words every stream head has generated beyond `next` when this is called (XRT_AHEAD: the
// canonical form between kernels; 624: an imported state after xrt_advance_kernel; 0: unknown)
static int run_iteration(const xrt_scene_t* sc, const KScene& ks, char* ws, size_t ws_bytes, KArgs a, int n_runs, bool hist,
                         int ahead, hipStream_t stream, bool force_staged = false)
{
    const bool canonical = ahead == (int)XRT_AHEAD;
    const int nh = count_heads(sc);
    KStream* streams = reinterpret_cast<KStream*>(ws + ws_off_streams(sc, n_runs));
    KStream* heads = reinterpret_cast<KStream*>(ws + ws_off_heads(sc, n_runs));
    const int64_t N = ks.src.n_rays;
    if (plasma_fused(sc) && !hist && !force_staged) {
        // ---- plasma source on the fused path: scout (one wave per run), then the fused kernel ---------------
        g_paths |= XRT_PATH_FUSED | XRT_PATH_PLASMA_SCOUT;
        const int slots = plasma_slots(sc, n_runs);
        const size_t B = (size_t)(sc->source.bundle_count > 0 ? sc->source.bundle_count : 0), NN = (size_t)(N > 0 ? N : 1);
        char* base = ws + ws_off_plasma_rays(sc, n_runs);
        KScout ps;
        memset(&ps, 0, sizeof(ps));
        char* p = base;
        ps.n_src = reinterpret_cast<long long*>(p);                     p += al256((size_t)slots * 8);
        ps.gauss0 = reinterpret_cast<double*>(p);                       p += al256((size_t)slots * 8);
        ps.btab = reinterpret_cast<double*>(p);                         p += (size_t)slots * al256(B * XRT_PB_ROWS * 8);
        ps.bpos = reinterpret_cast<long long*>(p);                      p += (size_t)slots * al256(B * XRT_PP_ROWS * 8);
        ps.ray_bundle = reinterpret_cast<uint32_t*>(p);                 p += (size_t)slots * al256(NN * 4);
        ps.wl = reinterpret_cast<double*>(p);                           p += (size_t)slots * al256(NN * 8);
        ps.dump = reinterpret_cast<uint32_t*>(p);
        ps.dump_words = (uint32_t)plasma_dump_words(sc);
        // (slot strides: the arrays are indexed [slot][...] with the unpadded sizes; al256 only pads the regions)
        ps.gauss_state = reinterpret_cast<KState*>(ws + ws_off_gauss(sc, n_runs));
        ps.flags = reinterpret_cast<uint32_t*>(ws + XRT_WS_STATUS_BYTE);
        a.n_src_heads = 0;
        a.run_counter = reinterpret_cast<uint32_t*>(ws);
        a.progress = env_on("XICSRT_NO_PRIORITY_FEEDBACK") ? nullptr : reinterpret_cast<unsigned long long*>(ws + 32);
        a.plasma.btab = ps.btab; a.plasma.bpos = ps.bpos; a.plasma.ray_bundle = ps.ray_bundle; a.plasma.wl = ps.wl;
        a.plasma.dump = ps.dump; a.plasma.n_src = ps.n_src; a.plasma.dump_words = ps.dump_words; a.plasma.gauss0 = ps.gauss0;
        const bool ext = needs_ext(sc);
        const size_t lds = plan_queue(ks, 0, ext, false, &a);
        for (int base_run = 0; base_run < n_runs; base_run += slots) {
            const int nb = (n_runs - base_run) < slots ? (n_runs - base_run) : slots;
            ps.run_base = base_run; ps.n_runs = nb;
            const bool voigt = sc->source.plasma && sc->source.plasma->voigt_gamma > 0.0;
            // (bundles with a Voigt table of their own: one wave -- one run -- per workgroup, its table in LDS; see the kernel)
            const size_t scout_lds = voigt ? 2 * XRT_VOIGT_GRID * sizeof(double) : 0;
            if (voigt) {
                HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(xrt_plasma_scout_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)scout_lds));
                hipLaunchKernelGGL((xrt_plasma_scout_kernel<true, true>), dim3(nb), dim3(64), scout_lds, stream, device_scene(ws), streams, ps);
            } else if (sc->source.plasma)
                hipLaunchKernelGGL((xrt_plasma_scout_kernel<false, true>), dim3((nb + 3) / 4), dim3(256), 0, stream, device_scene(ws), streams, ps);
            else
                hipLaunchKernelGGL((xrt_plasma_scout_kernel<false, false>), dim3((nb + 3) / 4), dim3(256), 0, stream, device_scene(ws), streams, ps);
            HIP_TRY(hipGetLastError());
            a.streams = streams + base_run; a.heads = heads; a.n_runs = nb;
            HIP_TRY(hipMemsetAsync(ws, 0, 64, stream));
            const int st = ext ? launch_variant<false, 2>(device_scene(ws), a, nb, lds, stream)
                               : launch_variant<false, 1>(device_scene(ws), a, nb, lds, stream);
            if (st) return st;
        }
        return 0;
    }
#ifndef XRT_DEV_ONLY_LEAN
    if (!force_staged && mosaic_fused_ok(sc, n_runs, hist)) {
        // ---- a mosaic crystal: the fused kernel's first phase parks the rays on the crystal, xrt_mosaic_kernel does the rest
        g_paths |= XRT_PATH_FUSED | XRT_PATH_MOSAIC_FUSED;
        {
            const int st = position_heads(sc, ks, ws, n_runs, nh, N, canonical, streams, heads, stream);
            if (st) return st;
        }
        int be = -1;
        for (int e = 0; e < sc->n_optics; e++) if (sc->optics[e].interact == XRT_INTERACT_MOSAIC) be = e;
        const bool has_wl = !(ks.src.wavelength_dist == XRT_WL_CONST && !ks.src.has_velocity);
        const int ncomp = has_wl ? 7 : 6;
        const size_t cap = cand_capacity(sc);
        // (the arena: the staged path's slots, one per run of a batch -- staged_slot_bytes >= what a run takes here; a call
        //  with more runs than slots -- 1024, or what the budget holds -- goes through them in equal batches)
        const int slots = staged_slots(sc, n_runs);
        const int n_batches = (n_runs + slots - 1) / slots, per_batch = (n_runs + n_batches - 1) / n_batches;
        char* base = ws + ws_off_staged(sc, n_runs);
        const size_t per_run = cap * (8 * (size_t)ncomp + 4 + 24 + 4);
        if (per_run + 16 > staged_slot_bytes(sc)) return fail(-5, "%s", "mosaic route: slot too small");
        KMosaic mo;
        memset(&mo, 0, sizeof(mo));
        char* p = base;
        mo.cand = reinterpret_cast<double*>(p);  p += (size_t)per_batch * cap * 8 * (size_t)ncomp;
        mo.draws = reinterpret_cast<double*>(p); p += (size_t)per_batch * cap * 24;
        mo.list = reinterpret_cast<uint32_t*>(p); p += (size_t)per_batch * cap * 4;
        mo.mark = reinterpret_cast<uint32_t*>(p); p += (size_t)per_batch * cap * 4;
        // (the runs' counts: in the staged path's per-slot words behind the slots -- 16 bytes per slot)
        uint32_t* d_flag = reinterpret_cast<uint32_t*>(base + (size_t)per_batch * (staged_slot_bytes(sc) - 16));
        if ((char*)d_flag < p) return fail(-5, "%s", "mosaic route: slot layout");
        mo.n_cand = d_flag;
        mo.cap = (int64_t)cap; mo.be = be; mo.ncomp = ncomp;
        mo.dbg = reinterpret_cast<unsigned long long*>(ws + XRT_WS_STATUS_BYTE + 64);     // (development builds: the header's spare words)
        a.n_src_heads = nh;
        a.run_counter = reinterpret_cast<uint32_t*>(ws);
        a.progress = env_on("XICSRT_NO_PRIORITY_FEEDBACK") ? nullptr : reinterpret_cast<unsigned long long*>(ws + 32);
        // first phase: every run one unit (one segment, one part), the plain layout of the heads
        const int64_t Lr = (N + XRT_TILE - 1) / XRT_TILE * XRT_TILE;
        a.n_seg = 1; a.seg_len = Lr; a.n_sub = 1; a.sub_len = Lr; a.run_stride = nh; a.mode = 2;
        a.chunk_heads = nullptr; a.chunk_words = 1;
        a.cand = mo.cand; a.cand_id = nullptr; a.cand_aux = nullptr; a.cand_cap = (int64_t)cap; a.unit_flag = d_flag;
        a.unit_o = nullptr; a.split_interp = 0; a.dir_lds_bytes = 0;
        a.qcap = XRT_TILE; a.bragg_batch = 128u;
        const size_t lds3 = lds_bytes(nh, false, false, has_wl, XRT_TILE);
        // (pixel bins in LDS for the reflected rays, where the scene's bins fit 40 KB of LDS together with the rest -- four
        //  workgroups per CU -- as 16-bit counters: they take the place of the ring, which is done by then, and what they need beyond)
        const uint32_t ring_words = XRT_PC_RING + XRT_PC_MIRROR + 4u, small_words = 2 * (XRT_DEV_MAX_OPTICS + 2) + 8 + XRT_PC_CTL + 4;
        const uint32_t bins_words = (a.images && sc->image_bins > 0 && sc->image_bins <= 19000 && !env_on("XICSRT_NO_LDS_BINS")) ? (uint32_t)((sc->image_bins + 1) / 2) : 0u;
        mo.lbins_words = bins_words;
        const size_t lds_mo = sizeof(uint32_t) * ((size_t)small_words + (bins_words > ring_words ? bins_words : ring_words));
        int ti = -1;
        if (timing_on && timing_n < TIMING_MAX) {
            ti = timing_n++;
            HIP_TRY(hipEventCreate(&timing_ev[ti][0]));
            HIP_TRY(hipEventCreate(&timing_ev[ti][1]));
            HIP_TRY(hipEventRecord(timing_ev[ti][0], stream));
        }
        for (int base_run = 0; base_run < n_runs; base_run += per_batch) {
            const int nb = (n_runs - base_run) < per_batch ? (n_runs - base_run) : per_batch;
            a.streams = streams + base_run; a.heads = heads + (size_t)base_run * (size_t)nh; a.n_runs = nb;
            mo.gauss_state = reinterpret_cast<KState*>(ws + ws_off_gauss(sc, n_runs)) + base_run;
            HIP_TRY(hipMemsetAsync(d_flag, 0, sizeof(uint32_t) * (size_t)nb, stream));
            HIP_TRY(hipMemsetAsync(ws, 0, 64, stream));
            const int st = needs_full(sc) ? launch_variant<false, 1, 3>(device_scene(ws), a, nb, lds3, stream)
                                          : launch_variant<false, 0, 3>(device_scene(ws), a, nb, lds3, stream);
            if (st) return st;
            // the layers and the elements behind the crystal: a workgroup per run
            KArgs am = a;
            if (am.image_rep > 1u && am.images) am.images = am.images_rep;
            HIP_TRY(hipMemsetAsync(ws, 0, 64, stream));
            hipLaunchKernelGGL(xrt_mosaic_kernel, dim3((unsigned)(nb < 1024 ? nb : 1024)), dim3(XRT_TILE), lds_mo, stream, device_scene(ws), am, mo);
            HIP_TRY(hipGetLastError());
        }
        if (ti >= 0) HIP_TRY(hipEventRecord(timing_ev[ti][1], stream));
        return 0;
    }
#endif
#ifdef XRT_DEV_ONLY_LEAN
    if (needs_staged(sc) || force_staged) return fail(-3, "%s", "development build: lean kernel only");
#else
    if (needs_staged(sc) || force_staged) {
        // general path: array-at-a-time passes with one sequential stream head per run
        KStaged g;
        memset(&g, 0, sizeof(g));
        const int slots = staged_slots(sc, n_runs);
        char* base = ws + ws_off_staged(sc, n_runs);
        g.arr = reinterpret_cast<double*>(base);
        g.ids = reinterpret_cast<uint32_t*>(base + (size_t)slots * (size_t)N * XRT_ST_ARRAYS * sizeof(double));
        g.aux = g.ids + (size_t)slots * (size_t)N;
        g.act = g.aux + (size_t)slots * (size_t)N;
        g.bundle_off = reinterpret_cast<double*>(base + (size_t)slots * (size_t)N * (XRT_ST_ARRAYS * sizeof(double) + 3 * sizeof(uint32_t)));
        g.flags = reinterpret_cast<uint32_t*>(ws + XRT_WS_STATUS_BYTE);        // status word in the 256-byte workspace header
        if (sc->source.plasma && sc->source.plasma->voigt_gamma > 0.0)      // behind the bundle tables and the per-slot counts
            g.voigt_tab = g.bundle_off + (size_t)slots * XRT_ST_BUNDLE_ROWS * (size_t)(sc->source.bundle_count > 0 ? sc->source.bundle_count : 0) + 2 * (size_t)slots;
        g.gauss_state = reinterpret_cast<KState*>(ws + ws_off_gauss(sc, n_runs));
        for (int i = 0; i < 9; i++) g.spatial_A[i] = sc->source.spatial_A[i];
        g.spatial_gaussian = sc->source.spatial_dist == XRT_SPATIAL_GAUSSIAN;
        a.streams = streams; a.heads = heads; a.n_runs = n_runs; a.n_src_heads = nh;
        a.run_counter = reinterpret_cast<uint32_t*>(ws);
        if (a.image_rep > 1u && a.images) a.images = a.images_rep;      // (the replicas of the bins, summed behind the call's last launch)
        HIP_TRY(hipMemsetAsync(ws, 0, 64, stream));
        const size_t lds = sizeof(double) * XRT_TILE_COMP * XRT_TILE + sizeof(uint32_t) * (2 * XRT_TILE + XRT_RING + 16 + 2 * (XRT_DEV_MAX_OPTICS + 2) + 16 + XRT_ST_FSPEC);
        // (the instances that run a mosaic crystal's layers with the waves in roles -- ST_PC -- have the larger ring and the hand-over words)
        const size_t lds_pc = lds + sizeof(uint32_t) * (XRT_PC_RING + XRT_PC_MIRROR + 4u - XRT_RING + XRT_PC_CTL + 4);
        int ti = -1;
        if (timing_on && timing_n < TIMING_MAX) {
            ti = timing_n++;
            HIP_TRY(hipEventCreate(&timing_ev[ti][0]));
            HIP_TRY(hipEventCreate(&timing_ev[ti][1]));
            HIP_TRY(hipEventRecord(timing_ev[ti][0], stream));
        }
        const int src = sc->source.kind == XRT_SRC_PLASMA ? 1 : (sc->source.kind == XRT_SRC_EXTERNAL ? 2 : 0);
        bool special = false, has_mesh = false;
        for (int e = 0; e < sc->n_optics; e++) {
            special = special || sc->optics[e].interact == XRT_INTERACT_MOSAIC || sc->optics[e].shape == XRT_SHAPE_MESH ||
                      (sc->optics[e].flags & XRT_F_TRACE_LOCAL);
            has_mesh = has_mesh || sc->optics[e].shape == XRT_SHAPE_MESH;
        }
        if (src == 2 && !hist) return fail(-2, "%s", "external rays are traced through xrt_trace_history");
        g_paths |= XRT_PATH_STAGED;
        if (src != 2 && !hist && !env_on("XICSRT_NO_STAGE_SPLIT")) {
            g_paths |= XRT_PATH_STAGE_SPLIT;
            // source and optics in separate launches, a batch of `slots` runs at a time
            g.n_src_slot = reinterpret_cast<int64_t*>(g.bundle_off + (size_t)slots * XRT_ST_BUNDLE_ROWS * (size_t)(sc->source.bundle_count > 0 ? sc->source.bundle_count : 0));
            void (*k1)(const KScene*, const KArgs, const KStaged) = src == 1 ? xrt_staged_kernel<false, 1, 0, 1>
                                                                            : xrt_staged_kernel<false, 0, 0, 1>;
            // (the optics stage without the mesh code -- mosaic crystals, local frames -- needs half the registers of the one with it)
            void (*k2)(const KScene*, const KArgs, const KStaged) =
                src == 1 ? (has_mesh ? xrt_staged_kernel<false, 1, 2, 2> : special ? xrt_staged_kernel<false, 1, 1, 2> : xrt_staged_kernel<false, 1, 0, 2>)
                         : (has_mesh ? xrt_staged_kernel<false, 0, 2, 2> : special ? xrt_staged_kernel<false, 0, 1, 2> : xrt_staged_kernel<false, 0, 0, 2>);
            // (batches of equal size: a workgroup's time hardly depends on how many share its CU -- the stages are chains
            //  of barriers and memory round trips -- so a last batch of a few runs would cost as much as a full one)
            const int n_batches = (n_runs + slots - 1) / slots, per_batch = (n_runs + n_batches - 1) / n_batches;
            for (int base_run = 0; base_run < n_runs; base_run += per_batch) {
                const int nb = (n_runs - base_run) < per_batch ? (n_runs - base_run) : per_batch;
                g.run_base = base_run;
                a.n_runs = nb;
                const int g1 = src == 1 ? 768 : 1024, g2 = has_mesh ? 512 : (special ? 1024 : 768);     // workgroups per launch: 3-4 (source) and 2 - 4 (optics) per CU
                int grid1 = nb < g1 ? nb : g1, grid2 = nb < g2 ? nb : g2;
                HIP_TRY(hipMemsetAsync(ws, 0, 64, stream));
                hipLaunchKernelGGL(k1, dim3(grid1), dim3(XRT_TILE), lds, stream, device_scene(ws), a, g);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipMemsetAsync(ws, 0, 64, stream));
                hipLaunchKernelGGL(k2, dim3(grid2), dim3(XRT_TILE), (has_mesh || special) ? lds_pc : lds, stream, device_scene(ws), a, g);
                HIP_TRY(hipGetLastError());
            }
            if (ti >= 0) HIP_TRY(hipEventRecord(timing_ev[ti][1], stream));
            return 0;
        }
        void (*kern)(const KScene*, const KArgs, const KStaged);
        if (hist) kern = src == 2 ? (special ? xrt_staged_kernel<true, 2, 2> : xrt_staged_kernel<true, 2, 0>)
                       : src == 1 ? (special ? xrt_staged_kernel<true, 1, 2> : xrt_staged_kernel<true, 1, 0>)
                                  : (special ? xrt_staged_kernel<true, 0, 2> : xrt_staged_kernel<true, 0, 0>);
        else      kern = src == 1 ? (special ? xrt_staged_kernel<false, 1, 2> : xrt_staged_kernel<false, 1, 0>)
                                  : (special ? xrt_staged_kernel<false, 0, 2> : xrt_staged_kernel<false, 0, 0>);
        hipLaunchKernelGGL(kern, dim3(slots), dim3(XRT_TILE), (special && !hist) ? lds_pc : lds, stream, device_scene(ws), a, g);
        HIP_TRY(hipGetLastError());
        if (ti >= 0) HIP_TRY(hipEventRecord(timing_ev[ti][1], stream));
        return 0;
    }
#endif
    SegPlan plan = ahead > 0 ? plan_segments(sc, n_runs) : SegPlan{1, 0, 0, 0, 0};
    if (plan.mesh_split && hist) {          // (histories: the routes without the split)
        if (plan.n_seg * plan.n_sub == 1 && plan.n_gchunks == 0) plan = SegPlan{1, 0, 0, 0, 0};
        else plan.mesh_split = false;
    }
    if (seg_active(plan)) {
        // ---- segmented runs -------------------------------------------------------------------------
        g_paths |= XRT_PATH_FUSED | XRT_PATH_SEGMENTED | XRT_PATH_JUMP;
        const int S = plan.n_seg, M = plan.n_sub, nj = seg_jobs(sc, plan);
        if (env_on("XICSRT_PLAN_DEBUG"))
            fprintf(stderr, "[xicsrt] plan: runs %d, %d segments x %d parts of %lld rays, %d chunk heads every %lld words, %d jump jobs per run\n",
                    n_runs, S, M, (long long)plan.sub_len, plan.n_chunk_heads, (long long)plan.chunk_words, nj);
        const int64_t L = plan.seg_len, CH = plan.chunk_words;
        const int be = bragg_element(sc);
        char* base = ws + ws_off_seg(sc, n_runs);
        KStream* dst = reinterpret_cast<KStream*>(base);
        uint32_t* d_polys = reinterpret_cast<uint32_t*>(base + al256(sizeof(KStream) * (size_t)nj * (size_t)n_runs));
        uint64_t* d_off = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(d_polys) + jump_partial_bytes(n_runs, (size_t)nj));
        uint32_t* d_cnt = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(d_off) + al256(sizeof(uint64_t) * (size_t)nj));
        double* d_wl = reinterpret_cast<double*>(reinterpret_cast<char*>(d_cnt) + al256(sizeof(uint32_t) * (size_t)n_runs * (size_t)S * (size_t)M) + 256);
        uint32_t* d_gacc = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(d_wl) + al256(sizeof(double) * (size_t)n_runs * (size_t)N));
        uint64_t* d_gend = reinterpret_cast<uint64_t*>(reinterpret_cast<char*>(d_gacc) + al256(sizeof(uint32_t) * (size_t)n_runs * (size_t)(plan.n_gchunks > 0 ? plan.n_gchunks : 1)));
        const int n_ch = plan.n_chunk_heads, n_gch = plan.n_gchunks;
        const uint64_t GCH = 4ull * (uint64_t)plan.gpairs;
        // What the jump needs of a plan -- per group of eight jobs the positions of the polynomials' terms sorted by class
        // (see xrt_jump_groups_kernel), and the jobs' offsets -- is derived once, kept in device memory the library owns
        // and reused by later calls (a plan = ray count, segmentation and source arrays; not the seeds)
        struct PlanCache { int64_t N, L, gpairs, CH; int S, nj, n_ch; uint32_t used; int n_arrays, ahead, dev; uint16_t* pos; uint32_t* cls; uint64_t* offs; };
        static std::mutex plan_mu;
        static std::vector<PlanCache> plans;
        PlanCache hit;
        hit.pos = nullptr; hit.cls = nullptr; hit.offs = nullptr;
        bool plan_once = false;
        int dev_now = 0;
        HIP_TRY(hipGetDevice(&dev_now));
        {
            std::lock_guard<std::mutex> lock(plan_mu);
            for (const PlanCache& c : plans)
                if (c.N == N && c.L == L && c.S == S && c.nj == nj && c.used == ks.src.array_used && c.n_arrays == ks.src.n_arrays &&
                    c.ahead == ahead && c.n_ch == n_ch && c.gpairs == plan.gpairs && c.CH == CH && c.dev == dev_now) hit = c;
        }
        const int n_groups = (nj + XRT_JG - 1) / XRT_JG;
        if (!hit.pos) {
        static thread_local std::vector<uint64_t> offs;
        static thread_local std::vector<uint32_t> polys;
        offs.assign((size_t)nj, 0);
        int j = 0;
        for (int sgm = 0; sgm < S; sgm++)
            for (int k = 0; k < ks.src.n_arrays; k++)
                if ((ks.src.array_used >> k) & 1u) offs[j++] = 2ull * ((uint64_t)k * (uint64_t)N + (uint64_t)sgm * (uint64_t)L);
        const uint64_t behind = 2ull * (uint64_t)ks.src.n_arrays * (uint64_t)N;
        for (int c = 0; c < n_ch; c++) offs[j++] = behind + (uint64_t)c * (uint64_t)CH;
        for (int c = 0; c < n_gch; c++) offs[j++] = behind + (uint64_t)c * GCH;
        if (j != nj) return fail(-5, "%s", "segment job count mismatch");
        polys.assign((size_t)nj * 624, 0);
        {
            std::vector<uint64_t> Js;
            std::vector<int> where;
            for (int q = 0; q < nj; q++) if (offs[q] >= (uint64_t)ahead) { Js.push_back(offs[q] - (uint64_t)ahead); where.push_back(q); }
            std::vector<uint32_t> tmp(Js.size() * 624);
            if (!mtjump::jump_polys(Js.data(), (int)Js.size(), tmp.data()))
                return fail(-5, "%s", "MT19937 characteristic polynomial could not be derived");
            for (size_t q = 0; q < Js.size(); q++) memcpy(&polys[(size_t)where[q] * 624], &tmp[q * 624], 624 * sizeof(uint32_t));
        }
        // positions by class, every class padded to whole blocks of 16
        std::vector<uint16_t> pos((size_t)n_groups * XRT_JG_POS_CAP, (uint16_t)XRT_JG_ZERO);
        std::vector<uint32_t> cls((size_t)n_groups * 257, 0u);
        {
            std::vector<uint8_t> key(19937);
            std::vector<uint32_t> count(256);
            for (int g = 0; g < n_groups; g++) {
                std::fill(key.begin(), key.end(), (uint8_t)0);
                for (int i = 0; i < XRT_JG && XRT_JG * g + i < nj; i++) {
                    const uint32_t* poly = &polys[(size_t)(XRT_JG * g + i) * 624];
                    for (int jj = 0; jj < 19937; jj++) if ((poly[jj >> 5] >> (jj & 31)) & 1u) key[jj] |= (uint8_t)(1u << i);
                }
                std::fill(count.begin(), count.end(), 0u);
                for (int jj = 0; jj < 19937; jj++) count[key[jj]]++;
                uint32_t* c = &cls[(size_t)g * 257];
                uint32_t blk = 0;
                c[0] = 0;
                for (int k = 1; k < 256; k++) { c[k] = blk; blk += (count[k] + 15u) / 16u; }
                c[256] = blk;
                if ((size_t)blk * 16 > XRT_JG_POS_CAP) return fail(-5, "%s", "jump position table overflow");
                std::vector<uint32_t> at(256);
                for (int k = 1; k < 256; k++) at[k] = c[k] * 16u;
                uint16_t* pp = &pos[(size_t)g * XRT_JG_POS_CAP];
                for (int jj = 0; jj < 19937; jj++) if (key[jj]) pp[at[key[jj]]++] = (uint16_t)jj;
            }
        }
        PlanCache c;
        c.N = N; c.L = L; c.S = S; c.nj = nj; c.used = ks.src.array_used; c.n_arrays = ks.src.n_arrays; c.ahead = ahead;
        c.n_ch = n_ch; c.gpairs = plan.gpairs; c.CH = CH; c.dev = dev_now;
        c.pos = nullptr; c.cls = nullptr; c.offs = nullptr;
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c.pos), pos.size() * sizeof(uint16_t)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c.cls), cls.size() * sizeof(uint32_t)));
        HIP_TRY(hipMalloc(reinterpret_cast<void**>(&c.offs), sizeof(uint64_t) * (size_t)nj));
        // (host temporaries: synchronous copies, once per plan)
        HIP_TRY(hipMemcpy(c.pos, pos.data(), pos.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c.cls, cls.data(), cls.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(c.offs, offs.data(), sizeof(uint64_t) * (size_t)nj, hipMemcpyHostToDevice));
        {
            // (cached plans are never freed: another thread may be about to launch with them; the cache is bounded
            //  instead, and a plan beyond the bound is released by this call once its launch is through)
            std::lock_guard<std::mutex> lock(plan_mu);
            if (plans.size() < 256) plans.push_back(c); else plan_once = true;
        }
        hit = c;
        }
        KJumpGroups jobs;
        jobs.pos = hit.pos; jobs.cls = hit.cls; jobs.offsets = hit.offs; jobs.dst = dst; jobs.n_jobs = nj; jobs.n_runs = n_runs;
        jobs.ahead = (uint64_t)ahead;
        const size_t jl = sizeof(uint32_t) * ((size_t)XRT_JG_S_WORDS + XRT_JG * XRT_JG_OUT + 264);
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(xrt_jump_groups_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)jl));
        // One workgroup per CU (105 KB of LDS), each with an equal share of the work items (run, group, share of the
        // group's classes); the groups are cut into 1 - 8 shares so that a workgroup gets at least two items
        {
            static thread_local int c_dev2 = -1, c_cus2 = 256;
            if (c_dev2 != dev_now) { HIP_TRY(hipDeviceGetAttribute(&c_cus2, hipDeviceAttributeMultiprocessorCount, dev_now)); c_dev2 = dev_now; }
            const int shares = jump_shares(n_runs, nj);
            jobs.n_shares = shares; jobs.pad = 0;
            jobs.partial = reinterpret_cast<uint32_t*>(d_polys);       // (behind the destination heads)
            const long long items = (long long)n_runs * n_groups * shares;
            const int grid = (int)(items < c_cus2 ? items : c_cus2);
            hipLaunchKernelGGL(xrt_jump_groups_kernel, dim3((unsigned)grid), dim3(XRT_JUMP_THREADS), jl, stream, streams, jobs);
            HIP_TRY(hipGetLastError());
            if (shares > 1) {
                hipLaunchKernelGGL(xrt_jump_combine_kernel, dim3((unsigned)(n_runs * nj)), dim3(640), 0, stream, streams, jobs);
                HIP_TRY(hipGetLastError());
            }
            if (plan_once) {
                HIP_TRY(hipStreamSynchronize(stream));
                (void)hipFree(hit.pos); (void)hipFree(hit.cls); (void)hipFree(hit.offs);
            }
        }
        a.streams = streams; a.heads = dst; a.n_runs = n_runs; a.n_src_heads = nh;
        a.run_counter = reinterpret_cast<uint32_t*>(ws);
        a.progress = env_on("XICSRT_NO_PRIORITY_FEEDBACK") ? nullptr : reinterpret_cast<unsigned long long*>(ws + 32);
        a.n_seg = S; a.seg_len = L; a.unit_count = d_cnt; a.chunk_heads = dst + (size_t)S * nh; a.chunk_words = CH;
        a.run_stride = nj;
        a.n_sub = M; a.sub_len = plan.sub_len;
        // one pass (candidates parked in HBM behind everything else) when the workspace holds them
        const size_t cand_need = cand_bytes(sc, n_runs, XRT_CAND_BUDGET);
        const bool one_pass = be >= 0 && (S * M > 1 || plan.mesh_split) && cand_need > 0 && ws_base_bytes(sc, n_runs) + cand_need <= ws_bytes;
        const bool split = one_pass && plan.mesh_split;
        if (split) {
            char* cb = ws + ws_base_bytes(sc, n_runs);
            a.cand = reinterpret_cast<double*>(cb);
            a.cand_id = nullptr;
            a.cand_aux = reinterpret_cast<uint32_t*>(cb + mesh_split_off_aux(sc, n_runs));
            a.batch_alive = reinterpret_cast<uint32_t*>(cb + mesh_split_off_batch_alive(sc, n_runs));
            a.unit_alive = reinterpret_cast<uint32_t*>(cb + mesh_split_off_unit_alive(sc, n_runs));
            a.slow_q = reinterpret_cast<uint32_t*>(cb + mesh_split_off_slow_q(sc, n_runs, plan));
            a.slow_cnt = reinterpret_cast<uint32_t*>(cb + mesh_split_off_slow_cnt(sc, n_runs, plan));
            a.slow_q2 = reinterpret_cast<uint32_t*>(cb + mesh_split_off_slow_q2(sc, n_runs, plan));
            a.unit_slow = reinterpret_cast<uint32_t*>(cb + mesh_split_off_unit_slow(sc, n_runs, plan));
            a.slow_pass = 0;
            a.split_interp = (sc->optics[be].mesh && sc->optics[be].mesh->interpolate) ? 1u : 0u;
            {   // the mesh straight behind a source without extent: one origin for every parked ray (KArgs.unit_o)
                const bool point = split_shared_origin(sc);
                a.unit_o = point ? reinterpret_cast<double*>(cb + mesh_split_off_unit_o(sc, n_runs, plan)) : nullptr;
                a.split_lean = (point && !a.split_interp) ? 1u : 0u;
            }
            a.cand_cap = (int64_t)cand_capacity(sc);
            a.unit_flag = d_cnt;
            HIP_TRY(hipMemsetAsync(d_cnt, 0, sizeof(uint32_t) * (size_t)n_runs * (size_t)S * (size_t)M, stream));
            HIP_TRY(hipMemsetAsync(a.unit_alive, 0, sizeof(uint32_t) * (size_t)n_runs * (size_t)S * (size_t)M, stream));
            HIP_TRY(hipMemsetAsync(a.unit_slow, 0, sizeof(uint32_t) * (size_t)n_runs * (size_t)S * (size_t)M, stream));
            g_paths |= XRT_PATH_ONE_PASS | XRT_PATH_MESH_SPLIT;
        } else
        if (one_pass && ws_base_bytes(sc, n_runs) + al256((size_t)n_runs * cand_capacity(sc) * 64) + 256 > ws_bytes)
            return fail(-4, "%s", "workspace too small for the parked candidates of the one-pass route");
        else
        if (one_pass) {
            const size_t cap = cand_capacity(sc);
            char* cb = ws + ws_base_bytes(sc, n_runs);
            a.cand = reinterpret_cast<double*>(cb);
            a.cand_id = reinterpret_cast<uint32_t*>(cb + (size_t)n_runs * cap * 56);
            a.cand_aux = a.cand_id + (size_t)n_runs * cap;
            a.cand_cap = (int64_t)cap;
            a.unit_flag = d_cnt;
            HIP_TRY(hipMemsetAsync(d_cnt, 0, sizeof(uint32_t) * (size_t)n_runs * (size_t)S * (size_t)M, stream));
            g_paths |= XRT_PATH_ONE_PASS;
        }
        a.dbg = nullptr;
#ifdef XRT_UNIT_CLOCKS      // development builds only (XRT_EXTRA_FLAGS=-DXRT_UNIT_CLOCKS=1): a device buffer [units][8] of phase stamps
        if (const char* e = getenv("XICSRT_UNIT_CLOCKS")) a.dbg = reinterpret_cast<unsigned long long*>(strtoull(e, nullptr, 0));
#endif

        a.wl_array = nullptr; a.base_words = nullptr;
        if (n_gch > 0) {
            g_paths |= XRT_PATH_GAUSS_PREPARED;
            // np.random.normal wavelengths of every run as an array (count pass only when a run has several chunks)
            KGauss gk;
            memset(&gk, 0, sizeof(gk));
            gk.heads = dst + (size_t)S * nh + n_ch; gk.run_stride = nj; gk.pairs_per_chunk = plan.gpairs; gk.n_values = N;
            gk.n_chunks = n_gch; gk.n_runs = n_runs; gk.acc = d_gacc; gk.wl = d_wl; gk.end_words = d_gend;
            gk.loc = sc->source.wavelength; gk.sigma = sc->source.wl_a;
            gk.counter = reinterpret_cast<uint32_t*>(ws); gk.flags = reinterpret_cast<uint32_t*>(ws + XRT_WS_STATUS_BYTE);
            int gunits = n_runs * n_gch, ggrid = gunits < 1024 ? gunits : 1024;
            for (int mode = (n_gch > 1 ? 1 : 2); mode <= 2; mode++) {
                gk.mode = mode;
                HIP_TRY(hipMemsetAsync(ws, 0, 64, stream));
                hipLaunchKernelGGL(xrt_gauss_kernel, dim3(ggrid), dim3(XRT_TILE), 0, stream, gk);
                HIP_TRY(hipGetLastError());
            }
            a.wl_array = d_wl; a.base_words = d_gend;
        }
        int variant = needs_ext(sc) ? 2 : (needs_full(sc) ? 1 : 0);
        if (variant == 0 && n_gch > 0) variant = 3;         // lean geometry, wavelength per ray from the prepared array
        if (variant == 0 && lds_bins_wanted(sc, a.images != nullptr, hist)) variant = 4;
        const size_t lds = plan_queue(ks, nh, variant == 2, hist, &a, variant == 4 ? (uint32_t)((sc->image_bins + 1) / 2) : 0u, split ? 4 : 0);
        if (variant == 4) g_paths |= XRT_PATH_LDS_BINS;
        if (split) {
            // first phase, the rest of the mesh intersection for every parked ray, second phase
            a.mode = 2;
            HIP_TRY(hipMemsetAsync(ws, 0, 64, stream));
            // (the first phase needs the record buffer for the compactions of one tile only, and room for the direction grid)
            KArgs a3 = a;
            a3.qcap = XRT_TILE; a3.bragg_batch = 128u;
            a3.dir_lds_bytes = (uint32_t)(ks.opt[be].mesh_dir_bytes > 0 ? ks.opt[be].mesh_dir_bytes : 0);
            const bool has_wl3 = !(ks.src.wavelength_dist == XRT_WL_CONST && !ks.src.has_velocity);
            const size_t lds3 = lds_bytes(nh, true, false, has_wl3, XRT_TILE) + a3.dir_lds_bytes;
            int st = launch_variant<false, 2, 3>(device_scene(ws), a3, n_runs, lds3, stream);
            if (st) return st;
            const uint32_t bpu = (uint32_t)(plan.sub_len / XRT_TILE);
            const unsigned long long blocks = (unsigned long long)n_runs * (unsigned long long)(S * M) * bpu;
            if (blocks > 0x7fffffffull) return fail(-5, "%s", "mesh split: too many blocks");
            int ti = -1;
            if (timing_on && timing_n < TIMING_MAX) {
                ti = timing_n++;
                HIP_TRY(hipEventCreate(&timing_ev[ti][0]));
                HIP_TRY(hipEventCreate(&timing_ev[ti][1]));
                HIP_TRY(hipEventRecord(timing_ev[ti][0], stream));
            }
            KArgs am = a;
            const bool ct = sc->optics[be].mesh && sc->optics[be].mesh->interpolate;
            const int lds_m = ks.opt[be].mesh_lds_bytes, lds_ct = ct ? ks.opt[be].mesh_ct_lds_bytes : 0;
            static thread_local int c_dev3 = -1, c_cus3 = 256;
            if (c_dev3 != dev_now) { HIP_TRY(hipDeviceGetAttribute(&c_cus3, hipDeviceAttributeMultiprocessorCount, dev_now)); c_dev3 = dev_now; }
            // (the kernels with tables in LDS: one workgroup of 1024 per CU, blocks of 1024 rays round robin)
            const uint32_t bpu4 = (bpu + 3u) / 4u;
            const unsigned long long items = (unsigned long long)n_runs * (unsigned long long)(S * M) * bpu4;
            const unsigned grid = (unsigned)(items < (unsigned long long)c_cus3 ? items : (unsigned long long)c_cus3);
            const uint32_t n_units = (uint32_t)(n_runs * S * M);
            // the work of the launches with tables in LDS: the blocks of 1024 records that hold rays (KArgs.items)
            uint4* d_items = nullptr;
            uint32_t* d_nitems = nullptr;
            const int lds_st = ks.opt[be].mesh_star_lds_bytes;
            const bool fans_lds = lds_m > 0 && lds_st > 0, fans_glb = !fans_lds && lds_st != 0;
            if (lds_m > 0 || lds_ct > 0 || fans_glb) {
                char* cb = ws + ws_base_bytes(sc, n_runs);
                d_items = reinterpret_cast<uint4*>(cb + mesh_split_off_items(sc, n_runs, plan));
                d_nitems = reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(d_items) + 2 * al256(16 * mesh_split_items_max(sc, n_runs, plan)));
                hipLaunchKernelGGL(xrt_mesh_items_kernel, dim3(1), dim3(1024), 0, stream, am.unit_flag, 1u, n_units, d_items, d_nitems);
                HIP_TRY(hipGetLastError());
                am.items = d_items; am.n_items = d_nitems;
            }
            // a mesh with fans: what one face of the nearest point's fan settles first, the launch below for the rays left over
            if (fans_lds || fans_glb) {
                if (fans_glb) {
                    // (a mesh beyond the LDS: the same launch over the tables in global memory)
                    if (ct) hipLaunchKernelGGL((xrt_mesh_star_kernel<true>), dim3(grid), dim3(XRT_MESH_LDS_THREADS), 0, stream, device_scene(ws), am, be);
                    else    hipLaunchKernelGGL((xrt_mesh_star_kernel<false>), dim3(grid), dim3(XRT_MESH_LDS_THREADS), 0, stream, device_scene(ws), am, be);
                } else
                if (ct) {
                    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(xrt_mesh_star_lds_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_st));
                    hipLaunchKernelGGL((xrt_mesh_star_lds_kernel<true>), dim3(grid), dim3(XRT_MESH_LDS_THREADS), (size_t)lds_st, stream, device_scene(ws), am, be);
                } else {
                    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(xrt_mesh_star_lds_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_st));
                    hipLaunchKernelGGL((xrt_mesh_star_lds_kernel<false>), dim3(grid), dim3(XRT_MESH_LDS_THREADS), (size_t)lds_st, stream, device_scene(ws), am, be);
                }
                HIP_TRY(hipGetLastError());
                hipLaunchKernelGGL(xrt_mesh_slow_compact_kernel, dim3(n_units < 2048u ? n_units : 2048u), dim3(256), 0, stream, device_scene(ws), am, n_units);
                HIP_TRY(hipGetLastError());
                am.slow_pass = 1;
                am.slow_q = am.slow_q2;
                // (the list walk's own items: the blocks of the units' lists of rays left over)
                uint4* d_items2 = reinterpret_cast<uint4*>(reinterpret_cast<char*>(d_items) + al256(16 * mesh_split_items_max(sc, n_runs, plan)));
                hipLaunchKernelGGL(xrt_mesh_items_kernel, dim3(1), dim3(1024), 0, stream, am.unit_slow, 0u, n_units, d_items2, d_nitems + 1);
                HIP_TRY(hipGetLastError());
                am.items = d_items2; am.n_items = d_nitems + 1;
                g_paths |= XRT_PATH_MESH_FANS;
            }
            // up to the hit face (an interpolated mesh: the rest is the next launch's)
            if (lds_m > 0) {
                if (ct) {
                    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(xrt_mesh_rest_lds_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_m));
                    hipLaunchKernelGGL((xrt_mesh_rest_lds_kernel<true>), dim3(grid), dim3(XRT_MESH_LDS_THREADS), (size_t)lds_m, stream, device_scene(ws), am, be);
                } else {
                    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(xrt_mesh_rest_lds_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_m));
                    hipLaunchKernelGGL((xrt_mesh_rest_lds_kernel<false>), dim3(grid), dim3(XRT_MESH_LDS_THREADS), (size_t)lds_m, stream, device_scene(ws), am, be);
                }
            } else if (ct)
                hipLaunchKernelGGL((xrt_mesh_rest_kernel<true>), dim3((unsigned)blocks), dim3(XRT_TILE), 0, stream, device_scene(ws), am, be, bpu);
            else
                hipLaunchKernelGGL((xrt_mesh_rest_kernel<false>), dim3((unsigned)blocks), dim3(XRT_TILE), 0, stream, device_scene(ws), am, be, bpu);
            HIP_TRY(hipGetLastError());
            am.slow_pass = 0;
            am.items = d_items; am.n_items = d_nitems;
            // interpolation, bounds, counts
            if (ct) {
                if (lds_ct > 0) {
                    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(xrt_mesh_ct_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_ct));
                    hipLaunchKernelGGL(xrt_mesh_ct_lds_kernel, dim3(grid), dim3(XRT_MESH_LDS_THREADS), (size_t)lds_ct, stream, device_scene(ws), am, be);
                } else
                    hipLaunchKernelGGL(xrt_mesh_ct_kernel, dim3((unsigned)blocks), dim3(XRT_TILE), 0, stream, device_scene(ws), am, be, bpu);
            }
            HIP_TRY(hipGetLastError());
            // the rays every unit has left: the sums of its counts per 64 records
            hipLaunchKernelGGL(xrt_mesh_unit_alive_kernel, dim3(n_units < 2048u ? n_units : 2048u), dim3(256), 0, stream, device_scene(ws), am, n_units);
            HIP_TRY(hipGetLastError());
            if (ti >= 0) HIP_TRY(hipEventRecord(timing_ev[ti][1], stream));
            HIP_TRY(hipMemsetAsync(ws, 0, 64, stream));
            st = launch_variant<false, 2, 4>(device_scene(ws), a, n_runs, lds, stream);
            return st;
        }
        // (a run that is one unit has nothing in front of it: no count pass)
        for (int mode = ((be >= 0 && S * M > 1 && !one_pass) ? 1 : 2); mode <= 2; mode++) {
            a.mode = mode;
            HIP_TRY(hipMemsetAsync(ws, 0, 64, stream));
            int st;
#define XRT_LAUNCH_SEG(SEGV)                                                                                                     \
            if (variant == 4) st = launch_variant<false, 4, SEGV>(device_scene(ws), a, n_runs, lds, stream);                     \
            else if (hist) st = variant == 3 ? launch_variant<true, 3, SEGV>(device_scene(ws), a, n_runs, lds, stream)           \
                         : variant == 2 ? launch_variant<true, 2, SEGV>(device_scene(ws), a, n_runs, lds, stream)                \
                         : variant == 1 ? launch_variant<true, 1, SEGV>(device_scene(ws), a, n_runs, lds, stream)                \
                                        : launch_variant<true, 0, SEGV>(device_scene(ws), a, n_runs, lds, stream);               \
            else      st = variant == 3 ? launch_variant<false, 3, SEGV>(device_scene(ws), a, n_runs, lds, stream)               \
                         : variant == 2 ? launch_variant<false, 2, SEGV>(device_scene(ws), a, n_runs, lds, stream)               \
                         : variant == 1 ? launch_variant<false, 1, SEGV>(device_scene(ws), a, n_runs, lds, stream)               \
                                        : launch_variant<false, 0, SEGV>(device_scene(ws), a, n_runs, lds, stream);
            if (one_pass) { XRT_LAUNCH_SEG(2) } else { XRT_LAUNCH_SEG(1) }
#undef XRT_LAUNCH_SEG
            if (st) return st;
        }
        return 0;
    }
    g_paths |= XRT_PATH_FUSED;
    {
        const int st = position_heads(sc, ks, ws, n_runs, nh, N, canonical, streams, heads, stream);
        if (st) return st;
    }
    a.streams = streams; a.heads = heads; a.n_runs = n_runs; a.n_src_heads = nh;
    a.run_counter = reinterpret_cast<uint32_t*>(ws);
    a.progress = env_on("XICSRT_NO_PRIORITY_FEEDBACK") ? nullptr : reinterpret_cast<unsigned long long*>(ws + 32);
    HIP_TRY(hipMemsetAsync(ws, 0, 64, stream));
    int variant = needs_ext(sc) ? 2 : (needs_full(sc) ? 1 : 0);
    if (variant == 0 && lds_bins_wanted(sc, a.images != nullptr, hist)) variant = 4;
    const size_t lds = plan_queue(ks, nh, variant == 2, hist, &a, variant == 4 ? (uint32_t)((sc->image_bins + 1) / 2) : 0u);
    if (variant == 4) {
        g_paths |= XRT_PATH_LDS_BINS;
        return launch_variant<false, 4>(device_scene(ws), a, n_runs, lds, stream, true);
    }
    if (hist) {
        if (variant == 2) return launch_variant<true, 2>(device_scene(ws), a, n_runs, lds, stream);
        return variant == 1 ? launch_variant<true, 1>(device_scene(ws), a, n_runs, lds, stream) : launch_variant<true, 0>(device_scene(ws), a, n_runs, lds, stream);
    }
    if (variant == 2) return launch_variant<false, 2>(device_scene(ws), a, n_runs, lds, stream, true);
    return variant == 1 ? launch_variant<false, 1>(device_scene(ws), a, n_runs, lds, stream, true) : launch_variant<false, 0>(device_scene(ws), a, n_runs, lds, stream, true);
}

// diagnostic: g(t) = t^J mod phi(t) as 624 words (bit j of word j/32 = g_j); host only
extern "C" int xrt_mt_jump_poly(uint64_t J, uint32_t* out624)
{
    if (!out624) return fail(-1, "%s", "NULL argument");
    if (!mtjump::jump_poly(J, out624)) return fail(-5, "%s", "MT19937 characteristic polynomial could not be derived");
    return 0;
}

static int trace_runs(const xrt_scene_t* sc, const uint32_t* seeds, int32_t n_runs, int32_t n_iter,
                      uint64_t* num_out, uint64_t* images, void* workspace, size_t ws_bytes, void* stream_, bool clear_status)
{
    int st = 0;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    char* ws = reinterpret_cast<char*>(workspace);
    uint32_t* d_seeds = reinterpret_cast<uint32_t*>(ws + ws_off_seeds(sc));
    KStream* streams = reinterpret_cast<KStream*>(ws + ws_off_streams(sc, n_runs));
    if (n_runs <= 512) {
        // (few seeds: through the kernel-argument path like the scene, no staged copy from pageable memory)
        KBlob b;
        memcpy(b.w, seeds, sizeof(uint32_t) * (size_t)n_runs);
        hipLaunchKernelGGL(xrt_put_kernel, dim3(1), dim3(256), 0, stream, d_seeds, b, n_runs);
        HIP_TRY(hipGetLastError());
    } else
    HIP_TRY(hipMemcpyAsync(d_seeds, seeds, sizeof(uint32_t) * (size_t)n_runs, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(xrt_seed_kernel, dim3(n_runs), dim3(64), 0, stream, d_seeds, streams,
                       reinterpret_cast<KState*>(ws + ws_off_gauss(sc, n_runs)), n_runs);
    HIP_TRY(hipGetLastError());
    st = upload_tables(sc, ws, stream);
    if (st) return st;
    if (clear_status) HIP_TRY(hipMemsetAsync(ws + XRT_WS_STATUS_BYTE, 0, 64, stream));           // status word
    KScene ks;
    build_kscene(sc, ws, &ks);
    st = upload_meshes(sc, ws, n_runs, &ks, stream);
    if (st) return st;
    st = upload_plasma(sc, ws, n_runs, &ks, stream);
    if (st) return st;
    st = upload_scene(ks, ws, stream);
    if (st) return st;
    KArgs a;
    memset(&a, 0, sizeof(a));
    a.num_out = reinterpret_cast<unsigned long long*>(num_out);
    a.images = reinterpret_cast<unsigned long long*>(images);
    st = image_rep_begin(sc, ws, n_runs, &a, stream);
    if (st) return st;
    // iterations share each run's stream (xicsrt_raytrace.py:153): the stream head left by
    // iteration i is where iteration i+1 starts
    for (int it = 0; it < n_iter; it++) {
        st = run_iteration(sc, ks, ws, ws_bytes, a, n_runs, false, (int)XRT_AHEAD, stream);
        if (st) return st;
    }
    return image_rep_end(sc, ws, n_runs, a, a.images, stream);
}

extern "C" int xrt_trace(const xrt_scene_t* sc, const uint32_t* seeds, int32_t n_runs, int32_t n_iter,
                         uint64_t* num_out, uint64_t* images, void* workspace, size_t workspace_bytes, void* stream_)
{
    int st = xrt_scene_check(sc);
    if (st) return st;
    if (n_runs <= 0 || n_iter <= 0) return 0;
    if (!seeds || !num_out || !workspace) return fail(-1, "%s", "NULL argument");
    // (the layout of this call; xrt_workspace_bytes() also covers the layouts of the possible second pass, which is
    //  checked when its run count is known -- a sweep over all of them here would cost a millisecond per call)
    if (workspace_bytes < ws_base_bytes(sc, n_runs)) return fail(-4, "%s", "workspace too small");
    {   // a mesh crystal whose parked rays would not fit the budget: the runs in equal batches (mesh_batch_runs)
        const int per = mesh_batch_runs(sc, n_runs);
        if (per < n_runs) {
            t_tail = -1;
            for (int r0 = 0; r0 < n_runs; r0 += per) {
                const int nb = (n_runs - r0) < per ? (n_runs - r0) : per;
                if (workspace_bytes < ws_base_bytes(sc, nb) + cand_bytes(sc, nb, XRT_CAND_BUDGET)) return fail(-4, "%s", "workspace too small for a batch of runs");
                st = trace_runs(sc, seeds + r0, nb, n_iter, num_out, images, workspace, workspace_bytes, stream_, r0 == 0);
                if (st) return st;
            }
            return 0;
        }
    }
    // (see t_tail: the unsegmented launch may leave the last < 256 runs to a second pass over the same workspace)
    // (a second pass costs about a millisecond of set-up: only where a run takes longer than that)
    t_tail = tail_split_possible(sc) ? 0 : -1;
    st = trace_runs(sc, seeds, n_runs, n_iter, num_out, images, workspace, workspace_bytes, stream_, true);
    const int tail = t_tail;
    t_tail = -1;
    if (st || tail <= 0) return st;
    // (the tail's layout is part of xrt_workspace_bytes(sc, n_runs); checked all the same)
    if (workspace_bytes < ws_base_bytes(sc, tail)) return fail(-4, "%s", "workspace too small for the second pass over the last runs");
    return trace_runs(sc, seeds + (n_runs - tail), tail, n_iter, num_out, images, workspace, workspace_bytes, stream_, false);
}

extern "C" int xrt_trace_history(const xrt_scene_t* sc, const xrt_rng_state_t* state_in,
                                 uint64_t* num_out, uint64_t* images, double* rays, uint8_t* mask, void* state_out,
                                 void* workspace, size_t workspace_bytes, void* stream_)
{
    int st = xrt_scene_check(sc);
    if (st) return st;
    if (!state_in || !num_out || !workspace || !rays || !mask) return fail(-1, "%s", "NULL argument");
    if (state_in->pos < 0 || state_in->pos > 624) return fail(-2, "%s", "generator position out of range");
    if (workspace_bytes < xrt_workspace_bytes(sc, 1)) return fail(-4, "%s", "workspace too small");
    static_assert(sizeof(KState) == sizeof(xrt_rng_state_t), "state layout");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    char* ws = reinterpret_cast<char*>(workspace);
    KState* d_state = reinterpret_cast<KState*>(ws + ws_off_state(sc));
    KStream* streams = reinterpret_cast<KStream*>(ws + ws_off_streams(sc, 1));
    HIP_TRY(hipMemcpyAsync(d_state, state_in, sizeof(KState), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(xrt_import_state_kernel, dim3(1), dim3(64), 0, stream, d_state, streams,
                       reinterpret_cast<KState*>(ws + ws_off_gauss(sc, 1)));
    HIP_TRY(hipGetLastError());
    st = upload_tables(sc, ws, stream);
    if (st) return st;
    HIP_TRY(hipMemsetAsync(ws + XRT_WS_STATUS_BYTE, 0, 64, stream));           // status word
    KScene ks;
    build_kscene(sc, ws, &ks);
    st = upload_meshes(sc, ws, 1, &ks, stream);
    if (st) return st;
    st = upload_plasma(sc, ws, 1, &ks, stream);
    if (st) return st;
    st = upload_scene(ks, ws, stream);
    if (st) return st;
    KArgs a;
    memset(&a, 0, sizeof(a));
    a.num_out = reinterpret_cast<unsigned long long*>(num_out);
    a.images = reinterpret_cast<unsigned long long*>(images);
    a.hist = rays; a.hmask = mask;
    int ahead = 0;
    // np.random.normal with a cached second value pending: the array cannot be prepared in pairs
    const bool force_staged = sc->source.wavelength_dist == XRT_WL_NORMAL && state_in->has_gauss != 0;
    if (seg_active(plan_segments(sc, 1)) && !force_staged) {
        // segmented history run: the imported state gets the fixed lead the jump polynomials are made for
        hipLaunchKernelGGL(xrt_advance_kernel, dim3(1), dim3(64), 0, stream, streams);
        HIP_TRY(hipGetLastError());
        ahead = 624;
    }
    st = image_rep_begin(sc, ws, 1, &a, stream);
    if (st) return st;
    st = run_iteration(sc, ks, ws, workspace_bytes, a, 1, true, ahead, stream, force_staged);
    if (st) return st;
    st = image_rep_end(sc, ws, 1, a, a.images, stream);
    if (st) return st;
    if (state_out) {
        hipLaunchKernelGGL(xrt_export_state_kernel, dim3(1), dim3(64), 0, stream, streams,
                           reinterpret_cast<const KState*>(ws + ws_off_gauss(sc, 1)),
                           reinterpret_cast<KState*>(state_out));
        HIP_TRY(hipGetLastError());
    }
    return 0;
}

__global__ void xrt_image_kernel(const KOptic op, int64_t n, const double* rays, const uint8_t* mask, unsigned long long* images)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !mask[i]) return;
    V3 X;
    X.x = rays[0 * n + i]; X.y = rays[1 * n + i]; X.z = rays[2 * n + i];
    image_hit(op, X, images);
}

extern "C" int xrt_make_image(const xrt_optic_t* optic, int64_t n, const double* rays, const uint8_t* mask,
                              uint64_t* images, void* stream_)
{
    if (!optic || !images || (n > 0 && (!rays || !mask))) return fail(-1, "%s", "NULL argument");
    if (!(optic->flags & XRT_F_IMAGE) || optic->pixel_nx <= 0 || optic->pixel_ny <= 0) return fail(-2, "%s", "optic makes no image");
    if (n <= 0) return 0;
    KOptic q;
    memset(&q, 0, sizeof(q));
    for (int i = 0; i < 3; i++) q.origin[i] = optic->origin[i];
    for (int i = 0; i < 9; i++) q.R[i] = optic->orientation[i];
    q.flags = optic->flags;
    q.pixel_size = optic->pixel_size; q.pixel_xoff = optic->pixel_xoff; q.pixel_yoff = optic->pixel_yoff;
    q.pixel_nx = optic->pixel_nx; q.pixel_ny = optic->pixel_ny; q.image_offset = optic->image_offset;
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    hipLaunchKernelGGL(xrt_image_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, q, n, rays, mask,
                       reinterpret_cast<unsigned long long*>(images));
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- TraceObject.intersect / check_bounds / interact as separate calls (optics/_TraceObject.py:157-180) --------
// One optic, a caller's ray array (device, component-major [8][n] as xrt_source_t.ext_rays); xloc and norm
// are [3][n].  The same device functions as in the propagation kernels, one ray per thread.
struct KStep { KOptic op; xrt_aperture_t ap[XRT_MAX_APERTURES]; };

__global__ void xrt_step_intersect_kernel(const KStep a, int64_t n, const double* rays, const uint8_t* mask_in,
                                          double* xloc, double* norm, uint8_t* mask_out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    KOptic op = a.op;
    op.apertures = a.ap;
    const double nan = __builtin_nan("");
    V3 X, nr;
    X.x = X.y = X.z = nan; nr = X;
    if (op.shape == XRT_SHAPE_TORUS) nr.x = nr.y = nr.z = 0.0;      // ShapeTorus.intersect_normal starts from zeros (:196)
    bool m = mask_in[i] != 0;
    if (m) {
        Ray ray;
        ray.o.x = rays[0 * n + i]; ray.o.y = rays[1 * n + i]; ray.o.z = rays[2 * n + i];
        ray.d.x = rays[3 * n + i]; ray.d.y = rays[4 * n + i]; ray.d.z = rays[5 * n + i];
        ray.wl = rays[6 * n + i];
        V3 P;
        m = intersect_point<true>(op, ray, P, false, PointPre());
        if (m) { X = P; nr = surface_normal<true>(op, P); }
    }
    xloc[0 * n + i] = X.x; xloc[1 * n + i] = X.y; xloc[2 * n + i] = X.z;
    norm[0 * n + i] = nr.x; norm[1 * n + i] = nr.y; norm[2 * n + i] = nr.z;
    mask_out[i] = m ? 1 : 0;
}

__global__ void xrt_step_bounds_kernel(const KStep a, int64_t n, const double* xloc, uint8_t* mask)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !mask[i]) return;
    KOptic op = a.op;
    op.apertures = a.ap;
    V3 X; X.x = xloc[0 * n + i]; X.y = xloc[1 * n + i]; X.z = xloc[2 * n + i];
    if (!check_bounds<true>(op, X)) mask[i] = 0;
}

__global__ void xrt_step_interact_kernel(const KStep a, int64_t n, double* rays, const double* xloc, const double* norm,
                                         uint8_t* mask, const double* test)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const KOptic& op = a.op;
    // InteractObject / InteractMirror.reflect_vectors (optics/_InteractMirror.py:29-42): O[:] = xloc[:] for every ray
    rays[0 * n + i] = xloc[0 * n + i]; rays[1 * n + i] = xloc[1 * n + i]; rays[2 * n + i] = xloc[2 * n + i];
    bool m = mask[i] != 0;
    if (m && op.interact != XRT_INTERACT_NONE) {
        Ray ray;
        ray.d.x = rays[3 * n + i]; ray.d.y = rays[4 * n + i]; ray.d.z = rays[5 * n + i];
        ray.wl = rays[6 * n + i];
        V3 nr; nr.x = norm[0 * n + i]; nr.y = norm[1 * n + i]; nr.z = norm[2 * n + i];
        if (op.interact == XRT_INTERACT_CRYSTAL && (op.flags & XRT_F_CHECK_BRAGG))
            m = bragg_accept(op, ray, nr, 0.0 + (1.0 - 0.0) * test[i], false, 0.0);      // InteractCrystal.angle_check (:126-145)
        if (m) {
            const double dt = dot_e(ray.d, nr);
            rays[3 * n + i] = ray.d.x - 2.0 * (dt * nr.x);
            rays[4 * n + i] = ray.d.y - 2.0 * (dt * nr.y);
            rays[5 * n + i] = ray.d.z - 2.0 * (dt * nr.z);
        }
    }
    mask[i] = m ? 1 : 0;
}

static int make_step(const xrt_optic_t* optic, KStep* k)
{
    if (!optic) return fail(-1, "%s", "NULL argument");
    if (optic->shape < XRT_SHAPE_PLANE || optic->shape > XRT_SHAPE_TORUS)
        return fail(-3, "%s", "step-wise calls are implemented for the analytic shapes (plane, sphere, cylinder, torus)");
    if (optic->n_apertures < 0 || optic->n_apertures > XRT_MAX_APERTURES) return fail(-2, "%s", "bad aperture count");
    fill_koptic(*optic, k->op);
    memset(k->ap, 0, sizeof(k->ap));
    for (int i = 0; i < optic->n_apertures; i++) k->ap[i] = optic->apertures[i];
    return 0;
}

extern "C" int xrt_optic_intersect(const xrt_optic_t* optic, int64_t n, const double* rays, const uint8_t* mask_in,
                                   double* xloc, double* norm, uint8_t* mask_out, void* stream_)
{
    static thread_local KStep k;
    int st = make_step(optic, &k);
    if (st) return st;
    if (n <= 0) return 0;
    if (!rays || !mask_in || !xloc || !norm || !mask_out) return fail(-1, "%s", "NULL argument");
    hipLaunchKernelGGL(xrt_step_intersect_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_),
                       k, n, rays, mask_in, xloc, norm, mask_out);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int xrt_optic_check_bounds(const xrt_optic_t* optic, int64_t n, const double* xloc, uint8_t* mask, void* stream_)
{
    static thread_local KStep k;
    int st = make_step(optic, &k);
    if (st) return st;
    if (n <= 0) return 0;
    if (!xloc || !mask) return fail(-1, "%s", "NULL argument");
    hipLaunchKernelGGL(xrt_step_bounds_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_),
                       k, n, xloc, mask);
    HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int xrt_optic_interact(const xrt_optic_t* optic, int64_t n, double* rays, const double* xloc, const double* norm,
                                  uint8_t* mask, const double* test, void* stream_)
{
    static thread_local KStep k;
    int st = make_step(optic, &k);
    if (st) return st;
    if (optic->interact == XRT_INTERACT_MOSAIC)
        return fail(-3, "%s", "the mosaic interaction draws whole arrays per layer: use trace_global()");
    if (n <= 0) return 0;
    if (!rays || !xloc || !norm || !mask) return fail(-1, "%s", "NULL argument");
    if (optic->interact == XRT_INTERACT_CRYSTAL && (optic->flags & XRT_F_CHECK_BRAGG) && !test)
        return fail(-1, "%s", "a Bragg test needs one uniform deviate per ray");
    hipLaunchKernelGGL(xrt_step_interact_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_),
                       k, n, rays, xloc, norm, mask, test);
    HIP_TRY(hipGetLastError());
    return 0;
}

// Self-test of div3_rn against the division operator: num [3][n], den [n] (device); counts the quotients whose bits differ.
__global__ void xrt_selftest_div3_kernel(const double* num, const double* den, int64_t n, unsigned long long* bad)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V3 v; v.x = num[i]; v.y = num[n + i]; v.z = num[2 * n + i];
    const double m = den[i];
    const V3 q = div3_rn(v, m);
    const double r0 = v.x / m, r1 = v.y / m, r2 = v.z / m;
    int c = (__double_as_longlong(q.x) != __double_as_longlong(r0) && !(q.x != q.x && r0 != r0))
          + (__double_as_longlong(q.y) != __double_as_longlong(r1) && !(q.y != q.y && r1 != r1))
          + (__double_as_longlong(q.z) != __double_as_longlong(r2) && !(q.z != q.z && r2 != r2));
    // ... and of ct_div3 (xrt_mesh.inc) against / 3 on the same operands
    const double t0 = ct_div3_checked(v.x), t1 = ct_div3_checked(v.y), t2 = ct_div3_checked(m), u0 = v.x / 3, u1 = v.y / 3, u2 = m / 3;
    c += (__double_as_longlong(t0) != __double_as_longlong(u0) && !(t0 != t0 && u0 != u0))
       + (__double_as_longlong(t1) != __double_as_longlong(u1) && !(t1 != t1 && u1 != u1))
       + (__double_as_longlong(t2) != __double_as_longlong(u2) && !(t2 != t2 && u2 != u2));
    if (c) atomicAdd(bad, (unsigned long long)c);
}

extern "C" int xrt_selftest_div3(const double* num, const double* den, int64_t n, uint64_t* bad, void* stream_)
{
    if (!num || !den || !bad || n <= 0) return fail(-1, "%s", "bad argument");
    hipLaunchKernelGGL(xrt_selftest_div3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream_),
                       num, den, n, reinterpret_cast<unsigned long long*>(bad));
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- np.random.shuffle(np.arange(n))[:m] of numpy's legacy generator, on the host (xicsrt_raytrace.py:264-266: the
// reference shuffles the indices of ALL lost rays to keep `history_max_lost` of them).  The draws are sequential by
// contract -- Fisher-Yates from the back, random_interval(i) by masked rejection on 32-bit words
// (numpy/random/_legacy: legacy_random_interval via mt19937 next_uint32) -- but they do not depend on the array, so they
// are made 64 at a time and the cells they name are prefetched before the swaps: the walk is bound by the generator, not by
// cache misses.  `state`: key[624], pos, has_gauss, gauss in and out.
extern "C" int xrt_legacy_shuffle_head(xrt_rng_state_t* state, int64_t n, int64_t m, int64_t* out)
{
    if (!state || (m > 0 && !out) || n < 0 || m < 0) return fail(-1, "%s", "bad argument");
    if (state->pos < 0 || state->pos > 624) return fail(-2, "%s", "generator position out of range");
    if (n > 0xffffffffll) return fail(-3, "%s", "more than 2^32 - 1 indices");
    if (m > n) m = n;
    uint32_t* mt = state->key;
    int pos = state->pos;
    auto next32 = [&]() -> uint32_t {
        if (pos >= 624) {
            int k;
            for (k = 0; k < 624 - 397; k++) { const uint32_t y = (mt[k] & 0x80000000u) | (mt[k + 1] & 0x7fffffffu); mt[k] = mt[k + 397] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u); }
            for (; k < 623; k++) { const uint32_t y = (mt[k] & 0x80000000u) | (mt[k + 1] & 0x7fffffffu); mt[k] = mt[k + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u); }
            const uint32_t y = (mt[623] & 0x80000000u) | (mt[0] & 0x7fffffffu);
            mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            pos = 0;
        }
        uint32_t y = mt[pos++];
        y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18);
        return y;
    };
    if (n > 1 && m * 16 <= n) {
        // few indices wanted: keep the draws (streamed, 4 bytes each) and follow only the m wanted cells backwards in
        // time -- cell k of the result holds what stood at sigma_{n-1}(...sigma_2(sigma_1(k))) at the start, sigma_i the
        // swap (i j_i) -- with a bitmap of the followed cells (n/8 bytes, cache resident) in front of the map.
        std::unique_ptr<uint32_t[]> J(new (std::nothrow) uint32_t[(size_t)n]);     // not zeroed: every cell from 1 up is written
        if (!J) return fail(-4, "%s", "out of host memory");
        {   // the draws: whole blocks of 624 tempered words (loops the compiler vectorises), consumed with the mask of
            // the current power-of-two span of i held constant and the rejection folded into the index step
            uint32_t buf[624];
            int have = 0, used = 0;
            auto refill = [&]() {
                if (pos >= 624) {
                    auto tw = [](uint32_t u, uint32_t l, uint32_t far) -> uint32_t {
                        const uint32_t y = (u & 0x80000000u) | (l & 0x7fffffffu);
                        return far ^ (y >> 1) ^ ((0u - (y & 1u)) & 0x9908b0dfu);
                    };
                    for (int k = 0; k < 227; k++) mt[k] = tw(mt[k], mt[k + 1], mt[k + 397]);
                    for (int k = 227; k < 454; k++) mt[k] = tw(mt[k], mt[k + 1], mt[k - 227]);
                    for (int k = 454; k < 623; k++) mt[k] = tw(mt[k], mt[k + 1], mt[k - 227]);
                    mt[623] = tw(mt[623], mt[0], mt[396]);
                    pos = 0;
                }
                have = 624 - pos; used = 0;
                for (int k = 0; k < have; k++) {
                    uint32_t y = mt[pos + k];
                    y ^= (y >> 11); y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= (y >> 18);
                    buf[k] = y;
                }
            };
            int64_t i = n - 1;
            while (i >= 1) {
                const int64_t lo = (int64_t)1 << (63 - __builtin_clzll((unsigned long long)i));   // span [lo, 2 lo)
                const uint32_t mask = (uint32_t)(2 * lo - 1);
                while (i >= lo) {
                    if (used == have) { pos += used; refill(); }
                    int q = used;
                    int64_t ii = i;
                    for (; q < have && ii >= lo; q++) {
                        const uint32_t v = buf[q] & mask;
                        J[(size_t)ii] = v;
                        ii -= (v <= (uint32_t)ii);
                    }
                    used = q; i = ii;
                }
            }
            pos += used;
        }
        std::vector<uint64_t> bits((size_t)(n + 63) / 64, 0);
        std::unordered_map<uint32_t, uint32_t> cell;           // followed position -> index of the result
        cell.reserve((size_t)m * 2 + 16);
        for (int64_t k = 0; k < m; k++) { cell[(uint32_t)k] = (uint32_t)k; bits[(size_t)k >> 6] |= 1ull << (k & 63); }
        auto move = [&](uint32_t from, uint32_t to) {
            auto it = cell.find(from);
            const uint32_t k = it->second;
            cell.erase(it);
            cell[to] = k;
            bits[from >> 6] &= ~(1ull << (from & 63));
            bits[to >> 6] |= 1ull << (to & 63);
        };
        for (int64_t i = 1; i < m; i++) {                        // both cells of a swap can be followed ones
            const uint32_t j = J[(size_t)i];
            if (j == (uint32_t)i) continue;
            if ((bits[j >> 6] >> (j & 63)) & 1u) std::swap(cell[(uint32_t)i], cell[j]);
            else move((uint32_t)i, j);
        }
        for (int64_t i = (m > 1 ? m : 1); i < n; i++) {          // cell i itself is never a followed one from here on
            const uint32_t j = J[(size_t)i];
            if ((bits[j >> 6] >> (j & 63)) & 1u) move(j, (uint32_t)i);
        }
        for (const auto& kv : cell) out[kv.second] = (int64_t)kv.first;
        state->pos = pos;
        return 0;
    }
    std::vector<uint32_t> x((size_t)n);
    for (int64_t i = 0; i < n; i++) x[(size_t)i] = (uint32_t)i;
    const int B = 64;
    uint32_t js[B];
    int64_t i = n - 1;
    while (i >= 1) {
        const int cnt = (int)(i < B ? i : B);       // iterations i, i-1, ..., i-cnt+1 (all >= 1)
        for (int q = 0; q < cnt; q++) {
            const uint32_t mx = (uint32_t)(i - q);
            uint32_t mask = mx;
            mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
            uint32_t v;
            do { v = next32() & mask; } while (v > mx);
            js[q] = v;
            __builtin_prefetch(&x[v], 1, 0);
        }
        for (int q = 0; q < cnt; q++) {
            const size_t a = (size_t)(i - q), b = (size_t)js[q];
            const uint32_t t = x[a]; x[a] = x[b]; x[b] = t;
        }
        i -= cnt;
    }
    for (int64_t q = 0; q < m; q++) out[q] = (int64_t)x[(size_t)q];
    state->pos = pos;
    return 0;
}

// diagnostic: n such polynomials at once, as a plan's jump jobs get them (families of exponents through products); host only
extern "C" int xrt_mt_jump_polys(const uint64_t* J, int32_t n, uint32_t* out)
{
    if (!J || !out || n < 0) return fail(-1, "%s", "NULL argument");
    if (!mtjump::jump_polys(J, n, out)) return fail(-5, "%s", "MT19937 characteristic polynomial could not be derived");
    return 0;
}

extern "C" uint32_t xrt_last_path(int32_t reset)
{
    const uint32_t p = g_paths;
    if (reset) g_paths = 0;
    return p;
}

extern "C" size_t xrt_status_offset(void) { return XRT_WS_STATUS_BYTE; }

extern "C" void xrt_set_workspace_budget(size_t bytes) { g_budget_cap = bytes; }

extern "C" int xrt_check(void* workspace, void* stream_)
{
    if (!workspace) return fail(-1, "%s", "NULL argument");
    hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
    uint32_t flags = 0;
    HIP_TRY(hipStreamSynchronize(stream));
    HIP_TRY(hipMemcpy(&flags, reinterpret_cast<char*>(workspace) + XRT_WS_STATUS_BYTE, sizeof(flags), hipMemcpyDeviceToHost));
    if (flags & 2u)
        return fail(-7, "%s", "intensity of less than one encountered. Turn on poisson statistics.");
    if (flags & 4u)
        return fail(-8, "%s", "Gaussian wavelength sampler ran out of provisioned candidates (a > 8 sigma event)");
    if (flags & 8u)
        return fail(-9, "%s", "Voight CDF calculation does not have enough resolution or its domain is too small.");
    if (flags & 1u)
        return fail(-6, "%s", "plasma source produced more rays than the declared capacity (Poisson tail): results are truncated");
    if (flags & 16u)
        return fail(-10, "%s", "No rays generated. Check plasma input parameters");
    return 0;
}

extern "C" int xrt_timing_begin(void)
{
    for (int i = 0; i < timing_n; i++) { (void)hipEventDestroy(timing_ev[i][0]); (void)hipEventDestroy(timing_ev[i][1]); }
    timing_n = 0; timing_ms = 0.0; timing_launches = 0; timing_on = true;
    return 0;
}

extern "C" int xrt_timing_end(double* kernel_ms, int64_t* launches)
{
    timing_on = false;
    double total = 0.0;
    for (int i = 0; i < timing_n; i++) {
        HIP_TRY(hipEventSynchronize(timing_ev[i][1]));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, timing_ev[i][0], timing_ev[i][1]));
        total += ms;
        (void)hipEventDestroy(timing_ev[i][0]); (void)hipEventDestroy(timing_ev[i][1]);
    }
    if (kernel_ms) *kernel_ms = total;
    if (launches) *launches = timing_n;
    timing_n = 0;
    return 0;
}
