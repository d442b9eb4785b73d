/*
 * xicsrt_hip.h -- C ABI of the MI355X-native photon propagation path.
 *
 * Boundary: this is what the host side (Python, ctypes) binds in place of the
 * reference's NumPy hot path.  Everything crossing it is plain C: fixed-layout
 * structs, pointers and sizes.  No torch / numpy / C++ types.
 *
 *   reference interface replaced                      entry point here
 *   ------------------------------------------------  ---------------------------
 *   xicsrt_raytrace.raytrace          (xicsrt/xicsrt_raytrace.py:28)   xrt_trace (run loop, seeds per run)
 *   xicsrt_raytrace.raytrace_single   (xicsrt/xicsrt_raytrace.py:87)   xrt_trace (np.random.seed :111, iteration loop :153)
 *   xicsrt_raytrace._raytrace_iter    (xicsrt/xicsrt_raytrace.py:178)  xrt_trace / xrt_trace_history
 *   Dispatcher.generate_rays          (xicsrt/objects/_Dispatcher.py:142)   source stage of the kernel
 *   Dispatcher.trace                  (xicsrt/objects/_Dispatcher.py:166)   optic stages, num_out, images
 *   TraceObject.make_image            (xicsrt/optics/_TraceObject.py:234)   images[] accumulation
 *   history deepcopy per element      (xicsrt/objects/_Dispatcher.py:162,187)  xrt_trace_history
 *   combine_raytrace (sum of meta/images over runs and iterations,
 *                     xicsrt/xicsrt_raytrace.py:328-356)   accumulation into num_out[] / images[]
 *
 * The scene structs below are the flattened `param` dictionaries of the
 * reference's initialised objects (ConfigObject.param, objects/_ConfigObject.py:33):
 * all scalar set-up arithmetic (cos(spread), xsize/2, 2*d, 2*sigma^2, sphere
 * centre ...) is done by the host exactly as the reference's setup()/initialize()
 * do it, and handed over as doubles.
 *
 * Conventions: return value 0 = ok, negative = error (text via xrt_last_error()).
 * The library never throws across the ABI, never allocates device memory on behalf
 * of the caller (it keeps small pinned host caches of jump polynomials) and does not
 * synchronise the device except in xrt_check / xrt_timing_end (table uploads for mesh
 * optics are blocking host-to-device copies): the caller owns every buffer (device
 * pointers are ordinary HIP device addresses, e.g. torch tensors' data_ptr()),
 * passes the HIP stream to launch on, and synchronises that stream itself.
 * A handle-free design: every call is self-contained; calls on different
 * streams/devices may be made from different threads.
 */
#ifndef XICSRT_HIP_H
#define XICSRT_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XRT_ABI_VERSION 19

#define XRT_MAX_OPTICS     64
#define XRT_MAX_APERTURES  32

/* ---- enumerations -------------------------------------------------------- */

/* cone-axis rule of the source (sources/_XicsrtSourceGeneric.py:262,
 * _XicsrtSourceDirected.py:46, _XicsrtSourceFocused.py:40) */
enum { XRT_SRC_GENERIC = 0, XRT_SRC_DIRECTED = 1, XRT_SRC_FOCUSED = 2,
       XRT_SRC_PLASMA = 3,  /* XicsrtPlasma*: bundles of focused sources (sources/_XicsrtPlasmaGeneric.py:252-382) */
       XRT_SRC_EXTERNAL = 4 /* rays supplied by the caller: TraceObject.trace_global(rays) on a user's ray
                             * array (optics/_TraceObject.py:135-154); xrt_trace_history only */ };
/* spatial_dist (_XicsrtSourceGeneric.py:231,237) */
enum { XRT_SPATIAL_UNIFORM = 0, XRT_SPATIAL_GAUSSIAN = 1 };
/* angular_dist (tools/xicsrt_spread.py:21) */
enum { XRT_ANG_ISOTROPIC = 0, XRT_ANG_ISOTROPIC_XY = 1, XRT_ANG_FLAT = 2, XRT_ANG_FLAT_XY = 3 };
/* effective wavelength sampler after the reference's own case analysis
 * (_XicsrtSourceGeneric.py:295-367): CONST consumes no random numbers. */
enum { XRT_WL_CONST = 0, XRT_WL_UNIFORM = 1, XRT_WL_NORMAL = 2, XRT_WL_VOIGT = 3,
       XRT_WL_VOIGT_BUNDLE = 4 /* internal: a plasma bundle with its own Voigt table (never in xrt_source_t) */ };

/* Shape* classes (optics/_ShapePlane.py, _ShapeSphere.py, _ShapeCylinder.py, _ShapeTorus.py) */
enum { XRT_SHAPE_PLANE = 0, XRT_SHAPE_SPHERE = 1, XRT_SHAPE_CYLINDER = 2, XRT_SHAPE_TORUS = 3,
       XRT_SHAPE_MESH = 4   /* optics/_ShapeMesh.py */ };
/* Interact* classes (optics/_InteractNone.py, _InteractMirror.py, _InteractCrystal.py) */
enum { XRT_INTERACT_NONE = 0, XRT_INTERACT_MIRROR = 1, XRT_INTERACT_CRYSTAL = 2,
       XRT_INTERACT_MOSAIC = 3   /* optics/_InteractMosaicCrystal.py:53-139 */ };
/* rocking_type (optics/_InteractCrystal.py:138-149) */
enum { XRT_ROCKING_STEP = 0, XRT_ROCKING_GAUSS = 1 };

/* aperture shapes and logic (tools/xicsrt_aperture.py:13-204) */
enum { XRT_AP_NONE = 0, XRT_AP_CIRCLE = 1, XRT_AP_SQUARE = 2, XRT_AP_RECTANGLE = 3,
       XRT_AP_ELLIPSE = 4, XRT_AP_TRIANGLE = 5 };
enum { XRT_LOGIC_AND = 0, XRT_LOGIC_NOT = 1, XRT_LOGIC_OR = 2, XRT_LOGIC_NAND = 3,
       XRT_LOGIC_NOR = 4, XRT_LOGIC_XOR = 5, XRT_LOGIC_XNOR = 6 };

/* optic flags */
enum {
    XRT_F_CHECK_SIZE     = 1 << 0,   /* param['check_size']      (_TraceObject.py:204) */
    XRT_F_CHECK_APERTURE = 1 << 1,   /* param['check_aperture']  (_TraceObject.py:227) */
    XRT_F_HAS_XSIZE      = 1 << 2,   /* xsize is not None        (_TraceObject.py:205) */
    XRT_F_HAS_YSIZE      = 1 << 3,
    XRT_F_HAS_ZSIZE      = 1 << 4,
    XRT_F_CONVEX         = 1 << 5,   /* param['convex']          (_ShapeSphere.py:95)  */
    XRT_F_CHECK_BRAGG    = 1 << 6,   /* param['check_bragg']     (_InteractCrystal.py:120) */
    XRT_F_IMAGE          = 1 << 7,   /* param['enable_image']    (_TraceObject.py:131) */
    XRT_F_TRACE_LOCAL    = 1 << 8    /* param['trace_local']     (_TraceObject.py:146) */
};

/* ---- scene --------------------------------------------------------------- */

typedef struct xrt_aperture {
    int32_t shape;
    int32_t logic;
    double  origin[2];
    double  size[2];          /* circle: size[0]=radius; square: size[0]; rectangle/ellipse: x,y */
    double  vertices[6];      /* triangle: x0,y0,x1,y1,x2,y2 with the aperture origin already added */
} xrt_aperture_t;

/* One XicsrtBundleFilterSightline (filters/_XicsrtBundleFilterSightline.py:31-56): a bundle is
 * kept when radius >= |l_0 - zaxis (zaxis . l_0)|, l_0 = origin - bundle centre; origin and zaxis
 * are the filter's config values as given (zaxis is NOT normalised by the reference). */
#define XRT_MAX_BUNDLE_FILTERS 16
typedef struct xrt_bundle_filter {
    double origin[3];
    double zaxis[3];
    double radius;
} xrt_bundle_filter_t;

enum { XRT_PLASMA_BOX = 0,       /* XicsrtPlasmaGeneric / Cubic: no flux coordinate           */
       XRT_PLASMA_TOROIDAL = 1   /* XicsrtPlasmaToroidal(+Datafile) (_XicsrtPlasmaToroidal.py:36-45) */ };

/* Per-bundle plasma model.  When xrt_source_t.plasma is non-NULL every bundle's mask,
 * emissivity, temperature and spread are evaluated from its centre, in the reference's order:
 * bundle_filter (sightlines) -> bundle_generate (rho, profiles, isfinite(temperature)) ->
 * create_sources (intensity, Poisson count) (sources/_XicsrtPlasmaGeneric.py:384-393).
 * NULL: all bundles share the source-level constants (bundle_intensity, ang, wl_*). */
typedef struct xrt_plasma {
    int32_t geometry;          /* XRT_PLASMA_*                                               */
    int32_t has_spread_radius; /* spread_b = arctan(spread_radius / |centre - target|) (:218-221) */
    int32_t n_filters;
    int32_t n_emissivity;      /* profile points of get_emissivity(rho); 0: the constant below */
    int32_t n_temperature;     /* profile points of get_temperature(rho); 0: constant (then the
                                * source-level wavelength fields already describe every bundle) */
    int32_t pad;
    double  torus_origin[3];
    double  major_radius, minor_radius;
    double  emissivity;        /* get_emissivity() constant                                  */
    double  emissivity_scale, temperature_scale;
    const double* emissivity_rho;   /* HOST pointers, n_emissivity / n_temperature doubles each; */
    const double* emissivity_val;   /* np.interp(rho, x, y, left=0, right=0)                     */
    const double* temperature_rho;  /* (_XicsrtPlasmaToroidalDatafile.py:31-45)                  */
    const double* temperature_val;
    double  spread_radius;
    double  solid_angle;       /* constant-spread case: 4 pi sin^2(spread/2) (xicsrt_spread.py:127) */
    double  time_resolution, bundle_volume, four_pi, volume_ratio;  /* intensity factors (:301-319),
                                * volume_ratio = volume / (bundle_count * bundle_volume)      */
    double  mass_number, amu_kg, c_squared, ev_J;   /* per-bundle Doppler width, c_squared = c**2 (_XicsrtSourceGeneric.py:363-365) */
    xrt_bundle_filter_t filters[XRT_MAX_BUNDLE_FILTERS];
    /* A temperature profile together with a natural line width (wavelength_dist = XRT_WL_VOIGT, voigt_n = 0 at
     * the source level): every bundle samples its own Voigt profile, whose CDF table of 1000 points
     * (tools/xicsrt_voigt.py:30-130, voigt_cdf_tab with the bundle's sigma) is built per bundle.  The
     * Faddeeva function Re w(z) of that table (scipy.special.wofz in the reference) is evaluated with
     * Weideman's rational approximation (SIAM J. Numer. Anal. 31 (1994) 1497): coefficients from the host. */
    double  voigt_gamma;       /* linewidth * wavelength**2 / (4 pi c 1e10) (_XicsrtSourceGeneric.py:346); 0: unused */
    double  weideman_L;
    const double* weideman_a;  /* HOST pointer, n_weideman polynomial coefficients, highest power first */
    int32_t n_weideman, pad2;
} xrt_plasma_t;

typedef struct xrt_source {
    int32_t kind;             /* XRT_SRC_*      */
    int32_t spatial_dist;     /* XRT_SPATIAL_*  */
    int32_t angular_dist;     /* XRT_ANG_*      */
    int32_t wavelength_dist;  /* XRT_WL_*       */
    int64_t intensity;        /* rays per iteration, param['intensity'] after initialize() */
    double  origin[3];        /* GeometryObject.origin                                     */
    double  orientation[9];   /* rows xaxis, yaxis = zaxis x xaxis, zaxis (_GeometryObject.py:94) */
    double  size[3];          /* xsize, ysize, zsize (full widths)                         */
    double  spatial_A[9];     /* GAUSSIAN: sqrt(s)[:,None]*v of svd(cov), row-major (np.random.multivariate_normal) */
    double  axis[3];          /* GENERIC: param zaxis; DIRECTED: param direction; FOCUSED: target */
    double  basis[9];         /* GENERIC/DIRECTED: rows o_2, o_1, normal of the emission frame,
                               * computed by the host as _XicsrtSourceGeneric.py:262-285 does
                               * (FOCUSED builds the frame per ray on the device)            */
    /* angular distribution constants, computed by the host as the reference does:
     *   ISOTROPIC    ang[0] = cos(spread)            (xicsrt_spread.py:102)
     *   FLAT         ang[0] = tan(spread)            (xicsrt_spread.py:235)
     *   FLAT_XY      ang[0..3] = tan(theta[0..3])    (xicsrt_spread.py:282-284)
     *   ISOTROPIC_XY ang[0] = cos(theta_max), ang[1..4] = sin(theta[0..3]) (xicsrt_spread.py:173-186) */
    double  ang[5];
    double  two_pi;           /* 2*np.pi as the host computes it                           */
    double  wavelength;       /* CONST / NORMAL loc / VOIGT centre                         */
    double  wl_a, wl_b;       /* UNIFORM: low, high-low; NORMAL: sigma in wl_a; VOIGT: cdf min, (cdf max - cdf min) */
    int32_t has_velocity;     /* not np.all(velocity == 0) (_XicsrtSourceGeneric.py:314)   */
    int32_t voigt_n;          /* number of table points                                    */
    double  velocity[3];
    double  light_speed;      /* scipy.constants c                                         */
    const double* voigt_cdf;  /* HOST pointers, voigt_n doubles each; copied per call      */
    const double* voigt_x;
    /* XRT_SRC_PLASMA only.  `intensity` is then the ray CAPACITY per iteration (the
     * actual count is Poisson distributed and reported in num_out[0]); `size` is the
     * bundle voxel edge (x3), `axis` the target, the cone/wavelength fields describe
     * every bundle (XicsrtPlasmaCubic: constant temperature, emissivity, spread). */
    int64_t bundle_count;     /* param['bundle_count']       (_XicsrtPlasmaGeneric.py:166-168) */
    double  plasma_size[3];   /* xsize, ysize, zsize of the plasma box (bundle centres, :195-197) */
    double  bundle_intensity; /* expected rays per bundle     (:301-319)                    */
    int32_t use_poisson;      /* np.random.poisson(intensity) per bundle, else int(intensity) */
    int32_t pad_plasma;
    const xrt_plasma_t* plasma;   /* HOST pointer or NULL: per-bundle model, see xrt_plasma_t   */
    /* XRT_SRC_EXTERNAL only: `intensity` rays in DEVICE memory, component-major
     * ext_rays[c * intensity + i], c = ox,oy,oz,dx,dy,dz,wavelength,weight; ext_mask[i] != 0: alive.
     * Rays with mask 0 do not take part (RayArray 'mask', objects/_RayArray.py:12-96). */
    const double*  ext_rays;
    const uint8_t* ext_mask;
    /* XicsrtSourceGeneric.ray_filter (_XicsrtSourceGeneric.py:223, :393-396) for the non-plasma sources: the
     * sightline filters attached to the source switch off (mask = False) the generated rays whose ORIGIN lies outside
     * (filters/_XicsrtBundleFilterSightline.py:31-56 applied to the ray dictionary); the rays keep their places in
     * the ray order and in the stream.  0: none. */
    int32_t n_ray_filters;
    int32_t pad_filters;
    xrt_bundle_filter_t ray_filters[XRT_MAX_BUNDLE_FILTERS];
} xrt_source_t;

/* A triangulated-mesh surface (optics/_ShapeMesh.py:198-261 _mesh_precalc output), all HOST
 * pointers, copied to the device per call.  Built by the host with SciPy exactly where the
 * reference uses it (Delaunay, CloughTocher2DInterpolator gradients). */
typedef struct xrt_mesh {
    int32_t n_points, n_faces;          /* fine mesh                                           */
    int32_t n_coarse_faces;             /* > 0: pre-selection mesh (mesh_refine), 0: search the fine mesh */
    int32_t interpolate;                /* mesh_interpolate                                    */
    int32_t n_simplices, pad;           /* x-y Delaunay used by the interpolators              */
    const double*  points;              /* [n_points][3]                                       */
    const double*  p0;                  /* [n_faces][3] vertices of every face                 */
    const double*  p1;
    const double*  p2;
    const double*  edge1;               /* [n_faces][3] p1-p0, p2-p0 (Moller-Trumbore, :289-348) */
    const double*  edge2;
    const double*  faces_normal;        /* [n_faces][3]  (:240-241)                            */
    const double*  faces_area;          /* [n_faces] |(p0-p1)x(p0-p2)|  (:405)                 */
    const int32_t* p_faces_idx;         /* [8][n_points] faces around each point (:446-462)    */
    const uint8_t* p_faces_mask;        /* [8][n_points]                                       */
    const double*  c_p0;                /* [n_coarse_faces][3] coarse mesh faces               */
    const double*  c_edge1;
    const double*  c_edge2;
    const int32_t* ct_simplices;        /* [n_simplices][3]                                    */
    const int32_t* ct_neighbors;        /* [n_simplices][3]                                    */
    const double*  ct_transform;        /* [n_simplices][3][2] barycentric transforms          */
    const double*  ct_points;           /* [n_points][2]                                       */
    const double*  ct_values;           /* [4][n_points]  z, normal_x, normal_y, normal_z      */
    const double*  ct_grad;             /* [4][n_points][2] vertex gradients                   */
    const int32_t* ct_vertex_simplex;   /* [n_points] a simplex containing the vertex          */
} xrt_mesh_t;

typedef struct xrt_optic {
    int32_t shape;            /* XRT_SHAPE_*    */
    int32_t interact;         /* XRT_INTERACT_* */
    int32_t flags;            /* XRT_F_*        */
    int32_t rocking_type;     /* XRT_ROCKING_*  */
    double  origin[3];
    double  orientation[9];
    double  half_size[3];     /* xsize/2, ysize/2, zsize/2                                 */
    double  radius;           /* sphere / cylinder                                         */
    double  radius2;          /* radius**2 as the host computes it                         */
    double  center[3];        /* sphere / cylinder / torus centre (_ShapeSphere.py:43)     */
    double  torus_major;      /* torus frame major radius (_ShapeTorus.py:72-91)           */
    double  torus_minor;
    double  torus_k[5];       /* host constants of the quartic (_ShapeTorus.py:139-159):
                               * r_sq = R^2+r^2, 2*r_sq, 4*R^2, 8*R^2, (R^2-r^2)^2         */
    int32_t torus_root;       /* quartic root column (_ShapeTorus.py:72-89)                */
    int32_t n_apertures;
    double  two_d;            /* 2*crystal_spacing            (_InteractCrystal.py:110)    */
    double  reflectivity;
    double  rocking_half_fwhm;/* STEP: rocking_fwhm/2         (_InteractCrystal.py:141)    */
    double  rocking_2sigma2;  /* GAUSS: 2*sigma**2            (_InteractCrystal.py:146-149)*/
    double  half_pi;          /* np.pi/2                      (_InteractCrystal.py:112)    */
    /* XRT_INTERACT_MOSAIC (optics/_InteractMosaicCrystal.py) */
    int32_t mosaic_depth;     /* param['mosaic_depth']                                     */
    int32_t mosaic_has_cutoff;/* param['mosaic_cutoff'] is not None                        */
    double  mosaic_cutoff_angle; /* sqrt(-log(cutoff)*2*sigma**2)  (:67-72)                */
    double  mosaic_A[4];      /* sqrt(s)[:,None]*v of svd(diag(sin(sigma_h)^2)), row-major:
                               * factor of multivariate_normal (tools/xicsrt_spread.py:317-332) */
    double  pixel_size;       /* (_TraceObject.py:107)                                     */
    double  pixel_xoff;       /* (pixel_xsize-1)/2            (_TraceObject.py:269)        */
    double  pixel_yoff;
    int32_t pixel_nx;
    int32_t pixel_ny;
    int64_t image_offset;     /* first bin of this optic's image in images[] (row-major [nx][ny]) */
    const xrt_mesh_t* mesh;   /* XRT_SHAPE_MESH only (HOST pointer)                        */
    xrt_aperture_t apertures[XRT_MAX_APERTURES];
} xrt_optic_t;

typedef struct xrt_scene {
    xrt_source_t source;
    int32_t      n_optics;
    int32_t      reserved;
    int64_t      image_bins;  /* total number of bins of all images                        */
    xrt_optic_t  optics[XRT_MAX_OPTICS];
} xrt_scene_t;

/* numpy legacy RandomState (MT19937 key + position + cached gaussian), the
 * state np.random.get_state()/set_state() exchange. */
typedef struct xrt_rng_state {
    uint32_t key[624];
    int32_t  pos;             /* 0..624; 624 = regenerate before the next word */
    int32_t  has_gauss;
    double   gauss;
} xrt_rng_state_t;

/* History snapshot of ONE run/iteration (Dispatcher history deepcopy,
 * objects/_Dispatcher.py:162,187): for element e (0 = source, 1.. = optics) and
 * original ray index i, structure-of-arrays
 *     rays[(e*8 + c)*n_rays + i],  c = ox,oy,oz,dx,dy,dz,wavelength,weight
 *     mask[e*n_rays + i]
 * Only rays alive after element e are written; the caller pre-fills the rest
 * (the reference leaves NaN origins there). */
#define XRT_HIST_COMPONENTS 8

/* ---- entry points -------------------------------------------------------- */

int         xrt_abi_version(void);
const char* xrt_last_error(void);

/* Size of struct xrt_scene as compiled, so a binding can verify its layout. */
size_t      xrt_sizeof_scene(void);

/* Static validation of a scene (no device work): 0 if the device path implements it. */
int         xrt_scene_check(const xrt_scene_t* scene);

/* Number of HIP devices visible (hipGetDeviceCount). */
int         xrt_device_count(int* count);

/* Bytes of device scratch xrt_trace needs for this problem. */
size_t      xrt_workspace_bytes(const xrt_scene_t* scene, int32_t n_runs);

/*
 * Trace n_runs independent runs (one MT19937 stream each, seeded
 * init_genrand(seeds[r]) like np.random.seed, xicsrt_raytrace.py:111), each of
 * n_iter iterations sharing the run's stream (xicsrt_raytrace.py:153), on the
 * current device.  Adds into
 *     num_out[0 .. n_optics]          device u64, rays alive after each element
 *     images[0 .. scene->image_bins)  device u64, per-optic pixel counts
 * (the caller zeroes them; repeated calls accumulate, which is the reference's
 * combine_raytrace sum).  `images` may be NULL when keep_images is off.
 * Asynchronous on `stream` (a hipStream_t; NULL = default stream).  A scene with mesh optics first waits for the
 * work already queued on `stream` (its packed mesh tables are copied into the workspace synchronously).
 */
int xrt_trace(const xrt_scene_t* scene,
              const uint32_t* seeds, int32_t n_runs, int32_t n_iter,
              uint64_t* num_out, uint64_t* images,
              void* workspace, size_t workspace_bytes,
              void* stream);

/*
 * One iteration from an explicit generator state, keeping the per-element
 * history (keep_history=True; also the single-object plug-in calls, which take
 * the global np.random state in and hand the advanced state back): as
 * xrt_trace with n_runs = n_iter = 1, and additionally fills the device buffers
 * `rays` (double, (n_optics+1)*8*n_rays), `mask` (uint8, (n_optics+1)*n_rays)
 * described above and `state_out` (one xrt_rng_state_t in device memory: the
 * generator state after the iteration).  `state_in` is host memory.
 * A ray that dies at element e is written at e with mask 0 and the point it
 * died at (NaN origin if it had no intersection), direction unchanged.
 */
int xrt_trace_history(const xrt_scene_t* scene, const xrt_rng_state_t* state_in,
                      uint64_t* num_out, uint64_t* images,
                      double* rays, uint8_t* mask, void* state_out,
                      void* workspace, size_t workspace_bytes,
                      void* stream);

/* TraceObject.make_image(rays) (optics/_TraceObject.py:234-293) for a caller's ray array: every ray
 * with mask[i] != 0 adds one count to the optic's pixel bin of its origin, images[image_offset +
 * cx * pixel_ny + cy].  rays: DEVICE, component-major [8][n] as xrt_source_t.ext_rays (only the
 * three origin rows are read); mask, images: DEVICE.  Asynchronous on `stream`. */
int xrt_make_image(const xrt_optic_t* optic, int64_t n, const double* rays, const uint8_t* mask,
                   uint64_t* images, void* stream);

/* Synchronises `stream` and reports conditions the asynchronous calls could only flag on the
 * device: -6 a plasma source produced more rays than the declared capacity; -7 a plasma bundle has an
 * intensity below one without Poisson statistics (the reference's ValueError,
 * _XicsrtSourceGeneric.py:193-194); -8 the candidate reserve of the Gaussian wavelength sampler was
 * exhausted (a > 8 sigma event); -9 a per-bundle Voigt table without enough resolution (the reference's ValueError,
 * tools/xicsrt_voigt.py); -10 a plasma source generated no ray at all in some iteration ("No rays generated",
 * _XicsrtPlasmaGeneric.py:368-369).  0 = clean. */
int xrt_check(void* workspace, void* stream);

/* The device status word xrt_check reads: a uint32 at this byte offset of the workspace (0 = nothing to report; the
 * bits are private to xrt_check, which words them).  Cleared at the start of every xrt_trace / xrt_trace_history call.
 * Exported so that a binding which already copies its results back can fetch the word in the same batch of
 * asynchronous copies and call xrt_check only when it is non-zero.  xrt_status_offset() returns the offset the
 * library was built with; a binding verifies it equals XRT_WS_STATUS_BYTE at load time.  No reference counterpart. */
#define XRT_WS_STATUS_BYTE 64
size_t xrt_status_offset(void);

/* Caps the budgets the big regions of a workspace are sized to (slots of the staged path, parked candidates of the one-pass
 * routes; beyond a budget a call works through its runs in batches or takes a leaner route).  The defaults suit a 288 GB
 * part and scale down with the current device's total memory; a caller whose allocation of xrt_workspace_bytes() failed
 * passes what it can spare and asks again.  0: back to the defaults.  Process-wide; affects xrt_workspace_bytes and the
 * calls that follow alike.  No reference counterpart (the reference's working set is whatever NumPy allocates). */
void   xrt_set_workspace_budget(size_t bytes);

/* TraceObject.intersect / check_bounds / interact as separate calls on a caller's ray array
 * (optics/_TraceObject.py:157-180 `trace`: xloc, norm, mask = intersect(rays); mask = check_bounds(xloc, mask);
 * rays = interact(rays, xloc, norm, mask)).  Analytic shapes (Shape{Plane,Sphere,Cylinder,Torus}.intersect,
 * optics/_Shape*.py); all pointers DEVICE, rays component-major [8][n] as xrt_source_t.ext_rays, xloc and norm
 * [3][n], masks one byte per ray.  Asynchronous on `stream`.
 *   xrt_optic_intersect     rays with mask_in: intersection point and surface normal (NaN where there is none),
 *                           mask_out = mask_in & has-intersection
 *   xrt_optic_check_bounds  mask &= inside xsize/ysize/zsize and the aperture list (_TraceObject.py:180-232)
 *   xrt_optic_interact      origin <- xloc for EVERY ray (_InteractMirror.py:38), rays with mask: mirror reflection
 *                           (_InteractMirror.py:29-42) or, for a crystal with check_bragg, first the rocking-curve
 *                           test p >= test[i] (_InteractCrystal.py:96-196; `test` = the uniform deviates the
 *                           reference draws for the live rays, scattered to ray positions by the caller); mask out.
 *                           Mosaic crystals (whole-array draws per layer) are refused: use xrt_trace_history. */
int xrt_optic_intersect(const xrt_optic_t* optic, int64_t n, const double* rays, const uint8_t* mask_in,
                        double* xloc, double* norm, uint8_t* mask_out, void* stream);
int xrt_optic_check_bounds(const xrt_optic_t* optic, int64_t n, const double* xloc, uint8_t* mask, void* stream);
int xrt_optic_interact(const xrt_optic_t* optic, int64_t n, double* rays, const double* xloc, const double* norm,
                       uint8_t* mask, const double* test, void* stream);

/* Self-test: the shared-reciprocal division the kernels use to normalise 3-vectors (three IEEE quotients with one
 * reciprocal) against the division operator, on caller-supplied operands (device: num [3][n], den [n]); *bad (device,
 * caller-zeroed) counts the quotients whose bits differ.  No reference counterpart. */
int xrt_selftest_div3(const double* num, const double* den, int64_t n, uint64_t* bad, void* stream);

/* Diagnostics: the device routes taken by this thread's xrt_trace / xrt_trace_history calls since the last
 * call with reset != 0 (bit set below).  The routes are picked per scene; the environment switches
 * XICSRT_NO_JUMP, XICSRT_NO_STAGE_SPLIT, XICSRT_STAGED_GAUSS, XICSRT_SEGMENTS, XICSRT_NO_MOSAIC_FUSED force the fallbacks and are read
 * at every call.  No reference counterpart. */
#define XRT_PATH_FUSED          1u   /* source -> optics -> histogram in one kernel                    */
#define XRT_PATH_STAGED         2u   /* array-at-a-time passes over a ray SoA in HBM                    */
#define XRT_PATH_STAGE_SPLIT    4u   /* staged: source and optics as two launches per batch of runs     */
#define XRT_PATH_JUMP           8u   /* MT19937 heads positioned by polynomial jump-ahead               */
#define XRT_PATH_SEEK          16u   /* heads positioned by walking the stream                          */
#define XRT_PATH_SEGMENTED     32u   /* runs split into segments (few runs of many rays)                */
#define XRT_PATH_GAUSS_PREPARED 64u  /* np.random.normal wavelengths prepared as an array               */
#define XRT_PATH_PLASMA_SCOUT 128u   /* plasma source: stream walked by one wave per run, rays rebuilt in the fused kernel */
#define XRT_PATH_LDS_BINS     256u   /* fused kernel with the pixel bins pre-aggregated in LDS (image-heavy scenes without a Bragg test) */
#define XRT_PATH_ONE_PASS     512u   /* segmented runs in one pass: Bragg candidates parked in HBM, stream offsets by look-back */
#define XRT_PATH_MESH_SPLIT   1024u  /* a mesh crystal: first phase, the rest of the mesh intersection per parked ray, second phase as three launches */
#define XRT_PATH_MESH_FANS    2048u  /* ... the second pass over the faces around the nearest point first from that point's fan, the list walk for the rays left over */
#define XRT_PATH_MOSAIC_FUSED 4096u  /* a mosaic crystal: the fused kernel's first phase parks the rays on the crystal, one kernel runs the layers and the elements behind over them */
uint32_t xrt_last_path(int32_t reset);

/* np.random.shuffle(np.arange(n))[:m] of numpy's legacy generator (host only, no device): what
 * _sort_raytrace draws to sample the lost rays (xicsrt/xicsrt_raytrace.py:264-266).  state in / out:
 * np.random.get_state() layout (has_gauss / gauss pass through untouched).  out: m indices. */
int xrt_legacy_shuffle_head(xrt_rng_state_t* state, int64_t n, int64_t m, int64_t* out);

/* Diagnostic, host only: the MT19937 jump-ahead polynomial g(t) = t^J mod phi(t)
 * (phi = characteristic polynomial of the generator, degree 19937) that positions
 * the per-array generator heads: out624 receives 624 words, bit j of word j/32 = g_j,
 * such that state_word[n + J] = XOR over {j : g_j = 1} of state_word[n + j].
 * Replaces nothing in the reference (np.random walks its stream sequentially). */
int xrt_mt_jump_poly(uint64_t J, uint32_t* out624);
/* ... n of them at once (out: n x 624 words), the way a plan's jump jobs get theirs: the exponents of a plan come in
 * arithmetic families (segment s of array k, chunk head c), and a polynomial is its neighbour's times t^step -- one
 * carry-less product and reduction instead of a square-and-multiply from scratch.  Same results as xrt_mt_jump_poly. */
int xrt_mt_jump_polys(const uint64_t* J, int32_t n, uint32_t* out);

/* Name and average duration (ms) bookkeeping of the propagation kernel for the
 * benchmark: brackets the kernel launches of the next xrt_trace calls with HIP
 * events on the launch stream.  xrt_timing_begin resets, xrt_timing_end
 * synchronises the events and returns total kernel milliseconds and launches. */
int xrt_timing_begin(void);
int xrt_timing_end(double* kernel_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* XICSRT_HIP_H */
