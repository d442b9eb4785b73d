"""
GPU tests at and near BASELINE sizes (-m gpu): where the oracle is too slow to follow, parity
is carried by size-independent properties of the domain (exact integer sums, partition and
repetition invariance, histogram = counter identities).
"""
import copy

import numpy as np
import pytest

import helpers
from xicsrt_amd import config as xconfig
from xicsrt_amd import xicsrt_raytrace as xrt
from xicsrt_amd import capi

pytestmark = pytest.mark.gpu


def _spectrometer(n_rays, runs, seed=0, **optic_over):
    cfg, _ = helpers.load_golden('C_sphere_1e5')
    cfg = copy.deepcopy(cfg)
    cfg['sources']['source']['intensity'] = n_rays
    cfg['general'].update(number_of_runs=runs, random_seed=seed)
    cfg['optics']['crystal'].update(optic_over)
    return cfg


def _trace(flat, seeds, n_iter=1):
    from xicsrt_amd import xicsrt_raytrace as xrt
    dev = xrt.DeviceTrace(flat)
    dev.trace(seeds, n_iter)
    dev.torch.cuda.synchronize()
    return dev.num_out.cpu().numpy().copy(), dev.images.cpu().numpy().copy()


def test_million_ray_runs_equal_oracle_bit_for_bit():
    """16 runs x 1e6 rays of the bench scene: counters and both images identical to the CPU oracle."""
    from xicsrt_amd import xicsrt_raytrace as xrt
    config, elements, flat = helpers.build(_spectrometer(1000000, 16, seed=123))
    seeds = xrt.run_seeds(123, 16)
    n_gpu, i_gpu = _trace(flat, seeds)
    n_cpu, i_cpu = helpers.oracle_counts(flat, seeds, 1, threads=16)
    assert np.array_equal(n_gpu, n_cpu)
    assert np.array_equal(i_gpu[:flat.image_bins], i_cpu[:flat.image_bins])
    assert int(n_gpu[0]) == 16000000 and 0.040 < n_gpu[2] / n_gpu[0] < 0.044


def test_full_size_invariants_1e9_photons():
    """The bench workload itself (1000 runs x 1e6 rays): exact identities that do not need an oracle."""
    from xicsrt_amd import xicsrt_raytrace as xrt
    config, elements, flat = helpers.build(_spectrometer(1000000, 1000))
    seeds = xrt.run_seeds(0, 1000)
    n_all, i_all = _trace(flat, seeds)
    assert int(n_all[0]) == 10 ** 9
    off_c, nx_c, ny_c = flat.image_slices['crystal']
    off_d, nx_d, ny_d = flat.image_slices['detector']
    # every reflected ray is binned on the crystal image, every detector hit on the detector image
    assert int(i_all[off_c:off_c + nx_c * ny_c].sum()) == int(n_all[1])
    assert int(i_all[off_d:off_d + nx_d * ny_d].sum()) == int(n_all[2])
    # repetition: the same launch again gives the same integers (atomics commute exactly)
    n_again, i_again = _trace(flat, seeds)
    assert np.array_equal(n_all, n_again) and np.array_equal(i_all, i_again)
    # partition: three unequal shards of the runs sum to the whole (what the multi-GPU path relies on)
    parts = [seeds[0:100], seeds[100:617], seeds[617:]]
    n_sum = np.zeros_like(n_all)
    i_sum = np.zeros_like(i_all)
    for p in parts:
        n, i = _trace(flat, p)
        n_sum += n
        i_sum += i
    assert np.array_equal(n_sum, n_all) and np.array_equal(i_sum, i_all)
    # the first 8 runs of the 1e9-photon job equal the reference's own published count for
    # 8 x 1e6 photons, seed 0 (335 789 detector hits)
    n8, _ = _trace(flat, seeds[:8])
    assert int(n8[2]) == 335789 and int(n8[1]) == 335789


def test_iterations_chain_the_stream():
    """number_of_iter=3 equals one iteration thrice only through the shared stream: compare with the oracle."""
    from xicsrt_amd import xicsrt_raytrace as xrt
    config, elements, flat = helpers.build(_spectrometer(200000, 6, seed=9))
    seeds = xrt.run_seeds(9, 6)
    n_gpu, i_gpu = _trace(flat, seeds, n_iter=3)
    n_cpu, i_cpu = helpers.oracle_counts(flat, seeds, 3, threads=6)
    assert np.array_equal(n_gpu, n_cpu) and np.array_equal(i_gpu[:flat.image_bins], i_cpu[:flat.image_bins])


def _random_scene(rs):
    """A random 3- or 4-element scene from the implemented classes (geometry of integrated_test_01, jittered)."""
    optic_classes = [
        ('XicsrtOpticPlanarMirror', {}), ('XicsrtOpticSphericalMirror', {'radius': 1.0}),
        ('XicsrtOpticPlanarCrystal', 'bragg'), ('XicsrtOpticSphericalCrystal', 'bragg+r'),
        ('XicsrtOpticCylindricalCrystal', 'bragg+r'), ('XicsrtOpticCylindricalMirror', {'radius': 1.0}),
        ('XicsrtOpticToroidalCrystal', 'bragg+t')]
    cls, extra = optic_classes[rs.randint(len(optic_classes))]
    crystal = {'class_name': cls, 'origin': [0.0, 0.0, 0.80374151], 'zaxis': [0.0, 0.59497864, -0.80374151],
               'xsize': float(rs.uniform(0.05, 0.3)), 'ysize': float(rs.uniform(0.05, 0.3))}
    if isinstance(extra, dict):
        crystal.update(extra)
        if rs.rand() < 0.3 and 'radius' in extra:
            crystal['convex'] = True
    else:
        crystal.update(crystal_spacing=2.45676, rocking_type=['gaussian', 'step'][rs.randint(2)],
                       rocking_fwhm=float(10 ** rs.uniform(-4.3, -2.0)), reflectivity=float(rs.uniform(0.3, 1.0)))
        if rs.rand() < 0.25:
            crystal['check_bragg'] = False
        if '+r' in extra:
            crystal['radius'] = float(rs.uniform(0.8, 1.5))
        if '+t' in extra:
            crystal.update(radius_major=float(rs.uniform(0.9, 1.3)), radius_minor=float(rs.uniform(0.1, 0.4)))
    if rs.rand() < 0.3:
        crystal['aperture'] = [{'shape': 'circle', 'size': [float(rs.uniform(0.03, 0.12))]},
                               {'shape': 'rectangle', 'size': [0.04, 0.02], 'origin': [0.01, -0.01],
                                'logic': ['not', 'or', 'xor'][rs.randint(3)]}]
    source = {'class_name': ['XicsrtSourceDirected', 'XicsrtSourceGeneric', 'XicsrtSourceFocused'][rs.randint(3)],
              'intensity': int(rs.randint(3000, 9000)), 'wavelength': 3.9492,
              'spread': float(np.radians(rs.uniform(2.0, 12.0))),
              'xsize': float(rs.choice([0.0, 0.002])), 'ysize': float(rs.choice([0.0, 0.004])), 'zsize': 0.0}
    if source['class_name'] == 'XicsrtSourceFocused':
        source['target'] = [0.0, 0.0, 0.80374151]
    wl = rs.randint(4)
    if wl == 1:
        source.update(wavelength_dist='uniform', wavelength_range=[3.9480, 3.9500])
    elif wl == 2:
        source.update(temperature=float(rs.uniform(200, 2000)), mass_number=39.948)
    elif wl == 3:
        source.update(linewidth=1.129e14, temperature=1000.0, mass_number=39.948)
    if rs.rand() < 0.25:
        source['angular_dist'] = ['flat', 'flat_xy', 'isotropic_xy'][rs.randint(3)]
        if source['angular_dist'] != 'flat':
            source['spread'] = [float(rs.uniform(0.03, 0.12)), float(rs.uniform(0.03, 0.12))]
    cfg = {'general': {'number_of_iter': int(rs.randint(1, 3)), 'number_of_runs': int(rs.randint(1, 4)),
                       'random_seed': int(rs.randint(0, 2 ** 31)), 'keep_history': False, 'print_results': False},
           'sources': {'source': source},
           'optics': {'crystal': crystal,
                      'detector': {'class_name': 'XicsrtOpticDetector', 'origin': [0.0, 0.76871290, 0.56904832],
                                   'zaxis': [0.0, -0.95641806, 0.29200084], 'xsize': 0.4, 'ysize': 0.2}}}
    return cfg


@pytest.mark.parametrize('case', range(24))
def test_random_scenes_equal_oracle(case):
    """Randomised sources / optics / seeds: device == oracle exactly (counts, images) and in history."""
    from xicsrt_amd import xicsrt_raytrace as xrt
    rs = np.random.RandomState(1000 + case)
    cfg = _random_scene(rs)
    config, elements, flat = helpers.build(cfg)
    g = config['general']
    seeds = xrt.run_seeds(g['random_seed'], g['number_of_runs'])
    n_gpu, i_gpu = _trace(flat, seeds, g['number_of_iter'])
    n_cpu, i_cpu = helpers.oracle_counts(flat, seeds, g['number_of_iter'])
    assert np.array_equal(n_gpu, n_cpu), (cfg, n_gpu, n_cpu)
    assert np.array_equal(i_gpu[:flat.image_bins], i_cpu[:flat.image_bins])
    dev = xrt.DeviceTrace(flat)
    rays, mask, st = dev.trace_history(xrt.rng_state_from_seed(seeds[0]))
    o_num, o_img, o_rays, o_mask, o_st = helpers.oracle_history(flat, helpers.seed_state(seeds[0]))
    assert np.array_equal(mask, o_mask)
    assert np.array_equal(np.isnan(rays), np.isnan(o_rays))
    ok = ~np.isnan(o_rays)
    assert np.max(np.abs(rays[ok] - o_rays[ok])) <= 1e-12 * max(1.0, np.max(np.abs(o_rays[ok])))
    rs2 = np.random.RandomState(0)
    rs2.set_state(('MT19937',) + tuple(st))
    assert rs2.random_sample() == helpers.state_next_double(o_st)


@pytest.mark.gpu
@pytest.mark.parametrize('route', ['one_pass', 'two_pass', 'parts', 'parts_two_pass', 'many_chunks'])
@pytest.mark.parametrize('segments', ['3', '7'])
@pytest.mark.parametrize('name', ['C_sphere_runs_iters', 'B_mirror_runs', 'C_sphere_2e5_s3', 'D_ToroidalCrystal_1e5',
                                  'E_mesh_interp_counts', 'A_example00_1e5', 'S_focused_1e5', 'W_voigt_1e5',
                                  'P_aperture_1e5', 'D_PlanarCrystal_bragg_1e5', 'G_flat_xy_5e4'])
def test_segmented_runs_equal_reference(name, segments, route, monkeypatch):
    """Runs split into work units -- segments with their own jump-positioned heads, parts of a segment that walk the
    segment's heads to their rays -- give the reference's integers (golden num_out and images, several iterations
    included) on both routes: one pass (candidates parked in HBM, stream offsets by look-back over the units in front)
    and two passes (count, then propagate), with few and with many chunk heads in the Bragg stream."""
    if name not in helpers.golden_names('counts'):
        pytest.skip('no such golden')
    monkeypatch.setenv('XICSRT_SEGMENTS', segments)
    if route in ('two_pass', 'parts_two_pass'):
        monkeypatch.setenv('XICSRT_SEG_TWO_PASS', '1')
    if route in ('parts', 'parts_two_pass'):
        monkeypatch.setenv('XICSRT_SUBUNITS', '3')
    if route == 'many_chunks':
        monkeypatch.setenv('XICSRT_CHUNK_HEADS', '11')
        monkeypatch.setenv('XICSRT_SUBUNITS', '2')
    cfg, gold = helpers.load_golden(name)
    config, elements, flat = helpers.build(cfg)
    g = config['general']
    seeds = xrt.run_seeds(g['random_seed'], g['number_of_runs'])
    dev = xrt.DeviceTrace(flat)
    dev.trace(seeds, g['number_of_iter'], keep_images=True)
    meta, image = dev.results()
    for nm in flat.names:
        assert int(meta[nm]['num_out']) == int(gold['num_out/' + nm]), nm
    for nm in flat.names[1:]:
        if image[nm] is not None:
            assert np.array_equal(image[nm].astype(np.int64), gold['image/' + nm]), nm
    path = capi.lib().xrt_last_path(1)
    if path & capi.PATH_SEGMENTED:
        has_bragg = any(o.interact == 2 and (o.flags & helpers.xscene.F_CHECK_BRAGG) for o in flat.struct.optics[:flat.struct.n_optics])
        assert bool(path & capi.PATH_ONE_PASS) == (has_bragg and route not in ('two_pass', 'parts_two_pass')), (path, route)


@pytest.mark.gpu
@pytest.mark.parametrize('route', ['split', 'split_tables_in_global', 'split_every_face', 'split_batches', 'split_list_walk', 'split_own_origins',
                                   'fused'])
@pytest.mark.parametrize('name', ['E_cfg5_mesh_flat_1e5', 'E_cfg5_mesh_interp_1e5', 'E_mesh_interp_counts', 'E_mesh_norefine_counts',
                                  'E_mesh_81_coarse17_counts', 'E_cfg5_wide_flat_1e5', 'E_cfg5_wide_interp_1e5', 'E_cfg5_wide_norefine_1e5'])
def test_mesh_crystal_routes_equal_reference(name, route, monkeypatch):
    """A mesh crystal that makes the Bragg test goes through three launches -- rays up to the first pass over the faces
    (behind a point source: through the grid over the directions), the rest of ShapeMesh.intersect per parked ray (small
    meshes: tables in LDS, faces classified from their vertices; first from the fan of faces around the nearest point, with the
    tables in LDS or in global memory; interpolation in a launch of its own), Bragg test and
    the elements behind.  Every variant of it, and the one-kernel route, give the reference's integers."""
    if route == 'fused':
        monkeypatch.setenv('XICSRT_NO_MESH_SPLIT', '1')
    if route == 'split_tables_in_global':
        monkeypatch.setenv('XICSRT_NO_MESH_LDS', '1')
    if route == 'split_every_face':
        monkeypatch.setenv('XICSRT_NO_DIR_GRID', '1')
    cfg, gold = helpers.load_golden(name)
    if route == 'split_own_origins':                                # every parked ray keeps its origin in its record (as behind an extended source)
        monkeypatch.setenv('XICSRT_NO_SHARED_ORIGIN', '1')
    if route == 'split_list_walk':                                  # without the fans: every parked ray walks its point's face list
        monkeypatch.setenv('XICSRT_NO_MESH_FANS', '1')
    config, elements, flat = helpers.build(cfg)
    if route == 'split_batches':                                    # a budget that holds one of the two runs: batches of runs
        cap = (flat.n_rays + 255) // 256 * 256
        whole = capi.lib().xrt_workspace_bytes(flat.byref(), 2)
        # (36 - 96 bytes per parked ray, by the mesh and the source; a single run's region holds 64 B per ray at least: what a
        #  history call of the scene parks)
        for mb in range(int(2 * cap * 100) // (1 << 20) + 2, 0, -1):
            monkeypatch.setenv('XICSRT_WORKSPACE_BUDGET_MB', str(mb))
            if capi.lib().xrt_workspace_bytes(flat.byref(), 2) < whole - cap * 4:
                break
        assert capi.lib().xrt_workspace_bytes(flat.byref(), 2) < whole - cap * 4
    g = config['general']
    seeds = xrt.run_seeds(g['random_seed'], g['number_of_runs'])
    capi.lib().xrt_last_path(1)
    dev = xrt.DeviceTrace(flat)
    dev.trace(seeds, g['number_of_iter'], keep_images=True)
    meta, image = dev.results()
    for nm in flat.names:
        assert int(meta[nm]['num_out']) == int(gold['num_out/' + nm]), nm
    for nm in flat.names[1:]:
        if image[nm] is not None:
            assert np.array_equal(image[nm].astype(np.int64), gold['image/' + nm]), nm
    path = capi.lib().xrt_last_path(1)
    assert bool(path & capi.PATH_MESH_SPLIT) == (route != 'fused'), (path, route)
    if route in ('split_list_walk', 'fused') or name in ('E_mesh_norefine_counts', 'E_cfg5_wide_norefine_1e5'):
        assert not (path & capi.PATH_MESH_FANS), (path, route)          # (no coarse level: no fans)
    else:
        # (also with the tables in global memory -- a mesh beyond the LDS, 81 x 81 points, or XICSRT_NO_MESH_LDS: xrt_mesh_star_kernel)
        assert path & capi.PATH_MESH_FANS, (path, route)


@pytest.mark.gpu
def test_few_long_runs_are_segmented_and_equal_the_oracle():
    """3 runs x 3e6 rays x 2 iterations of the bench scene: the library segments them on its own;
    counts and both images equal the oracle's, and the next iteration continues the right streams."""
    import bench
    config = xconfig.get_config(bench.spectrometer_config(3000000, 3, seed=21))
    config['general']['number_of_iter'] = 2
    elements = xrt.Elements(config)
    flat = elements.flatten()
    seeds = xrt.run_seeds(21, 3)
    dev = xrt.DeviceTrace(flat)
    dev.trace(seeds, 2, keep_images=True)
    meta, image = dev.results()
    o_num, o_img = helpers.oracle_counts(flat, seeds, 2, threads=3)
    assert [int(meta[nm]['num_out']) for nm in flat.names] == [int(v) for v in o_num]
    got = np.concatenate([image[nm].ravel() for nm in flat.names[1:]]).astype(np.int64)
    assert np.array_equal(got, o_img[:flat.image_bins])


@pytest.mark.gpu
@pytest.mark.parametrize('segments', ['4', '1'])
@pytest.mark.parametrize('bragg', [False, True])
def test_segmented_runs_chain_iterations(bragg, segments, monkeypatch):
    """Segmented (4) and whole (1) runs over several iterations (the stream head an iteration leaves must be the
    one the next iteration's jump expects), with and without Bragg draws (without: the kernel that counts its
    pixels in LDS and never draws from the stream head): equal to the oracle."""
    monkeypatch.setenv('XICSRT_SEGMENTS', segments)
    cfg = _spectrometer(20000, 2, seed=77, check_bragg=bragg, rocking_fwhm=3e-3)
    cfg['general']['number_of_iter'] = 3
    config, elements, flat = helpers.build(cfg)
    seeds = xrt.run_seeds(77, 2)
    n_gpu, i_gpu = _trace(flat, seeds, 3)
    n_cpu, i_cpu = helpers.oracle_counts(flat, seeds, 3)
    assert np.array_equal(n_gpu, n_cpu), (n_gpu, n_cpu)
    assert np.array_equal(i_gpu[:flat.image_bins], i_cpu[:flat.image_bins])


def _thermal(n_rays, runs, seed, iters=1):
    cfg = _spectrometer(n_rays, runs, seed=seed, rocking_fwhm=2e-3)
    cfg['sources']['source'].update(temperature=1500.0, mass_number=39.948, linewidth=0.0)
    cfg['general']['number_of_iter'] = iters
    return cfg


@pytest.mark.gpu
@pytest.mark.parametrize('n_rays,segments', [(300000, None), (100000, '5'), (100001, None), (4096, '3')])
def test_gaussian_wavelengths_prepared_or_staged_equal_oracle(n_rays, segments, monkeypatch):
    """np.random.normal wavelengths: even ray counts go through the prepared array + fused kernels (chunked
    candidate stream, several iterations chained), odd ones through the staged path; both equal the oracle."""
    if segments:
        monkeypatch.setenv('XICSRT_SEGMENTS', segments)
    cfg = _thermal(n_rays, 3, seed=31, iters=2)
    config, elements, flat = helpers.build(cfg)
    seeds = xrt.run_seeds(31, 3)
    n_gpu, i_gpu = _trace(flat, seeds, 2)
    n_cpu, i_cpu = helpers.oracle_counts(flat, seeds, 2, threads=3)
    assert np.array_equal(n_gpu, n_cpu), (n_gpu, n_cpu)
    assert np.array_equal(i_gpu[:flat.image_bins], i_cpu[:flat.image_bins])


@pytest.mark.gpu
def test_gaussian_wavelengths_with_a_cached_value_pending():
    """A generator state with a cached gauss value (has_gauss = 1) cannot be served by the pairwise prepared
    array: the library falls back to the staged path and still equals the oracle ray for ray."""
    cfg = _thermal(20000, 1, seed=5)
    config, elements, flat = helpers.build(cfg)
    rs = np.random.RandomState(77)
    rs.standard_normal(3)                                   # leaves one value cached
    st = rs.get_state()
    assert st[3] == 1
    state = helpers.xscene.RngState()
    import ctypes as C
    C.memmove(state.key, np.ascontiguousarray(st[1], dtype=np.uint32).ctypes.data, 624 * 4)
    state.pos, state.has_gauss, state.gauss = int(st[2]), int(st[3]), float(st[4])
    dev = xrt.DeviceTrace(flat)
    rays, mask, st_out = dev.trace_history((st[1], st[2], st[3], st[4]))
    o_num, o_img, o_rays, o_mask, o_st = helpers.oracle_history(flat, state)
    assert np.array_equal(mask, o_mask)
    both = ~np.isnan(o_rays)
    assert np.array_equal(np.isnan(rays), np.isnan(o_rays))
    assert np.max(np.abs(rays[both] - o_rays[both])) <= 1e-12
    assert (st_out[1], st_out[2]) == (int(o_st.pos), int(o_st.has_gauss))


@pytest.mark.gpu
@pytest.mark.parametrize('env', ['XICSRT_NO_JUMP', 'XICSRT_NO_STAGE_SPLIT', 'XICSRT_STAGED_GAUSS', 'XICSRT_PLASMA_STAGED', 'XICSRT_NO_LDS_BINS'])
@pytest.mark.parametrize('name', ['C_sphere_runs_iters', 'W_normal_1e5', 'F_plasma_counts', 'F_datafile_filter_counts',
                                  'Q_four_counts', 'B_mirror_runs', 'A_example00_1e5', 'B_cfg2_mirror_1e6'])
def test_alternative_device_paths_equal_reference(name, env, monkeypatch):
    """The library's fallbacks (sequential walk instead of jump-ahead, one-launch staged kernel, staged Gaussian
    wavelengths) are alternative routes to the same integers."""
    from xicsrt_amd import capi
    cfg, gold = helpers.load_golden(name)
    config, elements, flat = helpers.build(cfg)
    g = config['general']
    seeds = xrt.run_seeds(g['random_seed'], g['number_of_runs'])
    lib = capi.lib()
    # the route the scene takes by itself, then the one the switch forces: they must differ where the
    # switch applies to the scene (the switches are read at every call, not latched at first use)
    dev0 = xrt.DeviceTrace(flat)
    lib.xrt_last_path(1)
    dev0.trace(seeds, g['number_of_iter'], keep_images=True)
    dev0.results()
    default_path = lib.xrt_last_path(1)
    monkeypatch.setenv(env, '1')
    dev = xrt.DeviceTrace(flat)
    dev.trace(seeds, g['number_of_iter'], keep_images=True)
    meta, image = dev.results()
    forced_path = lib.xrt_last_path(1)
    if env == 'XICSRT_NO_JUMP' and (default_path & capi.PATH_JUMP):
        # heads by walking the stream; a scene whose Gaussian wavelengths were prepared through jump-positioned
        # chunk heads goes to the staged path instead
        assert not (forced_path & (capi.PATH_JUMP | capi.PATH_SEGMENTED))
        assert forced_path & (capi.PATH_SEEK | capi.PATH_STAGED)
    if env == 'XICSRT_PLASMA_STAGED' and (default_path & capi.PATH_PLASMA_SCOUT):
        assert (forced_path & capi.PATH_STAGED) and not (forced_path & capi.PATH_PLASMA_SCOUT)
    if env == 'XICSRT_NO_STAGE_SPLIT' and (default_path & capi.PATH_STAGED):
        assert (default_path & capi.PATH_STAGE_SPLIT) and not (forced_path & capi.PATH_STAGE_SPLIT)
    if env == 'XICSRT_NO_STAGE_SPLIT' and (default_path & capi.PATH_PLASMA_SCOUT):
        # the staged kernels of a plasma scene: both switches together
        monkeypatch.setenv('XICSRT_PLASMA_STAGED', '1')
        dev2 = xrt.DeviceTrace(flat)
        dev2.trace(seeds, g['number_of_iter'], keep_images=True)
        meta, image = dev2.results()
        forced_path = lib.xrt_last_path(1)
        assert (forced_path & capi.PATH_STAGED) and not (forced_path & capi.PATH_STAGE_SPLIT)
    if env == 'XICSRT_NO_LDS_BINS':
        # scenes without a Bragg test count their pixels in LDS first (when the bins fit); the switch sends them to the u64 bins directly
        assert bool(default_path & capi.PATH_LDS_BINS) == name.startswith(('B_', 'A_'))
        assert not (forced_path & capi.PATH_LDS_BINS)
    if env == 'XICSRT_STAGED_GAUSS' and (default_path & capi.PATH_GAUSS_PREPARED):
        assert (forced_path & capi.PATH_STAGED) and not (forced_path & capi.PATH_GAUSS_PREPARED)
    for nm in flat.names:
        assert int(meta[nm]['num_out']) == int(gold['num_out/' + nm]), nm
    for nm in flat.names[1:]:
        if image[nm] is not None:
            assert np.array_equal(image[nm].astype(np.int64), gold['image/' + nm]), nm


# ---------------------------------------------------------------------------------------------------------
# BASELINE.json configurations at their stated sizes.  Their geometry is pinned against the reference by
# the goldens B_cfg2_mirror_1e6 / C_cfg3_sphere_1e6 / F_cfg4_plasma_counts / E_cfg5_mesh_*_1e5 (small run
# counts, reference output); here the full-size jobs are held to the identities that need no oracle
# (histogram sums = counters, repetition, partition of the runs) and a subset of their runs to the oracle.
# ---------------------------------------------------------------------------------------------------------

def _full_size(name, runs, per_run_update=None):
    cfg, _ = helpers.load_golden(name)
    cfg = copy.deepcopy(cfg)
    cfg['general'].update(number_of_runs=runs, number_of_iter=1)
    if per_run_update:
        cfg['sources']['source'].update(per_run_update)
    return cfg


def _invariants(flat, seeds, parts, oracle_runs, threads=16):
    n_all, i_all = _trace(flat, seeds)
    for k, name in enumerate(flat.names[1:], start=1):
        sl = flat.image_slices[name]
        if sl is None:
            continue
        off, nx, ny = sl
        obj = flat.optic_objs[k - 1]
        if obj.param.get('check_size', True) and obj.param.get('xsize') and obj.param.get('ysize'):
            # every ray that leaves a size-checked element lands inside its own pixel grid
            assert int(i_all[off:off + nx * ny].sum()) == int(n_all[k]), name
    n_again, i_again = _trace(flat, seeds)
    assert np.array_equal(n_all, n_again) and np.array_equal(i_all, i_again)
    n_sum, i_sum = np.zeros_like(n_all), np.zeros_like(i_all)
    lo = 0
    for hi in list(parts) + [len(seeds)]:
        n, i = _trace(flat, seeds[lo:hi])
        n_sum += n
        i_sum += i
        lo = hi
    assert np.array_equal(n_sum, n_all) and np.array_equal(i_sum, i_all)
    sub = seeds[:oracle_runs]
    n_gpu, i_gpu = _trace(flat, sub)
    n_cpu, i_cpu = helpers.oracle_counts(flat, sub, 1, threads=threads)
    assert np.array_equal(n_gpu, n_cpu)
    assert np.array_equal(i_gpu[:flat.image_bins], i_cpu[:flat.image_bins])
    return n_all


def test_cfg2_planar_mirror_1e8_photons():
    """BASELINE cfg2: point source -> planar mirror -> detector, 100 runs x 1e6 rays."""
    from xicsrt_amd import capi
    config, elements, flat = helpers.build(_full_size('B_cfg2_mirror_1e6', 100))
    seeds = xrt.run_seeds(config['general']['random_seed'], 100)
    capi.lib().xrt_last_path(1)
    n = _invariants(flat, seeds, parts=(7, 60), oracle_runs=8)
    assert capi.lib().xrt_last_path(1) & capi.PATH_LDS_BINS           # pixel bins pre-aggregated in LDS
    assert int(n[0]) == 10 ** 8 and 0.51 < n[1] / n[0] < 0.53 and 0.31 < n[2] / n[0] < 0.325
    # the first two runs are the reference's own B_cfg2_mirror_1e6 job
    cfg, gold = helpers.load_golden('B_cfg2_mirror_1e6')
    n2, _ = _trace(flat, seeds[:2])
    assert [int(v) for v in n2] == [int(gold['num_out/' + nm]) for nm in flat.names]


def test_cfg4_plasma_2000_bundles_800x400_detector():
    """BASELINE cfg4 per GPU: 2000 bundles per run, Poisson statistics, 800 x 400-bin detector; 1024 runs
    (1.5e8 photons; the 1e10 of the 8-GPU statement are 65 536 such runs, sharded by run index)."""
    config, elements, flat = helpers.build(_full_size('F_cfg4_plasma_counts', 1024))
    off, nx, ny = flat.image_slices['detector']
    assert (nx, ny) == (800, 400) and flat.struct.source.bundle_count == 2000
    seeds = xrt.run_seeds(config['general']['random_seed'], 1024)
    n = _invariants(flat, seeds, parts=(100, 700), oracle_runs=16)
    assert 1.4e8 < int(n[0]) < 1.7e8
    cfg, gold = helpers.load_golden('F_cfg4_plasma_counts')
    n3, _ = _trace(flat, seeds[:3])
    assert [int(v) for v in n3] == [int(gold['num_out/' + nm]) for nm in flat.names]


@pytest.mark.parametrize('tag', ['flat', 'interp'])
def test_cfg5_toroidal_mesh_crystal_41x41_1e9_photons(tag):
    """BASELINE cfg5: 41 x 41 toroidal mesh crystal, 1000 runs x 1e6 rays (interpolation off and on)."""
    config, elements, flat = helpers.build(_full_size('E_cfg5_mesh_%s_1e5' % tag, 1000, {'intensity': 1000000}))
    seeds = xrt.run_seeds(config['general']['random_seed'], 1000)
    n_all, i_all = _trace(flat, seeds)
    assert int(n_all[0]) == 10 ** 9
    off, nx, ny = flat.image_slices['detector']
    assert int(i_all[off:off + nx * ny].sum()) == int(n_all[2])
    # partition of the runs (what the multi-GPU path relies on)
    n_sum, i_sum = np.zeros_like(n_all), np.zeros_like(i_all)
    for lo, hi in ((0, 333), (333, 1000)):
        n, i = _trace(flat, seeds[lo:hi])
        n_sum += n
        i_sum += i
    assert np.array_equal(n_sum, n_all) and np.array_equal(i_sum, i_all)
    # four of the runs against the oracle (the mesh path is ~100x slower on the CPU)
    n_gpu, i_gpu = _trace(flat, seeds[:4])
    n_cpu, i_cpu = helpers.oracle_counts(flat, seeds[:4], 1, threads=4)
    assert np.array_equal(n_gpu, n_cpu) and np.array_equal(i_gpu[:flat.image_bins], i_cpu[:flat.image_bins])


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['flat', 'interp', 'norefine'])
def test_cfg5_geometry_with_many_reflections_equals_the_oracle(tag):
    """The cfg5 geometry with a rocking curve wide enough to reflect (goldens E_cfg5_wide_*: 2.6 - 2.8e3 detector hits per 2e5 rays,
    where cfg5's own 48 urad curve gives 0 - 7): 64 runs x 1e6 rays against the oracle -- the three counters and all 15 000
    pixels -- and the reference's golden on its own two runs.  (What tests/soak_mesh.py does by hand.)"""
    name = 'E_cfg5_wide_%s_1e5' % tag
    config, elements, flat = helpers.build(_full_size(name, 64, {'intensity': 1000000}))
    seeds = xrt.run_seeds(config['general']['random_seed'], 64)
    capi.lib().xrt_last_path(1)
    n_gpu, i_gpu = _trace(flat, seeds)
    assert capi.lib().xrt_last_path(1) & capi.PATH_MESH_SPLIT
    n_cpu, i_cpu = helpers.oracle_counts(flat, seeds, 1, threads=16)
    assert int(n_gpu[0]) == 64 * 10 ** 6 and int(n_gpu[2]) > 500000
    assert np.array_equal(n_gpu, n_cpu)
    assert flat.image_bins == 15000 and np.array_equal(i_gpu[:flat.image_bins], i_cpu[:flat.image_bins])
    cfg, gold = helpers.load_golden(name)
    config2, elements2, flat2 = helpers.build(cfg)
    n2, i2 = _trace(flat2, xrt.run_seeds(config2['general']['random_seed'], 2))
    assert [int(v) for v in n2] == [int(gold['num_out/' + nm]) for nm in flat2.names]
    for nm in flat2.names[1:]:
        off, nx, ny = flat2.image_slices[nm]
        assert np.array_equal(i2[off:off + nx * ny].reshape(nx, ny), gold['image/' + nm]), nm


@pytest.mark.gpu
@pytest.mark.parametrize('name', ['E_mesh_norefine_counts', 'E_mesh_81_coarse17_counts'])
def test_meshes_with_many_first_pass_faces_at_full_size(name, monkeypatch):
    """First passes over 3200 faces (41 x 41, mesh_refine off) and 512 faces (81 x 81 behind a 17 x 17 coarse level),
    1000 runs x 1e6 rays: the device finds a ray's faces through an x-y grid over them; the same integers as the walk
    over every face, the oracle's on a few runs, the reference's on the small golden."""
    config, elements, flat = helpers.build(_full_size(name, 1000, {'intensity': 1000000}))
    seeds = xrt.run_seeds(config['general']['random_seed'], 1000)
    n_all, i_all = _trace(flat, seeds)
    assert int(n_all[0]) == 10 ** 9
    off, nx, ny = flat.image_slices['detector']
    assert int(i_all[off:off + nx * ny].sum()) == int(n_all[2])
    n_gpu, i_gpu = _trace(flat, seeds[:3])
    n_cpu, i_cpu = helpers.oracle_counts(flat, seeds[:3], 1, threads=3)
    assert np.array_equal(n_gpu, n_cpu) and np.array_equal(i_gpu[:flat.image_bins], i_cpu[:flat.image_bins])
    monkeypatch.setenv('XICSRT_NO_FACE_GRID', '1')
    n_walk, i_walk = _trace(flat, seeds[:40])
    monkeypatch.delenv('XICSRT_NO_FACE_GRID')
    n_grid, i_grid = _trace(flat, seeds[:40])
    assert np.array_equal(n_walk, n_grid) and np.array_equal(i_walk, i_grid)


def test_plasma_runs_in_several_batches(monkeypatch):
    """More runs than run slots for the scout's stream dumps: batches of slots, same integers."""
    cfg, gold = helpers.load_golden('F_cfg4_plasma_counts')
    cfg = copy.deepcopy(cfg)
    cfg['general'].update(number_of_runs=37)
    config, elements, flat = helpers.build(cfg)
    seeds = xrt.run_seeds(config['general']['random_seed'], 37)
    n_one, i_one = _trace(flat, seeds)
    monkeypatch.setenv('XICSRT_PLASMA_SLOTS', '8')
    n_many, i_many = _trace(flat, seeds)
    assert np.array_equal(n_one, n_many) and np.array_equal(i_one, i_many)
    n_cpu, i_cpu = helpers.oracle_counts(flat, seeds, 1, threads=8)
    assert np.array_equal(n_many, n_cpu) and np.array_equal(i_many[:flat.image_bins], i_cpu[:flat.image_bins])


@pytest.mark.gpu
def test_lds_pixel_bins_cannot_wrap():
    """A pencil beam puts every ray of a run into one pixel: the 16-bit counters of the workgroup's LDS copy of the
    bins are flushed before they can wrap (every 255 tiles), also across segments and runs."""
    from xicsrt_amd import capi
    cfg, _ = helpers.load_golden('A_example00_1e5')
    cfg = copy.deepcopy(cfg)
    cfg['general'].update(number_of_runs=3, number_of_iter=2)
    cfg['sources']['source'].update(intensity=300000, spread=0.0)
    config, elements, flat = helpers.build(cfg)
    seeds = xrt.run_seeds(config['general']['random_seed'], 3)
    capi.lib().xrt_last_path(1)
    n_gpu, i_gpu = _trace(flat, seeds, n_iter=2)
    assert capi.lib().xrt_last_path(1) & capi.PATH_LDS_BINS
    assert int(n_gpu[0]) == 1800000 and int(n_gpu[1]) == 1800000
    assert int(i_gpu[:flat.image_bins].max()) == 1800000 and int(i_gpu[:flat.image_bins].sum()) == 1800000
    n_cpu, i_cpu = helpers.oracle_counts(flat, seeds, 2, threads=3)
    assert np.array_equal(n_gpu, n_cpu) and np.array_equal(i_gpu[:flat.image_bins], i_cpu[:flat.image_bins])


@pytest.mark.gpu
@pytest.mark.parametrize('n_rays', [2500, 30000])
@pytest.mark.parametrize('words', [1, 79, 111, 112, 400, 625])
def test_history_from_any_generator_position(n_rays, words):
    """xrt_trace_history from a numpy state whose position is anywhere in its block (what the lost-ray shuffle between
    two iterations leaves): the fused kernel's heads read up to 112 words behind their position, which for pos < 112
    lie in front of the imported block and are reconstructed by running MT19937 backwards.  Whole runs (2500 rays)
    and segmented ones (30000), against the oracle.  (Found by tests/fuzz_raytrace.py.)"""
    import ctypes as C
    config, elements, flat = helpers.build(_spectrometer(n_rays, 1, seed=31, rocking_fwhm=3e-3))
    rng = np.random.RandomState(31)
    for _ in range(words):
        rng.bytes(4)                                    # one 32-bit word each
    st = rng.get_state()
    assert st[2] == (words if words < 624 else words - 624)
    state = helpers.xscene.RngState()
    C.memmove(state.key, np.ascontiguousarray(st[1], dtype=np.uint32).ctypes.data, 624 * 4)
    state.pos, state.has_gauss, state.gauss = int(st[2]), int(st[3]), float(st[4])
    dev = xrt.DeviceTrace(flat)
    dev._workspace(1)[0].fill_(0xA5)                    # nothing useful may be lying around in the workspace
    rays, mask, st_out = dev.trace_history((st[1], st[2], st[3], st[4]))
    o_num, o_img, o_rays, o_mask, o_st = helpers.oracle_history(flat, state)
    assert np.array_equal(mask, o_mask)
    assert np.array_equal(np.isnan(rays), np.isnan(o_rays))
    both = ~np.isnan(o_rays)
    assert np.max(np.abs(rays[both] - o_rays[both])) <= 1e-12
    rs2 = np.random.RandomState(0)
    rs2.set_state(('MT19937',) + tuple(st_out))
    assert rs2.random_sample() == helpers.state_next_double(o_st)


@pytest.mark.gpu
def test_runs_just_above_the_resident_workgroups_split_off_a_tail(monkeypatch):
    """1060 runs on 1024 workgroup slots: the unsegmented launch leaves the last 36 runs to a second pass (segmented
    route) instead of letting them hold a slot each for a whole run time; same integers either way, equal to the oracle."""
    from xicsrt_amd import capi
    cfg = _spectrometer(200000, 1060, seed=41, rocking_fwhm=3e-3)       # (only runs of >= 2e5 rays are worth a second pass)
    cfg['general']['number_of_iter'] = 2
    config, elements, flat = helpers.build(cfg)
    seeds = xrt.run_seeds(41, 1060)
    lib = capi.lib()
    lib.xrt_last_path(1)
    n_split, i_split = _trace(flat, seeds, 2)
    assert lib.xrt_last_path(1) & capi.PATH_SEGMENTED
    monkeypatch.setenv('XICSRT_NO_TAIL_SPLIT', '1')
    n_whole, i_whole = _trace(flat, seeds, 2)
    assert not (lib.xrt_last_path(1) & capi.PATH_SEGMENTED)
    assert np.array_equal(n_split, n_whole) and np.array_equal(i_split, i_whole)
    n_cpu, i_cpu = helpers.oracle_counts(flat, seeds, 2, threads=16)
    assert np.array_equal(n_split, n_cpu) and np.array_equal(i_split[:flat.image_bins], i_cpu[:flat.image_bins])


@pytest.mark.gpu
def test_tail_pass_fits_the_workspace_on_a_narrow_grid(monkeypatch):
    """A launch grid of 512 workgroups (two per CU, what the mesh / local-frame variant gets; forced here) leaves
    the last 2 of 514 runs to a second pass, which is segmented and lays the workspace out differently (more heads per
    run, jump tables): xrt_workspace_bytes covers that layout too, and the result equals the oracle's."""
    monkeypatch.setenv('XICSRT_MAX_WG_PER_CU', '2')
    import bench
    config = xconfig.get_config(bench.spectrometer_config(200000, 514, seed=11))
    flat = xrt.Elements(config).flatten()
    seeds = xrt.run_seeds(11, 514)
    lib = capi.lib()
    # (the second pass parks its candidates in HBM when they fit a 2 GiB share of the workspace -- here they do -- and
    #  counts in a pass of its own otherwise; both layouts are inside what the library asks for)
    assert lib.xrt_workspace_bytes(flat.byref(), 514) >= lib.xrt_workspace_bytes(flat.byref(), 2)
    dev = xrt.DeviceTrace(flat)
    lib.xrt_last_path(1)
    dev.trace(seeds, 1, keep_images=True)
    meta, image = dev.results()
    assert lib.xrt_last_path(1) & capi.PATH_SEGMENTED             # the tail went the segmented way
    o_num, o_img = helpers.oracle_counts(flat, seeds, 1, threads=16)
    assert [int(meta[nm]['num_out']) for nm in flat.names] == [int(v) for v in o_num]
    got = np.concatenate([image[nm].ravel() for nm in flat.names[1:]]).astype(np.int64)
    assert np.array_equal(got, o_img[:flat.image_bins])



@pytest.mark.parametrize('name', ['M_planar_mosaic_counts', 'M_spherical_mosaic_cutoff_1e5'])
def test_mosaic_layers_over_parked_rays_equal_reference(name, monkeypatch):
    """A mosaic crystal's layers over the rays the fused kernel's first phase parked (xrt_mosaic_kernel; scenes of at least
    65536 rays by themselves, smaller ones when XICSRT_MOSAIC_FUSED_MIN says so) and the staged kernel are two routes to the
    same integers: the reference's (optics/_InteractMosaicCrystal.py:53-139), runs x iterations and the cut-off included."""
    from xicsrt_amd import capi
    cfg, gold = helpers.load_golden(name)
    config, elements, flat = helpers.build(cfg)
    g = config['general']
    seeds = xrt.run_seeds(g['random_seed'], g['number_of_runs'])
    lib = capi.lib()
    got = {}
    for route in ('parked', 'staged'):
        if route == 'parked':
            monkeypatch.setenv('XICSRT_MOSAIC_FUSED_MIN', '1')
        else:
            monkeypatch.setenv('XICSRT_NO_MOSAIC_FUSED', '1')
        dev = xrt.DeviceTrace(flat)
        lib.xrt_last_path(1)
        dev.trace(seeds, g['number_of_iter'], keep_images=True)
        meta, image = dev.results()
        path = lib.xrt_last_path(1)
        if route == 'parked':
            assert (path & capi.PATH_MOSAIC_FUSED) and not (path & capi.PATH_STAGED)
        else:
            assert (path & capi.PATH_STAGED) and not (path & capi.PATH_MOSAIC_FUSED)
        for nm in flat.names:
            assert int(meta[nm]['num_out']) == int(gold['num_out/' + nm]), (route, nm)
        for nm in flat.names[1:]:
            if image[nm] is not None:
                assert np.array_equal(image[nm].astype(np.int64), gold['image/' + nm]), (route, nm)
        got[route] = dev.images.cpu().numpy().copy()
    assert np.array_equal(got['parked'], got['staged'])
    if g['number_of_runs'] >= 2:
        # a budget that holds one run's slot: the parked-ray route goes through the runs in batches
        monkeypatch.delenv('XICSRT_NO_MOSAIC_FUSED')
        monkeypatch.setenv('XICSRT_WORKSPACE_BUDGET_MB', str(max(1, int(flat.n_rays * 92 * 1.5) >> 20)))
        dev = xrt.DeviceTrace(flat)
        lib.xrt_last_path(1)
        dev.trace(seeds, g['number_of_iter'], keep_images=True)
        meta, image = dev.results()
        assert lib.xrt_last_path(1) & capi.PATH_MOSAIC_FUSED
        for nm in flat.names:
            assert int(meta[nm]['num_out']) == int(gold['num_out/' + nm]), ('batches', nm)
        assert np.array_equal(dev.images.cpu().numpy(), got['parked'])


def test_mosaic_layers_over_parked_rays_random_scenes(monkeypatch):
    """Random mosaic scenes (planar / spherical, cut-off, no Bragg test, extended and focused sources, wavelength
    distributions, several runs and iterations) through xrt_mosaic_kernel: counts and images of the oracle."""
    from xicsrt_amd import capi
    import fuzz_parity
    monkeypatch.setenv('FUZZ_KIND', '6')
    monkeypatch.setenv('FUZZ_MAX_ITER', '3')
    monkeypatch.setenv('FUZZ_MAX_RUNS', '5')
    monkeypatch.setenv('XICSRT_MOSAIC_FUSED_MIN', '1')
    lib = capi.lib()
    taken = 0
    for case in range(60):
        rs = np.random.RandomState(880000 + case)
        cfg = fuzz_parity.scene(rs)
        cfg['optics']['crystal']['trace_local'] = False
        cfg['sources']['source']['intensity'] = int(rs.choice([3000, 20000, 70000]))
        try:
            config, elements, flat = helpers.build(cfg)
        except Exception:
            continue
        g = config['general']
        seeds = xrt.run_seeds(g['random_seed'], g['number_of_runs'])
        try:
            n_cpu, i_cpu = helpers.oracle_counts(flat, seeds, g['number_of_iter'], threads=8)
        except AssertionError:
            continue
        dev = xrt.DeviceTrace(flat)
        lib.xrt_last_path(1)
        dev.trace(seeds, g['number_of_iter'])
        meta, image = dev.results()
        path = lib.xrt_last_path(1)
        taken += bool(path & capi.PATH_MOSAIC_FUSED)
        n_gpu = np.array([int(meta[nm]['num_out']) for nm in flat.names])
        assert np.array_equal(n_gpu, n_cpu), (case, n_gpu, n_cpu)
        assert np.array_equal(dev.images.cpu().numpy()[:flat.image_bins], i_cpu[:flat.image_bins]), case
    assert taken >= 20          # (sources with normal deviates or filters stay with the staged kernel)


def test_mosaic_pixel_counters_in_lds_spill_into_the_bins(monkeypatch):
    """xrt_mosaic_kernel counts the reflected rays' pixel hits in 16-bit LDS counters; the lane whose add takes a counter to 0x8000
    moves 0x8000 hits on into the u64 bin.  Images of a few coarse pixels: > 1e5 hits per counter and run, several crossings."""
    import bench
    from xicsrt_amd import capi
    cfg = bench.spectrometer_config(200000, 3, seed=11)
    cfg['optics']['crystal'].update(class_name='XicsrtOpticSphericalMosaicCrystal', mosaic_spread=float(np.radians(0.4)), mosaic_depth=8,
                                    rocking_fwhm=2e-3, pixel_size=0.2)
    cfg['optics']['detector'].update(pixel_size=0.4)
    config = xconfig.get_config(cfg)
    flat = xrt.Elements(config).flatten()
    assert flat.image_bins < 64
    seeds = xrt.run_seeds(11, 3)
    n_cpu, i_cpu = helpers.oracle_counts(flat, seeds, 1, threads=3)
    assert i_cpu[:flat.image_bins].max() > 6 * 0x8000         # (three runs: each takes its counter over 0x8000 at least twice)
    for lds_bins in (True, False):
        if not lds_bins:
            monkeypatch.setenv('XICSRT_NO_LDS_BINS', '1')
        dev = xrt.DeviceTrace(flat)
        capi.lib().xrt_last_path(1)
        dev.trace(seeds, 1)
        meta, image = dev.results()
        assert capi.lib().xrt_last_path(1) & capi.PATH_MOSAIC_FUSED
        assert [int(meta[nm]['num_out']) for nm in flat.names] == [int(v) for v in n_cpu]
        assert np.array_equal(dev.images.cpu().numpy()[:flat.image_bins], i_cpu[:flat.image_bins]), lds_bins
