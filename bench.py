#!/usr/bin/env python3
"""
Benchmark of the hot path: Mphotons/s launched->detector on the 3-element
spherical-crystal spectrometer (BASELINE.json metric; SURVEY.md section 8d cfg3:
point XicsrtSourceDirected, spread 10 deg -> XicsrtOpticSphericalCrystal with a
gaussian rocking curve -> XicsrtOpticDetector; 1e6 rays/run x 1000 runs = 1e9
photons per step and per GPU, keep_history=False, keep_images=True).

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by the driver through torch.distributed.run (one process per
GPU, RCCL): runs are sharded over the ranks (run i -> rank i mod N, weak
scaling: every rank traces `--runs` runs) and the integer histogram + counters
are summed with one all-reduce per step, inside the timed region.

One JSON line on rank 0.  `roofline.achieved` = algorithmic bytes per launch
(142 B/photon for this scene) / average duration of the
propagation kernel, measured with HIP events on the launch stream inside the
library (xrt_timing_begin/end).  `cpu_baseline` = the CPU oracle
(oracle/xrt_oracle.c, a port of the reference's xicsrt_multiprocessing path)
timed on the host cores on a bounded sample of the same workload, rank 0, N=1.
`raytrace_call_ms` / `raytrace_call_object_construction_ms` (the user-level
xicsrt_amd.raytrace(config) call of the same scene, outside the timed region) are
measured at N=1 without a process group only; in multi-GPU lines they are null.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_PHOTON = 142.0       # 68 + 70.9 + 2.9 + 0.7
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PEAK_CLOCK_GHZ = 2.4                 # MI355X_MICROARCH.md: peak engine clock; VALU issue peak = 1024 SIMDs x clock / 4


def spectrometer_config(n_rays, n_runs, seed=0):
    import numpy as np
    return {
        'general': {'number_of_iter': 1, 'number_of_runs': n_runs, 'random_seed': seed,
                    'keep_history': False, 'keep_images': True, 'print_results': False},
        'sources': {'source': {
            'class_name': 'XicsrtSourceDirected', 'intensity': n_rays, 'wavelength': 3.9492,
            'spread': float(np.radians(10.0)), 'xsize': 0.0, 'ysize': 0.0, 'zsize': 0.0}},
        'optics': {
            'crystal': {
                'class_name': 'XicsrtOpticSphericalCrystal', 'check_size': True,
                'origin': [0.0, 0.0, 0.80374151], 'zaxis': [0.0, 0.59497864, -0.80374151],
                'xsize': 0.2, 'ysize': 0.2, 'radius': 1.0,
                'crystal_spacing': 2.45676, 'rocking_type': 'gaussian', 'rocking_fwhm': 48.070e-6},
            'detector': {
                'class_name': 'XicsrtOpticDetector',
                'origin': [0.0, 0.76871290, 0.56904832], 'zaxis': [0.0, -0.95641806, 0.29200084],
                'xsize': 0.4, 'ysize': 0.2}},
    }


def cpu_baseline(flat, n_rays, target_seconds=8.0):
    """Oracle (port of the reference's pool-over-runs path) on the host cores, bounded sample."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import helpers
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    t0 = time.time()
    helpers.oracle_counts(flat, [12345], 1, threads=1)          # calibrate: one run, one core
    one = max(time.time() - t0, 1e-3)
    runs = int(max(threads, min(4096, round(target_seconds / one) * threads)))
    runs -= runs % threads
    runs = max(runs, threads)
    seeds = list(range(1000, 1000 + runs))
    t0 = time.time()
    helpers.oracle_counts(flat, seeds, 1, threads=threads)
    dt = time.time() - t0
    return {'value': runs * n_rays / dt / 1e6, 'unit': 'Mphotons/s', 'cores': threads, 'kind': 'port',
            'sample': '%d runs x %d rays of the same scene, %d threads (one run per thread at a time), %.1f s'
                      % (runs, n_rays, threads, dt)}


def committed_profile(rays, runs):
    """The newest committed rocprofv3 summary (profiles/r*.json, written by profiles/run_profile.sh) of this
    workload: HBM bytes per launch of the propagation kernel from the FETCH_SIZE / WRITE_SIZE passes and the
    SQ issue counters of one launch."""
    import glob
    for fn in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*.json')), reverse=True):
        try:
            d = json.load(open(fn))
            cfg = d.get('bench_fetch', {}).get('config', {})
            if cfg.get('rays_per_run') == rays and cfg.get('runs_per_gpu') == runs and 'hbm_traffic_bytes_per_launch' in d:
                return d, os.path.relpath(fn, ROOT)
        except Exception:
            continue
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--runs', type=int, default=1000, help='runs per step and per GPU')
    ap.add_argument('--rays', type=int, default=1000000, help='rays per run')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak',
                    help='weak: every rank traces --runs runs; strong: --runs runs in total, sharded over the ranks')
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from xicsrt_amd import xicsrt_raytrace as xrt
    from xicsrt_amd import config as xconfig
    from xicsrt_amd import capi

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # XICSRT_BENCH_FORCE_DIST=1 exercises the RCCL path with a single rank (launched through torchrun)
    use_dist = world > 1 or (os.environ.get('XICSRT_BENCH_FORCE_DIST') == '1' and 'RANK' in os.environ)
    # a line for N GPUs must come from N ranks: refuse, before any GPU work, rather than print a number that is not what it says
    if not use_dist and args.gpus != 1:
        sys.stderr.write('bench.py: --gpus %d needs one process per GPU (torch.distributed.run); no line printed\n' % args.gpus)
        sys.exit(2)
    if use_dist:
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend='nccl', device_id=torch.device('cuda', local_rank))
        if dist.get_world_size() != args.gpus:
            if rank == 0:
                sys.stderr.write('bench.py: --gpus %d but the process group has %d ranks; no line printed\n'
                                 % (args.gpus, dist.get_world_size()))
            dist.destroy_process_group()
            sys.exit(2)
    else:
        torch.cuda.set_device(0)
    lib = capi.lib()

    total_runs = args.runs * world if args.scaling == 'weak' else args.runs
    config = xconfig.get_config(spectrometer_config(args.rays, total_runs))
    elements = xrt.Elements(config)
    flat = elements.flatten()
    seeds = xrt.run_seeds(config['general']['random_seed'], total_runs)
    my_seeds = [seeds[i] for i in xrt.shard_runs(total_runs, rank, world)]
    dev = xrt.DeviceTrace(flat)

    def step():
        dev.num_out.zero_()
        dev.images.zero_()
        dev.trace(my_seeds, 1, keep_images=True)
        if use_dist:
            packed = torch.cat([dev.num_out, dev.images])
            dist.all_reduce(packed, op=dist.ReduceOp.SUM)
            return packed
        return dev.num_out

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    lib.xrt_timing_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    kernel_ms = C.c_double(0.0)
    launches = C.c_int64(0)
    lib.xrt_timing_end(C.byref(kernel_ms), C.byref(launches))
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # What users call: xicsrt_amd.raytrace(config) of the same scene (keep_history=False), whole call on the host clock,
    # and the part of it that is object construction (SURVEY 8d prices the call without it).  Outside the timed region.
    call_ms = setup_ms = None
    if rank == 0 and not use_dist:
        import copy
        cfg_user = spectrometer_config(args.rays, total_runs)
        xrt.raytrace(copy.deepcopy(cfg_user))                      # warm-up (plans, workspace)
        best = 1e30
        for _ in range(3):
            t1 = time.perf_counter(); xrt.raytrace(copy.deepcopy(cfg_user)); best = min(best, time.perf_counter() - t1)
        call_ms = best * 1e3
        best = 1e30
        for _ in range(3):
            t1 = time.perf_counter()
            c2, e2 = xrt._prepare(copy.deepcopy(cfg_user)); f2 = e2.flatten(); xrt.DeviceTrace(f2)
            best = min(best, time.perf_counter() - t1)
        setup_ms = best * 1e3

    if rank == 0:
        photons_per_step = float(total_runs) * float(args.rays)
        ms_per_step = elapsed / max(args.steps, 1) * 1e3
        value = photons_per_step / (ms_per_step * 1e-3) / 1e6
        num_out = out[:flat.n_elements].cpu().numpy()
        per_launch_photons = float(len(my_seeds)) * float(args.rays)
        kavg_s = (kernel_ms.value / max(launches.value, 1)) * 1e-3
        achieved = ALGO_BYTES_PER_PHOTON * per_launch_photons / kavg_s / 1e9 if kavg_s > 0 else 0.0
        line = {
            'metric': 'Mphotons/sec (launched->detector) 3-element crystal spectrometer',
            'value': value, 'unit': 'Mphotons/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None,
            'raytrace_call_ms': call_ms, 'raytrace_call_object_construction_ms': setup_ms,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'cfg3: XicsrtSourceDirected(point, spread 10deg) -> XicsrtOpticSphericalCrystal'
                                   '(gaussian rocking curve) -> XicsrtOpticDetector; %d rays/run x %d runs/GPU, '
                                   '1 iteration, keep_history=False, keep_images=True' % (args.rays, len(my_seeds)),
                       'rays_per_run': args.rays, 'runs_per_gpu': len(my_seeds), 'photons_per_step': photons_per_step,
                       'rccl_ranks': (dist.get_world_size() if use_dist else 0),
                       'parallelism': 'runs sharded i mod N, one all-reduce of u64 histogram+counters per step',
                       'num_out': {n: int(v) for n, v in zip(flat.names, num_out)}},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': None,
                         'kernel': 'xrt_trace_kernel<false, 0, false>', 'kernel_ms_avg': kavg_s * 1e3,
                         'algorithmic_bytes_per_photon': ALGO_BYTES_PER_PHOTON,
                         'photons_per_launch': per_launch_photons},
        }
        prof, prof_file = committed_profile(args.rays, len(my_seeds))
        if prof is not None:
            # `frac` above prices the ALGORITHMIC bytes (what a ray-SoA-in-HBM design would move); the fused
            # kernel keeps rays in registers / LDS, so the bytes that really cross the HBM interface are the
            # pixel atomics: reported next to it, with the rate they amount to.
            line['roofline']['traffic'] = prof['hbm_traffic_bytes_per_launch']
            line['roofline']['traffic_source'] = prof_file
            line['roofline']['hbm_measured_GBps'] = prof['hbm_traffic_bytes_per_launch'] / kavg_s / 1e9 if kavg_s > 0 else None
            line['roofline']['hbm_measured_frac_of_peak'] = (line['roofline']['hbm_measured_GBps'] or 0.0) / HBM_PEAK_GBS
            sqd = prof.get('sq_derived') or {}
            if 'valu_wave_instr_per_64_photons' in sqd:
                # the real limiter: vector-instruction issue (binary64 geometry + MT19937), from the committed
                # SQ pass of the same command; `achieved` = wave-instructions issued per second on this run
                vpp = sqd['valu_wave_instr_per_64_photons'] / 64.0
                clock = PEAK_CLOCK_GHZ                     # devices of the pool sustain 2.2-2.3 GHz under this load
                issued = vpp * per_launch_photons / kavg_s if kavg_s > 0 else 0.0
                peak = 1024.0 * clock * 1e9 / 4.0          # 1024 SIMDs, one wave64 VALU instruction per 4 cycles
                line['roofline_valu'] = {
                    'bound': 'valu_issue', 'achieved': issued / 1e9, 'peak': peak / 1e9, 'unit': 'G wave-instr/s',
                    'frac': issued / peak if peak > 0 else None,
                    'valu_wave_instr_per_photon': vpp, 'peak_clock_GHz': clock,
                    'clock_GHz_under_load_in_profile_pass': sqd.get('clock_GHz_under_load'),
                    'busy_fraction_in_profile_pass': sqd.get('valu_issue_busy_fraction'),
                    'source': prof_file}
                line['roofline']['limiter'] = 'valu_issue (see roofline_valu); HBM carries only the histogram atomics'
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(flat, args.rays)
        # the real limiter first: `roofline_valu` (vector issue) in front of the BASELINE-defined HBM roofline
        head = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline',
                'roofline_valu', 'roofline')
        line = {**{k: line[k] for k in head if k in line}, **{k: v for k, v in line.items() if k not in head}}
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
