"""Host profile of raytrace(config) with keep_history=True (not a test): python tests/prof_history.py [rays]"""
import sys, os, time, cProfile, pstats, copy
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, xicsrt_amd
rays = int(sys.argv[1]) if len(sys.argv) > 1 else 10000000
cfg = bench.spectrometer_config(rays, 1, seed=3)
cfg['general']['keep_history'] = True
xicsrt_amd.raytrace(copy.deepcopy(cfg))
t0 = time.time(); xicsrt_amd.raytrace(copy.deepcopy(cfg)); print('call %.1f ms' % ((time.time() - t0) * 1e3))
pr = cProfile.Profile(); pr.enable(); xicsrt_amd.raytrace(copy.deepcopy(cfg)); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
