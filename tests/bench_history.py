"""Throughput probe of raytrace(config) as notebooks use it (one or a few runs, keep_history on): whole call, results
on the host.  Not a test: python tests/bench_history.py"""
import sys, os, json, copy, time
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import logging; logging.disable(logging.WARNING)
import numpy as np, bench, torch
import xicsrt_amd
for rays, runs, hist in ((1000000, 1, True), (1000000, 1, False), (10000000, 1, True), (1000000, 10, True), (100000, 1, True)):
    cfg = bench.spectrometer_config(rays, runs, seed=17)
    cfg['general'].update(keep_history=hist, print_results=False)
    xicsrt_amd.raytrace(copy.deepcopy(cfg))
    t0 = time.time(); res = xicsrt_amd.raytrace(copy.deepcopy(cfg)); dt = time.time() - t0
    nf = len(res['found']['history']['detector']['mask']) if hist else 0
    nl = len(res['lost']['history']['detector']['mask']) if hist else 0
    print(json.dumps({'rays': rays, 'runs': runs, 'keep_history': hist, 'raytrace_s': round(dt, 4), 'found': nf, 'lost_sample': nl, 'Mphot_s': round(rays * runs / dt / 1e6, 1)}), flush=True)
