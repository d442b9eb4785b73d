"""
`raytrace_mp(config, processes=N)` from ONE process (xicsrt/xicsrt_multiprocessing.py:12-81): the runs are fanned out over
the visible devices, one host thread per device.  No GPU here: helpers.OracleDeviceTrace stands in for the device and the
module's two device hooks (`visible_devices`, `device_scope`) are replaced, so the fan-out, the host sums, the history
order and the saving are the code that ships.  On hardware the same entry point is held to `raytrace` in
tests/test_gpu_parity.py.
"""
import contextlib
import os

import numpy as np
import pytest

import helpers
from xicsrt_amd import xicsrt_raytrace as xrt


def _config(tmp_path, keep_history, runs=5, save_images=False):
    import bench
    cfg = bench.spectrometer_config(3000, runs, seed=11)
    cfg['general'].update({'keep_history': keep_history, 'history_max_lost': 500, 'number_of_iter': 2,
                           'save_config': False, 'save_images': save_images, 'output_path': str(tmp_path),
                           'output_prefix': 'mp', 'print_results': False})
    return cfg


@pytest.fixture
def devices(monkeypatch):
    """`n` pretend devices whose compute is the oracle; records which device scopes were entered."""
    entered = []

    def install(n, cls=helpers.OracleDeviceTrace):
        monkeypatch.setattr(xrt, 'DeviceTrace', cls)
        monkeypatch.setattr(xrt, 'visible_devices', lambda: n)

        @contextlib.contextmanager
        def scope(index):
            entered.append(index)
            yield
        monkeypatch.setattr(xrt, 'device_scope', scope)
        return entered
    return install


def _same(a, b):
    assert {k: int(v['num_out']) for k, v in a['total']['meta'].items()} == \
           {k: int(v['num_out']) for k, v in b['total']['meta'].items()}
    for k, img in a['total']['image'].items():
        assert (img is None and b['total']['image'][k] is None) or np.array_equal(img, b['total']['image'][k])
    for group in ('found', 'lost'):
        assert list(a[group]['history']) == list(b[group]['history'])
        for name, rays in a[group]['history'].items():
            for key in ('origin', 'direction', 'wavelength', 'mask'):
                assert np.array_equal(rays[key], b[group]['history'][name][key], equal_nan=(key != 'mask')), (group, name, key)


@pytest.mark.parametrize('keep_history', [False, True])
@pytest.mark.parametrize('n_dev,processes,width', [(2, None, 2), (3, None, 3), (4, 2, 2), (8, None, 5), (2, 1, 1)])
def test_fan_out_over_devices_equals_raytrace(tmp_path, devices, keep_history, n_dev, processes, width):
    entered = devices(n_dev)
    cfg = _config(tmp_path, keep_history)
    many = xrt.raytrace_mp(cfg, processes=processes)
    assert sorted(set(entered)) == (list(range(width)) if width > 1 else [])
    entered.clear()
    one = xrt.raytrace(cfg)
    assert entered == []
    assert int(one['total']['meta']['detector']['num_out']) > 0
    _same(one, many)
    # the reference's quirk: raytrace_mp leaves output_run_suffix in random_seed (xicsrt_multiprocessing.py:69)
    assert many['config']['general']['random_seed'] == many['config']['general']['output_run_suffix']
    assert one['config']['general']['random_seed'] == 11
    if keep_history:
        assert len(many['found']['history']['detector']['mask']) == int(many['total']['meta']['detector']['num_out'])


def test_every_device_writes_the_images_of_its_own_runs(tmp_path, devices):
    devices(2)
    cfg = _config(tmp_path, False, runs=3, save_images=True)
    xrt.raytrace_mp(cfg)
    names = sorted(os.listdir(tmp_path))
    for run in range(3):
        assert any('%04d' % run in n and 'detector' in n for n in names), (run, names)


def test_a_failing_device_raises_what_the_reference_raises(tmp_path, devices):
    class Failing(helpers.OracleDeviceTrace):
        fail_code = -7
    devices(2, Failing)
    with pytest.raises(ValueError, match='intensity of less than one'):
        xrt.raytrace_mp(_config(tmp_path, False))
    assert os.listdir(tmp_path) == []


def test_combine_raytrace_components_and_mismatched_images(caplog):
    def one(n, img):
        out = xrt._empty_output({'general': {}})
        out['total']['meta'] = {'source': {'num_out': 10}, 'a': {'num_out': n}, 'b': {'num_out': 2 * n}}
        out['total']['image'] = {'a': img, 'b': None}
        return out
    res = xrt.combine_raytrace([one(1, np.ones((2, 3))), one(2, np.ones((2, 3)))], components=['a', 'b'])
    assert list(res['total']['meta']) == ['a', 'b'] and res['total']['meta']['b']['num_out'] == 6
    assert np.array_equal(res['total']['image']['a'], 2 * np.ones((2, 3))) and res['total']['image']['b'] is None
    assert res['found']['history'] == {} and res['lost']['history'] == {}
    res = xrt.combine_raytrace([one(1, np.ones((2, 3))), one(2, np.ones((3, 2)))])
    assert res['total']['image']['a'] is None                  # shapes differ: warning, no image
    res = xrt.combine_raytrace([one(1, np.ones((2, 3))), one(2, np.ones((2, 3)))], keep_images=False)
    assert res['total']['image'] == {}
