"""
Command line front end: `python -m xicsrt_amd config.json [options]`
(reference: xicsrt/__main__.py:21-176; same options).

The configuration file is an XICSRT config dictionary (json or pickle).  `--mp` is accepted for
compatibility: on this build runs are spread over the GPUs of the process group, not over
host processes, so it selects the same device path as the default.
"""
import argparse
import logging
import sys

from .config import __version__


def get_parser():
    parser = argparse.ArgumentParser(
        prog='xicsrt_amd',
        description=f'xicsrt_amd (XICSRT {__version__} interface): ray trace a configuration file on the GPU.',
        formatter_class=argparse.RawDescriptionHelpFormatter)
    parser.add_argument('config_file', type=str, nargs='?', default='config.json',
                        help='The path to the configuration file for this run.')
    parser.add_argument('--numruns', type=int, default=None, metavar='N', help='Number of runs.')
    parser.add_argument('--numiter', type=int, default=None, metavar='N', help='Number of iterations per run.')
    parser.add_argument('--seed', type=int, default=None, metavar='N', help='The random seed to use.')
    parser.add_argument('--save', action='store_true', help='Save the results.')
    parser.add_argument('--images', action='store_true', help='Save intersection images.')
    parser.add_argument('--suffix', type=str, default=None, metavar='STR', help='A suffix to add to the output files.')
    parser.add_argument('--path', type=str, default=None, metavar='STR', help='Directory in which to store output.')
    parser.add_argument('--multiprocessing', '--mp', action='store_true', help='Use multiprocessing.')
    parser.add_argument('--processes', type=int, default=None, metavar='N',
                        help='Number of processes to use for muliprocessing.')
    parser.add_argument('--version', action='store_true', help='Show the version number.')
    parser.add_argument('--debug', action='store_true', help='Show debugging output in the log.')
    return parser


def run(argv=None):
    args = get_parser().parse_args(argv)
    if args.version:
        print(f'{__version__}')
        return None
    logging.basicConfig(level=logging.DEBUG if args.debug else logging.INFO, force=True)
    from . import xicsrt_io, xicsrt_raytrace
    config = xicsrt_io.load_config(args.config_file)
    general = config.setdefault('general', {})
    # the reference applies an option only when it is truthy (__main__.py:150-163)
    for option, key in ((args.suffix, 'output_suffix'), (args.numruns, 'number_of_runs'),
                        (args.numiter, 'number_of_iter'), (args.seed, 'random_seed'), (args.path, 'output_path'),
                        (args.save, 'save_results'), (args.images, 'save_images')):
        if option:
            general[key] = option
    if args.multiprocessing:
        return xicsrt_raytrace.raytrace_mp(config, processes=args.processes)
    return xicsrt_raytrace.raytrace(config)


if __name__ == '__main__':
    run(sys.argv[1:])
