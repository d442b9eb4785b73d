"""
`import xicsrt` compatibility: existing scripts and user plug-ins written against the reference's
module layout keep working on this package.

    import xicsrt_amd.compat; xicsrt_amd.compat.install()
    import xicsrt                                   # -> xicsrt_amd
    from xicsrt import xicsrt_io, xicsrt_config
    from xicsrt.optics._InteractCrystal import InteractCrystal      # one module per class, as upstream
    from xicsrt.sources._XicsrtSourceFocused import XicsrtSourceFocused

The reference keeps every class in a file `_<ClassName>.py` (objects/_Dispatcher.py:63-113 finds plug-ins
by that file name); here the classes live in a few modules, so `install()` registers one synthetic module
per class under the upstream names.  Nothing is installed implicitly, and an `xicsrt` that is already
imported (the real reference) is never replaced.
"""
import sys
import types


def _class_modules(package_name, module, names):
    mods = {}
    for name in names:
        obj = getattr(module, name, None)
        if isinstance(obj, type):
            m = types.ModuleType('%s._%s' % (package_name, name))
            setattr(m, name, obj)
            mods[m.__name__] = m
    return mods


def install(name='xicsrt'):
    """Register this package (and the upstream-style submodules) under `name`; returns the package."""
    import xicsrt_amd
    from . import config, objects, optics, sources, filters, xicsrt_io, xicsrt_raytrace
    from .optics import mesh as _mesh
    existing = sys.modules.get(name)
    if existing is not None and existing is not xicsrt_amd:
        raise ImportError('%r is already imported from %s; refusing to shadow it'
                          % (name, getattr(existing, '__file__', '?')))
    mods = {name: xicsrt_amd,
            name + '.xicsrt_raytrace': xicsrt_raytrace,
            name + '.xicsrt_config': config,
            name + '.xicsrt_io': xicsrt_io,
            name + '.optics': optics, name + '.sources': sources, name + '.filters': filters}
    # xicsrt_multiprocessing.raytrace(config, processes=None) (xicsrt_multiprocessing.py:12)
    mp = types.ModuleType(name + '.xicsrt_multiprocessing')
    mp.raytrace = xicsrt_raytrace.raytrace_mp
    mods[mp.__name__] = mp
    pub = types.ModuleType(name + '.xicsrt_public')
    pub.get_element = xicsrt_amd.get_element
    mods[pub.__name__] = pub
    obj_pkg = types.ModuleType(name + '.objects')
    mods[obj_pkg.__name__] = obj_pkg
    mods.update(_class_modules(name + '.objects', objects, ('ConfigObject', 'GeometryObject', 'RayArray')))
    optic_names = [n for n in dir(optics) if n.startswith(('XicsrtOptic', 'Shape', 'Interact', 'TraceObject'))]
    mods.update(_class_modules(name + '.optics', optics, optic_names))
    mods.update(_class_modules(name + '.optics', _mesh, [n for n in dir(_mesh) if n.startswith('ShapeMesh')]))
    mods.update(_class_modules(name + '.sources', sources, [n for n in dir(sources) if n.startswith('Xicsrt')]))
    mods.update(_class_modules(name + '.filters', filters, [n for n in dir(filters) if n.startswith('Xicsrt')]))
    for full, m in mods.items():
        sys.modules[full] = m
        parent, _, leaf = full.rpartition('.')
        if parent in mods and parent != name and not hasattr(mods[parent], leaf):
            setattr(mods[parent], leaf, m)
    for leaf in ('xicsrt_multiprocessing', 'xicsrt_public', 'objects'):
        if not hasattr(xicsrt_amd, leaf):
            setattr(xicsrt_amd, leaf, mods[name + '.' + leaf])
    return xicsrt_amd
