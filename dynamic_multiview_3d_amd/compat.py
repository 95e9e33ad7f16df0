"""The reference's module paths, served by this package.

The reference's files import each other as `dyn_mult_view.<...>` (appearance_flow_model.py:5
`from dyn_mult_view.mv3d.utils.tf_utils import *`, train.py:9 `import dyn_mult_view`) and its conf files import the
model modules by bare name (tensorflowdata/*/conf.py: `from appearance_flow_model import AppearanceFlowModel`).
`install()` makes every one of those names resolve to the MI355X implementation, so reference-side code that imports the
path by its original names needs no edit:

    dyn_mult_view.mv3d.utils.tf_utils                      -> dynamic_multiview_3d_amd.tf_utils
    dyn_mult_view.mv3d.{nobg_nodm, nobg_dm, bg_nodm}        -> dynamic_multiview_3d_amd.mv3d
    dyn_mult_view.multi_view_model.<model file>            -> dynamic_multiview_3d_amd.<model file>
    dyn_mult_view.multi_view_model.train                   -> dynamic_multiview_3d_amd.train
    dyn_mult_view.multi_view_model.utils.read_tf_records[_multobj] -> dynamic_multiview_3d_amd.read_tf_records
    appearance_flow_model, highdim_angle, ... (bare names) -> the same modules

Only this path is served: the renderers, the ROS collector and the data download scripts of the reference are out of scope
 and stay unresolved.
"""
import importlib
import importlib.abc
import importlib.machinery
import os
import sys
import types

PKG = 'dynamic_multiview_3d_amd'
MODEL_FILES = ('appearance_flow_model', 'highdim_angle', 'lowdim_angle', 'appearance_flow_tinghui', 'main_model',
               'multiobject_appflow', 'multiobject_main_model')
ALIASES = {
    'dyn_mult_view.mv3d.utils.tf_utils': PKG + '.tf_utils',
    'dyn_mult_view.mv3d.nobg_nodm': PKG + '.mv3d',
    'dyn_mult_view.mv3d.nobg_dm': PKG + '.mv3d',
    'dyn_mult_view.mv3d.bg_nodm': PKG + '.mv3d',
    'dyn_mult_view.multi_view_model.train': PKG + '.train',
    'dyn_mult_view.multi_view_model.utils.read_tf_records': PKG + '.read_tf_records',
    'dyn_mult_view.multi_view_model.utils.read_tf_records_multobj': PKG + '.read_tf_records',
}
for _m in MODEL_FILES:
    ALIASES['dyn_mult_view.multi_view_model.' + _m] = PKG + '.' + _m
    ALIASES[_m] = PKG + '.' + _m
PACKAGES = ('dyn_mult_view', 'dyn_mult_view.mv3d', 'dyn_mult_view.mv3d.utils', 'dyn_mult_view.multi_view_model',
            'dyn_mult_view.multi_view_model.utils')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path=None, target=None):
        if name in ALIASES or name in PACKAGES:
            return importlib.machinery.ModuleSpec(name, self, is_package=name in PACKAGES)
        return None

    def create_module(self, spec):
        if spec.name in ALIASES:
            return importlib.import_module(ALIASES[spec.name])          # the very same module object under a second name
        mod = types.ModuleType(spec.name)
        mod.__path__ = []
        # conf files compute data_dir from dyn_mult_view.__file__ ('/'.join(str.split(dyn_mult_view.__file__, '/')[:-2]) + ...)
        mod.__file__ = os.path.join(ROOT, *spec.name.split('.'), '__init__.py')
        return mod

    def exec_module(self, module):
        pass


_finder = None


def install():
    """Idempotent; returns the finder."""
    global _finder
    if _finder is None:
        _finder = _Finder()
        sys.meta_path.insert(0, _finder)
        stale = sys.modules.get('dyn_mult_view')
        if stale is not None and not hasattr(stale, '__path__'):
            del sys.modules['dyn_mult_view']
    return _finder
