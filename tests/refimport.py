"""Import the reference (`/root/reference/src/dsnt`) in the build container.

Only used by CPU tests and the golden-vector generator; the reference never
travels to the GPU box, so everything here degrades to `None` when the path is
absent.  The reference imports torchvision / torchdata / PIL-side helpers at
module import time (`src/dsnt/model.py:13`, `src/dsnt/data.py:11-16`); empty stub
modules are enough for the hot path.
"""
import importlib
import os
import sys
import types

REF_SRC = '/root/reference/src'


def _stub(name, **attrs):
    if name in sys.modules:
        return sys.modules[name]
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def load_reference():
    """Returns the reference `dsnt` package as (nn, hourglass, model) or None."""
    if not os.path.isdir(os.path.join(REF_SRC, 'dsnt')):
        return None
    sys.dont_write_bytecode = True
    tv = _stub('torchvision')
    tv.models = _stub('torchvision.models')
    tv.transforms = _stub('torchvision.transforms')
    td = _stub('torchdata')
    td.mpii = _stub('torchdata.mpii', MpiiData=object, MPII_Joint_Horizontal_Flips=[],
                    MPII_Image_Mean=[0, 0, 0], MPII_Image_Stddev=[1, 1, 1],
                    transform_keypoints=None)
    try:
        import PIL  # noqa: F401
    except ImportError:
        pil = _stub('PIL')
        pil.Image = _stub('PIL.Image', Image=type('Image', (), {}))
        pil.ImageDraw = _stub('PIL.ImageDraw')
        pil.ImageFont = _stub('PIL.ImageFont')
    # The product package is also called `dsnt`; load the reference under its own
    # name from an explicit path without disturbing sys.path ordering.
    saved = {k: sys.modules.pop(k) for k in list(sys.modules)
             if k == 'dsnt' or k.startswith('dsnt.')}
    sys.path.insert(0, REF_SRC)
    try:
        ref_nn = importlib.import_module('dsnt.nn')
        ref_hg = importlib.import_module('dsnt.hourglass')
        ref_model = importlib.import_module('dsnt.model')
        assert ref_nn.__file__.startswith(REF_SRC)
    finally:
        sys.path.remove(REF_SRC)
        for k in [k for k in sys.modules if k == 'dsnt' or k.startswith('dsnt.')]:
            del sys.modules[k]
        sys.modules.update(saved)
    return ref_nn, ref_hg, ref_model


def load_reference_module(name):
    """One more module of the reference package (e.g. 'dsnt.util'), same isolation as above."""
    if load_reference() is None:
        return None
    saved = {k: sys.modules.pop(k) for k in list(sys.modules) if k == 'dsnt' or k.startswith('dsnt.')}
    sys.path.insert(0, REF_SRC)
    try:
        mod = importlib.import_module(name)
        assert mod.__file__.startswith(REF_SRC)
    finally:
        sys.path.remove(REF_SRC)
        for k in [k for k in sys.modules if k == 'dsnt' or k.startswith('dsnt.')]:
            del sys.modules[k]
        sys.modules.update(saved)
    return mod
