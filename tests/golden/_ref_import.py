"""Import the read-only reference package (/root/reference) in THIS container only.

Used solely by tests/golden/make_golden.py to generate golden input/output vectors.
Third-party packages the reference imports but this image lacks are replaced by
inert stubs (none of them implements reference arithmetic).
Never imported by the product, by tests, or on the GPU box.
"""
import sys
import types
import importlib

import numpy as np


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []  # behave as a package so that submodule imports resolve
    sys.modules[name] = m
    return m


class _Missing:
    def __init__(self, what):
        self._what = what

    def __call__(self, *a, **k):
        raise RuntimeError(f"stubbed third-party call: {self._what}")

    def __getattr__(self, item):
        return _Missing(f"{self._what}.{item}")


def import_reference():
    if not hasattr(np, "float"):
        np.float = float  # removed in numpy>=1.24; used at mincostflow_models.py:19
    for name in ["torchvision", "torchvision.models", "torchvision.transforms",
                 "torchvision.transforms.functional", "torchsummary", "cv2",
                 "motmetrics", "pyastar2d", "libmot", "libmot.data_association",
                 "tifffile", "skimage", "skimage.util", "skimage.exposure",
                 "skimage.filters", "skimage.morphology"]:
        if name not in sys.modules:
            _stub(name)
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
    sys.modules["torchsummary"].summary = _Missing("torchsummary.summary")
    sys.modules["libmot.data_association"].MinCostFlowTracker = _Missing("libmot.MinCostFlowTracker")
    sys.modules["tifffile"].imread = _Missing("tifffile.imread")
    sys.modules["skimage.util"].img_as_float32 = _Missing("img_as_float32")
    sys.modules["skimage.exposure"].adjust_log = _Missing("adjust_log")
    sys.modules["skimage.filters"].gaussian = _Missing("gaussian")
    sys.modules["skimage.morphology"].dilation = _Missing("dilation")
    sys.modules["skimage.morphology"].square = _Missing("square")
    cv2 = sys.modules["cv2"]
    cv2.HISTCMP_BHATTACHARYYA = 3
    cv2.NORM_MINMAX = 32
    # transition_model calls compareHist even when vis_sim_weight == 0; its value is
    # multiplied by 0 there (mincostflow_models.py:107-117), any finite float works.
    cv2.compareHist = lambda a, b, m: 0.0
    if "/root" not in sys.path:
        sys.path.insert(0, "/root")
    return importlib.import_module("reference")
