"""axtrack_amd: AxTrack's detect + associate hot path on AMD Instinct MI355X (gfx950).

Drop-in for the reference's public API on that path (reference __init__.py:1-16):
setup_inference, prepare_input_data, inference -> AxonDetections.IDed_dets_all; PKG_DIR, _compute_astar_path;
visualize_inference exists and says that plotting is out of scope.
The compute lives in csrc/libaxtrack_hip.so (C ABI: include/axtrack_hip.h); there is no CPU
fallback -- importing works anywhere, running needs the GPU and the built library.
"""
from .interface import setup_inference, prepare_input_data, inference, visualize_inference, PKG_DIR, DEPLOYED_MODEL_DIR
from .utils import _compute_astar_path
from .detections import AxonDetections
from .hotpath import Detector
from .timelapse import Timelapse

__all__ = ['setup_inference', 'prepare_input_data', 'inference', 'visualize_inference', 'PKG_DIR', 'DEPLOYED_MODEL_DIR',
           '_compute_astar_path', 'AxonDetections', 'Detector', 'Timelapse']
