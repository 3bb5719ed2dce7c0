"""Public API of the reference (axtrack/interface.py:38-215) on the MI355X hot path:

    parameters, model, stnd_scaler = setup_inference(dest_dir)
    timelapse = prepare_input_data(imseq_fname, parameters, dest_dir, inference_data_dir, stnd_scaler, ...)
    axon_dets = inference(timelapse, model, dest_dir, parameters, ...)
    axon_dets.IDed_dets_all

Same names, argument meaning and return values as the reference, so examples/test.py runs
against this package by changing its import. Differences are confined to where the reference
depends on assets or packages that are not redistributable (weights file, tifffile):
`setup_inference` takes the weights explicitly, `prepare_input_data` also accepts arrays.
"""
import glob
import os

import numpy as np
import torch

from . import params as _params
from .detections import AxonDetections
from .hotpath import Detector
from .timelapse import Timelapse, preprocess, pad_mask


def _load_state_dict(weights):
    if isinstance(weights, dict):
        return weights.get('state_dict', weights)
    if os.path.isdir(weights):
        files = sorted(glob.glob(f'{weights}/*.pth'))
        weights = files[0]                       # IndexError if none, like utils.py:270
    ckpt = torch.load(weights, map_location='cpu')
    return ckpt['state_dict']                    # utils.py:260,272


def setup_inference(dest_dir, print_params=False, num_workers=3, device='cuda:0', weights=None, max_batch=256):
    """interface.py:38-77. `weights`: a state_dict, a checkpoint file or a directory holding one
    (default: $AXTRACK_MODEL_DIR) -- the reference reads deployed_model/E1000.pth, which is an
    external download."""
    parameters = _params.load_parameters()
    parameters['NUM_WORKERS'] = num_workers
    parameters['DEVICE'] = device
    torch.manual_seed(parameters['SEED'])
    if weights is None:
        weights = os.environ.get('AXTRACK_MODEL_DIR')
        if not weights:
            raise FileNotFoundError('no detector weights: pass weights=<state_dict | .pth | dir> or set AXTRACK_MODEL_DIR')
    model = Detector(_load_state_dict(weights), max_batch=max_batch, device=device)
    if print_params:
        for k, v in parameters.items():
            print(f'{k:28} {v}')
    os.makedirs(dest_dir, exist_ok=True)
    return parameters, model, _params.DEPLOYED_STND_SCALER


def prepare_input_data(imseq_fname, parameters, dest_dir, inference_data_dir, stnd_scaler, mask_fname=None,
                       use_cached_datasets='to', check_preproc=False, input_metadata={}):
    """interface.py:79-168. imseq_fname: a .tif/.npy file name inside inference_data_dir or a raw
    uint16 array [T,H,W]; mask_fname: .npy file name, bool array [H,W] or [T,H,W], or None.
    input_metadata['pad'] = p adds p zero pixels on all four sides (interface.py:126-128, Timelapse.py:224-234).
    use_cached_datasets: 'to' writes '{dest_dir}/{name}_dataset_cached.pkl', 'from' reads it (this package's file or
    one the reference wrote) instead of preprocessing, None does neither (Timelapse.py:435-449).
    check_preproc=True compares with the training data's statistics by plotting (interface.py:159-167): the plotting
    side is out of scope, so it raises."""
    name = input_metadata.get('name', 'timelapse')
    if check_preproc:
        raise NotImplementedError('check_preproc plots against train_preproc_data.csv, an external asset (interface.py:159-167)')
    if use_cached_datasets not in ('to', 'from', None):
        raise ValueError(f"use_cached_datasets must be 'to', 'from' or None, got {use_cached_datasets!r}")
    if use_cached_datasets == 'from':
        return Timelapse.from_cache(dest_dir, name, device=parameters['DEVICE'])
    if isinstance(imseq_fname, str):
        path = os.path.join(inference_data_dir, imseq_fname)
        if path.endswith('.npy'):
            imseq = np.load(path)
        else:
            try:
                from tifffile import imread
            except ImportError as e:
                raise ImportError('reading .tif needs tifffile; pass a numpy array or a .npy file instead') from e
            imseq = imread(path)
    else:
        imseq = np.asarray(imseq_fname)
    mask = None
    if isinstance(mask_fname, str) and not mask_fname.endswith('None'):
        mask = np.load(os.path.join(inference_data_dir, mask_fname))
    elif mask_fname is not None and not isinstance(mask_fname, str):
        mask = np.asarray(mask_fname)
    pad = input_metadata.get('pad')
    pad = [pad] * 4 if pad else None                                     # interface.py:126-128
    frames = preprocess(imseq, mask, offset=input_metadata.get('intensity_offset'),
                        clip=input_metadata.get('clip_intensity'), log_correct=parameters.get('LOG_CORRECT', True),
                        scale=stnd_scaler[1][0], device=parameters['DEVICE'], pad=pad)
    if pad:
        mask = pad_mask(mask, pad, imseq.shape)
    timelapse = Timelapse(frames, name=name, mask=mask, temporal_context=parameters['TEMPORAL_CONTEXT'],
                          tilesize=parameters['TILESIZE'], device=parameters['DEVICE'],
                          pixelsize=input_metadata.get('pixelsize'), dt=input_metadata.get('dt_min'),
                          incubation_time=input_metadata.get('incubation_time_min'))
    if use_cached_datasets == 'to':
        timelapse.to_cache(dest_dir)
    return timelapse


def inference(timelapse, model, dest_dir, parameters, detections_cache='to', astar_paths_cache='to',
              assigedIDs_cache='to'):
    """interface.py:170-215: detect growth cones in every frame, then associate them over time."""
    axon_dets_dir = f'{dest_dir}/axon_dets' if dest_dir else None
    if axon_dets_dir is None:
        detections_cache = astar_paths_cache = assigedIDs_cache = None
    axon_detections = AxonDetections(model, timelapse, parameters, axon_dets_dir)
    axon_detections.detect_dataset(cache=detections_cache)
    axon_detections.assign_ids(astar_paths_cache=astar_paths_cache, assigedIDs_cache=assigedIDs_cache)
    return axon_detections
