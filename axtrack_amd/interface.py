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
import pickle
import warnings

import numpy as np
import torch

from . import params as _params
from .detections import AxonDetections
from .hotpath import Detector
from .timelapse import Timelapse, preprocess, pad_mask

# config.py:5,8 -- the directory that holds examples/ and deployed_model/ (the reference's package root)
PKG_DIR = os.path.abspath(os.path.join(os.path.dirname(__file__), '..')) + '/'
DEPLOYED_MODEL_DIR = PKG_DIR + 'deployed_model/'


def _load_state_dict(weights):
    if isinstance(weights, dict):
        return weights.get('state_dict', weights)
    if os.path.isdir(weights):
        files = sorted(glob.glob(f'{weights}/*.pth'))
        weights = files[0]                       # IndexError if none, like utils.py:270
    ckpt = torch.load(weights, map_location='cpu')
    return ckpt['state_dict']                    # utils.py:260,272


def setup_inference(dest_dir, print_params=False, num_workers=3, device='cuda:0', weights=None, max_batch=256):
    """interface.py:38-77. `weights`: a state_dict, a checkpoint file or a directory holding one. Default, in this order:
    $AXTRACK_MODEL_DIR, then {PKG_DIR}/deployed_model/ as in the reference (interface.py:29, utils.py:269-272: the first
    *.pth of the directory) -- the reference's E1000.pth is an external download, so that directory is empty until the
    user puts the checkpoint there. The standardization scaler is deployed_model/train_stnd_scaler.pkl when that file is
    there (interface.py:66-67), else its known content ('zscore', (0.015176106, 0.009456525))."""
    parameters = _params.load_parameters()
    parameters['NUM_WORKERS'] = num_workers
    parameters['DEVICE'] = device
    torch.manual_seed(parameters['SEED'])
    model_dir = None
    if weights is None:
        for cand in (os.environ.get('AXTRACK_MODEL_DIR'), DEPLOYED_MODEL_DIR):
            if cand and (os.path.isfile(cand) or glob.glob(f'{cand}/*.pth')):
                weights = model_dir = cand
                break
        else:
            raise FileNotFoundError(f'no detector weights: pass weights=<state_dict | .pth | dir>, set AXTRACK_MODEL_DIR, or put the '
                                    f'checkpoint (*.pth) into {DEPLOYED_MODEL_DIR} as the reference\'s README does')
    model = Detector(_load_state_dict(weights), max_batch=max_batch, device=device)
    if print_params:
        for k, v in parameters.items():
            print(f'{k:28} {v}')
    os.makedirs(dest_dir, exist_ok=True)
    stnd_scaler = _params.DEPLOYED_STND_SCALER
    for d in (model_dir if model_dir and os.path.isdir(model_dir) else None, DEPLOYED_MODEL_DIR):
        if d and os.path.exists(f'{d}/train_stnd_scaler.pkl'):
            with open(f'{d}/train_stnd_scaler.pkl', 'rb') as f:
                stnd_scaler = pickle.load(f)
            break
    return parameters, model, stnd_scaler


def prepare_input_data(imseq_fname, parameters, dest_dir, inference_data_dir, stnd_scaler, mask_fname=None,
                       use_cached_datasets='to', check_preproc=False, input_metadata={}):
    """interface.py:79-168. imseq_fname: a .tif/.npy file name inside inference_data_dir or a raw
    uint16 array [T,H,W]; mask_fname: .npy file name, bool array [H,W] or [T,H,W], or None.
    input_metadata['pad'] = p adds p zero pixels on all four sides (interface.py:126-128, Timelapse.py:224-234).
    use_cached_datasets: 'to' writes '{dest_dir}/{name}_dataset_cached.pkl', 'from' reads it (this package's file or
    one the reference wrote) instead of preprocessing, None does neither (Timelapse.py:435-449).
    check_preproc=True (interface.py:159-167): the reference samples every preprocessing step of the first and the last
    time point into '{dest_dir}/{name}_preproc_data.csv' (utils.save_preproc_metrics) and plots them against the training
    data's (train_preproc_data.csv, an external asset). The statistics file is written here in the same layout; the
    comparison plot is out of scope (plotting), which a warning says."""
    name = input_metadata.get('name', 'timelapse')
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
    mask_raw = mask
    if pad:
        mask = pad_mask(mask, pad, imseq.shape)
    timelapse = Timelapse(frames, name=name, mask=mask, temporal_context=parameters['TEMPORAL_CONTEXT'],
                          tilesize=parameters['TILESIZE'], device=parameters['DEVICE'],
                          pixelsize=input_metadata.get('pixelsize'), dt=input_metadata.get('dt_min'),
                          incubation_time=input_metadata.get('incubation_time_min'))
    if use_cached_datasets == 'to':
        timelapse.to_cache(dest_dir)
    if check_preproc:
        fname = save_preproc_metrics(dest_dir, name, imseq, mask_raw, input_metadata, parameters, stnd_scaler)     # (before padding, as the 'Original' step of the reference is taken after it: zero margins only add zeros)
        warnings.warn(f'check_preproc: the preprocessing statistics are in {fname}; the comparison plot against the training '
                      f'data (ml_plotting.plot_preprocessed_input_data, train_preproc_data.csv) is out of scope of axtrack_amd')
    return timelapse


def save_preproc_metrics(dest_dir, name, imseq, mask, input_metadata, parameters, stnd_scaler, n_samples=int(1e6)):
    """utils.save_preproc_metrics (utils.py:90-110) for the steps Timelapse keeps when it plots (Timelapse.py:238-241,250-253,
    260-263,316-319): 'Original' (scaled to [0,1], masked, offset), 'Clipped', 'Log-Adjusted', 'Standardized (frame-wize: ..)',
    each sampled at n_samples random pixels of the first and of the last time point (timepoints[0] / timepoints[-1]: input
    frames temporal_context and T - 1 - temporal_context). Columns (name, step, 't_0' | 't_-1') as in the reference.
    Every step is the fused preprocessing kernel with the later steps switched off (axt_preprocess_u16)."""
    import pandas as pd
    tc = int(parameters['TEMPORAL_CONTEXT'])
    a = np.asarray(imseq)
    two = np.ascontiguousarray(a[[tc, a.shape[0] - 1 - tc]])
    m = None if mask is None else (np.asarray(mask)[[tc, a.shape[0] - 1 - tc]] if np.asarray(mask).ndim == 3 else mask)
    off, clip = input_metadata.get('intensity_offset'), input_metadata.get('clip_intensity')
    log_correct, scale = parameters.get('LOG_CORRECT', True), stnd_scaler[1][0]
    steps = [('Original', dict(clip=0, log_correct=False, scale=1.0)), ('Clipped', dict(clip=clip, log_correct=False, scale=1.0))]
    if log_correct:
        steps.append(('Log-Adjusted', dict(clip=clip, log_correct=True, scale=1.0)))
    if stnd_scaler[0]:
        steps.append((f"Standardized (frame-wize: {parameters.get('STANDARDIZE_FRAMEWISE', False)})",
                      dict(clip=clip, log_correct=log_correct, scale=scale)))
    idx = np.random.default_rng().choice(two[0].size, int(n_samples))
    samples = []
    for step, kw in steps:
        out = preprocess(two, m, offset=off, device=parameters['DEVICE'], **kw).cpu().numpy()
        samples.append(pd.Series(out[0].ravel()[idx], name=(name, step, 't_0')))
        samples.append(pd.Series(out[1].ravel()[idx], name=(name, step, 't_-1')))
    fname = f'{dest_dir}/{name}_preproc_data.csv'
    pd.concat(samples, axis=1).to_csv(fname)
    return fname


def visualize_inference(axon_dets, which_dets='IDed', **kwargs):
    """interface.py:217-320 renders every frame with matplotlib and encodes a video (video_plotting.draw_all): plotting is
    out of scope of this package ("Out of scope"). Everything it would draw is on the object it is given."""
    raise NotImplementedError(
        'visualize_inference (frame rendering / video encoding, axtrack/video_plotting.py) is out of scope of axtrack_amd. '
        'What it draws is available: axon_dets.IDed_dets_all, axon_dets.get_frame_dets(which, t), axon_dets.dataset.mask; '
        "the caches written by inference(..., *_cache='to') are in the reference's layouts, so the reference's own "
        'visualize_inference can read them.')


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
