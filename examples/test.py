"""The reference's examples/test.py (three API calls, then print IDed_dets_all) against axtrack_amd.

The reference's example assets (example_timelapse.tif, its mask, the trained weights) are external downloads
that are not redistributable; this script therefore feeds a synthetic raw uint16 timelapse and seeded weights.
With the real assets, pass weights='deployed_model/' and the .tif / .npy file names instead."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import axtrack_amd as axtrack
from axtrack_amd import synth


def main(T=24, H=512, W=512, dest_dir=None):
    dest_dir = dest_dir or tempfile.mkdtemp(prefix='axtrack_example_')
    parameters, model, stnd_scaler = axtrack.setup_inference(dest_dir, weights=synth.synth_state_dict(42), max_batch=64)
    parameters.update({'MCF_MAX_FLOW': 140})        # as examples/test.py:19 does for its short example
    # a raw 16-bit timelapse: invert the preprocessing of the synthetic frames (offset 121, std scaling)
    frames = synth.synth_frames(T, H, W, seed=7)
    raw = np.clip((2.0 ** (frames * stnd_scaler[1][0]) - 1.0) * 65535.0 + 121.0 * (frames > 0), 0, 65535).astype(np.uint16)
    input_metadata = {'dt': 31, 'pixelsize': .62, 'intensity_offset': 121, 'clip_intensity': 55,
                      'incubation_time': 52, 'name': 'example_timelapse'}
    timelapse = axtrack.prepare_input_data(raw, parameters, dest_dir, dest_dir, stnd_scaler, mask_fname=None,
                                           use_cached_datasets='to', check_preproc=False, input_metadata=input_metadata)
    axon_dets = axtrack.inference(timelapse, model, dest_dir, parameters, detections_cache='to',
                                  astar_paths_cache='to', assigedIDs_cache='to')
    dets = axon_dets.IDed_dets_all
    print(dets)
    return axon_dets


if __name__ == '__main__':
    main()
