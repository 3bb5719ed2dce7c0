"""Hot-path parameters of the deployed AxTrack model.

Values are those the reference pickles in deployed_model/params.pkl (readable copy:
deployed_model/params.txt); only the keys read on the inference path are kept
(AxonDetections.py:57-75, interface.py:100-168). `setup_inference` returns a fresh dict, which
callers may edit between steps as examples/test.py:19 does (MCF_MAX_FLOW)."""
import pickle

DEPLOYED = dict(
    DEVICE='cuda:0', SEED=42, NUM_WORKERS=3,
    TILESIZE=512, SX=12, SY=12, TEMPORAL_CONTEXT=2, USE_MOTION_DATA='exclude',
    NON_MAX_SUPRESSION_DIST=23, BBOX_THRESHOLD=0.7,
    MCF_MIN_ID_LIFETIME=5, MCF_CONF_CAPPING_METHOD='scale_to_max', MCF_VIS_SIM_WEIGHT=0,
    MCF_MAX_CONF_COST=4.6, MCF_MAX_FLOW=450, MCF_MIN_FLOW=5, MCF_MAX_NUM_MISSES=1, MCF_MISS_RATE=0.6,
    MCF_ENTRY_EXIT_COST=2, MCF_EDGE_COST_THR=0.7,
    LOG_CORRECT=True, STANDARDIZE=('zscore', None), STANDARDIZE_FRAMEWISE=False, USE_SPARSE=False,
    CLIP_LOWERLIM=55 / 2 ** 16, OFFSET=None, PAD=[0, 300, 0, 300],
    # build-side switches (not in the reference's params.pkl): see INTEGRATION.md
    CNN_ARITH='f32',
)
# ('zscore', (var_scalar, mean_scalar)) as unpickled from deployed_model/train_stnd_scaler.pkl (interface.py:66-67)
DEPLOYED_STND_SCALER = ('zscore', (0.015176106, 0.009456525))


def load_parameters(path=None):
    """A parameter dict; `path` may point to a reference-style params.pkl (exp_parameters.py:110-117)."""
    p = dict(DEPLOYED)
    if path:
        with open(path, 'rb') as f:
            loaded = pickle.load(f)
        p.update({k: v for k, v in loaded.items()})
    return p
