#!/usr/bin/env python3
"""Identity of a rocprofv3 collection: which kernel sources and which bench.py command line it measured. Written next to
the raw output by collect.sh / collect_pmc.sh and committed with the summaries as profiles/<tag>_meta.json; bench.py puts
committed counters into its line only when both still match (sources_sha256 of the conv kernels, the workload arguments).
    python3 profiles/profile_meta.py <tag> [bench.py arguments of the profiled command]"""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ('axtrack_amd/csrc/cnn.hip', 'axtrack_amd/csrc/cnn_front.hip')


def sources_sha256(root=ROOT):
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(root, f), 'rb').read())
    return h.hexdigest()


def workload_key(argv):
    """The arguments that change which kernels run and on what: workload, association, arithmetic, input, size, frames."""
    key = {'workload': 'c3', 'assoc': 'hungarian', 'arith': 'f32', 'input': 'hbm', 'size': None, 'frames': None}
    it = iter(argv)
    for a in it:
        if a.startswith('--') and a[2:] in key:
            key[a[2:]] = next(it, None)
    if key['arith'] == 'f32_winograd':
        key['arith'] = 'f32'
    return key


if __name__ == '__main__':
    tag, argv = sys.argv[1], sys.argv[2:]
    meta = {'tag': tag, 'sources': list(KERNEL_SOURCES), 'sources_sha256': sources_sha256(), 'bench_args': argv,
            'workload_key': workload_key(argv), 'collected': time.strftime('%Y-%m-%d %H:%M:%S')}
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    json.dump(meta, open(os.path.join(ROOT, 'gpurun_out', f'{tag}_meta.json'), 'w'), indent=1)
    print(json.dumps(meta))
