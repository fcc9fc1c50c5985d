"""Compile the reference's OWN CPU voxelizer into oracle/_ref/ (test infrastructure only).

The three translation units are compiled where they lie under /root/reference
(mmdet3d/ops/voxel/src/{voxelization.cpp,voxelization_cpu.cpp,scatter_points_cpu.cpp});
nothing is copied into this repository and no stand-in headers are written.  The
result (a pybind11 torch extension exporting hard_voxelize / dynamic_voxelize) is
used by tests to validate oracle/sf_oracle.c's restatement.  oracle/_ref/ is
git-ignored but travels to the GPU box with the snapshot.

The reference's spconv 1.x sources (mmdet3d/ops/spconv/{src,include}) are NOT built:
include/tensorview/tensorview.h:16 needs <cuda_runtime_api.h>, absent from this image,
and the rules forbid stand-in headers -> treated as unbuildable.
"""
import os
import sys

REF_SRC = '/root/reference/mmdet3d/ops/voxel/src'
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '_ref')
NAME = 'voxel_layer_ref'


def build(verbose=False):
    if not os.path.isdir(REF_SRC):
        print('[build_ref] reference sources not present; skipping')
        return None
    os.makedirs(OUT, exist_ok=True)
    from torch.utils.cpp_extension import load
    srcs = [os.path.join(REF_SRC, f) for f in
            ('voxelization.cpp', 'voxelization_cpu.cpp', 'scatter_points_cpu.cpp')]
    mod = load(name=NAME, sources=srcs, build_directory=OUT, extra_cflags=['-O2'],
               with_cuda=False, verbose=verbose)
    return mod


def load_prebuilt():
    """Import oracle/_ref/voxel_layer_ref.so if it was built (no compilation)."""
    import importlib.util
    import torch  # noqa: F401  (resolves libtorch symbols)
    path = os.path.join(OUT, NAME + '.so')
    if not os.path.exists(path):
        return None
    spec = importlib.util.spec_from_file_location(NAME, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == '__main__':
    m = build(verbose='-v' in sys.argv)
    print('[build_ref]', 'ok' if m is not None else 'skipped')
