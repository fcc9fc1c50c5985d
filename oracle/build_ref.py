"""Compile the reference's OWN CPU code into oracle/_ref/ (test infrastructure only).

Two pybind11 torch extensions, both compiled from the reference's sources WHERE THEY LIE under
/root/reference; nothing is copied into this repository and no stand-in headers are written.
oracle/_ref/ is git-ignored but travels to the GPU box with the snapshot.

1. voxel_layer_ref   mmdet3d/ops/voxel/src/{voxelization.cpp,voxelization_cpu.cpp,
                     scatter_points_cpu.cpp}: hard_voxelize / dynamic_voxelize.
2. spconv_ref        mmdet3d/ops/spconv/src/reordering.cc + our driver TU
                     oracle/ref_spconv_driver.cc, which includes the reference's
                     include/spconv/geometry.h: the reference's CPU rulebook functions
                     (getValidOutPos, getIndicePairsConv, getIndicePairsSubM) and CPU gather /
                     scatter-add functors.  Their headers need
                     <cuda_runtime_api.h> (include/tensorview/tensorview.h:16); a genuine copy
                     ships in this image with triton (backends/nvidia/include), used via -I.
   NOT built: include/spconv/spconv_ops.h (getIndicePair<3>, indiceConv<float>) and
   src/indice.cc (three-line functor wrappers, includes spconv_ops.h) -- spconv_ops.h includes
   torch_utils.h -> ATen/cuda/CUDAContext.h -> <cusparse.h>, absent here ("fatal error:
   cusparse.h: No such file or directory") -> unbuildable; the driver TU restates its CPU branch
   around the reference's functors (see its header).
"""
import os
import sys

REF_VOXEL = '/root/reference/mmdet3d/ops/voxel/src'
REF_SPCONV = '/root/reference/mmdet3d/ops/spconv'
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '_ref')
NAME = 'voxel_layer_ref'
NAME_SPCONV = 'spconv_ref'


def _cuda_runtime_include():
    """Directory of a genuine cuda_runtime_api.h (triton ships NVIDIA's headers)."""
    try:
        import triton
    except ImportError:
        return None
    d = os.path.join(os.path.dirname(triton.__file__), 'backends', 'nvidia', 'include')
    return d if os.path.exists(os.path.join(d, 'cuda_runtime_api.h')) else None


def build(verbose=False):
    if not os.path.isdir(REF_VOXEL):
        print('[build_ref] reference sources not present; skipping')
        return None
    os.makedirs(OUT, exist_ok=True)
    from torch.utils.cpp_extension import load
    srcs = [os.path.join(REF_VOXEL, f) for f in
            ('voxelization.cpp', 'voxelization_cpu.cpp', 'scatter_points_cpu.cpp')]
    mod = load(name=NAME, sources=srcs, build_directory=OUT, extra_cflags=['-O2'],
               with_cuda=False, verbose=verbose)
    build_spconv(verbose)
    return mod


def build_spconv(verbose=False):
    inc = _cuda_runtime_include()
    if not os.path.isdir(REF_SPCONV) or inc is None:
        print('[build_ref] spconv reference sources or cuda_runtime_api.h not present; skipping')
        return None
    out = os.path.join(OUT, 'spconv')
    os.makedirs(out, exist_ok=True)
    from torch.utils.cpp_extension import load
    srcs = [os.path.join(HERE, 'ref_spconv_driver.cc'),
            os.path.join(REF_SPCONV, 'src', 'reordering.cc')]
    return load(name=NAME_SPCONV, sources=srcs, build_directory=out,
                extra_include_paths=[os.path.join(REF_SPCONV, 'include'), inc],
                extra_cflags=['-O2', '-w'], with_cuda=False, verbose=verbose)


def _load(path, name):
    import importlib.util
    import torch  # noqa: F401  (resolves libtorch symbols)
    if not os.path.exists(path):
        return None
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_prebuilt():
    """Import oracle/_ref/voxel_layer_ref.so if it was built (no compilation)."""
    return _load(os.path.join(OUT, NAME + '.so'), NAME)


def load_prebuilt_spconv():
    """Import oracle/_ref/spconv/spconv_ref.so if it was built (no compilation)."""
    return _load(os.path.join(OUT, 'spconv', NAME_SPCONV + '.so'), NAME_SPCONV)


if __name__ == '__main__':
    m = build(verbose='-v' in sys.argv)
    print('[build_ref]', 'ok' if m is not None else 'skipped')
