"""Builds the REFERENCE's own CPU operators (nms, roi_align_forward) into oracle/_ref/ref_C.so.

Recipe (nothing from the reference is copied into this repository):
  * sources are compiled from where they lie: /root/reference/maskrcnn_benchmark/csrc/{vision.cpp,cpu/*.cpp};
  * torch >= 2.x no longer converts `tensor.type()` to a ScalarType inside AT_DISPATCH_FLOATING_TYPES, so two
    tokens need `.scalar_type()` (csrc/cpu/nms_cpu.cpp:71 `dets.type()`, csrc/cpu/ROIAlign_cpu.cpp:242
    `input.type()`).  The edit is applied by `sed` into a throw-away directory under $TMPDIR at build time;
    only the resulting shared object is kept, under oracle/_ref/ (git-ignored, travels to the GPU box);
  * the CUDA half of the reference (csrc/cuda/*.cu) is NOT buildable here: it includes THC/THC.h, which
    PyTorch removed (csrc/cuda/nms.cu:5).
Used by tests to validate oracle/dadet_oracle.c and (optionally) as the `reference` CPU baseline.
Skips silently when /root/reference is absent (e.g. on the GPU box, where the prebuilt .so is used).
"""
import glob
import os
import re
import shutil
import sys
import tempfile

REF = "/root/reference/maskrcnn_benchmark/csrc"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, "_ref")
OUT_SO = os.path.join(OUT_DIR, "ref_C.so")


def build(verbose=False):
    if not os.path.isdir(REF):
        return OUT_SO if os.path.exists(OUT_SO) else None
    if os.path.exists(OUT_SO):
        return OUT_SO
    from torch.utils import cpp_extension

    os.makedirs(OUT_DIR, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="dadet_ref_build_")
    try:
        srcs = []
        for rel in ["vision.cpp", "cpu/nms_cpu.cpp", "cpu/ROIAlign_cpu.cpp"]:
            text = open(os.path.join(REF, rel)).read()
            text = re.sub(r"AT_DISPATCH_FLOATING_TYPES\((\w+)\.type\(\)", r"AT_DISPATCH_FLOATING_TYPES(\1.scalar_type()", text)
            dst = os.path.join(tmp, os.path.basename(rel))
            open(dst, "w").write(text)
            srcs.append(dst)
        build_dir = os.path.join(tmp, "build")
        os.makedirs(build_dir)
        cpp_extension.load(name="ref_C", sources=srcs, extra_include_paths=[REF], build_directory=build_dir,
                           extra_cflags=["-O2", "-w"], verbose=verbose, is_python_module=False)
        built = glob.glob(os.path.join(build_dir, "ref_C*.so"))
        shutil.copy(built[0], OUT_SO)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return OUT_SO


def load():
    """import the built extension as a module exposing nms / roi_align_forward (reference vision.cpp:7-15)"""
    import importlib.util

    import torch  # noqa: F401  (libtorch must be loaded first)

    if not os.path.exists(OUT_SO):
        return None
    spec = importlib.util.spec_from_file_location("ref_C", OUT_SO)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    print(build(verbose="-v" in sys.argv))
