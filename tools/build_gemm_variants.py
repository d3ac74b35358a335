"""Experiment libraries for GEMM schedule A/B runs (never shipped): librlaifv_hip_<tag>.so next to the product library,
selected per process with RV_HIP_LIB.  Usage: python tools/build_gemm_variants.py [tag ...]"""
import importlib.util
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("_b", os.path.join(REPO, "rlaif-v_amd", "build.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)

VARIANTS = {
    "p0": ("RV_GEMM_PRIO_MODE=0",), "p1": ("RV_GEMM_PRIO_MODE=1",), "p2": ("RV_GEMM_PRIO_MODE=2",),
    "s0": ("RV_GEMM_DMA_SLOT=0",), "s2": ("RV_GEMM_DMA_SLOT=2",), "s3": ("RV_GEMM_DMA_SLOT=3",),
    "p1s3": ("RV_GEMM_PRIO_MODE=1", "RV_GEMM_DMA_SLOT=3"),
}
os.environ["RV_BUILD_ONLY"] = "gemm.hip"
b.build_extension()
for tag in (sys.argv[1:] or list(VARIANTS)):
    print(b.build_extension(defines=VARIANTS[tag], tag="_" + tag))
