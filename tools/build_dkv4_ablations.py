"""Experiment libraries of the dK/dV kernel version 4, one per ablation body (tools/gen_attn_dkv4.py --ablations):
rlaif-v_amd/librlaifv_hip_abl<n>.so, selected with RV_HIP_LIB for timing runs.  Results are WRONG by construction."""
import concurrent.futures as cf
import importlib.util
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
subprocess.run([sys.executable, os.path.join(REPO, "tools", "gen_attn_dkv4.py"), "--ablations"], check=True)
spec = importlib.util.spec_from_file_location("b", os.path.join(REPO, "rlaif-v_amd", "build.py"))
b = importlib.util.module_from_spec(spec)
spec.loader.exec_module(b)
os.environ["RV_BUILD_ONLY"] = "attention.hip"
b.build_extension()
prof = "--prof" in sys.argv              # phase-stamped builds (RV_DKV4_PROF): librlaifv_hip_prof<n>.so, n = 0 is the full body
ns = [int(x) for x in sys.argv[1:] if x.isdigit()] or ([0, 3, 5, 6] if prof else list(range(1, 9)))


def build(n):
    defs = (("RV_DKV4_PROF",) if prof else ()) + ((f"RV_DKV4_ABL={n}",) if n else ())
    return b.build_extension(force=True, verbose=False, defines=defs, tag=f"_{'prof' if prof else 'abl'}{n}")


with cf.ThreadPoolExecutor(max_workers=4) as ex:
    for lib in ex.map(build, ns):
        print(lib)
