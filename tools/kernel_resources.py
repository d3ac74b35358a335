"""Register / LDS / scratch table of the built kernels: unbundles the gfx950 code objects from rlaif-v_amd/build*/*.o and reads
their metadata notes.  Usage: python tools/kernel_resources.py [substring] [--build build_exp]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def resources(build_dir):
    rows = []
    for o in sorted(os.listdir(build_dir)):
        if not o.endswith(".o"):
            continue
        with tempfile.TemporaryDirectory() as td:
            fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
            subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", os.path.join(build_dir, o)], check=True)
            subprocess.run([f"{LLVM}/clang-offload-bundler", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            f"--input={fat}", f"--output={co}", "--unbundle"], check=True)
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
            dem = {}
            cur = {}
            for line in notes.splitlines():
                m = re.match(r"\s+-?\s*\.(\w+):\s+(.*)", line)
                if not m:
                    continue
                k, v = m.group(1), m.group(2).strip().strip("'")
                if k == "agpr_count" and cur.get("name"):
                    pass
                if k in ("agpr_count",) and "agpr_count" in cur:
                    rows.append(cur)
                    cur = {}
                cur[k] = v
                if k == "wavefront_size":
                    rows.append(cur)
                    cur = {}
    out = []
    for r in rows:
        if "name" not in r:
            continue
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(anonymous namespace\)::|void ", "", name).split("(")[0]
        out.append((name, int(r.get("vgpr_count", 0)), int(r.get("agpr_count", 0)), int(r.get("sgpr_count", 0)),
                    int(r.get("vgpr_spill_count", 0)), int(r.get("group_segment_fixed_size", 0)),
                    int(r.get("private_segment_fixed_size", 0))))
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    bd = "build"
    if "--build" in sys.argv:
        bd = sys.argv[sys.argv.index("--build") + 1]
        args = [a for a in args if a != bd]
    sub = args[0] if args else ""
    print(f"{'kernel':100s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} {'spill':>5s} {'lds':>7s} {'scratch':>7s}")
    for r in resources(os.path.join(REPO, "rlaif-v_amd", bd)):
        if sub in r[0]:
            print(f"{r[0][:100]:100s} {r[1]:5d} {r[2]:5d} {r[3]:5d} {r[4]:5d} {r[5]:7d} {r[6]:7d}")
