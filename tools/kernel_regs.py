"""Register / spill / scratch / LDS figures of every gfx950 kernel in the built library (no GPU needed): extracts the code
object from libmips_hip.so's .hip_fatbin and reads its note records.
    python tools/kernel_regs.py [substring ...]      # only kernels whose demangled name contains one of the substrings"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(lib=os.path.join(ROOT, "retrieval-augmented-mds_amd", "lib", "libmips_hip.so")):
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "k.co")
        subprocess.check_call([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        subprocess.check_call([f"{LLVM}/clang-offload-bundler", "--type=o", "--unbundle", f"--input={fat}", f"--output={co}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"])
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    out = []
    for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        g = lambda key: int(re.search(rf"\.{key}:\s+(\d+)", blk).group(1))
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        out.append({"name": name, "agpr": int(blk.split("\n")[0].strip()), "vgpr": g("vgpr_count"), "sgpr": g("sgpr_count"),
                    "spill": g("vgpr_spill_count"), "scratch": g("private_segment_fixed_size"), "lds": g("group_segment_fixed_size")})
    names = subprocess.run(["c++filt"], input="\n".join(k["name"] for k in out), capture_output=True, text=True).stdout.split("\n")
    for k, n in zip(out, names):
        k["demangled"] = re.sub(r"^void ", "", n).split("(")[0]
    return out


if __name__ == "__main__":
    ks = kernels()
    want = sys.argv[1:]
    bad = 0
    for k in ks:
        flag = k["spill"] or k["scratch"]
        bad += 1 if flag else 0
        if (want and any(w in k["demangled"] for w in want)) or (not want and flag):
            print(f'{k["demangled"][:100]:100s} vgpr {k["vgpr"]:3d} agpr {k["agpr"]:3d} spill {k["spill"]:3d} scratch {k["scratch"]:5d} lds {k["lds"]}')
    print(f"{len(ks)} kernels, {bad} with spills or scratch")
