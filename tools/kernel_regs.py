#!/usr/bin/env python3
"""Register / spill report of the gfx950 code objects in concrete-ntt_amd/csrc/_obj (reads the ELF notes).
    python tools/kernel_regs.py [unit ...] [--all]      default: kernels with spills or >= 160 VGPRs"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
OBJ = os.path.join(ROOT, "concrete-ntt_amd", "csrc", "_obj")


def kernels(unit, tmp):
    obj = os.path.join(OBJ, unit + ".o")
    fat, co = os.path.join(tmp, unit + ".fat"), os.path.join(tmp, unit + ".co")
    subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", obj, fat], check=True)
    subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                    "--input=" + fat, "--output=" + co, "--unbundle"], check=True)
    notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
    for blk in notes.split("- .agpr_count")[1:]:
        g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = dem.replace("cntt::", "").replace("unsigned long", "u64").replace("unsigned int", "u32")
        yield dem.split("(")[0], g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    show_all = "--all" in sys.argv
    units = args or sorted(f[:-2] for f in os.listdir(OBJ) if f.endswith(".o") and f != "host.o")
    with tempfile.TemporaryDirectory() as tmp:
        for u in units:
            for name, vg, sg, sp, scratch, lds in kernels(u, tmp):
                if show_all or sp or vg >= 160:
                    print("%-24s %-64s vgpr %3d sgpr %3d spill %3d scratch %4d lds %6d" % (u, name[:64], vg, sg, sp, scratch, lds))


if __name__ == "__main__":
    try:
        main()
    except BrokenPipeError:  # `| head`
        sys.exit(0)
