"""Register / LDS / spill metadata of every kernel of a .hip translation unit, from the compiler's own listing (no GPU needed).
python tools/kernel_resources.py hippie_amd/csrc/conv_mfma.hip [extra hipcc flags]"""
import re
import subprocess
import sys

FIELDS = ("agpr_count", "group_segment_fixed_size", "name", "private_segment_fixed_size", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count")


def kernel_resources(src, extra=()):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-S", "--cuda-device-only", src, "-o", "-"] + list(extra)
    asm = subprocess.run(cmd, check=True, capture_output=True, text=True).stdout
    out = []
    for blk in asm.split("- .agpr_count:")[1:]:
        blk = ".agpr_count:" + blk
        rec = {}
        for f in FIELDS:
            m = re.search(r"\.%s:\s+(\S+)" % f, blk)
            rec[f] = m.group(1) if m else None
        out.append(rec)
    names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in out), capture_output=True, text=True).stdout.splitlines()
    for r, n in zip(out, names):
        r["demangled"] = n
        for f in FIELDS:
            if f != "name":
                r[f] = int(r[f])
    return out


if __name__ == "__main__":
    for r in kernel_resources(sys.argv[1], sys.argv[2:]):
        print(f"{r['demangled'][:86]:86s} vgpr {r['vgpr_count']:3d} agpr {r['agpr_count']:3d} vspill {r['vgpr_spill_count']:2d} sspill {r['sgpr_spill_count']:2d} "
              f"scratch {r['private_segment_fixed_size']:3d} lds {r['group_segment_fixed_size']}")
