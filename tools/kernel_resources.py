"""Per-kernel register / LDS / scratch use from a device assembly file (hipcc --cuda-device-only -S):
python tools/kernel_resources.py file.s [substring]"""
import re, subprocess, sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"- \.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?"
                     r"\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)\s+\.vgpr_spill_count:\s+(\d+)", txt, re.S):
    agpr, lds, name, scratch, sgpr, vgpr, spill = m.groups()
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("(eab_conv_desc)", "")
    if flt in dn:
        print(f"vgpr {vgpr:>3} agpr {agpr:>3} spill {spill:>3} scratch {scratch:>4} lds {lds:>6}  {dn}")
