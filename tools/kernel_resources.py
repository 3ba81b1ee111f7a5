"""Summarises hipcc -Rpass-analysis=kernel-resource-usage output (csrc/build/resource_usage.txt)."""
import re, sys
t = open(sys.argv[1] if len(sys.argv) > 1 else "depth_completion_mt_amd/csrc/build/resource_usage.txt").read()
for blk in t.split("Function Name: ")[1:]:
    name = blk.split()[0]
    def g(k):
        m = re.search(re.escape(k) + r": (\d+)", blk)
        return m.group(1) if m else "?"
    print(f"{name[:60]:60s} VGPR {g('VGPRs'):>3} SGPR {g('TotalSGPRs'):>3} vspill {g('VGPRs Spill')} scratch {g('ScratchSize [bytes/lane]')} "
          f"LDS {g('LDS Size [bytes/block]'):>6} occ {g('Occupancy [waves/SIMD]')}")
