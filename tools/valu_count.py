"""Static issue-cost estimate of a kernel's hot loop from the hipcc .s (make -C csrc asm).

Costs are the measured MI355X issue rates of tools/dpp_probe.hip (cycles per wave64 instruction
per SIMD with 8 waves resident): v_add/v_mul/v_fma/v_mov ~2.4; v_max/v_min/*3/med3/cmp/cndmask/bfi,
every DPP form, integer min/max ~4.4; ds_bpermute ~24 (LDS crossbar, per CU 6).
usage: python tools/valu_count.py <kernel-name-substring> [steps_per_loop_iteration]
Prints every loop of the kernel with its VALU instruction count and estimated cycles."""
import re, sys, collections
path = "depth_completion_mt_amd/csrc/build/dcmt-hip-amdgcn-amd-amdhsa-gfx950.s"
key = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
txt = open(path).read()
m = re.search(r"^(_Z\S*%s\S*):" % re.escape(key), txt, re.M)
if not m:
    sys.exit("kernel not found")
name = m.group(1)
body = txt[m.end():txt.index(".Lfunc_end", m.end())]
FAST = ("v_add_f32", "v_mul_f32", "v_fma_f32", "v_mov_b32", "v_mov_b64", "v_sub_f32", "v_fmac_f32", "v_pk_")
lines = body.split("\n")
labels = {}
for i, line in enumerate(lines):
    lab = re.match(r"^(\.LBB\d+_\d+):", line)
    if lab:
        labels[lab.group(1)] = i
loops = []   # (header label, first line, last line) for every backward branch
for i, line in enumerate(lines):
    br = re.match(r"^\s+s_c?branch\S*\s+(\.LBB\d+_\d+)", line)
    if br and br.group(1) in labels and labels[br.group(1)] < i:
        loops.append((br.group(1), labels[br.group(1)], i))
print(name)
for h, a0, a1 in sorted(loops, key=lambda t: t[1] - t[2]):
    c = collections.Counter()
    for line in lines[a0:a1 + 1]:
        ins = re.match(r"^\s+([a-z_0-9]+)", line)
        if ins:
            c[ins.group(1)] += 1
    valu = {k: v for k, v in c.items() if k.startswith("v_")}
    nv = sum(valu.values())
    if nv < 20:
        continue
    fast = sum(v for k, v in valu.items() if k.startswith(FAST) and "dpp" not in k)
    slow = nv - fast
    lds = sum(v for k, v in c.items() if k.startswith("ds_"))
    nop = c.get("s_nop", 0)
    sal = sum(v for k, v in c.items() if k.startswith("s_")) - nop
    cyc = fast * 2.4 + slow * 4.4
    print(f"  loop {h}: VALU {nv} (fast {fast}, half-rate {slow})  est {cyc:.0f} cyc  | per step: VALU {nv/steps:.1f}, {cyc/steps:.0f} cyc | LDS {lds} s_nop {nop} SALU {sal} total {sum(c.values())}")
    top = sorted(valu.items(), key=lambda kv: -kv[1])[:14]
    print("     ", ", ".join(f"{k.replace('_e32','').replace('_e64','')} {v}" for k, v in top))
