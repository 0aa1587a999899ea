#!/usr/bin/env python3
"""VGPRs / scratch of the fused step kernels in a -save-temps .s file: tools/kregs.py file.s [H ...]"""
import re, sys
txt = open(sys.argv[1]).read()
shapes = set(sys.argv[2:]) or {"25"}
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", txt, re.S):
    name, body = m.group(1), m.group(2)
    if "k_env_rollout_rowlane" not in name:
        continue
    a = re.search(r"ILi(\d+)ELi(\d+)ELb(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)E", name)
    if a and a.group(1) in shapes:
        v = re.search(r"next_free_vgpr (\d+)", body).group(1)
        sc = re.search(r"private_segment_fixed_size (\d+)", body).group(1)
        print("H=%s W=%s LDS_LUT=%s SPAWN=%s WRAP=%s LEAN=%s ONE=%s" % a.groups(), "vgpr", v, "scratch", sc)
