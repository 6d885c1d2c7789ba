#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel in a hipcc -save-temps .s file.   usage: python tools/kres.py file.s [filter]"""
import re, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
    name = g("name")
    if flt in name:
        print(f"{name[:70]:70s} vgpr {g('vgpr_count'):>4} agpr {blk.split()[0]:>3} spill {g('vgpr_spill_count'):>3} scratch {g('private_segment_fixed_size'):>4} lds {g('group_segment_fixed_size'):>6} sgpr {g('sgpr_count')}")
