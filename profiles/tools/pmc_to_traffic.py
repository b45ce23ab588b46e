#!/usr/bin/env python
"""pmc summary (profiles/summarize.py output of the --pmc FETCH_SIZE / WRITE_SIZE passes) -> latest_traffic.json, keyed by
the kernel names bench.py reports.  usage: pmc_to_traffic.py <pmc.txt> <out.json> <tag>"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from bench import csrc_sha16  # noqa: E402  (the key bench.py checks before it quotes these numbers)

KIND = {0: "phase", 1: "vfull", 2: "vu"}
NOISE = {0: "nb", 1: "poisson", 2: "lognormal"}
src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
vals = {}
pat = re.compile(r"(?:void )?vc_main_kernel<(\d+), (\d+), (\d+), (\d+), (\d+)(?:, (\d+))?>.*\| (FETCH_SIZE|WRITE_SIZE) \| (\d+) \| ([\d.]+)")
for line in open(src):
    m = pat.match(line)
    if m:
        h, nb, kind, noise, gpl = (int(x) for x in m.groups()[:5])
        c16 = int(m.group(6) or 0)
        # (last template argument: bit 0 = uint16 count storage, bit 1 = the gradient-only instantiation of vc_set_loss_every,
        # bit 2 = the U-only kernel with the nu_omega partials per lane)
        name = f"vc_main_kernel<{h},{nb},{KIND[kind]}_{NOISE[noise]}{'_gradonly' if c16 & 2 else ''},gpl{gpl}{',u16' if c16 & 1 else ''}{',pwl' if c16 & 4 else ''}>"
        vals.setdefault(name, {})[m.group(7) + "_KiB"] = float(m.group(9))
for v in vals.values():
    if "FETCH_SIZE_KiB" in v and "WRITE_SIZE_KiB" in v:
        v["traffic_bytes"] = int((2 * v["FETCH_SIZE_KiB"] + v["WRITE_SIZE_KiB"]) * 1024)
out = {"_comment": "HBM traffic per launch of the likelihood kernel from rocprofv3 PMC passes (separate --pmc FETCH_SIZE and "
                   f"--pmc WRITE_SIZE runs of the driver's bench command, {tag}). Unit: FETCH_SIZE/WRITE_SIZE are KiB; on gfx950 "
                   "FETCH_SIZE counts 128-B requests at 64 B for wide (16 B/lane) streaming reads, so the read side is doubled "
                   "(MI355X_MICROARCH.md, HBM section). bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.",
       "workload": "50000x2000", "csrc_sha16": csrc_sha16(), "kernels": vals}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(vals))
