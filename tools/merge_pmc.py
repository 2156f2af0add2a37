"""Merge the JSON lines of tools/pmc_pass.sh (one file per pass) into the per-precision summary kept under profiles/.

usage: python tools/merge_pmc.py <commit> bf16=a.jsonl,b.jsonl fp32=c.jsonl > profiles/rNN_pmc.json
mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)."""
import json
import sys


def main():
    out = {"commit": sys.argv[1],
           "what": "rocprofv3 --kernel-trace --pmc passes (tools/pmc_pass.sh) over bench.py --precision bf16 / fp32, means per launch; "
                   "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES counts cycles, "
                   "GRBM_GUI_ACTIVE is summed over the 8 XCDs"}
    for arg in sys.argv[2:]:
        prec, files = arg.split("=")
        per = {}
        for f in files.split(","):
            for line in open(f):
                line = line.strip()
                if not line.startswith("{"):
                    continue
                rec = json.loads(line)
                k = per.setdefault(rec.pop("kernel"), {})
                rec.pop("launches", None)
                k.update(rec)
        for k in per.values():
            if "SQ_VALU_MFMA_BUSY_CYCLES" in k and k.get("GRBM_GUI_ACTIVE"):
                k["mfma_busy_frac"] = k["SQ_VALU_MFMA_BUSY_CYCLES"] / (k["GRBM_GUI_ACTIVE"] / 8 * 1024)
        out[prec] = per
    json.dump(out, sys.stdout, indent=1)


main()
