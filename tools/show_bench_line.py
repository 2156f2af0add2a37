"""Print the few numbers of a bench.py line that the round's notes quote.  usage: python tools/show_bench_line.py line.json [...]"""
import json
import sys

for path in sys.argv[1:]:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    c, sg = d.get("cli_end_to_end") or {}, d.get("sharded_gather") or {}
    print("%s: value %s  ms/step %.4f  frac %.3f | host-to-host %s | sharded_gather %s | cli %s in %.4f s (listing %.4f, rank0 %s)" % (
        path, d.get("value") or d.get("rehearsal_value"), d["ms_per_step"], (d.get("roofline") or {}).get("frac", 0.0), d.get("host_to_host_value"),
        sg.get("value"), c.get("value"), c.get("seconds", 0.0), c.get("listing_s", 0.0),
        {k: round(v, 4) for k, v in (c.get("rank0") or {}).items() if isinstance(v, float)}))
    for p, o in (d.get("other_precisions") or {}).items():
        print("   %s %.4g samples/s, max|dp| %.2e" % (p, o["value"], o["max_abs_dp_vs_fp64_oracle"]))
    if isinstance(d.get("config4"), dict):
        print("   config4 %s samples/s, label match %.5f" % (d["config4"].get("value"), d["config4"]["label_match_vs_fp32_oracle"]))
