"""One line per bench.py JSON output: python tools/benchsum.py a.json b.json ...  (no arguments: reads stdin)."""
import json
import sys


def line(text, tag=""):
    d = json.loads(text.strip().splitlines()[-1])
    r = d["roofline"]
    if "launches" in r:
        lv = [(l["levels"], l["kind"], round(l["avg_launch_ms"], 3), round(l["achieved_GBs"]), l["threads"]) for l in r["launches"]]
    else:             # a line written by an earlier round's bench.py
        lv = [(l["level"], round(l["avg_launch_ms"], 3), round(l["achieved_GBs"]), l["threads"], l["lds_bytes"]) for l in r["levels"]]
    rt = d.get("reference_termination") or {}
    print(tag, round(d["value"]), round(d["ms_per_step"], 3), "frac", round(d["roofline"]["frac"], 3),
          "all", round(d["roofline"]["all_levels_frac"], 3), lv, "its", [round(v, 2) for v in d["iterations_per_pair"]],
          "ref-term", round(rt.get("value", 0)), "serial", round((d.get("one_enqueue_at_a_time") or {}).get("value", 0)))


if len(sys.argv) == 3 and not sys.argv[2].endswith(".json"):      # file, tag
    line(open(sys.argv[1]).read(), sys.argv[2])
elif len(sys.argv) > 1:
    for f in sys.argv[1:]:
        line(open(f).read(), f)
else:
    line(sys.stdin.read())
