import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["value"]), round(d["ms_per_step"],3), [ (l["level"], round(l["avg_launch_ms"],3), round(l["achieved_GBs"]), l["threads"], l["lds_bytes"]) for l in d["roofline"]["levels"]], "ref-term", round(d["reference_termination"]["value"]))
