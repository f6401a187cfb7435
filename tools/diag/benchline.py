import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["config"]["rows"], d.get("ms_first_step"), d["stage_ms_per_step"])
