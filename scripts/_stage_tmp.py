import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["roofline"]["stages_ms_per_frame"])
