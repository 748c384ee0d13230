import json,sys
d=json.load(sys.stdin); print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["per_kernel_ms"])
