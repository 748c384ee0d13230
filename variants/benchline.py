import json,sys
d=json.load(sys.stdin); r=d["roofline"]
print(sys.argv[1], d["value"], d["ms_per_step"], r["kernel"], r["frac"], r["whole_call"], r["per_kernel_ms"])
