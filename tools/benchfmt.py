import json,sys
for line in sys.stdin:
    line=line.strip()
    if line.startswith('{'):
        d=json.loads(line)
        print("ms/step %.4f  value %.4g"%(d["ms_per_step"], d["value"]))
        print({k:round(v["ms_per_launch"],4) for k,v in d["kernels"].items()})
        print("s64", round(d["minibatch_s64"]["ms_per_step"],4), "roofline", d["roofline"]["kernel"], round(d["roofline"]["frac"],3))
        if "cpu_baseline" in d: print(d["cpu_baseline"])
