import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic"], d["cpu_baseline"]["value"], d["cpu_baseline"]["gpu_picture_0_identical"])
print(json.dumps(d["random_access"])[:700])
