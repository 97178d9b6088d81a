import sys, json
tag = sys.argv[1] if len(sys.argv) > 1 else ""
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(tag, d["value"], d["roofline"]["kernel_ms"], d["roofline"]["frac"])
