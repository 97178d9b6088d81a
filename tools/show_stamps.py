import json, sys
d = json.load(open(sys.argv[1]))
print(d['cycles_per_block_per_wave'])
for k, v in d.items():
    if isinstance(v, dict):
        print("%-18s %.3f %7.0f  A %6.0f  B %6.0f" % (k, v['share'], v['cycles_per_block'], sum(v['by_wave'][:4]) / 4, sum(v['by_wave'][4:]) / 4))
