import sys, json
for line in sys.stdin:
    line = line.strip()
    if line.startswith('{'):
        d = json.loads(line); r = d["results"]["tile_batch"]; print(sys.argv[1], "prop_ms %.1f wall %.1f plans/s %.0f" % (r["prop_ms"], r["wall_ms"], r["plans_per_s"]))
    elif line.startswith('phase') or 'Error' in line: print(line)
