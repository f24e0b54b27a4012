import sys; sys.path.insert(0,'.')
import bench, json
for st in (64, 256, 1024):
    print(st, json.dumps(bench.scheme_floor(8*128**3, 2.61, steps=st)))
