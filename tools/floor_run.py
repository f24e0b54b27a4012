"""Development aid: the scheme's issue floor (bench.scheme_floor) by itself, for counter passes."""
import json, sys
sys.path.insert(0, ".")
import bench
print(json.dumps(bench.scheme_floor(8 * 128 ** 3, 2.6)))
