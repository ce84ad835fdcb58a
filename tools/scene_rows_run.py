import sys
sys.path.insert(0, '.')
import bench_rows, json
print(json.dumps(bench_rows.scene_rows()))
