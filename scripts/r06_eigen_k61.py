"""secondary.eigen_k61 of bench.py on its own: python scripts/r06_eigen_k61.py [levels] [k] [columns]"""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
a = [int(x) for x in sys.argv[1:]]
print(json.dumps(bench.eigen_k61_measurement(0, *a), indent=1))
