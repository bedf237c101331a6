import sys, json, argparse
sys.path.insert(0, '.')
import numpy as np, torch
import bench
r = bench.c1_leg(argparse.Namespace(), torch, np, torch.device('cuda', 0))
print(json.dumps(r)[:3000])
