#!/usr/bin/env python3
"""bench.py's producer leg on its own (for rocprofv3 --kernel-trace: which kernels a tile of the device feed costs).  GPU."""
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from basevarc_amd import Context

ctx = Context(0)
leg = bench.producer_leg(ctx, 0.001, np, torch, torch.device("cuda:0"))
print(json.dumps({k: v for k, v in leg.items() if k != "workload"}))
