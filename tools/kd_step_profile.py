"""rocprofv3 target: only the KD training-step leg of bench.py (BASELINE cfg 4)."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from semantic_search_kd_amd.bench_support import bench_kd_step  # noqa: E402

print(json.dumps(bench_kd_step(torch.device("cuda:0"), steps=3, warmup=1)))
