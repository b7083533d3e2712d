"""rocprofv3 target: only the teacher cross-encoder leg of bench.py (BASELINE cfg 5 model)."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from semantic_search_kd_amd.bench_support import bench_teacher  # noqa: E402

print(json.dumps(bench_teacher(torch.device("cuda:0"), 1, 3, 1, torch.cuda.synchronize)))   # the barrier MUST synchronise: the timed region is asynchronous
