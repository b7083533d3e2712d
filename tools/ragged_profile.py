"""Diagnostic: runs only the ragged-length encode leg (for rocprofv3 --kernel-trace --stats)."""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from semantic_search_kd_amd.bench_support import bench_encode_ragged  # noqa: E402
from semantic_search_kd_amd.encoder import Mi355xSentenceEncoder  # noqa: E402
from semantic_search_kd_amd.weights import BertConfig  # noqa: E402

dev = torch.device("cuda:0")
enc = Mi355xSentenceEncoder.from_synthetic(BertConfig(), device=str(dev))
print(json.dumps(bench_encode_ragged(enc, dev, passes=5)))
