"""MI355X-native embedding-and-search path behind the semantic-search-kd API.

Drop-in for the reference's ``StudentModel`` (sentence-transformers encoder) and
``FAISSIndexBuilder`` (FAISS index) — reference: src/serve/app.py:21-23 — with the
arithmetic in hand-written gfx950 HIP kernels reached through the C-ABI in
``include/sskd_amd.h``.  The directory is named ``semantic-search-kd_amd``; import it
as ``semantic_search_kd_amd`` (the sibling alias package points here).
"""
from . import _native  # noqa: F401
from .index import FAISSIndexBuilder, IndexHandle, read_flat_ip, write_flat_ip  # noqa: F401

from .weights import BertConfig, synthetic_state_dict  # noqa: F401
from .encoder import Mi355xSentenceEncoder, build_wordpiece_tokenizer  # noqa: F401
from .student import StudentModel  # noqa: F401
from .losses import CombinedKDLoss, ContrastiveLoss, ListwiseKDLoss, MarginMSELoss  # noqa: F401
from .mining import ANCEMiner, TeacherMiner  # noqa: F401
from .teacher import TeacherConfig, TeacherModel  # noqa: F401
from .bench_support import bench_encode  # noqa: F401

Mi355xIndexBuilder = FAISSIndexBuilder

__all__ = [
    "ANCEMiner",
    "TeacherMiner",
    "TeacherConfig",
    "TeacherModel",
    "CombinedKDLoss",
    "ContrastiveLoss",
    "ListwiseKDLoss",
    "MarginMSELoss",
    "StudentModel",
    "Mi355xSentenceEncoder",
    "FAISSIndexBuilder",
    "Mi355xIndexBuilder",
    "IndexHandle",
    "BertConfig",
    "synthetic_state_dict",
    "build_wordpiece_tokenizer",
    "read_flat_ip",
    "write_flat_ip",
]
