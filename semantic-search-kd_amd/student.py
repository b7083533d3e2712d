"""``StudentModel`` — the reference's bi-encoder wrapper, backed by the gfx950 encoder.

The reference file ``src/models/student.py`` is absent from its checkout; the interface below
is reconstructed line by line from its call sites and tests:

* ``StudentModel(model_name, device=None)``; ``device is None`` -> ``"cuda"`` when available
  else ``"cpu"`` — tests/test_student_model.py:12-36, src/serve/app.py:87-90
* ``.model`` (the sentence encoder), ``.device``, ``.embedding_dim``, ``.max_length`` —
  src/kd/train.py:126-127, src/serve/app.py:428, tests/conftest.py:80-81
* ``encode(texts, batch_size, show_progress, convert_to_numpy=True, normalize=True)``: ``str`` is
  wrapped into a list and passed as the first positional argument of exactly one
  ``self.model.encode(...)`` call — tests/test_student_model.py:38-70, src/serve/app.py:385-389
* ``encode_queries`` / ``encode_documents``: ``"query: "`` / ``"passage: "`` prefixes for E5 models —
  tests/test_student_model.py:72-102, src/kd/eval.py:67-75
* ``compute_similarity(q, d) -> q @ d.T`` — tests/test_student_model.py:104-124, src/kd/eval.py:75
* ``cleanup()`` — tests/test_student_model.py:126-137

``SentenceTransformer`` below is the module-level name the reference's tests patch
(``@patch("src.models.student.SentenceTransformer")``); here it is the MI355X encoder.
"""
from __future__ import annotations

from typing import List, Optional, Union

import numpy as np
import torch

from . import _native
from .encoder import Mi355xSentenceEncoder as SentenceTransformer

E5_QUERY_PREFIX = "query: "
E5_PASSAGE_PREFIX = "passage: "


class StudentModel:
    def __init__(self, model_name: str, device: Optional[str] = None, prefix_mode: str = "auto") -> None:
        """``prefix_mode``: ``"auto"`` (E5 when the model name contains "e5" — the only rule the
        reference's tests pin), ``"e5"`` or ``"none"`` (the production default is a local path,
        src/config.py:25-28, for which the reference's rule is unpinned)."""
        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"
        self.model_name = model_name
        self.device = device
        if prefix_mode not in ("auto", "e5", "none"):
            raise ValueError(f"prefix_mode={prefix_mode!r}")
        self.prefix_mode = prefix_mode
        self.model = SentenceTransformer(model_name, device=device)
        self.embedding_dim = self.model.get_sentence_embedding_dimension()
        self.max_length = getattr(self.model, "max_seq_length", 512)

    @classmethod
    def from_encoder(cls, encoder, model_name: str, prefix_mode: str = "auto") -> "StudentModel":
        """Wrap an already constructed sentence encoder (synthetic weights in benchmarks / tests; the
        reference's tests inject a mock the same way: tests/test_student_model.py:12-19)."""
        self = cls.__new__(cls)
        self.model_name = model_name
        self.device = str(encoder.device)
        self.prefix_mode = prefix_mode
        self.model = encoder
        self.embedding_dim = encoder.get_sentence_embedding_dimension()
        self.max_length = getattr(encoder, "max_seq_length", 512)
        return self

    # ------------------------------------------------------------------ encode
    @property
    def is_e5(self) -> bool:
        if self.prefix_mode == "auto":
            return "e5" in str(self.model_name).lower()
        return self.prefix_mode == "e5"

    def encode(
        self,
        texts: Union[str, List[str]],
        batch_size: int = 32,
        show_progress: bool = False,
        convert_to_numpy: bool = True,
        normalize: bool = True,
    ) -> np.ndarray:
        if isinstance(texts, str):
            texts = [texts]
        return self.model.encode(
            list(texts),
            batch_size=batch_size,
            show_progress_bar=show_progress,
            convert_to_numpy=convert_to_numpy,
            normalize_embeddings=normalize,
        )

    def encode_queries(self, queries: Union[str, List[str]], **kwargs) -> np.ndarray:
        if isinstance(queries, str):
            queries = [queries]
        if self.is_e5:
            queries = [E5_QUERY_PREFIX + q for q in queries]
        return self.encode(list(queries), **kwargs)

    def encode_documents(
        self, documents: Union[str, List[str]], batch_size: int = 32, show_progress: bool = False, **kwargs
    ) -> np.ndarray:
        if isinstance(documents, str):
            documents = [documents]
        if self.is_e5:
            documents = [E5_PASSAGE_PREFIX + d for d in documents]
        return self.encode(list(documents), batch_size=batch_size, show_progress=show_progress, **kwargs)

    def encode_documents_device(self, documents: List[str], batch_size: int = 32):
        """``encode_documents`` whose result stays in HBM (a ``torch`` fp32 ``[n, 384]`` device tensor): what the index
        build and the ANCE refresh feed straight into ``FAISSIndexBuilder.add`` - no 1.5 GB-per-million-rows round trip
        through host NumPy.  Not part of the reference's surface (its index lives on the CPU)."""
        if self.is_e5:
            documents = [E5_PASSAGE_PREFIX + d for d in documents]
        return self.model.encode(list(documents), batch_size=batch_size, convert_to_numpy=False, convert_to_tensor=True,
                                 normalize_embeddings=True)

    def encode_with_gradients(self, texts: List[str], normalize: bool = True):
        """Training entry point of the reference (src/kd/train.py:180-187): embeddings ``[n, 384]`` on
        ``self.device`` that carry gradients back to ``self.model.parameters()``.  Forward (saving
        activations) and backward are HIP kernels behind one autograd function (training.py); E5
        prefixes are the caller's business here exactly as in the reference's training loop."""
        if isinstance(texts, str):
            texts = [texts]
        tok = self.model.tokenize(list(texts))
        return self.model.trainable()(tok["input_ids"], tok["attention_mask"], normalize=normalize)

    # -------------------------------------------------------------- similarity
    def compute_similarity(self, query_embeddings: np.ndarray, doc_embeddings: np.ndarray) -> np.ndarray:
        """``q @ d.T`` in fp32 on the GPU (same fma order as the index scan)."""
        _native.require_gpu()
        lib = _native.load()
        q = np.ascontiguousarray(np.asarray(query_embeddings, np.float32))
        d = np.ascontiguousarray(np.asarray(doc_embeddings, np.float32))
        if q.ndim == 1:
            q = q[None]
        if d.ndim == 1:
            d = d[None]
        if q.shape[1] != d.shape[1]:
            raise ValueError(f"dimension mismatch: {q.shape} vs {d.shape}")
        if q.shape[1] % 8:
            raise ValueError("embedding dimension must be a multiple of 8")
        dev = torch.device(self.device if self.device != "cuda" else f"cuda:{torch.cuda.current_device()}")
        qd, dd = torch.from_numpy(q).to(dev), torch.from_numpy(d).to(dev)
        out = torch.empty((q.shape[0], d.shape[0]), dtype=torch.float32, device=dev)
        _native.check(
            lib.sskd_similarity(
                qd.data_ptr(), q.shape[0], dd.data_ptr(), d.shape[0], q.shape[1], out.data_ptr(),
                int(torch.cuda.current_stream(dev).cuda_stream),
            )
        )
        return out.cpu().numpy()

    def cleanup(self) -> None:
        """Release cached device buffers (no-op when nothing is cached)."""
        cleanup = getattr(self.model, "cleanup", None)
        if callable(cleanup):
            cleanup()
        if isinstance(self.device, str) and self.device.startswith("cuda") and torch.cuda.is_available():
            torch.cuda.empty_cache()
