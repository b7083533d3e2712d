"""``SentenceTransformer.encode()``-shaped bi-encoder running on the gfx950 HIP kernels.

This is the object the reference's ``StudentModel`` holds as ``.model`` (reference:
src/kd/train.py:126-127, tests/test_student_model.py:12-70): constructed from a model path and a
device, exposing ``encode(...)``, ``get_sentence_embedding_dimension()`` and
``max_seq_length``.  The pipeline is the e5-small-v2 one — Transformer -> Pooling(mean) ->
Normalize (tests/test_model_validation.py:80-89,256-262) — executed by
``sskd_encoder_forward`` (bf16 MFMA encoder + fused mean-pool / L2-normalise).
There is no CPU path: without an MI355X and the built library, construction fails.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from . import _native
from .weights import BertConfig, DeviceWeights, load_config, load_state_dict, synthetic_state_dict

CLS_ID, SEP_ID, PAD_ID = 101, 102, 0


class Mi355xSentenceEncoder:
    def __init__(
        self,
        model_name_or_path: Union[str, Path, None] = None,
        device: Optional[str] = None,
        *,
        config: Optional[BertConfig] = None,
        state_dict: Optional[Dict[str, np.ndarray]] = None,
        tokenizer=None,
        max_seq_length: Optional[int] = None,
    ) -> None:
        """Load weights from a LOCAL HF / sentence-transformers directory, or take an explicit
        ``config`` + ``state_dict`` (synthetic weights).  Model *names* are not fetched."""
        _native.require_gpu()
        self.device = _resolve_device(device)
        model_dir = None
        if state_dict is None:
            if model_name_or_path is None:
                raise ValueError("pass a local model directory or config= / state_dict=")
            model_dir = Path(model_name_or_path)
            if not model_dir.is_dir():
                raise FileNotFoundError(
                    f"{model_name_or_path!r} is not a local directory. This backend never downloads "
                    "checkpoints: point it at a directory holding config.json + model.safetensors + tokenizer.json"
                )
            config = load_config(model_dir)
            state_dict = load_state_dict(model_dir)
        self.config = config or BertConfig()
        self.weights = DeviceWeights(self.config, state_dict, self.device)
        self.tokenizer = tokenizer if tokenizer is not None else _load_tokenizer(model_dir)
        st_max = _read_st_max_len(model_dir)
        self.max_seq_length = int(
            max_seq_length or st_max or min(512, self.config.max_position_embeddings)
        )
        self._workspace: Optional[torch.Tensor] = None

    # ----------------------------------------------------------- constructors
    @classmethod
    def from_synthetic(
        cls, config: Optional[BertConfig] = None, device: Optional[str] = None, tokenizer=None,
        stress: bool = False, **kw
    ) -> "Mi355xSentenceEncoder":
        """Random-init weights of the e5-small-v2 architecture (deterministic recipe, weights.py);
        ``stress`` = the trained-checkpoint-like hard-case recipe (peaky attention, outlier channels)."""
        cfg = config or BertConfig()
        sd = synthetic_state_dict(cfg, stress=stress)
        return cls(None, device, config=cfg, state_dict=sd, tokenizer=tokenizer, **kw)

    # ------------------------------------------------------ SentenceTransformer API
    def get_sentence_embedding_dimension(self) -> int:
        return self.config.hidden_size

    def tokenize(self, texts: Sequence[str]) -> Dict[str, np.ndarray]:
        """WordPiece ids, padded to the longest text, truncated to ``max_seq_length``."""
        if self.tokenizer is None:
            raise RuntimeError(
                "no tokenizer: the model directory has neither tokenizer.json nor vocab.txt "
                "(use encode_token_ids for pre-tokenised input)"
            )
        encs = self.tokenizer.encode_batch(list(texts))
        rows = [e.ids[: self.max_seq_length] for e in encs]
        for r, e in zip(rows, encs):
            if len(e.ids) > self.max_seq_length and r:
                r[-1] = e.ids[-1]  # keep the closing [SEP] when truncating
        width = max((len(r) for r in rows), default=1) or 1
        ids = np.full((len(rows), width), PAD_ID, np.int32)
        mask = np.zeros((len(rows), width), np.int32)
        for i, r in enumerate(rows):
            ids[i, : len(r)] = r
            mask[i, : len(r)] = 1
        return {"input_ids": ids, "attention_mask": mask}

    def encode_token_ids(
        self, input_ids, attention_mask=None, normalize: bool = True, out: Optional[torch.Tensor] = None
    ) -> torch.Tensor:
        """One forward pass over pre-tokenised ``[B, S]`` int32 ids; returns fp32 ``[B, 384]`` on device.

        Everything is enqueued on the current stream; no host synchronisation.
        """
        lib = _native.load()
        ids = _as_device_i32(input_ids, self.device)
        if ids.dim() != 2:
            raise ValueError(f"input_ids must be [B, S], got {tuple(ids.shape)}")
        B, S = ids.shape
        mask = torch.ones_like(ids) if attention_mask is None else _as_device_i32(attention_mask, self.device)
        if mask.shape != ids.shape:
            raise ValueError("attention_mask shape differs from input_ids")
        if out is None:
            out = torch.empty((B, self.config.hidden_size), dtype=torch.float32, device=self.device)
        if B == 0:
            return out
        need = int(lib.sskd_encoder_workspace_bytes(self.weights.cstruct_cfg, B, S))
        if self._workspace is None or self._workspace.numel() < need:
            self._workspace = None
            self._workspace = torch.empty(need, dtype=torch.uint8, device=self.device)
        _native.check(
            lib.sskd_encoder_forward(
                self.weights.cstruct_cfg,
                self.weights.struct,
                ids.data_ptr(),
                mask.data_ptr(),
                B,
                S,
                int(bool(normalize)),
                out.data_ptr(),
                self._workspace.data_ptr(),
                self._workspace.numel(),
                int(torch.cuda.current_stream(self.device).cuda_stream),
            )
        )
        return out

    def hidden_states(self, input_ids, attention_mask=None) -> torch.Tensor:
        """bf16 ``[B, S, 384]`` output of the last encoder layer (test hook)."""
        lib = _native.load()
        ids = _as_device_i32(input_ids, self.device)
        B, S = ids.shape
        mask = torch.ones_like(ids) if attention_mask is None else _as_device_i32(attention_mask, self.device)
        out = torch.empty((B, S, self.config.hidden_size), dtype=torch.bfloat16, device=self.device)
        need = int(lib.sskd_encoder_workspace_bytes(self.weights.cstruct_cfg, B, S))
        ws = torch.empty(max(need, 1), dtype=torch.uint8, device=self.device)
        _native.check(
            lib.sskd_encoder_hidden(
                self.weights.cstruct_cfg, self.weights.struct, ids.data_ptr(), mask.data_ptr(), B, S,
                out.data_ptr(), ws.data_ptr(), ws.numel(), int(torch.cuda.current_stream(self.device).cuda_stream),
            )
        )
        return out

    def encode(
        self,
        sentences: Union[str, List[str]],
        batch_size: int = 32,
        show_progress_bar: bool = False,
        convert_to_numpy: bool = True,
        convert_to_tensor: bool = False,
        normalize_embeddings: bool = False,
        device: Optional[str] = None,
        **_ignored,
    ):
        """Same contract as ``SentenceTransformer.encode``: sort by length (longest first),
        batch, pad per batch, forward, restore the input order.  The e5 pipeline ends in a
        ``Normalize`` module, so embeddings are unit-norm whatever ``normalize_embeddings`` says."""
        del show_progress_bar, device, normalize_embeddings
        single = isinstance(sentences, str)
        texts = [sentences] if single else list(sentences)
        n = len(texts)
        out = torch.empty((n, self.config.hidden_size), dtype=torch.float32, device=self.device)
        if n:
            tok = self.tokenize(texts)
            lengths = tok["attention_mask"].sum(axis=1)
            order = np.argsort(-lengths, kind="stable")
            with torch.cuda.device(self.device):
                for lo in range(0, n, batch_size):
                    idx = order[lo : lo + batch_size]
                    width = max(int(lengths[idx].max()), 1)
                    emb = self.encode_token_ids(
                        tok["input_ids"][idx, :width], tok["attention_mask"][idx, :width], normalize=True
                    )
                    out[torch.from_numpy(idx).to(self.device)] = emb
        if convert_to_tensor and not convert_to_numpy:
            return out[0] if single else out
        arr = out.cpu().numpy()
        return arr[0] if single else arr

    def cleanup(self) -> None:
        self._workspace = None


# ---------------------------------------------------------------------- helpers
def _resolve_device(device: Optional[str]) -> torch.device:
    if device is None or device == "cuda":
        return torch.device("cuda", torch.cuda.current_device())
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError(
            f"device={device!r}: the encoder runs on MI355X only (PyTorch-ROCm spells it 'cuda[:N]'); no CPU path"
        )
    return dev


def _as_device_i32(x, device: torch.device) -> torch.Tensor:
    t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(np.asarray(x)))
    return t.to(device=device, dtype=torch.int32).contiguous()


def _load_tokenizer(model_dir: Optional[Path]):
    if model_dir is None:
        return None
    from tokenizers import Tokenizer

    tj = model_dir / "tokenizer.json"
    if tj.exists():
        tok = Tokenizer.from_file(str(tj))
        tok.no_padding()
        tok.no_truncation()
        return tok
    vocab = model_dir / "vocab.txt"
    if vocab.exists():
        return build_wordpiece_tokenizer(vocab.read_text().splitlines())
    return None


def build_wordpiece_tokenizer(vocab_tokens: Sequence[str]):
    """Uncased BERT WordPiece tokenizer over an explicit vocabulary (``[CLS] x [SEP]`` template)."""
    from tokenizers import Tokenizer, models, normalizers, pre_tokenizers, processors

    vocab = {t: i for i, t in enumerate(vocab_tokens)}
    tok = Tokenizer(models.WordPiece(vocab, unk_token="[UNK]", max_input_chars_per_word=100))
    tok.normalizer = normalizers.BertNormalizer(lowercase=True)
    tok.pre_tokenizer = pre_tokenizers.BertPreTokenizer()
    tok.post_processor = processors.TemplateProcessing(
        single="[CLS] $A [SEP]",
        pair="[CLS] $A [SEP] $B:1 [SEP]:1",
        special_tokens=[("[CLS]", vocab["[CLS]"]), ("[SEP]", vocab["[SEP]"])],
    )
    return tok


def _read_st_max_len(model_dir: Optional[Path]) -> Optional[int]:
    if model_dir is None:
        return None
    p = model_dir / "sentence_bert_config.json"
    if p.exists():
        try:
            return int(json.loads(p.read_text()).get("max_seq_length"))
        except Exception:
            return None
    return None
