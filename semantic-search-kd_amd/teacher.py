"""``TeacherModel`` - the reference's cross-encoder reranker, on the MI355X generic encoder.

``src/models/teacher.py`` is absent from the reference checkout; the surface below is reconstructed
from its call sites:

* ``TeacherModel(model_name="BAAI/bge-reranker-large", device=...)`` - src/serve/app.py:101-104,
  scripts/evaluate_production.py:182
* ``score(pairs, batch_size=32) -> sequence of float``; pairs are ``[q, d]`` lists or ``(q, d)``
  tuples - src/serve/app.py:325-326, src/mining/miners.py:135-137
* ``get_confidence(score) -> float in [0, 1]`` - src/mining/miners.py:148
* ``predict_score(q, d) -> float`` - scripts/evaluate_production.py:73; ``predict`` alias - tests/conftest.py:108

The model is ``XLMRobertaForSequenceClassification`` with one label (XLM-R-large: 24 layers, hidden
1024, 16 heads, FFN 4096 - docs/adr-002): encoder -> <s> hidden state -> dense + tanh -> out_proj.
``score`` returns the RAW logit (what the KD losses soften with a temperature) and ``get_confidence``
its sigmoid - the reference's choice is unpinned (its source is missing); INTEGRATION.md notes it.
Pure data parallel across GPUs: every rank scores its own slice of the pairs, no communication.
"""
from __future__ import annotations

import json
import math
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _native
from .weights import BertConfig, synthetic_state_dict, synthetic_tensor


@dataclass(frozen=True)
class TeacherConfig:
    """XLM-R-large defaults (bge-reranker-large): docs/adr-002, SURVEY.md section 8(f) rank 3."""

    vocab_size: int = 250002
    hidden_size: int = 1024
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    intermediate_size: int = 4096
    max_position_embeddings: int = 514
    type_vocab_size: int = 1
    layer_norm_eps: float = 1e-5
    pad_token_id: int = 1

    @staticmethod
    def from_json(path: Path) -> "TeacherConfig":
        raw = json.loads(Path(path).read_text())
        keys = TeacherConfig.__dataclass_fields__.keys()
        return TeacherConfig(**{k: raw[k] for k in keys if k in raw})

    def as_bert(self) -> BertConfig:
        return BertConfig(vocab_size=self.vocab_size, hidden_size=self.hidden_size,
                          num_hidden_layers=self.num_hidden_layers, num_attention_heads=self.num_attention_heads,
                          intermediate_size=self.intermediate_size, max_position_embeddings=self.max_position_embeddings,
                          type_vocab_size=self.type_vocab_size, layer_norm_eps=self.layer_norm_eps)


def synthetic_teacher_state_dict(cfg: TeacherConfig, recipe: str = "init") -> Dict[str, np.ndarray]:
    """Deterministic synthetic weights + classifier head.

    ``recipe="init"``: the random-init recipe of weights.synthetic_state_dict (every matrix at std 0.02).  Such a
    network barely mixes tokens - the <s> state, and with it the logit, is almost the same for every input.
    ``recipe="spread"``: gains shaped like a TRAINED checkpoint's - value / output projections and the FFN at
    unit gain (std 1 / sqrt(fan_in)) times 1.5 / 1.0, query / key at 1.5 / sqrt(H) (attention logits of O(2):
    peaky but not one-hot), word embeddings at unit variance - so that attention moves the <s> state by as much as
    the residual carries and the logits of different inputs spread over several units (std about 1.3).  This is
    what the score-level parity tests use: an ORDER of logits that a broken kernel cannot reproduce.
    """
    sd = synthetic_state_dict(cfg.as_bert())
    h, f = cfg.hidden_size, cfg.intermediate_size
    if recipe == "init":
        # head scales chosen so that logits of different inputs differ at O(0.1 - 1)
        sd["classifier.dense.weight"] = synthetic_tensor("classifier.dense.weight", (h, h), 4.0 / math.sqrt(h))
        sd["classifier.dense.bias"] = synthetic_tensor("classifier.dense.bias", (h,), 0.1)
        sd["classifier.out_proj.weight"] = synthetic_tensor("classifier.out_proj.weight", (1, h), 8.0 / math.sqrt(h))
        sd["classifier.out_proj.bias"] = synthetic_tensor("classifier.out_proj.bias", (1,), 0.1)
        return sd
    if recipe != "spread":
        raise ValueError(f"unknown weight recipe {recipe!r}")
    u = math.sqrt(3.0)  # uniform(-s, s) has std s / sqrt(3)

    def mat(name, shape, std):
        return synthetic_tensor(name + "#spread", shape, std * u)

    sd["embeddings.word_embeddings.weight"] = mat("embeddings.word_embeddings.weight", (cfg.vocab_size, h), 1.0)
    sd["embeddings.position_embeddings.weight"] = mat("embeddings.position_embeddings.weight", (cfg.max_position_embeddings, h), 0.3)
    for i in range(cfg.num_hidden_layers):
        p = f"encoder.layer.{i}."
        for nm, shape, std in (("attention.self.query", (h, h), 1.5 / math.sqrt(h)), ("attention.self.key", (h, h), 1.5 / math.sqrt(h)),
                               ("attention.self.value", (h, h), 1.5 / math.sqrt(h)), ("attention.output.dense", (h, h), 1.5 / math.sqrt(h)),
                               ("intermediate.dense", (f, h), 1.0 / math.sqrt(h)), ("output.dense", (h, f), 1.0 / math.sqrt(f))):
            sd[p + nm + ".weight"] = mat(p + nm + ".weight", shape, std)
    sd["classifier.dense.weight"] = mat("classifier.dense.weight", (h, h), 1.0 / math.sqrt(h))
    sd["classifier.dense.bias"] = synthetic_tensor("classifier.dense.bias", (h,), 0.1)
    sd["classifier.out_proj.weight"] = mat("classifier.out_proj.weight", (1, h), 2.0 / math.sqrt(h))
    sd["classifier.out_proj.bias"] = synthetic_tensor("classifier.out_proj.bias", (1,), 0.1)
    return sd


def synthetic_pair_token_ids(cfg: TeacherConfig, n_pairs: int, width: int, seed: int, lengths=None):
    """XLM-R pair sequences ``<s> q </s></s> d </s>`` of random token ids (4 .. vocab), right-padded:
    ``(ids int32 [n, width], mask int32 [n, width])``.  Every pair draws its own tokens, so different inputs really
    differ (the BERT-shaped ``synthetic_token_ids`` recipe clamps small vocabularies to ONE id)."""
    rng = np.random.default_rng(seed)
    if lengths is None:
        lengths = rng.integers(min(8, width), width + 1, size=n_pairs)
    ids = rng.integers(4, cfg.vocab_size, size=(n_pairs, width)).astype(np.int32)
    mask = np.zeros((n_pairs, width), np.int32)
    for b, n in enumerate(lengths):
        n = int(min(max(n, 6), width))
        mask[b, :n] = 1
        ids[b, n:] = cfg.pad_token_id
        ids[b, 0] = 0
        q_len = max(1, min(n - 5, int(rng.integers(2, 12))))
        ids[b, 1 + q_len] = 2
        ids[b, 2 + q_len] = 2
        ids[b, n - 1] = 2
    return ids, mask


class TeacherModel:
    def __init__(self, model_name: str = "BAAI/bge-reranker-large", device: Optional[str] = None, *,
                 config: Optional[TeacherConfig] = None, state_dict: Optional[Dict[str, np.ndarray]] = None,
                 tokenizer=None, max_length: int = 512) -> None:
        _native.require_gpu()
        if device is None or device == "cuda":
            self.torch_device = torch.device("cuda", torch.cuda.current_device())
        else:
            self.torch_device = torch.device(device)
            if self.torch_device.type != "cuda":
                raise RuntimeError(f"device={device!r}: the teacher runs on MI355X only (no CPU path)")
        self.device = str(self.torch_device)
        self.model_name = model_name
        if state_dict is None:
            mdir = Path(model_name)
            if not mdir.is_dir():
                raise FileNotFoundError(
                    f"{model_name!r} is not a local directory. This backend never downloads checkpoints: point it at "
                    "a directory holding config.json + model.safetensors + tokenizer.json"
                )
            from safetensors.numpy import load_file

            config = TeacherConfig.from_json(mdir / "config.json")
            raw = load_file(str(mdir / "model.safetensors"))
            state_dict = {(k[8:] if k.startswith("roberta.") else k): np.asarray(v, np.float32) for k, v in raw.items()}
            if tokenizer is None and (mdir / "tokenizer.json").exists():
                from tokenizers import Tokenizer

                tokenizer = Tokenizer.from_file(str(mdir / "tokenizer.json"))
        self.config = config or TeacherConfig()
        self.tokenizer = tokenizer
        self.max_length = min(max_length, self.config.max_position_embeddings - 2)
        self._upload(state_dict)
        self._workspace: Optional[torch.Tensor] = None

    @classmethod
    def from_synthetic(cls, config: Optional[TeacherConfig] = None, device: Optional[str] = None, **kw) -> "TeacherModel":
        cfg = config or TeacherConfig()
        return cls("synthetic-reranker", device, config=cfg, state_dict=synthetic_teacher_state_dict(cfg), **kw)

    @classmethod
    def from_random_device(cls, config: Optional[TeacherConfig] = None, device: Optional[str] = None, seed: int = 0):
        """Random-init weights drawn ON the device (benchmarks: the numpy recipe takes a minute for 560 M parameters)."""
        cfg = config or TeacherConfig()
        self = cls.__new__(cls)
        _native.require_gpu()
        self.torch_device = torch.device(device or f"cuda:{torch.cuda.current_device()}")
        self.device, self.model_name, self.config, self.tokenizer = str(self.torch_device), "random-reranker", cfg, None
        self.max_length = cfg.max_position_embeddings - 2
        g = torch.Generator(device=self.torch_device).manual_seed(seed)
        h, f = cfg.hidden_size, cfg.intermediate_size

        def mat(*shape):
            return (torch.randn(shape, generator=g, device=self.torch_device) * 0.02)

        sd = {"embeddings.word_embeddings.weight": mat(cfg.vocab_size, h),
              "embeddings.position_embeddings.weight": mat(cfg.max_position_embeddings, h),
              "embeddings.token_type_embeddings.weight": mat(cfg.type_vocab_size, h),
              "embeddings.LayerNorm.weight": torch.ones(h, device=self.torch_device),
              "embeddings.LayerNorm.bias": torch.zeros(h, device=self.torch_device)}
        for i in range(cfg.num_hidden_layers):
            p = f"encoder.layer.{i}."
            for nm, shape in (("attention.self.query", (h, h)), ("attention.self.key", (h, h)), ("attention.self.value", (h, h)),
                              ("attention.output.dense", (h, h)), ("intermediate.dense", (f, h)), ("output.dense", (h, f))):
                sd[p + nm + ".weight"] = mat(*shape)
                sd[p + nm + ".bias"] = torch.zeros(shape[0], device=self.torch_device)
            for nm in ("attention.output.LayerNorm", "output.LayerNorm"):
                sd[p + nm + ".weight"] = torch.ones(h, device=self.torch_device)
                sd[p + nm + ".bias"] = torch.zeros(h, device=self.torch_device)
        sd["classifier.dense.weight"], sd["classifier.dense.bias"] = mat(h, h), torch.zeros(h, device=self.torch_device)
        sd["classifier.out_proj.weight"], sd["classifier.out_proj.bias"] = mat(1, h), torch.zeros(1, device=self.torch_device)
        self._upload(sd)
        self._workspace = None
        return self

    # ------------------------------------------------------------------ weights
    def _upload(self, sd) -> None:
        cfg, dev = self.config, self.torch_device
        keep = []

        def t(x):
            return x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x, np.float32))

        def bf(x) -> int:
            y = t(x).to(dev).to(torch.bfloat16).contiguous()
            keep.append(y)
            return y.data_ptr()

        def f32(x) -> int:
            y = t(x).to(device=dev, dtype=torch.float32).contiguous()
            keep.append(y)
            return y.data_ptr()

        L = cfg.num_hidden_layers
        layers = (_native.GenericLayerWeights * max(L, 1))()
        for i in range(L):
            p = f"encoder.layer.{i}."
            lw = layers[i]
            lw.wqkv = bf(torch.cat([t(sd[p + f"attention.self.{n}.weight"]) for n in ("query", "key", "value")], dim=0))
            lw.bqkv = f32(torch.cat([t(sd[p + f"attention.self.{n}.bias"]) for n in ("query", "key", "value")]))
            lw.wo, lw.bo = bf(sd[p + "attention.output.dense.weight"]), f32(sd[p + "attention.output.dense.bias"])
            lw.w1, lw.b1 = bf(sd[p + "intermediate.dense.weight"]), f32(sd[p + "intermediate.dense.bias"])
            lw.w2, lw.b2 = bf(sd[p + "output.dense.weight"]), f32(sd[p + "output.dense.bias"])
            lw.ln1_g, lw.ln1_b = f32(sd[p + "attention.output.LayerNorm.weight"]), f32(sd[p + "attention.output.LayerNorm.bias"])
            lw.ln2_g, lw.ln2_b = f32(sd[p + "output.LayerNorm.weight"]), f32(sd[p + "output.LayerNorm.bias"])
        w = _native.GenericWeights()
        w.word_emb = bf(sd["embeddings.word_embeddings.weight"])
        w.pos_emb = bf(sd["embeddings.position_embeddings.weight"])
        w.type_emb = bf(sd["embeddings.token_type_embeddings.weight"])
        w.emb_ln_g, w.emb_ln_b = f32(sd["embeddings.LayerNorm.weight"]), f32(sd["embeddings.LayerNorm.bias"])
        w.layers = layers
        # the classification head stays fp32 end to end (a reranker's product is the ORDER of its logits)
        self._head = (f32(sd["classifier.dense.weight"]), f32(sd["classifier.dense.bias"]),
                      f32(sd["classifier.out_proj.weight"]), f32(sd["classifier.out_proj.bias"]))
        self._w, self._layers, self._keep = w, layers, keep
        self._cfg = _native.GenericConfig(cfg.vocab_size, cfg.hidden_size, L, cfg.num_attention_heads, cfg.intermediate_size,
                                          cfg.max_position_embeddings, cfg.type_vocab_size, float(cfg.layer_norm_eps),
                                          cfg.pad_token_id + 1)

    # ------------------------------------------------------------------ scoring
    def score_token_ids(self, input_ids, attention_mask, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Raw logits fp32 ``[B]`` (device) for pre-tokenised pair sequences ``<s> q </s></s> d </s>``
        (right-padded with ``pad_token_id``).  Enqueued on the current stream."""
        lib = _native.load()
        ids, mask = self._to_device_i32(input_ids, attention_mask)
        B, S = ids.shape
        Sp = max(32, -(-S // 32) * 32)
        if Sp != S:
            ids = torch.nn.functional.pad(ids, (0, Sp - S), value=self.config.pad_token_id)
            mask = torch.nn.functional.pad(mask, (0, Sp - S))
        if out is None:
            out = torch.empty(B, dtype=torch.float32, device=self.torch_device)
        if B == 0:
            return out
        # tokens per launch a multiple of 256 (8 rows x 32): the weight GEMMs then run on the 256-row tile kernel
        # (1.0-1.2 PFLOP/s) instead of the 128-row fallback for ragged shapes; the filler rows repeat row 0 and
        # their scores are dropped (a row's score does not depend on its batch-mates)
        Bp = -(-B // 8) * 8
        user_out = None
        if Bp != B:
            ids = torch.cat([ids, ids[:1].expand(Bp - B, -1)])
            mask = torch.cat([mask, mask[:1].expand(Bp - B, -1)])
            user_out, out = out, torch.empty(Bp, dtype=torch.float32, device=self.torch_device)
        ids, mask = ids.contiguous(), mask.contiguous()
        B_launch = Bp
        need = int(lib.sskd_teacher_workspace_bytes(self._cfg, B_launch, Sp))
        if self._workspace is None or self._workspace.numel() < need:
            self._workspace = None
            self._workspace = torch.empty(need, dtype=torch.uint8, device=self.torch_device)
        with torch.cuda.device(self.torch_device):
            _native.check(lib.sskd_teacher_score(
                self._cfg, self._w, *self._head, ids.data_ptr(), mask.data_ptr(), B_launch, Sp, out.data_ptr(),
                self._workspace.data_ptr(), self._workspace.numel(),
                int(torch.cuda.current_stream(self.torch_device).cuda_stream)))
        if user_out is not None:
            user_out.copy_(out[:B])
            return user_out
        return out

    def _to_device_i32(self, input_ids, attention_mask):
        """Host arrays travel as ONE pinned block with a non-blocking copy: a pageable ``.to(device)`` makes the host wait
        for everything already enqueued, which serialised tokenising chunk c + 1 behind the GPU work of chunk c."""
        if isinstance(input_ids, torch.Tensor) and isinstance(attention_mask, torch.Tensor):
            return (input_ids.to(device=self.torch_device, dtype=torch.int32),
                    attention_mask.to(device=self.torch_device, dtype=torch.int32))
        a, m = np.asarray(input_ids), np.asarray(attention_mask)
        stage = torch.empty((2,) + a.shape, dtype=torch.int32, pin_memory=True)
        stage[0].copy_(torch.from_numpy(np.ascontiguousarray(a, np.int32)))
        stage[1].copy_(torch.from_numpy(np.ascontiguousarray(m, np.int32)))
        dev = stage.to(self.torch_device, non_blocking=True)
        return dev[0], dev[1]

    def tokenize_pairs(self, pairs: Sequence[Union[Tuple[str, str], List[str]]]):
        if self.tokenizer is None:
            raise RuntimeError("no tokenizer: the model directory has no tokenizer.json (use score_token_ids)")
        encs = self.tokenizer.encode_batch([(p[0], p[1]) for p in pairs])
        rows = [e.ids[: self.max_length - 1] + e.ids[-1:] if len(e.ids) > self.max_length else e.ids for e in encs]
        width = max((len(r) for r in rows), default=1)
        ids = np.full((len(rows), width), self.config.pad_token_id, np.int32)
        mask = np.zeros((len(rows), width), np.int32)
        for i, r in enumerate(rows):
            ids[i, : len(r)] = r
            mask[i, : len(r)] = 1
        return ids, mask

    def data_parallel(self, enabled: bool = True, group=None) -> "TeacherModel":
        """Shard ``score`` over the ranks of a ``torch.distributed`` process group (BASELINE cfg 5: one process per
        GPU, each holding the full model): rank r scores the contiguous pair range ``shard_bounds(n, G, r)``, ONE
        all-gather of the scores is the final concat, every rank returns all n scores (``dist.sharded_scores``).
        Every rank must call ``score`` with the same pairs.  ``group=None`` = the default group."""
        self._dp_enabled, self._dp_group = bool(enabled), group
        return self

    SCORE_CHUNK = 2048   # pairs tokenised at a time: the GPU scores chunk c while the host tokenises chunk c + 1

    def _score_local(self, pairs) -> torch.Tensor:
        """fp32 [n] device tensor of raw logits for this process's pairs.  Pairs are taken in chunks: a chunk is
        tokenised (the Rust tokenizer, GIL released), sorted by length, cut into launches by a token budget and
        ENQUEUED - launches are asynchronous, so the GPU works on chunk c while the host tokenises chunk c + 1; nothing
        synchronises until the caller reads the scores.  Token ids travel as one pinned block per launch."""
        n = len(pairs)
        out = torch.empty(n, dtype=torch.float32, device=self.torch_device)
        budget = 64 * 512
        for c0 in range(0, n, self.SCORE_CHUNK):
            chunk = pairs[c0 : c0 + self.SCORE_CHUNK]
            ids, mask = self.tokenize_pairs(chunk)
            lengths = mask.sum(1)
            order = np.argsort(-lengths, kind="stable")
            ids, mask, lengths = ids[order], mask[order], lengths[order]
            m = len(chunk)
            sorted_out = torch.empty(m, dtype=torch.float32, device=self.torch_device)
            lo = 0
            while lo < m:
                width = max(int(lengths[lo]), 1)
                rows = max(8, budget // (-(-width // 32) * 32) // 8 * 8)   # whole 256-token multiples per launch
                self.score_token_ids(ids[lo : lo + rows, :width], mask[lo : lo + rows, :width], out=sorted_out[lo : lo + rows])
                lo += rows
            perm = torch.empty(m, dtype=torch.int64, pin_memory=True)
            perm.copy_(torch.from_numpy(order.astype(np.int64) + c0))
            out[perm.to(self.torch_device, non_blocking=True)] = sorted_out   # one un-permute per chunk
        return out

    def score(self, pairs: Sequence[Union[Tuple[str, str], List[str]]], batch_size: int = 32) -> List[float]:
        """``CrossEncoder.predict``-shaped: one float (raw logit) per (query, passage) pair.  Pairs are
        sorted by length and cut into launches by a token budget; ``batch_size`` (reference default 32)
        does not shape the GPU work.  After ``data_parallel()`` the pairs are split over the process group."""
        del batch_size
        n = len(pairs)
        if n == 0:
            return []
        if getattr(self, "_dp_enabled", False):
            from .dist import sharded_scores

            out = sharded_scores(lambda lo, hi: self._score_local(pairs[lo:hi]), n, self._dp_group, self.torch_device)
        else:
            out = self._score_local(pairs)
        return [float(x) for x in out.cpu().numpy()]

    def predict_score(self, query: str, document: str) -> float:
        return self.score([(query, document)])[0]

    def predict(self, pairs, batch_size: int = 32):
        return np.asarray(self.score(pairs, batch_size=batch_size), np.float32)

    @staticmethod
    def get_confidence(score: float) -> float:
        return 1.0 / (1.0 + math.exp(-float(score)))

    def cleanup(self) -> None:
        self._workspace = None
