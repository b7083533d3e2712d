"""Encoder weights: HF-directory loading, deterministic synthetic weights, MFMA fragment tiling.

The reference loads ``intfloat/e5-small-v2`` (or a fine-tuned copy) through
``SentenceTransformer(model_name)`` (reference: src/serve/app.py:87-90; files pinned by
tests/test_model_validation.py:243-262: ``config.json``, ``model.safetensors``,
``tokenizer.json`` ...).  This module reads the same directory layout with ``safetensors``
directly, and — because no checkpoint can be fetched offline — also provides the
version-independent synthetic weight recipe of SURVEY.md App. B (64-bit counter ->
splitmix64 -> uniform(-1, 1) * 0.02 * sqrt(3); LayerNorm gamma = 1, beta = 0, biases 0 by
default) so that tests and benchmarks use random-init weights of the real architecture.
"""
from __future__ import annotations

import json
import zlib
from dataclasses import dataclass
from pathlib import Path
from typing import Dict, Optional

import numpy as np


@dataclass(frozen=True)
class BertConfig:
    """Architecture constants (e5-small-v2 defaults)."""

    vocab_size: int = 30522
    hidden_size: int = 384
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 1536
    max_position_embeddings: int = 512
    type_vocab_size: int = 2
    layer_norm_eps: float = 1e-12
    hidden_act: str = "gelu"

    @staticmethod
    def from_json(path: Path) -> "BertConfig":
        raw = json.loads(Path(path).read_text())
        keys = BertConfig.__dataclass_fields__.keys()
        return BertConfig(**{k: raw[k] for k in keys if k in raw})

    def to_hf_dict(self) -> dict:
        d = {k: getattr(self, k) for k in self.__dataclass_fields__}
        d.update(model_type="bert", architectures=["BertModel"], pad_token_id=0, position_embedding_type="absolute")
        return d


# ------------------------------------------------------------------ synthetic weights
_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def synthetic_tensor(name: str, shape, scale: float) -> np.ndarray:
    """uniform(-scale, scale) keyed by (crc32(name), flat index): identical on every platform."""
    n = int(np.prod(shape))
    key = np.uint64(zlib.crc32(name.encode("utf-8"))) << np.uint64(32)
    bits = _splitmix64(key + np.arange(n, dtype=np.uint64))
    u = (bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))  # [0, 1)
    return ((2.0 * u - 1.0) * scale).astype(np.float32).reshape(shape)


STRESS_OUTLIER_CHANNELS = (7, 133, 300)


def synthetic_state_dict(cfg: BertConfig, nonzero_bias: bool = True, stress: bool = False,
                         recipe: str = "init") -> Dict[str, np.ndarray]:
    """fp32 state dict with HF ``BertModel`` parameter names (no pooler: mean pooling ignores it).

    ``nonzero_bias`` also draws biases / LayerNorm parameters (small, around their defaults) so
    that parity tests exercise every term of the forward pass.

    ``stress`` draws weights shaped like a TRAINED checkpoint's hard cases instead of the benign
    init: query / key projections scaled up so that attention logits are O(5) (peaky softmax rows:
    the online-softmax rescale and bf16 P.V paths matter), LayerNorm gains log-uniform in
    [0.3, 3], biases of +-0.5, and three outlier channels whose LayerNorm bias is +-10 (the
    near-constant massive-activation channels real BERT checkpoints have: 10x the typical value).

    ``recipe="spread"`` draws weights with the GAINS of a trained checkpoint instead of the 0.02 init (which barely
    mixes tokens: embeddings of different texts then lie within 1e-3 cosine of each other and no test on them can
    tell a wrong row from rounding): unit-variance word embeddings (token identity dominates position), every
    projection at unit gain (sqrt(3 / fan_in) uniform), query / key at 1.4x so that attention logits are O(2).
    Embeddings of different texts then separate by O(0.1 ... 1) in cosine.
    """
    if recipe not in ("init", "spread"):
        raise ValueError(f"unknown weight recipe {recipe!r}")
    h, f = cfg.hidden_size, cfg.intermediate_size
    w_scale = 0.02 * np.sqrt(3.0)

    def mat(name, shape):
        if recipe == "spread":
            if name.endswith("word_embeddings.weight"):
                return synthetic_tensor(name, shape, float(np.sqrt(3.0)))
            if "embeddings." in name and "encoder." not in name:
                return synthetic_tensor(name, shape, 0.1 * float(np.sqrt(3.0)))
            gain = 1.4 if (".query." in name or ".key." in name) else 1.0
            return synthetic_tensor(name, shape, gain * float(np.sqrt(3.0 / shape[1])))
        w = synthetic_tensor(name, shape, w_scale)
        if stress and (".query." in name or ".key." in name):
            w = w * np.float32(3.5)
        return w

    def vec(name, n, centre):
        if not nonzero_bias:
            return np.full(n, centre, np.float32)
        if stress:
            u = synthetic_tensor(name, (n,), 1.0)
            if centre == 1.0:  # LayerNorm gain
                return np.power(np.float32(3.0), u).astype(np.float32)
            b = (0.5 * u).astype(np.float32)
            if "LayerNorm" in name:  # near-constant massive activations in a few channels
                b[list(STRESS_OUTLIER_CHANNELS)] = np.float32(10.0) * np.sign(b[list(STRESS_OUTLIER_CHANNELS)])
            return b
        return (centre + synthetic_tensor(name, (n,), 0.1)).astype(np.float32)

    sd = {
        "embeddings.word_embeddings.weight": mat("embeddings.word_embeddings.weight", (cfg.vocab_size, h)),
        "embeddings.position_embeddings.weight": mat(
            "embeddings.position_embeddings.weight", (cfg.max_position_embeddings, h)
        ),
        "embeddings.token_type_embeddings.weight": mat(
            "embeddings.token_type_embeddings.weight", (cfg.type_vocab_size, h)
        ),
        "embeddings.LayerNorm.weight": vec("embeddings.LayerNorm.weight", h, 1.0),
        "embeddings.LayerNorm.bias": vec("embeddings.LayerNorm.bias", h, 0.0),
    }
    for i in range(cfg.num_hidden_layers):
        p = f"encoder.layer.{i}."
        for nm, shape in (
            ("attention.self.query", (h, h)),
            ("attention.self.key", (h, h)),
            ("attention.self.value", (h, h)),
            ("attention.output.dense", (h, h)),
            ("intermediate.dense", (f, h)),
            ("output.dense", (h, f)),
        ):
            sd[p + nm + ".weight"] = mat(p + nm + ".weight", shape)
            sd[p + nm + ".bias"] = vec(p + nm + ".bias", shape[0], 0.0)
        for nm in ("attention.output.LayerNorm", "output.LayerNorm"):
            sd[p + nm + ".weight"] = vec(p + nm + ".weight", h, 1.0)
            sd[p + nm + ".bias"] = vec(p + nm + ".bias", h, 0.0)
    return sd


def load_state_dict(model_dir: Path) -> Dict[str, np.ndarray]:
    """Read ``model.safetensors`` (sentence-transformers / HF layout), stripping a ``bert.`` prefix."""
    from safetensors.numpy import load_file

    model_dir = Path(model_dir)
    path = model_dir / "model.safetensors"
    if not path.exists() and (model_dir / "0_Transformer" / "model.safetensors").exists():
        path = model_dir / "0_Transformer" / "model.safetensors"
    if not path.exists():
        raise FileNotFoundError(
            f"{path} not found: pass a LOCAL model directory (config.json + model.safetensors + tokenizer); "
            "model names cannot be fetched offline"
        )
    sd = load_file(str(path))
    return {(k[5:] if k.startswith("bert.") else k): np.asarray(v, np.float32) for k, v in sd.items()}


def save_model_dir(model_dir: Path, cfg: BertConfig, sd: Dict[str, np.ndarray]) -> None:
    """Write an HF-style directory (used by tests / tooling to materialise synthetic models)."""
    from safetensors.numpy import save_file

    model_dir = Path(model_dir)
    model_dir.mkdir(parents=True, exist_ok=True)
    (model_dir / "config.json").write_text(json.dumps(cfg.to_hf_dict(), indent=1))
    save_file({k: np.ascontiguousarray(v) for k, v in sd.items()}, str(model_dir / "model.safetensors"))


# ------------------------------------------------------------------ fragment tiling
def tile_weight_fragments(w: np.ndarray) -> np.ndarray:
    """``W[N, K]`` (nn.Linear layout) -> A-operand fragments ``[N/32, K/16, 64, 8]`` of
    ``v_mfma_f32_32x32x16_bf16``: lane ``l`` of fragment ``(nt, s)`` holds
    ``W[32 nt + (l & 31)][16 s + 8 (l >> 5) + j]``, ``j = 0..7`` (csrc/encoder.hip)."""
    n, k = w.shape
    assert n % 32 == 0 and k % 16 == 0, (n, k)
    t = w.reshape(n // 32, 32, k // 16, 2, 8)  # [nt, r, s, h, j]
    t = t.transpose(0, 2, 3, 1, 4)             # [nt, s, h, r, j] -> lane = 32 h + r
    return np.ascontiguousarray(t.reshape(n // 32, k // 16, 64, 8))


def tile_w2_chunked(w2: np.ndarray) -> np.ndarray:
    """``W2[384, 1536]`` -> ``[48 chunks][12 tiles][2 k-steps][64 lanes][8]`` for the fused MLP: the
    consumer waves' A operand for hidden chunk ``c`` (features 32c..32c+31).  The k slots are
    permuted so that the producers' accumulators ARE the matching B fragments, lane for lane:
    lane ``l``, slot ``j`` of fragment ``(c, nt, s2)`` holds
    ``W2[32 nt + (l & 31)][32 c + 16 s2 + 8 (j >> 2) + 4 (l >> 5) + (j & 3)]``
    (csrc/encoder.hip fused_mlp_ln_kernel, "Hand-over")."""
    n, k = w2.shape
    assert n % 32 == 0 and k % 32 == 0, (n, k)
    t = w2.reshape(n // 32, 32, k // 32, 2, 2, 2, 4)   # [nt, r, c, s2, jh, h, jl]: unit = 16 s2 + 8 jh + 4 h + jl
    t = t.transpose(2, 0, 3, 5, 1, 4, 6)               # [c, nt, s2, h, r, jh, jl] -> lane = 32 h + r, j = 4 jh + jl
    return np.ascontiguousarray(t.reshape(k // 32, n // 32, 2, 64, 8))


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even fp32 -> bf16 bit pattern (uint16)."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    return ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16)


def bf16_round(x: np.ndarray) -> np.ndarray:
    """fp32 values rounded to the nearest bf16 (still stored as fp32)."""
    return (f32_to_bf16_bits(x).astype(np.uint32) << np.uint32(16)).view(np.float32)


class DeviceWeights:
    """bf16 / fp32 weight tensors resident in HBM plus the C structs that point at them."""

    def __init__(self, cfg: BertConfig, sd: Dict[str, np.ndarray], device) -> None:
        import torch

        from . import _native

        if cfg.hidden_size != 384 or cfg.num_attention_heads != 12 or cfg.intermediate_size != 1536:
            raise ValueError(
                "the gfx950 encoder kernels are specialised for hidden=384, heads=12, intermediate=1536 "
                f"(got {cfg.hidden_size}, {cfg.num_attention_heads}, {cfg.intermediate_size})"
            )
        if cfg.hidden_act != "gelu":
            raise ValueError(f"hidden_act={cfg.hidden_act!r}: only exact-erf 'gelu' is implemented")
        self.cfg = cfg
        self.device = device
        self._keep = []

        def bf16(a: np.ndarray):
            t = torch.from_numpy(f32_to_bf16_bits(a).view(np.int16)).to(device).view(torch.bfloat16)
            self._keep.append(t)
            return t.data_ptr()

        def f32(a: np.ndarray):
            t = torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(device)
            self._keep.append(t)
            return t.data_ptr()

        L = cfg.num_hidden_layers
        self.layers = (_native.EncoderLayerWeights * max(L, 1))()
        for i in range(L):
            p = f"encoder.layer.{i}."
            wqkv = np.concatenate(
                [sd[p + f"attention.self.{n}.weight"] for n in ("query", "key", "value")], axis=0
            )
            bqkv = np.concatenate([sd[p + f"attention.self.{n}.bias"] for n in ("query", "key", "value")])
            lw = self.layers[i]
            lw.wqkv = bf16(tile_weight_fragments(wqkv))
            lw.bqkv = f32(bqkv)
            lw.wo = bf16(tile_weight_fragments(sd[p + "attention.output.dense.weight"]))
            lw.bo = f32(sd[p + "attention.output.dense.bias"])
            lw.ln1_g = f32(sd[p + "attention.output.LayerNorm.weight"])
            lw.ln1_b = f32(sd[p + "attention.output.LayerNorm.bias"])
            lw.w1 = bf16(tile_weight_fragments(sd[p + "intermediate.dense.weight"]))
            lw.b1 = f32(sd[p + "intermediate.dense.bias"])
            lw.w2 = bf16(tile_w2_chunked(sd[p + "output.dense.weight"]))
            lw.b2 = f32(sd[p + "output.dense.bias"])
            lw.ln2_g = f32(sd[p + "output.LayerNorm.weight"])
            lw.ln2_b = f32(sd[p + "output.LayerNorm.bias"])
        self.struct = _native.EncoderWeights()
        self.struct.word_emb = bf16(sd["embeddings.word_embeddings.weight"])
        self.struct.pos_emb = bf16(sd["embeddings.position_embeddings.weight"])
        self.struct.type_emb = bf16(sd["embeddings.token_type_embeddings.weight"])
        self.struct.emb_ln_g = f32(sd["embeddings.LayerNorm.weight"])
        self.struct.emb_ln_b = f32(sd["embeddings.LayerNorm.bias"])
        self.struct.layers = self.layers
        self.cstruct_cfg = _native.EncoderConfig(
            cfg.vocab_size,
            cfg.hidden_size,
            cfg.num_hidden_layers,
            cfg.num_attention_heads,
            cfg.intermediate_size,
            cfg.max_position_embeddings,
            cfg.type_vocab_size,
            float(cfg.layer_norm_eps),
        )

    def nbytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self._keep)


def load_config(model_dir: Optional[Path]) -> BertConfig:
    if model_dir is None:
        return BertConfig()
    p = Path(model_dir) / "config.json"
    if not p.exists() and (Path(model_dir) / "0_Transformer" / "config.json").exists():
        p = Path(model_dir) / "0_Transformer" / "config.json"
    return BertConfig.from_json(p)
