"""Student forward / backward on the MI355X (BASELINE cfg 4): ``encode_with_gradients``.

The reference's KD step (src/kd/train.py:176-210) calls ``StudentModel.encode_with_gradients``
twice per query (query, then its positive + hard negatives), multiplies, applies
``CombinedKDLoss`` and lets torch autograd walk back through the sentence-transformers modules.
Here the encoder forward that saves its activations and the whole backward pass are HIP
(``sskd_generic_forward`` / ``sskd_generic_backward``, csrc/train.hip) behind ONE
``torch.autograd.Function``; torch holds the fp32 master parameters (what the optimizer updates)
and supplies device memory.  bf16 compute, fp32 accumulation and gradients.  No CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
from torch import nn

from . import _native
from .weights import BertConfig

_LAYER_MATS = (
    ("attention.self.query", "q"), ("attention.self.key", "k"), ("attention.self.value", "v"),
    ("attention.output.dense", "o"), ("intermediate.dense", "w1"), ("output.dense", "w2"),
)


def _pname(hf_name: str) -> str:
    return hf_name.replace(".", "__")


class _EncoderFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module: "TrainableEncoder", ids: torch.Tensor, mask: torch.Tensor, normalize: bool, *params):
        lib = _native.load()
        B, S = ids.shape
        module._refresh_device_weights()
        need = int(lib.sskd_generic_workspace_bytes(module.cfg_struct, B, S, 1))
        ws = torch.empty(max(need, 1), dtype=torch.uint8, device=ids.device)
        out = torch.empty((B, module.config.hidden_size), dtype=torch.float32, device=ids.device)
        with torch.cuda.device(ids.device):
            _native.check(lib.sskd_generic_forward(
                module.cfg_struct, module.w_struct, ids.data_ptr(), mask.data_ptr(), B, S, 1, 1, int(normalize),
                out.data_ptr(), ws.data_ptr(), ws.numel(), int(torch.cuda.current_stream(ids.device).cuda_stream)))
        ctx.module, ctx.ws, ctx.normalize = module, ws, bool(normalize)
        ctx.save_for_backward(ids, mask)
        ctx.weights_version = module._weights_version
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _native.load()
        module: TrainableEncoder = ctx.module
        ids, mask = ctx.saved_tensors
        B, S = ids.shape
        if ctx.weights_version != module._weights_version:
            raise RuntimeError("parameters changed between forward and backward of encode_with_gradients")
        grads, gstruct, keep = module._new_grad_buffers()
        dout = dout.to(torch.float32).contiguous()
        with torch.cuda.device(ids.device):
            _native.check(lib.sskd_generic_backward(
                module.cfg_struct, module.w_struct, gstruct, ids.data_ptr(), mask.data_ptr(), B, S, int(ctx.normalize),
                dout.data_ptr(), ctx.ws.data_ptr(), ctx.ws.numel(),
                int(torch.cuda.current_stream(ids.device).cuda_stream)))
        del keep
        ctx.ws = None
        return (None, None, None, None, *module._grads_in_param_order(grads))


class TrainableEncoder(nn.Module):
    """fp32 master parameters (HF ``BertModel`` names) + the bf16 device copies the kernels read."""

    def __init__(self, config: BertConfig, state_dict: Dict[str, np.ndarray], device, pos_offset: int = 0) -> None:
        super().__init__()
        _native.require_gpu()
        self.config = config
        self.device = torch.device(device)
        self.pos_offset = pos_offset
        self.names: List[str] = []
        for name, arr in state_dict.items():
            if name.endswith("position_ids") or name.startswith("pooler."):
                continue
            self.names.append(name)
            self.register_parameter(_pname(name), nn.Parameter(torch.from_numpy(np.ascontiguousarray(arr, np.float32)).to(self.device)))
        self.cfg_struct = _native.GenericConfig(
            config.vocab_size, config.hidden_size, config.num_hidden_layers, config.num_attention_heads,
            config.intermediate_size, config.max_position_embeddings, config.type_vocab_size,
            float(config.layer_norm_eps), pos_offset)
        self._weights_version = -1
        self._seen_versions: Optional[tuple] = None
        self._keep: list = []
        self.w_struct = None

    def p(self, hf_name: str) -> nn.Parameter:
        return getattr(self, _pname(hf_name))

    # -------------------------------------------------------------- device copies
    def _refresh_device_weights(self) -> None:
        versions = tuple(p._version for p in self.parameters())
        if versions == self._seen_versions and self.w_struct is not None:
            return
        keep = []

        def bf(t: torch.Tensor) -> int:
            t = t.detach().to(torch.bfloat16).contiguous()
            keep.append(t)
            return t.data_ptr()

        def f32(t: torch.Tensor) -> int:
            t = t.detach().contiguous()
            keep.append(t)
            return t.data_ptr()

        L = self.config.num_hidden_layers
        layers = (_native.GenericLayerWeights * max(L, 1))()
        for i in range(L):
            pre = f"encoder.layer.{i}."
            wqkv = torch.cat([self.p(pre + f"attention.self.{n}.weight") for n in ("query", "key", "value")], dim=0)
            lw = layers[i]
            lw.wqkv, lw.wqkv_t = bf(wqkv), bf(wqkv.detach().t())
            lw.bqkv = f32(torch.cat([self.p(pre + f"attention.self.{n}.bias") for n in ("query", "key", "value")]))
            for hf, short in (("attention.output.dense", "wo"), ("intermediate.dense", "w1"), ("output.dense", "w2")):
                w = self.p(pre + hf + ".weight")
                setattr(lw, short, bf(w))
                setattr(lw, short + "_t", bf(w.detach().t()))
            lw.bo = f32(self.p(pre + "attention.output.dense.bias"))
            lw.b1 = f32(self.p(pre + "intermediate.dense.bias"))
            lw.b2 = f32(self.p(pre + "output.dense.bias"))
            lw.ln1_g, lw.ln1_b = f32(self.p(pre + "attention.output.LayerNorm.weight")), f32(self.p(pre + "attention.output.LayerNorm.bias"))
            lw.ln2_g, lw.ln2_b = f32(self.p(pre + "output.LayerNorm.weight")), f32(self.p(pre + "output.LayerNorm.bias"))
        w = _native.GenericWeights()
        w.word_emb = bf(self.p("embeddings.word_embeddings.weight"))
        w.pos_emb = bf(self.p("embeddings.position_embeddings.weight"))
        w.type_emb = bf(self.p("embeddings.token_type_embeddings.weight"))
        w.emb_ln_g = f32(self.p("embeddings.LayerNorm.weight"))
        w.emb_ln_b = f32(self.p("embeddings.LayerNorm.bias"))
        w.layers = layers
        self.w_struct, self._layers_struct, self._keep = w, layers, keep
        self._seen_versions = versions
        self._weights_version += 1

    def _new_grad_buffers(self):
        """Zeroed fp32 buffers in the C-ABI's (fused QKV) shapes + the struct pointing at them."""
        cfg = self.config
        H, F, L = cfg.hidden_size, cfg.intermediate_size, cfg.num_hidden_layers
        dev = self.device
        z = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)  # noqa: E731
        g = {"word": z(cfg.vocab_size, H), "pos": z(cfg.max_position_embeddings, H), "type": z(cfg.type_vocab_size, H),
             "emb_g": z(H), "emb_b": z(H), "layers": []}
        layers = (_native.GenericLayerGrads * max(L, 1))()
        for i in range(L):
            lg = {"wqkv": z(3 * H, H), "bqkv": z(3 * H), "wo": z(H, H), "bo": z(H), "ln1_g": z(H), "ln1_b": z(H),
                  "w1": z(F, H), "b1": z(F), "w2": z(H, F), "b2": z(H), "ln2_g": z(H), "ln2_b": z(H)}
            for k, t in lg.items():
                setattr(layers[i], k, t.data_ptr())
            g["layers"].append(lg)
        s = _native.GenericGrads()
        s.word_emb, s.pos_emb, s.type_emb = g["word"].data_ptr(), g["pos"].data_ptr(), g["type"].data_ptr()
        s.emb_ln_g, s.emb_ln_b = g["emb_g"].data_ptr(), g["emb_b"].data_ptr()
        s.layers = layers
        return g, s, layers

    def _grads_in_param_order(self, g) -> List[torch.Tensor]:
        H = self.config.hidden_size
        out = []
        for name in self.names:
            if name == "embeddings.word_embeddings.weight":
                out.append(g["word"])
            elif name == "embeddings.position_embeddings.weight":
                out.append(g["pos"])
            elif name == "embeddings.token_type_embeddings.weight":
                out.append(g["type"])
            elif name == "embeddings.LayerNorm.weight":
                out.append(g["emb_g"])
            elif name == "embeddings.LayerNorm.bias":
                out.append(g["emb_b"])
            else:
                parts = name.split(".")
                lg = g["layers"][int(parts[2])]
                rest, kind = ".".join(parts[3:-1]), parts[-1]
                qkv = {"attention.self.query": 0, "attention.self.key": 1, "attention.self.value": 2}
                if rest in qkv:
                    src = lg["wqkv"] if kind == "weight" else lg["bqkv"]
                    out.append(src[qkv[rest] * H : (qkv[rest] + 1) * H])
                else:
                    key = {"attention.output.dense": ("wo", "bo"), "intermediate.dense": ("w1", "b1"),
                           "output.dense": ("w2", "b2"), "attention.output.LayerNorm": ("ln1_g", "ln1_b"),
                           "output.LayerNorm": ("ln2_g", "ln2_b")}[rest]
                    out.append(lg[key[0] if kind == "weight" else key[1]])
        return out

    # -------------------------------------------------------------- forward
    def forward(self, input_ids, attention_mask=None, normalize: bool = True) -> torch.Tensor:
        """ids / mask ``[B, S]`` -> embeddings ``[B, hidden]`` (fp32, on device, differentiable with
        respect to every parameter).  S is padded to a multiple of 32 here."""
        ids = torch.as_tensor(np.asarray(input_ids) if not isinstance(input_ids, torch.Tensor) else input_ids)
        ids = ids.to(device=self.device, dtype=torch.int32)
        mask = torch.ones_like(ids) if attention_mask is None else torch.as_tensor(
            np.asarray(attention_mask) if not isinstance(attention_mask, torch.Tensor) else attention_mask
        ).to(device=self.device, dtype=torch.int32)
        B, S = ids.shape
        Sp = max(32, -(-S // 32) * 32)
        if Sp != S:
            ids = torch.nn.functional.pad(ids, (0, Sp - S))
            mask = torch.nn.functional.pad(mask, (0, Sp - S))
        params = [self.p(n) for n in self.names]
        return _EncoderFunction.apply(self, ids.contiguous(), mask.contiguous(), normalize, *params)

    def state_dict_numpy(self) -> Dict[str, np.ndarray]:
        return {n: self.p(n).detach().cpu().numpy() for n in self.names}


def kd_step_scores(query_emb: torch.Tensor, doc_embs: torch.Tensor) -> torch.Tensor:
    """``torch.matmul(query_emb, doc_embs.T)[0]`` of the reference step (src/kd/train.py:189)."""
    return torch.matmul(query_emb, doc_embs.T)[0]
